#!/bin/bash
# Collect the judged evidence for one round on the GPU box (run from the repo root):
#   bash profiles/tools/collect_round.sh r01
# 1) rocprofv3 --kernel-trace --stats of the default bench command  -> gpurun_out/<tag>/stats
# 2) separate PMC passes (MFMA busy / FETCH_SIZE / WRITE_SIZE)       -> gpurun_out/<tag>/pmc
# Copy the summaries into profiles/ afterwards (gpurun_out/ is scratch).
set -e
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/stats.err
python bench.py > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA \
  --output-format csv -d $out/pmc/sq -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/pmc_sq.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc/fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc/write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/pmc_write.err
python profiles/summarize_pmc.py $out/pmc > $out/pmc_summary.txt
python - <<PY
import csv, glob, json
f = glob.glob("$out/stats/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
print(open(f).read()[:1500])
agg = {}
for g in glob.glob("$out/pmc/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(g)):
        if 'k_egnn_edge' in r['Kernel_Name'] and r['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE'):
            a = agg.setdefault(r['Counter_Name'], [0.0, 0])
            a[0] += float(r['Counter_Value']); a[1] += 1
fetch_kb = agg['FETCH_SIZE'][0] / agg['FETCH_SIZE'][1]
write_kb = agg['WRITE_SIZE'][0] / agg['WRITE_SIZE'][1]
json.dump({'kernel': 'k_egnn_edge', 'fetch_size_kb_raw': fetch_kb, 'write_size_kb_raw': write_kb,
           'hbm_bytes_per_launch': (2 * fetch_kb + write_kb) * 1024,
           'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, averaged per dispatch; FETCH_SIZE doubled '
                   '(gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md HBM section)'},
          open("$out/traffic.json", 'w'), indent=1)
print(open("$out/traffic.json").read())
PY
cat $out/bench.json
