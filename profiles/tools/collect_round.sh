#!/bin/bash
# Collect the judged evidence for one round on the GPU box (run from the repo root):
#   bash profiles/tools/collect_round.sh r02
# 1) the default bench line (contract line + secondary workloads + end-to-end)            -> <tag>/bench.json
# 2) rocprofv3 --kernel-trace --stats of the contract workload, of its opt-in f16x2 mode and of the two GVP workloads -> <tag>/stats_*
# 3) separate PMC passes per workload: SQ counters, FETCH_SIZE, WRITE_SIZE                  -> <tag>/pmc_*
# 4) traffic.json: HBM bytes per launch of the dominant kernel of every workload (2 x FETCH_SIZE + WRITE_SIZE)
# Copy the summaries into profiles/ afterwards (gpurun_out/ is scratch).
set -e
tag=${1:-r05}
root=$(cd "$(dirname "$0")/../.." && pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/$tag
rm -rf $out && mkdir -p $out
python bench.py > $out/bench.json 2> $out/bench.err && cp gpurun_out/bench_full_record.json $out/bench_full.json
short="--steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-secondary"
for wl in egnn_all_atom egnn_all_atom_f16x2 gvp_40kp gvp_all_atom_ragged gvp_40kp_f16x2 gvp_all_atom_ragged_f16x2; do
  case $wl in
    egnn_all_atom) args="";;
    egnn_all_atom_f16x2) args="--gemm f16x2";;
    gvp_40kp) args="--workload gvp_40kp";;
    gvp_all_atom_ragged) args="--workload gvp_all_atom --ragged";;
    gvp_40kp_f16x2) args="--workload gvp_40kp --gemm f16x2";;
    gvp_all_atom_ragged_f16x2) args="--workload gvp_all_atom --ragged --gemm f16x2";;
  esac
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$wl -- python bench.py --steps 40 --warmup 5 --repeats 1 --no-cpu-baseline --no-secondary $args > $out/bench_under_rocprof_$wl.json 2> $out/stats_$wl.err
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA \
    --output-format csv -d $out/pmc_$wl/sq -- python bench.py $short $args > /dev/null 2> $out/pmc_sq_$wl.err
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_$wl/fetch -- python bench.py $short $args > /dev/null 2> $out/pmc_fetch_$wl.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_$wl/write -- python bench.py $short $args > /dev/null 2> $out/pmc_write_$wl.err
  python profiles/summarize_pmc.py $out/pmc_$wl > $out/pmc_summary_$wl.txt
  cp $(ls $out/stats_$wl/*/*kernel_stats.csv | head -1) $out/kernel_stats_$wl.csv
  echo "== $wl done"
done
KPD_OUT=$out python - <<'PY'
import csv, glob, json, os
out = os.environ['KPD_OUT']
dom = {'egnn_all_atom': 'k_egnn_edge<4>', 'egnn_all_atom_f16x2': 'k_egnn_edge_h', 'gvp_40kp': 'k_gvp_chain<16, 0, 0>', 'gvp_all_atom_ragged': 'k_gvp_chain<16, 0, 0>',
       'gvp_40kp_f16x2': 'k_gvp_chain<16, 1, 0>', 'gvp_all_atom_ragged_f16x2': 'k_gvp_chain<16, 1, 0>'}
res = {}
for wl, kern in dom.items():
    agg = {}
    for g in glob.glob(f'{out}/pmc_{wl}/*/*/*counter_collection.csv'):
        for r in csv.DictReader(open(g)):
            if kern in r['Kernel_Name'] and r['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE'):
                a = agg.setdefault(r['Counter_Name'], [0.0, 0])
                a[0] += float(r['Counter_Value']); a[1] += 1
    fetch_kb = agg['FETCH_SIZE'][0] / agg['FETCH_SIZE'][1]
    write_kb = agg['WRITE_SIZE'][0] / agg['WRITE_SIZE'][1]
    res[wl] = {'kernel': kern, 'fetch_size_kb_raw': fetch_kb, 'write_size_kb_raw': write_kb, 'dispatches': agg['FETCH_SIZE'][1],
               'hbm_bytes_per_launch': (2 * fetch_kb + write_kb) * 1024}
res['note'] = ('rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, averaged over the dispatches of the dominant kernel (the final '
               'layer / conv launch, which runs ll + kl only, included); FETCH_SIZE doubled (gfx950 reports half the bytes of wide '
               'coalesced reads, MI355X_MICROARCH.md HBM section)')
json.dump(res, open(f'{out}/traffic.json', 'w'), indent=1)
print(open(f'{out}/traffic.json').read())
PY
rocm-smi --showuniqueid 2>/dev/null | grep -i 'unique id' > $out/gpu_id.txt
# the raw rocprofv3 trees are only needed for the summaries above (gpurun_out/ is capped at 64 MiB)
find $out -mindepth 1 -maxdepth 1 -type d \( -name 'pmc_*' -o -name 'stats_*' \) -exec rm -rf {} +
tail -c 600 $out/bench.json
