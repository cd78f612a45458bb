#!/bin/bash
# PMC counters of the big kernels of a training workload, three separate passes (SQ MFMA-busy, FETCH_SIZE, WRITE_SIZE) + the kernel statistics
# of the same command for the durations (run on the GPU box from the repo root):
#   bash profiles/tools/train_pmc.sh gvp_train 'gvp_chain|wgrad|node_chain'      -> gpurun_out/train_pmc_<workload>.txt
w=${1:-gvp_train}; pat=${2:-'gvp_chain|wgrad|node_chain'}
root=$(cd "$(dirname "$0")/../.." && pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/train_pmc_$w
rm -rf $out; mkdir -p $out
args="--workload $w --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py $args > $out/stats.json 2> $out/stats.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/sq -- python bench.py $args > /dev/null 2> $out/sq.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python bench.py $args > /dev/null 2> $out/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python bench.py $args > /dev/null 2> $out/write.err
KPD_OUT=$out KPD_PAT="$pat" KPD_W=$w python - <<'PY' | tee gpurun_out/train_pmc_$w.txt
import csv, glob, os, re, collections
out, pat, w = os.environ['KPD_OUT'], re.compile(os.environ['KPD_PAT']), os.environ['KPD_W']
dur = {}
for r in csv.DictReader(open(glob.glob(out + '/stats/*/*kernel_stats.csv')[0])):
    dur[r['Name'].replace('(anonymous namespace)::', '').split('(')[0]] = float(r['AverageNs']) / 1e3
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        a = agg[r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0]][r['Counter_Name']]
        a[0] += float(r['Counter_Value']); a[1] += 1
print(f'{w}: per dispatch, rocprofv3 --pmc in separate passes; FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md); duration from --stats of the same command')
for k in sorted(agg, key=lambda k: -dur.get(k, 0.0)):
    if not pat.search(k):
        continue
    m = agg[k]
    avg = lambda c: m[c][0] / m[c][1] if c in m else float('nan')
    util = 100 * avg('SQ_VALU_MFMA_BUSY_CYCLES') / (avg('GRBM_GUI_ACTIVE') / 8 * 1024)
    rd, wr, us = 2 * avg('FETCH_SIZE') * 1024, avg('WRITE_SIZE') * 1024, dur.get(k, float('nan'))
    print(f'  {k[:60]:60s} {us:8.1f} us  MfmaUtil {util:5.1f} %  HBM read {rd / 1e6:8.1f} MB  write {wr / 1e6:8.1f} MB  = {(rd + wr) / us / 1e6:5.2f} TB/s')
PY
find $out -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
