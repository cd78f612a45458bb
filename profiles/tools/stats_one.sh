#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench.py workload (run on the GPU box from the repo root):
#   bash profiles/tools/stats_one.sh <tag> [bench.py args...]     -> gpurun_out/<tag>_kernel_stats.csv
tag=${1:?tag}; shift
root=$(pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
rm -rf gpurun_out/_stats_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_stats_$tag -- python bench.py --steps 40 --warmup 5 --repeats 1 --no-cpu-baseline --no-secondary "$@" > gpurun_out/${tag}_under_rocprof.json 2> gpurun_out/${tag}_rocprof.err
cp $(ls gpurun_out/_stats_$tag/*/*kernel_stats.csv | head -1) gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/_stats_$tag
python - "$tag" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(f'gpurun_out/{sys.argv[1]}_kernel_stats.csv')))
for r in rows[:14]:
    n = r['Name'].split('(')[0].replace('void ', '').replace('kpd::', '').replace('(anonymous namespace)::', '')
    print(f"{n[:52]:52s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:8.1f} us  {r['Percentage']:>6s} %")
PY
