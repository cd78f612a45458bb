#!/bin/bash
# rocprofv3 kernel statistics of ONE training workload, printed as ms per step (run on the GPU box from the repo root):
#   bash profiles/tools/train_stats_one.sh <workload> [top]    -> gpurun_out/stats_<workload>.csv + a table on stdout
set -e
w=${1:-egnn_train}; top=${2:-28}
root=$(cd "$(dirname "$0")/../.." && pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/stats_one_$w
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $out/bench.json 2> $out/err.txt
cp $(ls $out/*/*kernel_stats.csv | head -1) gpurun_out/stats_$w.csv
python - "$w" "$top" <<'PY'
import csv, sys
w, top = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(f'gpurun_out/stats_{w}.csv')))
steps = 10.0        # 3 repeats x 3 steps + 1 warmup
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'{w}: kernel time {tot / 1e6 / steps:.2f} ms/step')
for r in rows[:top]:
    n = int(r['Calls'])
    print('  %-70s %5.1f calls/step %8.3f ms/step %7.1f us/call' % (r['Name'].replace('kpd::(anonymous namespace)::', '')[:70], n / steps,
          float(r['TotalDurationNs']) / 1e6 / steps, float(r['TotalDurationNs']) / 1e3 / n))
PY
