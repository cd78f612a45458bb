#!/bin/bash
# rocprofv3 kernel statistics + plain bench lines of the four training workloads (run on the GPU box from the repo root):
#   bash profiles/tools/train_profile.sh <tag>     -> gpurun_out/<tag>_train/{<workload>_bench.json, <workload>_kernel_stats.csv}
set -e
tag=${1:-r03}
root=$(cd "$(dirname "$0")/../.." && pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/${tag}_train
mkdir -p $out
for w in egnn_train gvp_train egnn_40kp_train gvp_40kp_train; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$w -- python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $out/${w}_bench_under_rocprof.json 2> $out/${w}_stats.err
  cp $(ls $out/stats_$w/*/*kernel_stats.csv | head -1) $out/${w}_kernel_stats.csv
  rm -rf $out/stats_$w
  python bench.py --workload $w --steps 8 --warmup 2 > $out/${w}_bench.json 2> $out/${w}_bench.err
  echo "== $w done"
done
