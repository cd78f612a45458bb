#!/bin/bash
# rocprofv3 kernel statistics of the two training workloads (run on the GPU box from the repo root):
#   bash profiles/tools/train_profile.sh <tag>     -> gpurun_out/<tag>_{egnn,gvp}_train/
set -e
tag=${1:-r01}
for w in egnn gvp; do
  out=gpurun_out/${tag}_${w}_train
  mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py --workload ${w}_train --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/stats.err
  find $out -name "*kernel_trace.csv" -delete
  python bench.py --workload ${w}_train --steps 8 --warmup 2 > $out/bench.json 2> $out/bench.err
done
