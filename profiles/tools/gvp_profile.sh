#!/bin/bash
# Kernel-trace stats of the two GVP bench workloads (run from the repo root on the GPU box).
set -e
tag=${1:-r01_gvp}
root=$(cd "$(dirname "$0")/../.." && pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/$tag
rm -rf $out && mkdir -p $out
for wl in gvp_all_atom gvp_40kp; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$wl -- python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline > $out/$wl.json 2> $out/$wl.err
  f=$(ls $out/$wl/*/*kernel_stats.csv)
  cp $f $out/${wl}_kernel_stats.csv
  cut -c1-150 $f | sed -n 1,14p
  cat $out/$wl.json
done
