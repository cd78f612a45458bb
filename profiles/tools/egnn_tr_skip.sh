#!/bin/bash
# timing sensitivity of the EGNN trainer's edge kernels to their stores (TOOLS build, KPD_TR_SKIP bits: 1 pre1 + a1, 2 pre2, 4 geometry, 8 dpre2, 16 dpre1)
root=$(cd "$(dirname "$0")/../.." && pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
export KPD_LIB=$root/keypoint-diffusion_amd/csrc/tools_build/libkpd_hip.so
for sk in 0 1 2 4 7 8 16 24; do
  out=gpurun_out/egnn_trskip_$sk; rm -rf $out; mkdir -p $out
  KPD_TR_SKIP=$sk rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python bench.py --workload egnn_train --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --tools > $out/bench.json 2> $out/err.txt
  f=$(ls $out/*/*kernel_stats.csv | head -1)
  echo "skip=$sk $(grep 'k_egnn_edge_train' $f | awk -F, '{print $(NF-4)}') ns/call fwd; $(grep 'k_egnn_edge_bwd' $f | awk -F, '{print $(NF-4)}') bwd"
done
