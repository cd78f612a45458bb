#!/bin/bash
# timing sensitivity of the training chain kernel to its stores (TOOLS build)
cd /tmp && export TMPDIR=/tmp && cd /root/repo
export KPD_LIB=/root/repo/keypoint-diffusion_amd/csrc/tools_build/libkpd_hip.so
for sk in 0 1 2 3 4 7; do
  out=gpurun_out/trskip_$sk; rm -rf $out; mkdir -p $out
  KPD_TR_SKIP=$sk rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python bench.py --workload gvp_train --steps 2 --warmup 1 --no-cpu-baseline --tools > $out/bench.json 2> $out/err.txt
  f=$(ls $out/*/*kernel_stats.csv | head -1)
  echo "skip=$sk $(grep 'k_gvp_chain<16, 0, 1>' $f | awk -F, '{print $4}') ns/call fwd; $(grep 'k_gvp_chain_bwd' $f | awk -F, '{print $4}') bwd"
done
