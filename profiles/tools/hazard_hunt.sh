#!/bin/bash
# Build the f16x2 edge kernel WITH the batched distance read (the arrangement that showed the first-forward deviation in round 2),
# ship it next to the production library and run the detectors on both in ONE call.  Run HERE (calls gpurun).
set -e
root=$(cd "$(dirname "$0")/../.." && pwd); csrc=$root/keypoint-diffusion_amd/csrc
make -C $csrc -j8 libkpd_hip.so > /dev/null
objs=$(sed -n 's/^SRCS = //p' $csrc/Makefile | sed 's/\.hip/.o/g')
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DKPD_H_BATCH_D=true -c $csrc/egnn_kernels.hip -o /tmp/hz_B.o
list=""; for o in $objs; do if [ $o = egnn_kernels.o ]; then list="$list /tmp/hz_B.o"; else list="$list $csrc/$o"; fi; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/libkpd_hz_B.so $list -lpthread
cp $csrc/libkpd_hip.so $root/libkpd_hz_A.so
cd $root
/usr/local/graft/bin/gpurun --timeout 900 -- "mkdir -p gpurun_out/r03; L=keypoint-diffusion_amd/csrc/libkpd_hip.so; ( for v in B A; do cp libkpd_hz_\$v.so \$L; for mode in f16x2 f32; do for rep in 1 2 3; do echo \"== lib \$v gemm \$mode rep \$rep: plain\"; KPD_GEMM=\$mode timeout -k 5 200 python profiles/tools/poison_check.py; echo \"== lib \$v gemm \$mode rep \$rep: flush\"; KPD_RUNS=8 KPD_FLUSH=1 KPD_GEMM=\$mode timeout -k 5 200 python profiles/tools/poison_check.py; echo \"== lib \$v gemm \$mode rep \$rep: poison\"; KPD_POISON=1 KPD_GEMM=\$mode timeout -k 5 200 python profiles/tools/poison_check.py; done; done; done; cp libkpd_hz_A.so \$L ) > gpurun_out/r03/hazard_hunt.log 2>&1; tail -5 gpurun_out/r03/hazard_hunt.log"
rm -f $root/libkpd_hz_A.so $root/libkpd_hz_B.so
