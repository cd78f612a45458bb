#!/bin/bash
# Run a diagnostic script in fresh processes against a VARIANT build of one translation unit (run HERE; calls gpurun once):
#   bash profiles/tools/hz_variant.sh "<extra hipcc flags>" <script.py> [reps] [env assignments for the script] [translation unit]
# The production library is restored before the call ends; the log lands in gpurun_out/r03/<script>.log
set -e
flags=$1; script=$2; reps=${3:-3}; envs=${4:-KPD_GEMM=f16x2}; tu=${5:-egnn_kernels.hip}
root=$(cd "$(dirname "$0")/../.." && pwd); csrc=$root/keypoint-diffusion_amd/csrc
make -C $csrc -j8 libkpd_hip.so > /dev/null
objs=$(sed -n 's/^SRCS = //p' $csrc/Makefile | sed 's/\.hip/.o/g')
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $flags -c $csrc/$tu -o /tmp/hz_B.o
list=""; for o in $objs; do if [ $o = ${tu%.hip}.o ]; then list="$list /tmp/hz_B.o"; else list="$list $csrc/$o"; fi; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/libkpd_hz_B.so $list -lpthread
cp $csrc/libkpd_hip.so $root/libkpd_hz_A.so
name=$(basename $script .py)
cd $root
/usr/local/graft/bin/gpurun --timeout 600 -- "mkdir -p gpurun_out/r03; L=keypoint-diffusion_amd/csrc/libkpd_hip.so; cp libkpd_hz_B.so \$L; for rep in \$(seq 1 $reps); do echo \"== rep \$rep ($flags)\"; env $envs timeout -k 5 150 python $script 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r03/$name.log 2>&1; cp libkpd_hz_A.so \$L; tail -2 gpurun_out/r03/$name.log" 2>&1 | tail -3
rm -f $root/libkpd_hz_A.so $root/libkpd_hz_B.so
