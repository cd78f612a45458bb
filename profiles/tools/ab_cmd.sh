#!/bin/bash
# Same-call A/B of two builds of one translation unit under an arbitrary command (run HERE, it calls gpurun itself):
#   (flags "@HEAD": A = the whole library as committed -- HEAD exported and built in /tmp --, B = the working tree's library; the
#    translation-unit argument is then only a label)
#   bash profiles/tools/ab_cmd.sh gvp_chain.hip "-DKPD_CHAIN_NBUF=4" "python bench.py --workload gvp_40kp --no-secondary --no-cpu-baseline --steps 100 --warmup 10" [reps]
# A = as committed, B = with the extra compile flags.  Both libraries go to ONE GPU box as libkpd_ab_{A,B}.so and are selected through
# KPD_LIB (hip.py); bench.py needs --tools then (KPD_LIB is a refused variable otherwise).  Output: gpurun_out/ab_cmd.log
set -e
tu=${1:?translation unit}; flags=${2:?extra compile flags of the B build}; cmd=${3:?command}; reps=${4:-2}
root=$(cd "$(dirname "$0")/../.." && pwd); csrc=$root/keypoint-diffusion_amd/csrc
make -C $csrc -j8 libkpd_hip.so > /dev/null
if [ "$flags" = "@HEAD" ]; then
  # A = the whole library as committed (a clean export of HEAD built in /tmp), B = the working tree's library
  rm -rf /tmp/ab_head && mkdir -p /tmp/ab_head && git -C $root archive HEAD keypoint-diffusion_amd/csrc include | tar -x -C /tmp/ab_head
  make -C /tmp/ab_head/keypoint-diffusion_amd/csrc -j8 libkpd_hip.so > /dev/null
  cp /tmp/ab_head/keypoint-diffusion_amd/csrc/libkpd_hip.so $root/libkpd_ab_A.so
  cp $csrc/libkpd_hip.so $root/libkpd_ab_B.so
else
objs=$(sed -n 's/^SRCS = //p' $csrc/Makefile | sed 's/\.hip/.o/g')
for v in A B; do
  f=""; [ $v = B ] && f="$flags"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $f -c $csrc/$tu -o /tmp/ab_$v.o
  list=""; for o in $objs; do if [ $o = ${tu%.hip}.o ]; then list="$list /tmp/ab_$v.o"; else list="$list $csrc/$o"; fi; done
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/libkpd_ab_$v.so $list -lpthread
done
fi
cd $root
/usr/local/graft/bin/gpurun --timeout 900 -- "( for rep in \$(seq $reps); do for v in A B; do echo \"== \$v (rep \$rep)\"; KPD_LIB=\$PWD/libkpd_ab_\$v.so $cmd 2>&1 | grep -v amdgpu.ids; done; done ) > gpurun_out/ab_cmd.log 2>&1; tail -60 gpurun_out/ab_cmd.log"
rm -f $root/libkpd_ab_A.so $root/libkpd_ab_B.so
