#!/bin/bash
# Same-call A/B of two builds of one translation unit (run HERE, it calls gpurun itself):
#   bash profiles/tools/ab_same_call.sh egnn_kernels.hip "-DKPD_SOME_SWITCH" [bench flags, e.g. "--gemm f16x2"]
# Builds the object twice (as committed / with the extra flags), links two libraries in the Makefile's object order, ships both
# to ONE GPU box and alternates them twice.  Only numbers from the same call are comparable: the GPUs of the pool differ (0.6 %
# on the exact path, 14 % on the f16x2 EGNN path).
set -e
tu=${1:?translation unit (e.g. egnn_kernels.hip)}; flags=${2:?extra compile flags of the B build}; bflags=${3:-}
root=$(cd "$(dirname "$0")/../.." && pwd); csrc=$root/keypoint-diffusion_amd/csrc
make -C $csrc -j8 libkpd_hip.so > /dev/null
objs=$(sed -n 's/^SRCS = //p' $csrc/Makefile | sed 's/\.hip/.o/g')
for v in A B; do
  f=""; [ $v = B ] && f="$flags"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $f -c $csrc/$tu -o /tmp/ab_$v.o
  list=""; for o in $objs; do if [ $o = ${tu%.hip}.o ]; then list="$list /tmp/ab_$v.o"; else list="$list $csrc/$o"; fi; done
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/libkpd_ab_$v.so $list -lpthread
done
cd $root
/usr/local/graft/bin/gpurun --timeout 900 -- "for rep in 1 2; do for v in A B; do cp libkpd_ab_\$v.so keypoint-diffusion_amd/csrc/libkpd_hip.so; echo -n \"\$v: \"; python bench.py --no-secondary --steps 150 --warmup 10 --no-cpu-baseline $bflags | python -c 'import json,sys; b=json.loads(sys.stdin.read()); print(round(b[\"value\"],2), \"steps/s  dominant kernel\", round(b[\"roofline\"][\"avg_launch_ms\"],4), \"ms\")'; done; done"
rm -f $root/libkpd_ab_A.so $root/libkpd_ab_B.so
