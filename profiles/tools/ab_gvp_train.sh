#!/bin/bash
# Same-call A/B of the GVP trainer's message / node path (TOOLS build; run on the GPU box from the repo root):
#   A = KPD_TRAIN_FUSED=0: one GVP at a time through the GEMM kernels (the path of rounds 3 - 4, still taken at hidden widths other than 256)
#   B = KPD_TRAIN_FUSED=1: the register-chained kernels + batched weight gradients (the default)
root=$(cd "$(dirname "$0")/../.." && pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
export KPD_LIB=$PWD/keypoint-diffusion_amd/csrc/tools_build/libkpd_hip.so
for rep in 1 2; do
  for f in 0 1; do
    for w in gvp_train gvp_40kp_train; do
      KPD_TRAIN_FUSED=$f python bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline --tools 2>/dev/null |
        python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('rep $rep  KPD_TRAIN_FUSED=$f  %-15s %7.3f ms/step  %6.3f steps/s  frac %.3f' % ('$w', d['ms_per_step'], d['value'], d['roofline']['frac']))"
    done
  done
done
