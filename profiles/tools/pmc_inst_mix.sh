#!/bin/bash
# Instruction mix of the bench workload's kernels (separate PMC pass, kernel-trace only)
root=$(cd "$(dirname "$0")/../.." && pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/${1:-pmc_mix}
mkdir -p $out
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_FLAT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES \
  --output-format csv -d $out/mix -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/mix.json 2> $out/mix.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FLOPS_FP32_TRANS SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA \
  --output-format csv -d $out/mix2 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/mix2.json 2> $out/mix2.err
python profiles/summarize_pmc.py $out | grep -A22 "k_egnn_edge"
