#!/bin/bash
# SQ counter pass for a GVP bench workload (run on the GPU box from the repo root).
set -e
wl=${1:-gvp_all_atom}
root=$(cd "$(dirname "$0")/../.." && pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
out=gpurun_out/pmc_$wl
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES \
  --output-format csv -d $out/sq -- python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > $out/sq.json 2> $out/sq.err
python profiles/summarize_pmc.py $out > $out/summary.txt
cat $out/summary.txt
