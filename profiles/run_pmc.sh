#!/bin/bash
# PMC passes for the bench workload (run on the GPU box from the repo root):
#   bash profiles/run_pmc.sh <tag>
# Separate rocprofv3 runs per counter group (TCC slots: FETCH_SIZE and WRITE_SIZE cannot share a pass).
set -e
tag=${1:-pmc}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p $out
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --output-format csv -d $out/sq -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/sq.json 2> $out/sq.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/fetch.json 2> $out/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/write.json 2> $out/write.err
python profiles/summarize_pmc.py $out > $out/summary.txt
cat $out/summary.txt
