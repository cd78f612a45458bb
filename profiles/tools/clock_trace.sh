#!/bin/bash
# On the GPU box: build the bare-MFMA trace and run it with a background sampler of the SMI clocks / power.
set -e
out=${1:-gpurun_out/r03}; mkdir -p $out profiles/bin
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o profiles/bin/mfma_clock_trace profiles/tools/mfma_clock_trace.hip
( for i in $(seq 1 60); do date +%s.%N; /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" ; sleep 0.15; done ) > $out/clock_trace_smi.txt 2>&1 &
smi=$!
for w in 1 2; do ./profiles/bin/mfma_clock_trace $w; done > $out/clock_trace.txt 2>&1
wait $smi || true
