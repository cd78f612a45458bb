#!/bin/bash
# phase breakdown of k_egnn_edge_train / k_egnn_edge_bwd by s_memtime stamps (TOOLS build, run on the GPU box from the repo root):
#   bash profiles/tools/train_stamps.sh        -> table on stdout
root=$(cd "$(dirname "$0")/../.." && pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
export KPD_LIB=$root/keypoint-diffusion_amd/csrc/tools_build/libkpd_hip.so
KPD_TRAIN_STAMPS=1 python bench.py --workload egnn_train --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --tools 2>&1 | grep -v "^{" | grep -v amdgpu
