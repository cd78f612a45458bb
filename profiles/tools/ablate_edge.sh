#!/bin/bash
# What each part of k_egnn_edge<4> costs: KPD_EDGE_ABLATE bits 1 = no GEMMs, 2 = no A-build (gather + first-layer SiLU), 4 = no
# epilogues (T-store, dots, segmented sums); 7 = tile prologue + barriers only.  Outputs are wrong by construction: timing only.
# (the switches below exist only in the TOOLS build of the library: `make -C keypoint-diffusion_amd/csrc tools` first; bench.py --tools marks the line as a diagnostic run)
export KPD_LIB=${KPD_LIB:-$PWD/keypoint-diffusion_amd/csrc/tools_build/libkpd_hip.so}
for m in 0 6 1 2 4 5 3 7; do
  echo -n "ablate=$m: "; KPD_EDGE_ABLATE=$m timeout -k 10 300 python bench.py --tools --steps 30 --warmup 5 --repeats 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step', round(d['ms_per_step'],3), ' edge kernel avg ms', round(d['roofline']['avg_launch_ms'],4))"
done
echo -n "one workgroup per CU (KPD_EDGE_LDS_PAD=24000), ablate=0: "; KPD_EDGE_LDS_PAD=24000 python bench.py --tools --steps 30 --warmup 5 --repeats 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step', round(d['ms_per_step'],3), ' edge kernel avg ms', round(d['roofline']['avg_launch_ms'],4))"
