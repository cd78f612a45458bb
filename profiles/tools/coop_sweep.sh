#!/bin/bash
# Row limit of the cooperative node-side GVP kernels (gvp_coop.hip): steps/s and per-kernel times for "never", the default and "always"
# on the two GVP workloads, one GPU box, same call.  Needs the TOOLS build (KPD_COOP_ROWS is read only there).
export KPD_LIB=$PWD/keypoint-diffusion_amd/csrc/tools_build/libkpd_hip.so
for wl in "gvp_40kp" "gvp_all_atom --ragged" "gvp_all_atom"; do
  for rows in 1 8192 32768 1073741824; do
    echo -n "$wl coop_rows=$rows: "
    KPD_COOP_ROWS=$rows python bench.py --tools --workload $wl --no-secondary --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | python profiles/tools/bench_brief.py
  done
done
