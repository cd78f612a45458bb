#!/bin/bash
# ring depth of the tiled training GEMM on the egnn_train step (TOOLS build, KPD_SGEMM_STAGES): total time of its kernels per step
root=$(cd "$(dirname "$0")/../.." && pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
export KPD_LIB=$root/keypoint-diffusion_amd/csrc/tools_build/libkpd_hip.so
for st in 3 4 5 6; do
  out=gpurun_out/sgemm_stages_$st; rm -rf $out; mkdir -p $out
  KPD_SGEMM_STAGES=$st rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python bench.py --workload egnn_train --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --tools > $out/bench.json 2> $out/err.txt
  f=$(ls $out/*/*kernel_stats.csv | head -1)
  python - "$f" "$st" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 3.0
tot = sum(float(r['TotalDurationNs']) for r in rows if 'k_sgemm<' in r['Name']) / 1e6 / steps
big = {r['Name'].split('k_sgemm')[1][:22]: round(float(r['TotalDurationNs']) / 1e6 / steps, 3) for r in rows if 'k_sgemm<' in r['Name'] and float(r['TotalDurationNs']) / 1e6 / steps > 0.3}
print(f'stages={sys.argv[2]}: k_sgemm<...> total {tot:.3f} ms/step', big)
PY
done
