# (the switches below exist only in the TOOLS build of the library: `make -C keypoint-diffusion_amd/csrc tools` first; bench.py --tools marks the line as a diagnostic run)
export KPD_LIB=${KPD_LIB:-$PWD/keypoint-diffusion_amd/csrc/tools_build/libkpd_hip.so}
root=$(cd "$(dirname "$0")/../.." && pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
for pad in 0 20000 50000 100000; do
  rm -rf gpurun_out/occ; KPD_NODE_LDS_PAD=$pad timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/occ -- python bench.py --tools --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  echo -n "pad=$pad: "; python - <<PY
import csv,glob
f=glob.glob("gpurun_out/occ/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if 'node_update8' in r['Name']: print(r['Calls'], 'avg us', round(float(r['AverageNs'])/1e3,1), 'min', r['MinNs'], 'max', r['MaxNs'])
PY
done
