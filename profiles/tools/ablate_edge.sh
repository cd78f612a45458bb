for m in 0 1 2 4 8 16 32 9 63; do
  echo -n "ablate=$m: "; KPD_EDGE_ABLATE=$m timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), round(d['roofline']['avg_launch_ms'],3))"
done
