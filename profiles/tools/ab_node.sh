for r in 1 2; do
  for f in 0 1; do
    echo -n "NODE_FUSED=$f: "; KPD_NODE_FUSED=$f timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],4))"
  done
done
