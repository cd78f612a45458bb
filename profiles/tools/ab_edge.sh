# A/B of edge-kernel variants in separate processes (3 rounds each, interleaved)
for r in 1 2 3; do
  for nw in 4 8; do
    echo -n "NW=$nw: "; KPD_EDGE_NW=$nw timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],3))"
  done
done
