#!/bin/bash
# A/B of the edge kernel builds (run on the GPU box): 8 waves (default) vs 4 waves with prefetched coordinate-branch gathers
for nw in 8 4; do
  echo "KPD_EDGE_NW=$nw"
  KPD_EDGE_NW=$nw python bench.py --steps 30 --warmup 3 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done
