#!/bin/bash
# A/B of the node-side orchestration (run on the GPU box; see egnn.hip KPD_NODE_MODE)
for m in split fused; do
  echo "KPD_NODE_MODE=$m"
  KPD_NODE_MODE=$m python bench.py --steps 20 --warmup 3 --no-cpu-baseline | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
done
