for cfg in "KPD_NODE_FUSED=1" "KPD_NODE_FUSED=0 KPD_PROJ_SPB=1" "KPD_NODE_FUSED=0 KPD_PROJ_SPB=2" "KPD_NODE_FUSED=0 KPD_PROJ_SPB=4" "KPD_NODE_FUSED=0 KPD_PROJ_SPB=8"; do
  echo -n "$cfg: "; env $cfg timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['ms_per_step'],3))"
done
