rocm-smi --showuniqueid --showmaxpower --showpower --showtemp 2>/dev/null | grep -E "Unique|Max Graphics|Power|Temperature \(Sensor junction" | head -8
python bench.py --no-secondary --gemm f16x2 --steps 100 --warmup 10 --no-cpu-baseline | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('f16x2', round(b['value'],1), round(b['roofline']['avg_launch_ms'],4))"
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | head -4
