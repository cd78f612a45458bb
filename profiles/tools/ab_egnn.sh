python bench.py --tools --no-secondary --no-cpu-baseline --steps 150 --warmup 15 2>/dev/null | python profiles/tools/bench_brief.py
python bench.py --tools --workload egnn_train --no-cpu-baseline 2>/dev/null | python profiles/tools/bench_brief.py
