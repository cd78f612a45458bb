python bench.py --tools --no-secondary --no-cpu-baseline --steps 150 --warmup 15 2>/dev/null | python profiles/tools/bench_brief.py
for wl in "gvp_40kp" "gvp_all_atom --ragged"; do python bench.py --tools --workload $wl --no-secondary --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | python profiles/tools/bench_brief.py; done
