python profiles/tools/sgemm_bench.py 2>&1 | grep -v amdgpu | awk '{print $1,$2,$3,"own",$10,$11,$12,$13}' | head -17
for wl in egnn_train gvp_train; do python bench.py --tools --workload $wl --no-cpu-baseline 2>/dev/null | python profiles/tools/bench_brief.py; done
