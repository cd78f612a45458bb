"""Does the CPU baseline get steadier below the cgroup CPU quota?  `bench.cpu_baseline` at 16 / 14 / 12 threads on a one-GPU box (quota 16 CPUs):
round 4 read B = 8 rates of 14.6 / 13.9 (16 threads, spread 30 - 46 %), 11.6 / 12.1 (14, 23 %), 13.0 (12, 19 %) complex-steps/s -- the spread is
the shared host, not oversubscription; the bench keeps the full quota."""
import sys, os, json
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, bench
orig = bench.host_cpu
for thr in (16, 14, 12, 16, 14):
    def hc(thr=thr):
        d = orig(); d['threads_used'] = thr; return d
    bench.host_cpu = hc
    r = bench.cpu_baseline('egnn_all_atom')
    print(thr, 'threads: B1', round(r['cases']['B1']['complex_steps_per_s'], 2), 'spread', round(r['cases']['B1']['spread_pct'], 1), '| B8', round(r['cases']['B8']['complex_steps_per_s'], 2), 'spread', round(r['cases']['B8']['spread_pct'], 1), flush=True)
