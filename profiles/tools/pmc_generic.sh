#!/bin/bash
# usage: pmc_generic.sh <tag> <counters...>   -- one PMC pass of the bench workload (contract line only, short)
# extra bench flags via $KPD_PMC_BENCH_ARGS; kernels reported: $KPD_PMC_KERNELS (regex, default egnn_edge|node_|proj)
root=$(cd "$(dirname "$0")/../.." && pwd); cd /tmp && export TMPDIR=/tmp && cd "$root"
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/p -- python bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-secondary $KPD_PMC_BENCH_ARGS > $out/p.json 2> $out/p.err
KPD_PMC_OUT=$out python - <<'PY'
import csv, glob, collections, os, re
out = os.environ['KPD_PMC_OUT']
pat = re.compile(os.environ.get('KPD_PMC_KERNELS', 'egnn_edge|node_|proj'))
agg=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0]))
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][:40]; a=agg[k][r['Counter_Name']]; a[0]+=float(r['Counter_Value']); a[1]+=1
for k in agg:
    if pat.search(k):
        print(k)
        for c,(t,n) in sorted(agg[k].items()): print(f'   {c:36s} {t/n:16.0f}  (n={n})')
PY
