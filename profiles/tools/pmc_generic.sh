#!/bin/bash
# usage: pmc_generic.sh <tag> <counters...>   -- one PMC pass of the bench workload
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/p.json 2> $out/p.err
python - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0]))
for f in glob.glob('$out/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:34]; a=agg[k][r['Counter_Name']]; a[0]+=float(r['Counter_Value']); a[1]+=1
for k in agg:
    if 'egnn_edge' in k or 'node_' in k:
        print(k)
        for c,(t,n) in sorted(agg[k].items()): print(f'   {c:36s} {t/n:16.0f}  (n={n})')
PY
