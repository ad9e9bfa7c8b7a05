#!/bin/bash
# scripts/pmc.sh <tag> "<bench args>" "<counters pass1>" ["<counters pass2>" ...]  (GPU box)
# counters-only rocprofv3 passes restricted to the compose kernel; prints per-counter means
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; args=$2; shift 2
i=0
for ctr in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-include-regex compose_kernel --output-format csv -d gpurun_out/pmc_$tag -o p$i -- python3 bench.py $args --cpu-sample 0 > /dev/null 2> gpurun_out/pmc_$tag.err || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/pmc_$tag/p*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print("%-28s mean %.4g  (n=%d)" % (k, sum(v)/len(v), len(v)))
PY
