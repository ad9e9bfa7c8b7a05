#!/bin/bash
# round-2 kernel experiments 1: LDS pattern ubench with conflict counters; layout / get_col variants A/B
set -o pipefail
R=$PWD; O=$R/gpurun_out/r02c; mkdir -p $O
./scripts/ubench/lds_patterns > $O/lds_patterns_time.txt 2>&1; cat $O/lds_patterns_time.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $O/lds_pmc -o p -- $R/scripts/ubench/lds_patterns > /dev/null 2> $O/lds_pmc.err || echo "pmc failed"
cd $R
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/r02c/lds_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"], r["Dispatch_Id"])
        acc[k][r["Counter_Name"]] = float(r["Counter_Value"])
rows = {}
for (name, d), c in acc.items():
    if c.get("SQ_INSTS_LDS", 0) > rows.get(name, {}).get("SQ_INSTS_LDS", -1):
        rows[name] = c
with open("gpurun_out/r02c/lds_patterns_pmc.txt", "w") as out:
    for name in sorted(rows):
        c = rows[name]
        line = "%-40s insts %.3g  idx_active %.4g  conflict %.4g  conflict/active %.3f  active/inst %.2f" % (
            name[:40], c.get("SQ_INSTS_LDS", 0), c.get("SQ_LDS_IDX_ACTIVE", 0), c.get("SQ_LDS_BANK_CONFLICT", 0),
            c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1), c.get("SQ_LDS_IDX_ACTIVE", 0) / max(c.get("SQ_INSTS_LDS", 1), 1))
        print(line); out.write(line + "\n")
PY
for R_ in A1 A2 A3; do
  bash scripts/ab_bench.sh "--regime $R_ --cpu-sample 16" base split cond splitcond 2>&1 | tee -a $O/ab_$R_.txt
done
bash scripts/ab_bench.sh "--regime A1 --cpu-sample 0" noblend noblendsplit 2>&1 | tee -a $O/ab_noblend.txt
