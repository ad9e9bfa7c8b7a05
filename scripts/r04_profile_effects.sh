#!/bin/bash
# round 4: HBM traffic of the stand-alone DC-effect pass (BASELINE configs[4]: 4096 x 4K resident, luminance then tint), separate
# --pmc passes, and its kernel stats
set -o pipefail
R=$PWD; O=$R/gpurun_out/r04prof_fx; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for pass in "stats --kernel-trace --stats" "fetch --pmc FETCH_SIZE" "write --pmc WRITE_SIZE"; do
  set -- $pass; tag=$1; shift
  rocprofv3 "$@" --output-format csv -d $O/$tag -o p -- python3 $R/bench.py --workload effects_4k_batch --steps 10 --cpu-sample 0 > $O/$tag.json 2> $O/$tag.err || echo "$tag FAILED"
  echo "done $tag"
done
cd $R
python3 - <<'PY'
import csv, glob, collections, json
O = "gpurun_out/r04prof_fx"
for f in glob.glob(f"{O}/stats/**/*kernel_stats.csv", recursive=True):
    for line in open(f).read().splitlines()[:4]: print("  ", line)
try:
    d = json.loads(open(f"{O}/stats.json").read().strip().splitlines()[-1])
    print("   bench.py:", d["ms_per_step"], "ms per step,", d["value"], d["unit"], d["parity"])
except Exception as e:
    print("bench line unreadable", e)
tot = {}
for tag in ("fetch", "write"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{O}/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "effect_dc" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        n = min(len(v), 20)
        print("   %-12s mean of last %d effect_dc dispatches: %.6g KiB (n=%d)" % (k, n, sum(v[-n:]) / n, len(v)))
        tot[k] = sum(v[-n:]) / n
if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
    print(json.dumps({"effects_4k_batch:4096": {"hbm_bytes_per_launch": int((tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024), "fetch_size_kib": tot["FETCH_SIZE"], "write_size_kib": tot["WRITE_SIZE"], "round": 4}}))
PY
