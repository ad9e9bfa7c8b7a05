#!/bin/bash
# round 4: co-run variants, kernel trace of the last steps (see r04_corun_trace.sh)
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04_corun2; mkdir -p $O
regime=${1:-A5}
run() { # tag, env...
  tag=$1; shift
  ( export "$@" DUMMY=1; rocprofv3 --kernel-trace --output-format csv -d $O/$tag -o p -- python3 $R/bench.py --regime $regime $BENCH_ARGS --steps 6 --warmup 2 --cpu-sample 2 --pristine-launches 0 > $O/$tag.json 2> $O/$tag.err ) || echo "$tag FAILED"
  python3 - $O/$tag $O/$tag.json <<'PY'
import csv, glob, sys, json
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "compose" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-3:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("   %-34s start %8.1f us  dur %7.1f us  end %8.1f us  vgpr %s" % (r["Kernel_Name"][10:44], (s - t0) / 1e3, (e - s) / 1e3, (e - t0) / 1e3, r.get("VGPR_Count", "?")))
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("   step %.4f ms  parity %s" % (d["ms_per_step"], d["parity"]))
except Exception as ex:
    print("   no json:", ex)
PY
}
echo "== serial"; run serial MJX_CORUN=0
for split in 0.6 0.7 0.8 0.9; do
echo "== streamer takes $split of the light runs"; run s_${split} MJX_CORUN=1 MJX_BLEND_LEAN=1 MJX_LIGHT_SHARED=1 MJX_LIGHT_SPLIT=$split
done
echo "== i16 tiles"; BENCH_ARGS="--tiles i16"
run serial_i16 MJX_CORUN=0
for split in 0.7 0.8 0.9; do
echo "== streamer takes $split"; run si_$split MJX_CORUN=1 MJX_BLEND_LEAN=1 MJX_LIGHT_SHARED=1 MJX_LIGHT_SPLIT=$split
done
