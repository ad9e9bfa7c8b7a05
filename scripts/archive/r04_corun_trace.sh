#!/bin/bash
# round 4: when do the two kernels of a step run?  rocprofv3 kernel trace of a short A5 bench, serial / co-run / co-run with
# the blend kernel capped at two workgroups per CU; prints start offset and duration of both kernels for the last steps
# usage (GPU box): scripts/r04_corun_trace.sh [regime]  -> gpurun_out/r04_corun_trace/
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04_corun_trace; mkdir -p $O
regime=${1:-A5}
run() { # tag, env...
  tag=$1; shift
  env "$@" true
  ( export "$@" DUMMY=1; rocprofv3 --kernel-trace --output-format csv -d $O/$tag -o p -- python3 $R/bench.py --regime $regime $BENCH_ARGS --steps 6 --warmup 2 --cpu-sample 0 --pristine-launches 0 > $O/$tag.json 2> $O/$tag.err ) || echo "$tag FAILED"
  python3 - $O/$tag <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "compose" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-8:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("   %-28s queue %-3s start %8.1f us  dur %7.1f us  end %8.1f us  lds %s vgpr %s" % (r["Kernel_Name"][:28], r.get("Queue_Id", "?"), (s - t0) / 1e3, (e - s) / 1e3, (e - t0) / 1e3, r.get("LDS_Block_Size", "?"), r.get("VGPR_Count", "?")))
PY
}
echo "== serial"; run serial MJX_CORUN=0
echo "== corun"; run corun MJX_CORUN=1
echo "== corun, blend capped (pad 4096)"; run corun_pad MJX_CORUN=1 MJX_BLEND_LDS_PAD=4096
echo "== serial, blend capped (pad 4096)"; run serial_pad MJX_CORUN=2 MJX_BLEND_LDS_PAD=4096
