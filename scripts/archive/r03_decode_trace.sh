#!/bin/bash
# per-kernel times of one GPU read (rocprofv3 kernel trace), lane-per-chunk and wave-per-chunk rounds
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
H=$PWD/libmodjpeg_amd/host
T=$PWD/tests/tools
OUT=$PWD/gpurun_out/dectrace
mkdir -p $OUT
Q=${Q:-85}
for lim in 0 100000000; do
  export MJX_DECODE_WAVE_CHUNKS=$lim BENCH_QUALITY=$Q
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/w$lim -- $T/bench_dropin $H/libmodjpeg.so.1 ${W:-1920} ${HH:-1080} 256 256 6 > $OUT/w$lim.log 2>&1)
  f=$(find $OUT/w$lim -name "*kernel_trace.csv" | head -1)
  echo "=== MJX_DECODE_WAVE_CHUNKS=$lim  ($f)"
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last image's decode: from the last huffdec_sync round 0 on
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "huffdec_sync" in n]
# find the start of the last run of sync kernels
last = idx[-1]
start = last
while start - 1 in idx: start -= 1
t0 = int(rows[start]["Start_Timestamp"])
prev_end = t0
for r in rows[start:start + 40]:
    n = r["Kernel_Name"].split("(")[0].replace("mjx::", "")
    if "huff" not in n: break
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"  {n:34s} start +{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f} us  gap {(s - prev_end) / 1e3:6.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}")
    prev_end = e
PY
done
