#!/bin/bash
# rocprofv3 kernel + memory-copy trace of the drop-in: read / compose / write of one image, every kernel and copy of the last image in order (start, duration, gap before it)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
H=$PWD/libmodjpeg_amd/host
T=$PWD/tests/tools
OUT=$PWD/gpurun_out/enctrace
rm -rf $OUT; mkdir -p $OUT
export BENCH_QUALITY=${Q:-85}
(cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/t -- $T/bench_dropin $H/libmodjpeg.so.1 ${W:-1920} ${HH:-1080} ${DW:-256} ${DH:-256} ${N:-6} > $OUT/t.log 2>&1)
python3 - $OUT <<'PY'
import csv, sys, glob
out = sys.argv[1]
k = glob.glob(out + "/t/**/*kernel_trace.csv", recursive=True)[0]
rows = [("K", r["Kernel_Name"].split("(")[0].replace("mjx::", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(k))]
m = glob.glob(out + "/t/**/*memory_copy_trace.csv", recursive=True)
if m:
    for r in csv.DictReader(open(m[0])):
        rows.append(("C", r.get("Direction", r.get("Name", "copy")), int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort(key=lambda r: r[2])
# the last image: from the last round-0 sync kernel on
idx = [i for i, r in enumerate(rows) if "huffdec_zero" in r[1]]
start = idx[-1] if idx else max(0, len(rows) - 60)
t0 = rows[start][2]
prev = t0
for kind, name, s, e in rows[start - 4:]:
    print(f"  {kind} {name[:44]:44s} +{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev) / 1e3:7.1f}")
    prev = e
PY
