#!/bin/bash
# hardware counters of the decoder's synchronisation rounds, one dispatch per line (lane-per-chunk kernel)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
H=$PWD/libmodjpeg_amd/host
T=$PWD/tests/tools
OUT=$PWD/gpurun_out/decpmc
mkdir -p $OUT
export MJX_DECODE_WAVE_CHUNKS=${LIM:-0} BENCH_QUALITY=${Q:-95}
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD"; do
  tag=$(echo $set | tr ' ' '_')
  (cd /tmp && rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- $T/bench_dropin $H/libmodjpeg.so.1 ${W:-3840} ${HH:-2160} 256 256 3 > $OUT/$tag.log 2>&1)
  f=$(find $OUT/$tag -name "*counter_collection.csv" | head -1)
  echo "=== $set"
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.OrderedDict()
for r in rows:
    if "huffdec" not in r["Kernel_Name"]: continue
    key = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0].replace("mjx::", ""))
    by.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
keys = sorted(by)
# the last image: the last 30 dispatches
for k in keys[-28:]:
    print(f"  {k[0]:5d} {k[1]:30s} " + "  ".join(f"{n}={v:.0f}" for n, v in by[k].items()))
PY
done
