#!/bin/bash
# VALU instructions by opcode class of the dense launch (A3), scalar build against the packed-pair build (scripts/r05_pk.sh)
# usage (GPU box): scripts/r05_pk_counters.sh -> gpurun_out/r05/pk_counters.txt
R=$PWD; O=$R/gpurun_out/r05/pkc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32"
for v in pk0 pk1; do
  i=0
  for set in "$SQ1" "$SQ2"; do
    i=$((i+1))
    MJX_HIP_LIB=$R/gpurun_variants/libmjx_$v.so rocprofv3 --pmc $set --output-format csv -d $O/${v}_$i -o p -- python3 $R/bench.py --regime A3 --steps 5 --warmup 1 --spinup-ms 50 --cpu-sample 0 --pristine-launches 0 --also none > $O/${v}_$i.json 2> $O/${v}_$i.err || echo "$v $i FAILED"
  done
done
cd $R
python3 - <<'PY' | tee gpurun_out/r05/pk_counters.txt
import csv, glob, collections
O = "gpurun_out/r05/pkc"
for v in ("pk0", "pk1"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{O}/{v}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "compose_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(v, "(A3, mjx::compose_kernel, per octet = counter / 786432, mean of the last 5 launches)")
    for k in sorted(acc):
        vals = acc[k][-5:]
        print("   %-28s %12.1f per octet" % (k, sum(vals) / len(vals) / 786432))
PY
