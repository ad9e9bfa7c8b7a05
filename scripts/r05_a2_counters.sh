#!/bin/bash
# SQ instruction counters of the headline launch (A2, low-column build), per octet
R=$PWD; O=$R/gpurun_out/r05/a2c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32"
SQ3="SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SENDMSG"
i=0
for set in "$SQ1" "$SQ2" "$SQ3"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/p$i -o p -- python3 $R/bench.py --steps 5 --warmup 1 --spinup-ms 50 --cpu-sample 0 --pristine-launches 0 --also none > $O/p$i.json 2> $O/p$i.err || echo "pass $i FAILED"
done
cd $R
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r05/a2c/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "compose_lowl_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("A2, mjx::compose_lowl_kernel, per octet = counter / 786432, mean of the last 5 launches")
for k in sorted(acc):
    vals = acc[k][-5:]
    print("   %-28s %12.1f per octet" % (k, sum(vals) / len(vals) / 786432))
PY
