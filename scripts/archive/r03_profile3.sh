#!/bin/bash
# round-3 profiles, third pass: SQ instruction counters of the final kernels (A2: compose_kernel; A1: compose_light_kernel)
set -o pipefail
R=$PWD; O=$R/gpurun_out/r03prof3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
SQ2="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for RG in A2 A1 A5; do
  rocprofv3 --pmc $SQ --output-format csv -d $O/sq_$RG -o p -- python3 $R/bench.py --cpu-sample 0 --regime $RG --steps 10 > $O/sq_$RG.json 2> $O/sq_$RG.err || echo "sq_$RG FAILED"
  rocprofv3 --pmc $SQ2 --output-format csv -d $O/sq2_$RG -o p -- python3 $R/bench.py --cpu-sample 0 --regime $RG --steps 10 > $O/sq2_$RG.json 2> $O/sq2_$RG.err || echo "sq2_$RG FAILED"
  echo "done $RG"
done
cd $R
python3 - <<'PY'
import csv, glob, collections, os
O = "gpurun_out/r03prof3"
for tag in sorted(os.listdir(O)):
    if not os.path.isdir(f"{O}/{tag}"): continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{O}/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for kn in ("compose_kernel", "compose_light_kernel"):
                if kn + "<" in r["Kernel_Name"]: acc[kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kn, cs in acc.items():
        print("==", tag, kn, " ".join("%s=%.6g" % (c, sum(v[-20:]) / len(v[-20:])) for c, v in sorted(cs.items())))
PY
