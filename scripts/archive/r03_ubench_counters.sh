#!/bin/bash
# counters of the run-length effect: the three-streams-in-place kernel of copy_roof.hip
# with runs of 1 and 16 octets per wave, under rocprofv3 --pmc (separate passes), dispatches told apart by grid size
set -o pipefail
R=$PWD; O=$R/gpurun_out/r03ubc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TA_TCP_STATE_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_TAG_STALL_sum" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-include-regex "run_kernel" --output-format csv -d $O/$tag -o p -- $R/scripts/ubench/copy_roof 2 > $O/$tag.out 2> $O/$tag.err || echo "$tag FAILED"
  echo "done $tag"
done
cd $R
python3 - <<'PY'
import csv, glob, collections
O = "gpurun_out/r03ubc"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{O}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "run_kernel" not in r["Kernel_Name"]: continue
        key = ("strided" if "Lb1" in r["Kernel_Name"] else "blocked", int(r["Grid_Size"]))
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
noct = 786432
print("three streams in place, nt/nt, 786432 octets; R = octets per wave = octets / (grid / 64 lanes)")
for key in sorted(acc):
    kind, grid = key
    R = round(noct / (grid / 64))
    if kind != "blocked" or R not in (1, 4, 16, 32): continue
    print(f"== {kind} R={R} (grid {grid})")
    for c in sorted(acc[key]):
        v = acc[key][c]
        print("   %-32s mean %.6g (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
