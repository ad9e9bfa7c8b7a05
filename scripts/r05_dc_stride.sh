#!/bin/bash
# What the memory system charges for touching the 2-byte DC of every 128-byte block (scripts/ubench/dc_stride.hip): rates of every
# variant, then how the requests leave L2 for a few of them (rocprofv3 --pmc, separate passes of at most four TCC counters).
# usage (GPU box): scripts/r05_dc_stride.sh  -> gpurun_out/r05/dc_stride/
set -e
cd "$(dirname "$0")/.."
out=$PWD/gpurun_out/r05/dc_stride
mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/ubench/dc_stride.hip -o $out/dc_stride
$out/dc_stride 8 | tee $out/rates.txt
export TMPDIR=/tmp
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_EA0_[A-Z0-9_]*\|TCC_[A-Z_]*REQ[A-Z0-9_]*\|TCP_[A-Z_]*TCC[A-Z_]*" | sort -u > $out/tcc_counters_available.txt || true
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | tr ' ' '+')
  for v in rd2_plain wr2_plain rmw2_plain rmw2_nt rmw2_system rd16_plain rd64_plain rd128_plain wr128_plain rmw128_plain dense2; do
    rm -rf /tmp/dcp
    rocprofv3 --pmc $set --kernel-trace -d /tmp/dcp -o p --output-format csv -- $out/dc_stride 2 $v > /dev/null 2>&1 || { echo "$tag $v: rocprofv3 failed" >> $out/pmc.txt; continue; }
    f=$(find /tmp/dcp -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 - "$f" "$v" "$tag" >> $out/pmc.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    k = r.get("Kernel_Name", "")
    if "head_kernel" in k or "wide_kernel" in k or "dense2" in k:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
nb = 2 * (1 << 30) // 128
print(sys.argv[2], " ".join("%s=%.3f/block" % (c, sum(v[-3:]) / len(v[-3:]) / nb) for c, v in sorted(acc.items())))
PY
  done
done
cat $out/pmc.txt
