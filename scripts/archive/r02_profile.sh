#!/bin/bash
# round-2 profiles (GPU box): rocprofv3 kernel stats of the default bench command, HBM traffic (FETCH_SIZE /
# WRITE_SIZE in separate --pmc passes), SQ counters (LDS conflicts, VALU by opcode class) per alpha regime,
# and the same for the effects workload.  No --kernel-include-regex: the whole run is instrumented
# (round 1's abort under --pmc came with 36 864 generator launches; a batch is now filled by 9).
# Output: gpurun_out/r02prof/ ; summaries are copied to profiles/r02/ by hand afterwards.
set -o pipefail
R=$PWD; O=$R/gpurun_out/r02prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
prof() { # tag, rocprof args..., --, bench args
  local tag=$1; shift
  local pa=(); while [ "$1" != "--" ]; do pa+=("$1"); shift; done; shift
  rocprofv3 "${pa[@]}" --output-format csv -d $O/$tag -o p -- python3 $R/bench.py "$@" > $O/$tag.json 2> $O/$tag.err || echo "$tag FAILED (rc $?)"
}
SQ1="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
SQ2="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32"
SQ3="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"
prof stats_A2 --kernel-trace --stats -- --cpu-sample 0
prof fetch_A2 --pmc FETCH_SIZE -- --cpu-sample 0
prof write_A2 --pmc WRITE_SIZE -- --cpu-sample 0
prof grbm_A2 --pmc GRBM_GUI_ACTIVE -- --cpu-sample 0
for RG in A1 A2 A3 A4; do
  prof sq1_$RG --pmc $SQ1 -- --cpu-sample 0 --regime $RG --steps 10
  prof sq2_$RG --pmc $SQ2 -- --cpu-sample 0 --regime $RG --steps 10
  prof sq3_$RG --pmc $SQ3 -- --cpu-sample 0 --regime $RG --steps 10
done
prof stats_fx --kernel-trace --stats -- --workload effects_4k_batch --steps 10 --cpu-sample 0
prof fetch_fx --pmc FETCH_SIZE -- --workload effects_4k_batch --steps 10 --cpu-sample 0
prof write_fx --pmc WRITE_SIZE -- --workload effects_4k_batch --steps 10 --cpu-sample 0
prof stats_4k --kernel-trace --stats -- --workload compose_4k_fullframe --cpu-sample 0
cd $R
python3 - <<'PY'
import csv, glob, collections, os, json
O = "gpurun_out/r02prof"
def counters(tag, kernel):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{O}/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}
with open(f"{O}/summary.txt", "w") as out:
    def P(*a):
        s = " ".join(str(x) for x in a); print(s); out.write(s + "\n")
    for tag in sorted(os.listdir(O)):
        if not os.path.isdir(f"{O}/{tag}"): continue
        if tag.startswith("stats"):
            for f in glob.glob(f"{O}/{tag}/**/*kernel_stats.csv", recursive=True):
                P("==", tag, "kernel_stats"); 
                for line in open(f).read().splitlines()[:8]: P("  ", line)
        else:
            kern = "effect_dc_kernel" if tag.endswith("_fx") else "compose_kernel"
            c = counters(tag, kern)
            P("==", tag, kern)
            for k in sorted(c): P("   %-28s mean %.6g (n=%d)" % (k, c[k][0], c[k][1]))
PY
# same-box A/B of the sz0 register cache (variants built by scripts/build_variant.sh)
if [ -f gpurun_variants/libmjx_n2nosz.so ]; then
  for RG in A2 A3 A4; do bash scripts/ab_bench.sh "--regime $RG --cpu-sample 16 --pristine-launches 0" n2 n2nosz 2>&1 | tee -a $O/ab_sz0_cache.txt; done
fi
