#!/bin/bash
# round-4 profiles (GPU box), final kernels of the round:
#  1. rocprofv3 --kernel-trace --stats + FETCH_SIZE / WRITE_SIZE (separate --pmc passes) for the headline launch (A2, 4096 images)
#  2. the same for the real-mask regimes A5 and A4 as the library now launches them (blend kernel and stream kernel side by
#     side) and one kernel after the other (--launch serial): the overlap shows as the sum of the kernels' durations exceeding
#     the step, FETCH / WRITE per step must not change
#  3. A3 (dense): VALU instructions by opcode class, VALU busy cycles, wave cycles - what the blend kernel is bound by
# Output: gpurun_out/r04prof/ ; summary.txt at the end
set -o pipefail
R=$PWD; O=$R/gpurun_out/r04prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
prof() { # tag, rocprof args..., --, bench args
  local tag=$1; shift
  local pa=(); while [ "$1" != "--" ]; do pa+=("$1"); shift; done; shift
  rocprofv3 "${pa[@]}" --output-format csv -d $O/$tag -o p -- python3 $R/bench.py "$@" > $O/$tag.json 2> $O/$tag.err || echo "$tag FAILED (rc $?)"
  echo "done $tag"
}
prof stats_A2_auto --kernel-trace --stats -- --cpu-sample 0
prof fetch_A2_auto --pmc FETCH_SIZE -- --cpu-sample 0
prof write_A2_auto --pmc WRITE_SIZE -- --cpu-sample 0
for RG in A5 A4; do for L in auto serial; do
  prof stats_${RG}_$L --kernel-trace --stats -- --cpu-sample 0 --regime $RG --launch $L
  prof fetch_${RG}_$L --pmc FETCH_SIZE -- --cpu-sample 0 --regime $RG --launch $L
  prof write_${RG}_$L --pmc WRITE_SIZE -- --cpu-sample 0 --regime $RG --launch $L
done; done
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"
SQ2="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32"
SQ3="SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVES GRBM_GUI_ACTIVE"
for RG in A3 A2; do
  prof sq1_${RG}_auto --pmc $SQ1 -- --cpu-sample 0 --regime $RG --steps 10
  prof sq2_${RG}_auto --pmc $SQ2 -- --cpu-sample 0 --regime $RG --steps 10
  prof sq3_${RG}_auto --pmc $SQ3 -- --cpu-sample 0 --regime $RG --steps 10
done
cd $R
python3 - <<'PY'
import csv, glob, collections, os, json
O = "gpurun_out/r04prof"
def counters(tag):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{O}/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "compose" in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("(")[0].replace("void mjx::", "")
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc
traffic = {}
with open(f"{O}/summary.txt", "w") as out:
    def P(*a):
        s = " ".join(str(x) for x in a); print(s); out.write(s + "\n")
    for tag in sorted(os.listdir(O)):
        if not os.path.isdir(f"{O}/{tag}"): continue
        if tag.startswith("stats"):
            for f in glob.glob(f"{O}/{tag}/**/*kernel_stats.csv", recursive=True):
                P("==", tag, "kernel_stats")
                for line in open(f).read().splitlines()[:7]: P("  ", line)
            try:
                d = json.loads(open(f"{O}/{tag}.json").read().strip().splitlines()[-1])
                P("   bench.py: ms_per_step", d["ms_per_step"], "kernel_ms_mean (events around the K steps / K)", d["roofline"]["kernel_ms_mean"], "frac", d["roofline"]["frac"], d["parity"], "|", d["config"]["launch"])
            except Exception as e:
                P("   bench line unreadable", e)
        else:
            acc = counters(tag)
            P("==", tag)
            kind, regime, launch = tag.split("_")
            for kern in sorted(acc):
                for k in sorted(acc[kern]):
                    v = acc[kern][k]
                    n = min(len(v), 20)  # the timed steps (the first dispatches are the parity pass and the spin-up, same size)
                    P("   %-36s %-28s mean of last %d: %.6g (n=%d)" % (kern, k, n, sum(v[-n:]) / n, len(v)))
                    if k in ("FETCH_SIZE", "WRITE_SIZE"):
                        traffic.setdefault((regime, launch), {}).setdefault(k, 0.0)
                        traffic[(regime, launch)][k] += sum(v[-n:]) / n
    P("== HBM bytes per step: sum over the step's kernels of (FETCH_SIZE KiB x 2 on gfx950 + WRITE_SIZE KiB) x 1024")
    for (regime, launch), e in sorted(traffic.items()):
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            hbm = int(round((2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024))
            P(json.dumps({f"compose_1080p_batch:{regime}:4096:{launch}": {"hbm_bytes_per_step": hbm, "fetch_size_kib": e["FETCH_SIZE"], "write_size_kib": e["WRITE_SIZE"],
                                                                           "blocks_per_step": 4096 * 1536, "bytes_per_block": round(hbm / (4096 * 1536), 2), "round": 4}}))
PY
