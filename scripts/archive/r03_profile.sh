#!/bin/bash
# round-3 profiles (GPU box): rocprofv3 kernel stats of the default bench command; HBM traffic (FETCH_SIZE /
# WRITE_SIZE in separate --pmc passes) for the headline launch AND for the 8/4/2-GPU shard sizes (512 / 1024 /
# 2048 images); kernel stats + traffic of the real-mask regime A5 (zero-alpha blocks travel without tiles) and
# A4; the 4K batch (BASELINE configs[1]'s kernel on a launch that fills the device).  Whole run instrumented.
# Output: gpurun_out/r03prof/ ; the summary block at the end prints the traffic.json entries.
set -o pipefail
R=$PWD; O=$R/gpurun_out/r03prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
prof() { # tag, rocprof args..., --, bench args
  local tag=$1; shift
  local pa=(); while [ "$1" != "--" ]; do pa+=("$1"); shift; done; shift
  rocprofv3 "${pa[@]}" --output-format csv -d $O/$tag -o p -- python3 $R/bench.py "$@" > $O/$tag.json 2> $O/$tag.err || echo "$tag FAILED (rc $?)"
  echo "done $tag"
}
prof stats_A2_4096 --kernel-trace --stats -- --cpu-sample 0
prof fetch_A2_4096 --pmc FETCH_SIZE -- --cpu-sample 0
prof write_A2_4096 --pmc WRITE_SIZE -- --cpu-sample 0
for IM in 512 1024 2048; do
  prof stats_A2_$IM --kernel-trace --stats -- --cpu-sample 0 --images $IM
  prof fetch_A2_$IM --pmc FETCH_SIZE -- --cpu-sample 0 --images $IM
  prof write_A2_$IM --pmc WRITE_SIZE -- --cpu-sample 0 --images $IM
done
for RG in A1 A4 A5; do
  prof stats_${RG}_4096 --kernel-trace --stats -- --cpu-sample 0 --regime $RG
  prof fetch_${RG}_4096 --pmc FETCH_SIZE -- --cpu-sample 0 --regime $RG
  prof write_${RG}_4096 --pmc WRITE_SIZE -- --cpu-sample 0 --regime $RG
done
prof stats_4kbatch --kernel-trace --stats -- --workload compose_4k_batch --cpu-sample 0 --steps 20
prof stats_4k --kernel-trace --stats -- --workload compose_4k_fullframe --cpu-sample 0
cd $R
python3 - <<'PY'
import csv, glob, collections, os, json
O = "gpurun_out/r03prof"
def counters(tag, kernel):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{O}/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    # the last 40 dispatches: the timed steps (the first are the parity pass and the spin-up, same size anyway)
    return {k: (sum(v[-40:]) / len(v[-40:]), len(v)) for k, v in acc.items()}
traffic = {}
with open(f"{O}/summary.txt", "w") as out:
    def P(*a):
        s = " ".join(str(x) for x in a); print(s); out.write(s + "\n")
    for tag in sorted(os.listdir(O)):
        if not os.path.isdir(f"{O}/{tag}"): continue
        if tag.startswith("stats"):
            for f in glob.glob(f"{O}/{tag}/**/*kernel_stats.csv", recursive=True):
                P("==", tag, "kernel_stats")
                for line in open(f).read().splitlines()[:6]: P("  ", line)
            try:
                d = json.loads(open(f"{O}/{tag}.json").read().strip().splitlines()[-1])
                P("   bench.py:", "kernel_ms_mean", d["roofline"]["kernel_ms_mean"], "frac", d["roofline"]["frac"], "value", d["value"], d["parity"])
            except Exception as e:
                P("   bench line unreadable", e)
        else:
            c = counters(tag, "compose_kernel")
            P("==", tag)
            for k in sorted(c): P("   %-28s mean %.6g (n=%d)" % (k, c[k][0], c[k][1]))
            kind, regime, images = tag.split("_")
            ent = traffic.setdefault((regime, images), {})
            if "FETCH_SIZE" in c: ent["fetch_size_kib"] = c["FETCH_SIZE"][0]
            if "WRITE_SIZE" in c: ent["write_size_kib"] = c["WRITE_SIZE"][0]
    P("== traffic.json entries (FETCH_SIZE KiB x2 on gfx950 + WRITE_SIZE KiB)")
    for (regime, images), e in sorted(traffic.items()):
        if "fetch_size_kib" in e and "write_size_kib" in e:
            hbm = int(round((2 * e["fetch_size_kib"] + e["write_size_kib"]) * 1024))
            blocks = int(images) * 1536
            P(json.dumps({f"compose_1080p_batch:{regime}:{images}": {"hbm_bytes_per_launch": hbm, "fetch_size_kib": e["fetch_size_kib"], "write_size_kib": e["write_size_kib"],
                                                                      "blocks_per_launch": blocks, "bytes_per_block": round(hbm / blocks, 2), "round": 3}}))
PY
