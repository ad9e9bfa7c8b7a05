#!/bin/bash
# round-5 profiles (GPU box), final tree of the round:
#  1. rocprofv3 --kernel-trace --stats of the DEFAULT command (`python bench.py --steps 20 --warmup 5`: the headline launch and the
#     `also` block - A2 without the promise, A5, A4, A3, the 4K full-frame image, the effects batch), and of the headline alone
#  2. FETCH_SIZE / WRITE_SIZE (separate --pmc passes) of the headline launch, of the same launch planned without the promise that
#     alpha is final, of A5 / A4 as the library launches them, of A3, and of the effects batch (+ how its requests leave L2)
# Output: gpurun_out/r05prof/ ; summary.txt at the end
set -o pipefail
R=$PWD; O=$R/gpurun_out/r05prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
prof() { # tag, rocprof args..., --, bench args
  local tag=$1; shift
  local pa=(); while [ "$1" != "--" ]; do pa+=("$1"); shift; done; shift
  rocprofv3 "${pa[@]}" --output-format csv -d $O/$tag -o p -- python3 $R/bench.py "$@" > $O/$tag.json 2> $O/$tag.err || echo "$tag FAILED (rc $?)"
  echo "done $tag"
}
prof stats_default --kernel-trace --stats -- --steps 20 --warmup 5
H="--also none --cpu-sample 0"
prof stats_A2 --kernel-trace --stats -- $H
prof fetch_A2 --pmc FETCH_SIZE -- $H
prof write_A2 --pmc WRITE_SIZE -- $H
prof stats_A2nopromise --kernel-trace --stats -- $H --keep-zero-alpha
prof fetch_A2nopromise --pmc FETCH_SIZE -- $H --keep-zero-alpha
prof write_A2nopromise --pmc WRITE_SIZE -- $H --keep-zero-alpha
for RG in A1 A5 A4 A3; do
  prof stats_$RG --kernel-trace --stats -- $H --regime $RG
  prof fetch_$RG --pmc FETCH_SIZE -- $H --regime $RG
  prof write_$RG --pmc WRITE_SIZE -- $H --regime $RG
done
E="--workload effects_4k_batch --steps 10 --cpu-sample 0"
prof stats_effects --kernel-trace --stats -- $E
prof fetch_effects --pmc FETCH_SIZE -- $E
prof write_effects --pmc WRITE_SIZE -- $E
prof tcc_effects --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -- $E
cd $R
python3 - <<'PY'
import csv, glob, collections, os, json
O = "gpurun_out/r05prof"
def counters(tag, want):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{O}/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if want in r["Kernel_Name"]:
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
                for line in open(f).read().splitlines()[:(16 if tag == "stats_default" else 6)]: P("  ", line)
            try:
                d = json.loads(open(f"{O}/{tag}.json").read().strip().splitlines()[-1])
                r = d["roofline"]
                P("   bench.py: ms_per_step", d["ms_per_step"], "kernel_ms_mean", r["kernel_ms_mean"], "frac", r["frac"], d["parity"])
                if "also" in d:
                    P("   plan_ms", json.dumps({k: v for k, v in d["plan_ms"].items() if k != "note"}), " frac_without_promise", r.get("frac_without_promise"), "kernel_ms_without_promise", r.get("kernel_ms_without_promise"))
                    for k, v in d["also"].items():
                        if isinstance(v, dict): P("   also", k, "ms_per_step %.4f" % v["ms_per_step"], v["parity"], "frac_on_bytes_moved", v.get("frac_on_bytes_moved"))
            except Exception as e:
                P("   bench line unreadable", e)
        else:
            kind, what = tag.split("_", 1)
            acc = counters(tag, "effect_dc" if what == "effects" else "compose")
            P("==", tag)
            for kern in sorted(acc):
                for k in sorted(acc[kern]):
                    v = acc[kern][k]
                    n = min(len(v), 20)
                    P("   %-36s %-28s mean of last %d: %.6g (n=%d)" % (kern, k, n, sum(v[-n:]) / n, len(v)))
                    if k in ("FETCH_SIZE", "WRITE_SIZE"):
                        traffic.setdefault(what, {}).setdefault(k, 0.0)
                        traffic[what][k] += sum(v[-n:]) / n
    P("== HBM bytes per step: sum over the step's kernels of (FETCH_SIZE KiB x 2 on gfx950 + WRITE_SIZE KiB) x 1024  (effects: FETCH_SIZE uncorrected AND x 2: a DC read is one request for a whole 128-byte line, tallied at 64)")
    for what, e in sorted(traffic.items()):
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            hbm = int(round((2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024))
            P(json.dumps({what: {"hbm_bytes": hbm, "fetch_size_kib": e["FETCH_SIZE"], "write_size_kib": e["WRITE_SIZE"], "round": 5}}))
PY
