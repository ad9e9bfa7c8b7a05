#!/bin/bash
# rocprofv3 passes for the headline launch after the low-column build learnt to leave the second line of the alpha blocks alone
# (runs without weights in rows 4..7): kernel stats + FETCH_SIZE / WRITE_SIZE (separate --pmc passes) for A2 with 4096 images
# and with the 512-image shard of the 8-GPU split, and for the 256 x 4K batch.  Output: gpurun_out/r04prof_lowrows/ ; summary.txt
set -o pipefail
R=$PWD; O=$R/gpurun_out/r04prof_lowrows; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
prof() { # tag, rocprof args..., --, bench args
  local tag=$1; shift
  local pa=(); while [ "$1" != "--" ]; do pa+=("$1"); shift; done; shift
  rocprofv3 "${pa[@]}" --output-format csv -d $O/$tag -o p -- python3 $R/bench.py "$@" > $O/$tag.json 2> $O/$tag.err || echo "$tag FAILED (rc $?)"
  echo "done $tag"
}
for W in "A2x4096:" "A2x512:--images 512" "4Kx256:--workload compose_4k_batch"; do
  tag=${W%%:*}; args=${W#*:}
  prof stats_$tag --kernel-trace --stats -- --cpu-sample 0 $args
  prof fetch_$tag --pmc FETCH_SIZE -- --cpu-sample 0 $args
  prof write_$tag --pmc WRITE_SIZE -- --cpu-sample 0 $args
done
cd $R
python3 - <<'PY'
import csv, glob, collections, os, json
O = "gpurun_out/r04prof_lowrows"
def counters(tag):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{O}/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "compose" in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("(")[0].replace("void mjx::", "")
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc
traffic, blocks = {}, {}
with open(f"{O}/summary.txt", "w") as out:
    def P(*a):
        s = " ".join(str(x) for x in a); print(s); out.write(s + "\n")
    for tag in sorted(os.listdir(O)):
        if not os.path.isdir(f"{O}/{tag}"): continue
        kind, what = tag.split("_")
        if kind == "stats":
            for f in glob.glob(f"{O}/{tag}/**/*kernel_stats.csv", recursive=True):
                P("==", tag, "kernel_stats")
                for line in open(f).read().splitlines()[:6]: P("  ", line)
            try:
                d = json.loads(open(f"{O}/{tag}.json").read().strip().splitlines()[-1])
                r = d["roofline"]
                blocks[what] = d["config"]["blocks_per_step_total"]
                P("   bench.py: ms_per_step", d["ms_per_step"], "kernel_ms_mean", r["kernel_ms_mean"], "frac", r["frac"], "frac_on_bytes_moved", r["frac_on_bytes_moved"],
                  "bytes_moved_per_block", r["bytes_moved_per_block"], d["parity"], "| plan:", {k: d["config"]["plan"][k] for k in ("runs", "runs_low_columns", "low_column_octets", "low_row_octets")})
            except Exception as e:
                P("   bench line unreadable", e)
        else:
            acc = counters(tag)
            P("==", tag)
            for kern in sorted(acc):
                for k in sorted(acc[kern]):
                    v = acc[kern][k]
                    n = min(len(v), 20)
                    P("   %-36s %-28s mean of last %d: %.6g (n=%d)" % (kern, k, n, sum(v[-n:]) / n, len(v)))
                    if k in ("FETCH_SIZE", "WRITE_SIZE"):
                        traffic.setdefault(what, {}).setdefault(k, 0.0)
                        traffic[what][k] += sum(v[-n:]) / n
    P("== HBM bytes per step: (FETCH_SIZE KiB x 2 on gfx950 + WRITE_SIZE KiB) x 1024")
    for what, e in sorted(traffic.items()):
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e and what in blocks:
            hbm = int(round((2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024))
            P(json.dumps({what: {"hbm_bytes_per_step": hbm, "fetch_size_kib": e["FETCH_SIZE"], "write_size_kib": e["WRITE_SIZE"], "blocks_per_step": blocks[what],
                                 "bytes_per_block": round(hbm / blocks[what], 2), "round": 4}}))
PY
