#!/bin/bash
# round-3 profiles, second pass (after the DC-only runs): FETCH_SIZE / WRITE_SIZE and kernel stats of A1 / A3 / A4 / A5 with the final kernel
set -o pipefail
R=$PWD; O=$R/gpurun_out/r03prof2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
prof() { local tag=$1; shift; local pa=(); while [ "$1" != "--" ]; do pa+=("$1"); shift; done; shift
  rocprofv3 "${pa[@]}" --output-format csv -d $O/$tag -o p -- python3 $R/bench.py "$@" > $O/$tag.json 2> $O/$tag.err || echo "$tag FAILED (rc $?)"; echo "done $tag"; }
for RG in A1 A3 A4 A5; do
  prof stats_${RG}_4096 --kernel-trace --stats -- --cpu-sample 0 --regime $RG
  prof fetch_${RG}_4096 --pmc FETCH_SIZE -- --cpu-sample 0 --regime $RG
  prof write_${RG}_4096 --pmc WRITE_SIZE -- --cpu-sample 0 --regime $RG
done
cd $R
python3 - <<'PY'
import csv, glob, collections, os, json
O = "gpurun_out/r03prof2"
def counters(tag):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{O}/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "compose_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
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
                for line in open(f).read().splitlines()[:3]: P("  ", line[:200])
            d = json.loads(open(f"{O}/{tag}.json").read().strip().splitlines()[-1])
            P("   bench.py:", "kernel_ms_mean", d["roofline"]["kernel_ms_mean"], "frac", d["roofline"]["frac"], d["config"]["plan"])
        else:
            c = counters(tag); P("==", tag, {k: v[0] for k, v in c.items()})
            kind, regime, images = tag.split("_")
            e = traffic.setdefault((regime, images), {})
            if "FETCH_SIZE" in c: e["fetch_size_kib"] = c["FETCH_SIZE"][0]
            if "WRITE_SIZE" in c: e["write_size_kib"] = c["WRITE_SIZE"][0]
    for (regime, images), e in sorted(traffic.items()):
        if len(e) == 2:
            hbm = int(round((2 * e["fetch_size_kib"] + e["write_size_kib"]) * 1024)); blocks = int(images) * 1536
            P(json.dumps({f"compose_1080p_batch:{regime}:{images}": {"hbm_bytes_per_launch": hbm, **e, "blocks_per_launch": blocks, "bytes_per_block": round(hbm / blocks, 2), "round": 3}}))
PY
