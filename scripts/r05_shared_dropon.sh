#!/bin/bash
# round 5: ONE overlay for the whole batch (SURVEY 8d's secondary variant; what the reference's callers do): step times with the
# non-temporal load policy (nt1, the product) and with plain loads (nt0) on per-image and shared tiles, then FETCH_SIZE / WRITE_SIZE
# of the shared launches
# usage (GPU box): scripts/r05_shared_dropon.sh > gpurun_out/r05/shared_dropon.txt
R=$PWD; O=$R/gpurun_out/r05/shared; mkdir -p $O
for round in 1 2; do for rg in A2 A1 A5; do for sh in "" "--shared-dropon"; do for v in nt1 nt0; do
  MJX_HIP_LIB=$R/gpurun_variants/libmjx_$v.so python bench.py --regime $rg $sh --also none --steps 30 --cpu-sample 4 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read())
print('%-4s %-3s %-16s %8.3f Gblk/s  step %.4f ms  %s' % ('$v', d['config']['alpha_regime'], '$sh', d['value']/1e9, d['ms_per_step'], d['parity']))"
done; done; done; done
cd /tmp && export TMPDIR=/tmp
for rg in A2 A1 A5; do for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/${rg}_$c -o p -- python3 $R/bench.py --regime $rg --shared-dropon --also none --cpu-sample 0 --steps 20 > /dev/null 2> $O/${rg}_$c.err || echo "$rg $c FAILED"
done; done
cd $R
python3 - <<'PY'
import csv, glob, collections
O = "gpurun_out/r05/shared"
for rg in ("A2", "A1", "A5"):
    tot = 0.0
    for c, mult in (("FETCH_SIZE", 2), ("WRITE_SIZE", 1)):
        acc = collections.defaultdict(list)
        for f in glob.glob(f"{O}/{rg}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "compose" in r["Kernel_Name"]: acc[r["Kernel_Name"].split("(")[0].replace("void mjx::", "")].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            m = sum(v[-20:]) / len(v[-20:]); tot += m * mult * 1024
            print("%s shared  %-36s %-11s mean of last 20: %.0f KiB" % (rg, k, c, m))
    print("%s shared  HBM bytes per step (FETCH x 2 + WRITE): %.0f = %.1f B per block" % (rg, tot, tot / 6291456))
PY
