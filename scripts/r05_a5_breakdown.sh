#!/bin/bash
# round 5: where a step of A5 / A4 (real overlay / mixed classes) spends its time after the opaque blocks left the DC-only stream:
# kernel durations (rocprofv3 --kernel-trace --stats) with the kernels side by side (default) and one after the other
R=$PWD; O=$R/gpurun_out/r05/a5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for RG in A5 A4; do for L in auto serial; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${RG}_$L -o p -- python3 $R/bench.py --regime $RG --launch $L --also none --cpu-sample 0 --steps 40 > $O/${RG}_$L.json 2> $O/${RG}_$L.err || echo "$RG $L FAILED"
  echo "== $RG --launch $L"
  python3 - $O/${RG}_$L.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("   ms_per_step %.4f  %s  plan: %s" % (d["ms_per_step"], d["parity"], {k: d["config"]["plan"][k] for k in ("octets", "zero_alpha_octets", "dc_only_octets", "opaque_octets", "runs", "runs_beside")}))
PY
  f=$(find $O/${RG}_$L -name '*kernel_stats.csv' | head -1); head -6 $f | cut -c1-200 | sed 's/^/   /'
done; done
