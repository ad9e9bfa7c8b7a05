#!/bin/bash
# round-2 GPU check 4: real-mask regime A5, plan-level grouping fuzz, A5 bench grouped / not
set -o pipefail
O=gpurun_out/r02i; mkdir -p $O
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q -k "a5 or grouped or density" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for G in "" "--no-group"; do python bench.py --regime A5 $G --cpu-sample 64 > $O/bench_A5$G.json 2> $O/bench_A5$G.err || echo "bench A5 $G failed"; done
for f in $O/bench_A5*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[1].split('/')[-1], "value=%.4g"%d["value"], "kern=%.4f"%r["kernel_ms_mean"], "frac=%.3f"%r["frac"], d["parity"], "gather_octets", d["config"]["gather_octets_per_gpu"])
PY
done
