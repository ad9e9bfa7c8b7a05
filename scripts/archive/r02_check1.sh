#!/bin/bash
# round-2 GPU check 1: new tests, all four alpha regimes, 2-rank rehearsal on one GPU, effects at full size
set -o pipefail
O=gpurun_out/r02b; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "fullsize or fuzz or synth" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for R in A1 A2 A3 A4; do python bench.py --regime $R > $O/bench_$R.json 2> $O/bench_$R.err || echo "bench $R failed"; done
MJX_BENCH_DEVICES=0,0 python bench.py --gpus 2 > $O/bench_2ranks_strong.json 2> $O/bench_2ranks_strong.err || echo "2-rank failed"
MJX_BENCH_DEVICES=0,0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --scaling weak --images 1024 > $O/bench_2ranks_weak.json 2> $O/bench_2ranks_weak.err || echo "2-rank weak failed"
python bench.py --workload effects_4k_batch --steps 10 > $O/bench_effects_4096.json 2> $O/bench_effects_4096.err || echo "effects failed"
python bench.py --workload compose_4k_fullframe --regime A2 > $O/bench_4k_A2.json 2> $O/bench_4k_A2.err || echo "4k failed"
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get("roofline") or {}
    print(sys.argv[1].split('/')[-1], "value=%.4g"%d["value"], "ms=%.4f"%d["ms_per_step"], "frac=%s"%r.get("frac"), "kern=%s"%r.get("kernel_ms_mean"), "prist=%s"%r.get("kernel_ms_pristine"), d.get("parity"), d.get("scaling"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
