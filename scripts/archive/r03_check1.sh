#!/bin/bash
# round-3 GPU check 1: the copy yardstick, N-rank rehearsals on one GPU (item 1a:
# 4 and 6 ranks — the pool's process guard allows at most 6 processes on the card, so 8 cannot be rehearsed here),
# and the shard-size baseline of the round-2 kernel on this box.
set -o pipefail
O=gpurun_out/r03a; mkdir -p $O
rocm-smi --showclocks > $O/clocks.txt 2>&1 || true
timeout -k 10 300 scripts/ubench/copy_roof 2 > $O/copy_roof.txt 2>&1; echo "copy_roof rc=$?"
timeout -k 10 120 scripts/ubench/stream_pattern > $O/stream_pattern.txt 2>&1; echo "stream_pattern rc=$?"
for N in 512 1024 2048 4096; do
  timeout -k 10 300 python bench.py --images $N --cpu-sample 8 > $O/bench_images_$N.json 2> $O/bench_images_$N.err || echo "bench images $N failed"
done
MJX_BENCH_DEVICES=0,0,0,0 timeout -k 10 400 python bench.py --gpus 4 --cpu-sample 4 > $O/bench_4ranks_strong.json 2> $O/bench_4ranks_strong.err || echo "4-rank failed"
MJX_BENCH_DEVICES=0,0,0,0,0,0 timeout -k 10 400 python bench.py --gpus 6 --cpu-sample 4 > $O/bench_6ranks_strong.json 2> $O/bench_6ranks_strong.err || echo "6-rank failed"
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    txt=open(sys.argv[1]).read().strip().splitlines()
    d=json.loads(txt[-1])
    r=d.get("roofline") or {}
    print(sys.argv[1].split('/')[-1], "lines=%d"%len(txt), "value=%.4g"%d["value"], "ms=%.4f"%d["ms_per_step"], "frac=%s"%r.get("frac"), "kern=%s"%r.get("kernel_ms_mean"), d.get("parity"), d.get("scaling"), "n=%s"%d.get("n_gpus"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
cat $O/copy_roof.txt
