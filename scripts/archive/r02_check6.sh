#!/bin/bash
# round-2 GPU check 6: int16 tiles (MJX_JOB_TILES_I16) - parity with the float tiles, and the labelled secondary bench lines
set -o pipefail
O=gpurun_out/r02k; mkdir -p $O
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for RG in A1 A2 A3 A4 A5; do for T in f32 i16; do
  python bench.py --regime $RG --tiles $T --cpu-sample 32 > $O/bench_${RG}_$T.json 2> $O/bench_${RG}_$T.err || echo "bench $RG $T failed"
  python - "$O/bench_${RG}_$T.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[1].split('/')[-1], "%.3f G blocks/s" % (d["value"]/1e9), "kern %.4f ms" % r["kernel_ms_mean"], "%.0f GB/s (%d B/block) frac %.3f" % (r["achieved"], r["bytes_per_block"], r["frac"]), d["parity"])
except Exception as e: print(sys.argv[1], "unreadable", e)
PY
done; done
