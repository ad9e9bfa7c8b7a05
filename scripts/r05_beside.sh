#!/bin/bash
# side-by-side launches: round 4's arrangement (lean blend 144 registers + one stream wave of 80 per SIMD), the 120-register lean build
# with one stream wave (w1) and with two of 72 (w2); then the share of the stream runs that goes beside the blend kernel, swept
# usage (GPU box): scripts/r05_beside.sh -> gpurun_out/r05/beside.txt   (variants: scripts/build_variant.sh r04 / w1 / w2)
cd "$(dirname "$0")/.."
out=gpurun_out/r05/beside.txt
mkdir -p gpurun_out/r05
: > $out
one() { # variant, regime, extra args
  MJX_HIP_LIB=$PWD/gpurun_variants/libmjx_$1.so python bench.py --regime $2 --steps 40 --also none --cpu-sample 8 --pristine-launches 0 $3 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); r=d['roofline']; p=d['config']['plan']
print('%-4s %-3s %-28s step %.4f ms  frac(moved) %.3f  beside %6d of %6d light runs  parity %s' % ('$1', '$2', '$3', r['kernel_ms_mean'], r['frac'], p['runs_beside'], p['runs'], d['parity']))" >> $out
}
for round in 1 2; do
  for reg in A5 A4; do
    for v in r04 w1 w2; do one $v $reg ""; done
  done
done
for reg in A5 A4; do
  for share in 0.5 0.6 0.7 0.8 0.9 1.0; do one w2 $reg "--beside-share $share"; done
done
for reg in A5 A4; do
  for v in r04 w2; do one $v $reg "--tiles i16"; done
  for share in 0.7 0.85 1.0; do one w2 $reg "--tiles i16 --beside-share $share"; done
done
for v in r04 w2; do one $v A2 ""; one $v A3 ""; one $v A1 ""; done
cat $out
