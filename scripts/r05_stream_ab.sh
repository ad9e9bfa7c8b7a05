#!/bin/bash
# the stream kernel's table words by v_writelane + DPP (new) against seven selects per word (head): real-mask regimes, same box, alternating
# usage (GPU box): scripts/r05_stream_ab.sh [variants...] -> gpurun_out/r05/stream_pick_ab.txt   (scripts/build_variant.sh head / new)
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/r05
out=gpurun_out/r05/stream_pick_ab.txt; : > $out
V=${@:-head new}
one() { MJX_HIP_LIB=$PWD/gpurun_variants/libmjx_$1.so python bench.py --regime $2 --steps 40 --also none --cpu-sample 8 --pristine-launches 0 $3 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); r=d['roofline']; p=d['config']['plan']
print('%-5s %-3s %-26s step %.4f ms  frac(moved) %.3f  beside %6d of %6d  parity %s' % ('$1', '$2', '$3', r['kernel_ms_mean'], r['frac'], p['runs_beside'], p['runs'], d['parity']))" >> $out; }
for round in 1 2 3; do for reg in A5 A4; do for v in $V; do one $v $reg ""; done; done; done
for reg in A5 A4; do for v in $V; do one $v $reg "--launch serial"; one $v $reg "--tiles i16"; done; done
for v in $V; do one $v A1 ""; one $v A2 ""; done
for reg in A5 A4; do for share in 0.6 0.7 0.8 0.9; do one new $reg "--beside-share $share"; done; done
cat $out
