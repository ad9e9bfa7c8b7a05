#!/bin/bash
# the blend kernels of the tree (MJX_PK_F32=0: paired z storage, scalar terms) against those of the commit before (scripts/build_variant.sh head / new)
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/r05
out=gpurun_out/r05/pk0_structure_ab.txt; : > $out
one() { MJX_HIP_LIB=$PWD/gpurun_variants/libmjx_$1.so python bench.py --regime $2 --steps 40 --also none --cpu-sample 8 --pristine-launches 0 $3 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-4s %-3s %-22s step %.4f ms  frac(moved) %.3f  parity %s' % ('$1', '$2', '$3', r['kernel_ms_mean'], r['frac'], d['parity']))" >> $out; }
for round in 1 2 3; do for reg in A2 A3 A5 A4; do for v in head new; do one $v $reg ""; done; done; done
for v in head new head new; do one $v A2 "--tiles i16"; one $v A2 "--images 512"; done
cat $out
