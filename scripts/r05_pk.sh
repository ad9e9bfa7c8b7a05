#!/bin/bash
# packed f32 pairs of terms (vstage_pair, MJX_PK_F32) against the scalar build, same box, alternating
# usage (GPU box): scripts/r05_pk.sh -> gpurun_out/r05/pk_ab.txt   (variants: scripts/build_variant.sh pk1 ; scripts/build_variant.sh pk0 -DMJX_PK_F32=0)
cd "$(dirname "$0")/.."
out=gpurun_out/r05/pk_ab.txt
mkdir -p gpurun_out/r05
: > $out
one() { # variant, regime, extra args
  MJX_HIP_LIB=$PWD/gpurun_variants/libmjx_$1.so python bench.py --regime $2 --steps 40 --also none --cpu-sample 8 --pristine-launches 0 $3 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); r=d['roofline']; p=d['config']['plan']
print('%-4s %-3s %-22s step %.4f ms  frac(moved) %.3f  parity %s' % ('$1', '$2', '$3', r['kernel_ms_mean'], r['frac'], d['parity']))" >> $out
}
for round in 1 2 3; do
  for reg in A3 A5 A4 A2; do
    for v in pk0 pk1; do one $v $reg ""; done
  done
done
for reg in A3 A5 A2; do for v in pk0 pk1; do one $v $reg "--tiles i16"; done; done
for v in pk0 pk1; do one $v A2 "--no-low-column-build"; one $v A2 "--workload compose_4k_batch"; done
cat $out
