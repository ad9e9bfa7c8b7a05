#!/bin/bash
# small launches (one round of waves): the run length from the slot count of the build the runs go to (new) against the general build's (head)
# usage (GPU box): scripts/r05_small_launches.sh -> gpurun_out/r05/small_launches_ab.txt   (scripts/build_variant.sh head / new)
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/r05
out=gpurun_out/r05/small_launches_ab.txt; : > $out
one() { MJX_HIP_LIB=$PWD/gpurun_variants/libmjx_$1.so python bench.py --steps 400 --also none --cpu-sample 1 --pristine-launches 0 $2 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); r=d['roofline']; p=d['config']['plan']
print('%-5s %-44s step %.4f ms  frac %.3f  runs %6d  run_octets %d..%d  parity %s' % ('$1', '$2', r['kernel_ms_mean'], r['frac'], p['runs'], p['run_octets_min'], p['run_octets_max'], d['parity']))" >> $out; }
for round in 1 2 3; do for v in head new; do one $v "--workload compose_4k_fullframe"; done; done
for n in 8 32 64 128 256 512; do for v in head new; do one $v "--images $n"; done; done
for v in head new; do one $v "--workload compose_4k_fullframe --tiles i16"; one $v "--workload compose_4k_fullframe --regime A3"; one $v "--workload compose_4k_fullframe --regime A5"; done
cat $out
