#!/bin/bash
# run bench.py for every variant .so under gpurun_variants/ (on the GPU box), interleaved rounds
# usage: scripts/ab_bench.sh "<bench args>" name1 name2 ...
args=$1; shift
for round in 1 2; do
for n in "$@"; do
  MJX_HIP_LIB=$PWD/gpurun_variants/libmjx_$n.so python bench.py $args --steps 30 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-14s %-4s value %.3f Gblk/s  kern %.4f ms  %.0f GB/s  frac %.3f  parity %s' % ('$n', d['config']['alpha_regime'], d['value']/1e9, r['kernel_ms_mean'], r['achieved'], r['frac'], d['parity']))"
done; done
