#!/bin/bash
# round 5: what the gather form of the blend kernel costs.  The headline launch (A2: every block needs all 64 terms, octets of 8
# consecutive blocks) against the same blocks through the gather form (MJX_EXP_GATHER=1: same octets, named by the table) and with
# the blocks of every rectangle shuffled (=2: an octet's eight blocks lie anywhere in their rectangle).  Test build of the library.
export MJX_HIP_LIB=$PWD/libmodjpeg_amd/libmjx_hip_faultinj.so
for round in 1 2; do for g in 0 1 2; do for t in f32 i16; do
  MJX_EXP_GATHER=$g python bench.py --also none --cpu-sample 4 --steps 30 --tiles $t 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); p=d['config']['plan']
print('MJX_EXP_GATHER=$g %-4s step %.4f ms  %s  gather octets %d of %d  runs %d' % ('$t', d['ms_per_step'], d['parity'], p['gather_octets'], p['octets'], p['runs']))"
done; done; done
