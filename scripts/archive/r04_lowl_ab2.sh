#!/bin/bash
# round 4: low-column build of the blend kernel (16 waves per CU) at A2: run length per wave, against the general build, alternating, 200 steps each
T=$PWD/libmodjpeg_amd/libmjx_hip_faultinj.so
line() { python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-26s step %.4f ms  frac %.3f  runs %s (%s..%s octets)' % ('$1', d['ms_per_step'], r['frac'], d['config']['plan']['runs'], d['config']['plan']['run_octets_min'], d['config']['plan']['run_octets_max']))"; }
for round in 1 2; do
  MJX_HIP_LIB=$T MJX_LOWL=0 python bench.py --steps 200 --cpu-sample 0 --pristine-launches 0 2>/dev/null | line "general, runs of 16"
  for len in 16 8 4 2; do
    MJX_HIP_LIB=$T MJX_RUN_OCTETS=$len python bench.py --steps 200 --cpu-sample 0 --pristine-launches 0 2>/dev/null | line "low-column, runs of $len"
  done
  MJX_HIP_LIB=$T MJX_LOWL=0 MJX_RUN_OCTETS=8 python bench.py --steps 200 --cpu-sample 0 --pristine-launches 0 2>/dev/null | line "general, runs of 8"
done
