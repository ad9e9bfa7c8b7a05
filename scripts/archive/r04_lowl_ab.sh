#!/bin/bash
# round 4: the low-column build of the blend kernel (four workgroups per CU) against the general one on the regimes whose weights lie in
# columns 0..3 (A2: all of them), interleaved on one box.  The switch MJX_LOWL exists in the test build of the library only.
O=gpurun_out/r04_lowl; mkdir -p $O
T=$PWD/libmodjpeg_amd/libmjx_hip_faultinj.so
line() { python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-10s %-3s img %-5s step %.4f ms  kern %.4f ms  frac %.3f  parity %s  low-column runs %s of %s' % ('$1', d['config']['alpha_regime'], '$2', d['ms_per_step'], r['kernel_ms_mean'], r['frac'], d['parity'], d['config']['plan'].get('runs_low_columns'), d['config']['plan'].get('runs')))"; }
for round in 1 2 3; do for images in 4096 512; do for regime in ${@:-A2}; do
  MJX_HIP_LIB=$T MJX_LOWL=0 python bench.py --regime $regime --images $images --steps 40 --cpu-sample 2 --pristine-launches 0 2>/dev/null | line general $images
  python bench.py --regime $regime --images $images --steps 40 --cpu-sample 2 --pristine-launches 0 2>/dev/null | line low-column $images
done; done; done 2>&1 | tee $O/lowl_ab.txt
