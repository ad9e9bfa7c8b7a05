#!/bin/bash
# round 4: the blend kernel and the lean streaming kernel of one step side by side (two streams joined by events, MJX_CORUN)
# against one after the other, same box, interleaved; MJX_BLEND_LDS_PAD caps the blend kernel's workgroups per CU
# usage (GPU box): scripts/r04_corun.sh [regimes...]   -> gpurun_out/r04_corun/
O=gpurun_out/r04_corun; mkdir -p $O
line() { python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-22s %-3s %-4s step %.4f ms  kern %.4f ms (pristine %s)  frac(plan bytes) %.3f  frac@768 %s  parity %s' % ('$1', d['config']['alpha_regime'], d['config'].get('tiles','f32'), d['ms_per_step'], r['kernel_ms_mean'], r.get('kernel_ms_pristine'), r['frac'], r.get('frac_at_nominal_bytes_per_block'), d['parity']))"; }
for regime in ${@:-A5 A4}; do
for round in 1 2; do
  python bench.py --regime $regime --steps 40 --cpu-sample 2 2>/dev/null | line "serial"
  MJX_CORUN=1 python bench.py --regime $regime --steps 40 --cpu-sample 2 2>/dev/null | line "corun"
  MJX_CORUN=1 MJX_BLEND_LDS_PAD=4096 python bench.py --regime $regime --steps 40 --cpu-sample 2 2>/dev/null | line "corun, blend 2 wg/cu"
done; done 2>&1 | tee $O/corun.txt
