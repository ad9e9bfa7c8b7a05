#!/bin/bash
# round 4: every regime one kernel after the other and as the library decides (side by side where both kernels fill the device),
# interleaved on one box; then the parity tests that force either mode
O=gpurun_out/r04_sidebyside; mkdir -p $O
line() { python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-8s %-3s %-5s step %.4f ms  kern %.4f ms  frac(plan bytes) %.3f  frac@768 %.3f  parity %s  | %s' % ('$1', d['config']['alpha_regime'], '$3', d['ms_per_step'], r['kernel_ms_mean'], r['frac'], r.get('frac_at_nominal_bytes_per_block') or 0, d['parity'], d['config']['launch']))"; }
for round in 1 2; do for tiles in f32 i16; do for regime in ${@:-A5 A4 A1 A2 A3}; do
  for launch in serial auto; do
    python bench.py --regime $regime --tiles $tiles --launch $launch --steps 40 --cpu-sample 2 --pristine-launches 0 2>/dev/null | line $launch $regime $tiles
  done
done; done; done 2>&1 | tee $O/regimes.txt
