#!/bin/bash
# round 4: the lean streaming kernel (blocks that need no blend) with a real pipeline inside the wave (scalar table reads, global
# instead of flat loads, 2-4 octets in flight) against round 3's (flat loads, a vmcnt(0) per octet): A1 (all blocks DC-only) and A5
O=gpurun_out/r04_light_ab; mkdir -p $O
line() { python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-12s %-3s %-5s step %.4f ms  kern %.4f ms  frac(plan bytes) %.3f  frac@768 %.3f  parity %s' % ('$1', d['config']['alpha_regime'], '$3', d['ms_per_step'], r['kernel_ms_mean'], r['frac'], r.get('frac_at_nominal_bytes_per_block') or 0, d['parity']))"; }
for round in 1 2; do for tiles in f32 i16; do for regime in A1 A5; do
  for v in r03light base cap64 cap64d3 d3; do
    lib=$PWD/gpurun_variants/libmjx_$v.so; [ $v = base ] && lib=$PWD/libmodjpeg_amd/libmjx_hip.so
    MJX_HIP_LIB=$lib python bench.py --regime $regime --tiles $tiles --steps 40 --cpu-sample 2 --pristine-launches 0 2>/dev/null | line $v $regime $tiles
  done
done; done; done 2>&1 | tee $O/light_ab.txt
