#!/bin/bash
# round 4: the 4K workloads (BASELINE configs[1]: one 3840x2160 image, full-frame dropon; and 256 of them) with the low-column build of the
# blend kernel (default) and with the general build (MJX_LOWL=0, test build of the library), interleaved
T=$PWD/libmodjpeg_amd/libmjx_hip_faultinj.so
line() { python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-12s %-22s step %.5f ms  kern %.5f ms  frac %.3f  parity %s' % ('$1', d['config']['workload'], d['ms_per_step'], r['kernel_ms_mean'], r['frac'], d['parity']))"; }
for round in 1 2 3; do for w in compose_4k_fullframe compose_4k_batch; do
  MJX_HIP_LIB=$T MJX_LOWL=0 python bench.py --workload $w --steps 50 --cpu-sample 1 --pristine-launches 0 2>/dev/null | line general
  python bench.py --workload $w --steps 50 --cpu-sample 1 --pristine-launches 0 2>/dev/null | line low-column
done; done
