#!/bin/bash
# round 5, after the opaque blocks: the share of the stream runs that goes beside the blend kernel, swept (A5 / A4, float and int16
# tiles), against the library's own estimate (share 0)
# usage (GPU box): scripts/r05_beside_opaque.sh > gpurun_out/r05/beside_opaque.txt
for t in f32 i16; do for reg in A5 A4; do for share in 0 0.5 0.6 0.7 0.8 0.9 1.0; do
  python bench.py --regime $reg --tiles $t --steps 40 --also none --cpu-sample 0 --pristine-launches 0 --beside-share $share 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); p=d['config']['plan']
print('%-3s %-4s share %-4s step %.4f ms  beside %6d of %6d light runs' % ('$reg', '$t', '$share', d['ms_per_step'], p['runs_beside'], (p['zero_alpha_octets'] + p['opaque_octets'] + p['dc_only_octets'] + 3) // 4))"
done; done; done
