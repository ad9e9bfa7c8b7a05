#!/bin/bash
# round 5: opaque blocks (MJX_PLAN_TILES_ARE_FINAL / int16 tiles) and the stream kernels without their spill, against the tree
# before both (gpurun_variants/libmjx_head.so, built from a worktree of the previous commit).  Same box, interleaved.
# usage (on the GPU box): scripts/r05_opaque_ab.sh > gpurun_out/r05/opaque_ab.txt
line() {
python -c "
import sys, json
d=json.loads(sys.stdin.read()); r=d['roofline']; st=d['config'].get('plan', {})
print('%-6s %-4s %-4s %8.3f Gblk/s  step %.4f ms  parity %s  opaque octets %s' % ('$1', d['config']['alpha_regime'], d['config'].get('tiles', '?')[:5], d['value']/1e9, d['ms_per_step'], d['parity'], st.get('opaque_octets')))"
}
for round in 1 2; do
for regime in A1 A5 A4; do
for tiles in f32 i16; do
  MJX_HIP_LIB=$PWD/gpurun_variants/libmjx_head.so python bench.py --regime $regime --tiles $tiles --also none --steps 30 --cpu-sample 8 --tiles-not-final 2>/dev/null | line head
  python bench.py --regime $regime --tiles $tiles --also none --steps 30 --cpu-sample 8 2>/dev/null | line new
  [ $tiles = f32 ] && python bench.py --regime $regime --tiles $tiles --also none --steps 30 --cpu-sample 8 --tiles-not-final 2>/dev/null | line new-nf
done; done; done
true
