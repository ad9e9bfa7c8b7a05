#!/bin/bash
# round 5: the two halves of the decoder fold on their own.  d = the store pass writes the DC differences (no gather kernel),
# s = each prefix scan is one launch (the last workgroup makes the totals); variants d0s0 (seven launches, as round 4) .. d1s1 (four)
# usage (GPU box): scripts/r05_decoder_fold_halves.sh   (variants: scripts/build_variant.sh d1s0 -DMJX_DEC_STORE_DIFF=1 -DMJX_DEC_FUSED_SCAN=0 ...)
cd "$(dirname "$0")/.."
work=$(mktemp -d /tmp/mjdec.XXXX)
H=libmodjpeg_amd/host; T=tests/tools
for v in d0s0 d1s0 d0s1 d1s1; do
  mkdir -p $work/tree_$v/libmodjpeg_amd; cp -r $H $work/tree_$v/libmodjpeg_amd/host
  cp gpurun_variants/libmjx_$v.so $work/tree_$v/libmodjpeg_amd/libmjx_hip.so
done
for cfg in "1920 1080 85" "3840 2160 85"; do set -- $cfg
  for rep in 1 2 3; do for v in d0s0 d1s0 d0s1 d1s1; do
    printf "%-6s %sx%s q%s  " $v $1 $2 $3
    BENCH_QUALITY=$3 $T/bench_dropin $work/tree_$v/libmodjpeg_amd/host/libmodjpeg.so.1 $1 $2 256 256 ${IMAGES:-150} 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('read %.4f ms  (%d images, %s)' % (d['read_ms'], d['images'], d['fnv1a']))"
  done; done
done
rm -rf $work
