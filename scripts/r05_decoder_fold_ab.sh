#!/bin/bash
# round 5: the decoder's passes after the synchronisation rounds folded from seven launches to four (each prefix scan one launch, the
# DC differences written by the store pass): read stage of the drop-in, before (gpurun_variants/libmjx_<before>.so) against the
# library as built, same box, alternating; then the batch CLI rate
# usage (GPU box): scripts/r05_decoder_fold_ab.sh <before-variant>
cd "$(dirname "$0")/.."
before=${1:-new}
work=$(mktemp -d /tmp/mjdec.XXXX)
H=libmodjpeg_amd/host; T=tests/tools
for v in $before none; do
  mkdir -p $work/tree_$v/libmodjpeg_amd; cp -r $H $work/tree_$v/libmodjpeg_amd/host
  if [ $v = none ]; then cp libmodjpeg_amd/libmjx_hip.so $work/tree_$v/libmodjpeg_amd/; else cp gpurun_variants/libmjx_$v.so $work/tree_$v/libmodjpeg_amd/libmjx_hip.so; fi
done
for cfg in "1920 1080 85" "3840 2160 85" "640 480 85"; do set -- $cfg
  for rep in 1 2 3 4; do for v in $before none; do
    printf "%-6s %sx%s q%s  " $v $1 $2 $3
    BENCH_QUALITY=$3 $T/bench_dropin $work/tree_$v/libmodjpeg_amd/host/libmodjpeg.so.1 $1 $2 256 256 ${IMAGES:-200} 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('read %.4f ms  compose %.4f  write %.4f  (%d images, %s)' % (d['read_ms'], d['compose_ms'], d['write_ms'], d['images'], d['fnv1a']))"
  done; done
done
rm -rf $work
scripts/cli_decoder_ab.sh 1024 16 $before none
