#!/bin/bash
# damaged-file soak of the GPU decoder in every round mode: whatever libjpeg makes of a broken stream, the drop-in returns the same arrays or error
cd "$(dirname "$0")/.."
H=libmodjpeg_amd/host
T=tests/tools
work=$(mktemp -d /tmp/mjsoak.XXXX)
mk() { BENCH_SAMPLING=$2 BENCH_DUMP_INPUT=$1 MJ_GPU_DECODE=0 MJ_GPU_ENCODE=0 env $5 $T/bench_dropin $H/libmodjpeg.so.1 $3 $4 16 16 1 > /dev/null 2>&1; }
mk $work/a.jpg 420 800 600 ""
mk $work/b.jpg 444 400 300 "BENCH_RESTART=7"
mk $work/c.jpg gray 640 480 "BENCH_OPTIMIZE=1"
mk $work/d.jpg 422 1280 720 "BENCH_QUALITY=96"
for mode in "" "MJX_DECODE_WAVE_CHUNKS=0" "MJX_DECODE_WAVE_CHUNKS=1 MJX_DECODE_LANE_ROUNDS=1"; do
  for f in a b c d; do
    for seed in 11 12; do
      echo -n "mode [$mode] $f.jpg seed $seed: "
      env $mode MJ_GPU_DECODE_MIN_BYTES=0 timeout -k 10 300 $T/mjhost_probe decodefuzz $work/$f.jpg ${N:-400} $seed 2>&1 | tail -1
    done
  done
done
rm -rf $work
