#!/bin/bash
# where one lane per chunk overtakes one wavefront per chunk: stage times of single reads with the rounds both ways
cd "$(dirname "$0")/.."
H=libmodjpeg_amd/host
T=tests/tools
for cfg in "1920 1080 85" "1920 1080 95" "3840 2160 85" "3840 2160 95" "1920 1080 100" "7680 4320 85"; do
  set -- $cfg
  for lim in 0 100000000; do
    echo "-- $1x$2 q$3 MJX_DECODE_WAVE_CHUNKS=$lim"
    MJX_DECODE_WAVE_CHUNKS=$lim BENCH_QUALITY=$3 MJX_HUFFDEC_DEBUG=1 timeout -k 10 120 $T/bench_dropin $H/libmodjpeg.so.1 $1 $2 256 256 12 2>&1 | grep -E "mjx_huff_decode_host" | tail -6 | grep -o "chunks\|[0-9]* chunks\|batch [0-9.]* ms" | paste -sd' '
    MJX_DECODE_WAVE_CHUNKS=$lim BENCH_QUALITY=$3 timeout -k 10 120 $T/bench_dropin $H/libmodjpeg.so.1 $1 $2 256 256 24 2>/dev/null | tail -1 | grep -o '"read_ms": [0-9.]*'
  done
done
