#!/bin/bash
# wave-per-chunk synchronisation rounds of the entropy decoder against lane-per-chunk: parity tests under both, then stage times
cd "$(dirname "$0")/.."
H=libmodjpeg_amd/host
T=tests/tools
for lim in 0 100000000; do
  echo "=== MJX_DECODE_WAVE_CHUNKS=$lim"
  MJX_DECODE_WAVE_CHUNKS=$lim timeout -k 10 600 python -m pytest tests/test_gpu_huffdec.py tests/test_dropin_shim.py -m gpu -x -q -k "huff or decoder or entropy" 2>&1 | tail -3 || exit 1
done
for lim in 0 100000000; do
  echo "=== MJX_DECODE_WAVE_CHUNKS=$lim"
  for cfg in ${CFGS:-"1920 1080 85" "1920 1080 95" "3840 2160 85" "3840 2160 95" "1920 1080 100"}; do
    set -- $cfg
    echo "-- $1x$2 q$3"
    MJX_DECODE_WAVE_CHUNKS=$lim BENCH_QUALITY=$3 MJX_HUFFDEC_DEBUG=1 timeout -k 10 120 $T/bench_dropin $H/libmodjpeg.so.1 $1 $2 256 256 12 2>&1 | grep -E "mjx_huff_decode_host" | tail -4 | sed 's/mjx_huff_decode_host: //'
    MJX_DECODE_WAVE_CHUNKS=$lim BENCH_QUALITY=$3 timeout -k 10 120 $T/bench_dropin $H/libmodjpeg.so.1 $1 $2 256 256 40 2>/dev/null | tail -1 | grep -o '"read_ms": [0-9.]*'
  done
done
