#!/bin/bash
# hybrid decoder rounds (lane per chunk first, wavefront per chunk after): parity in every mode, single reads by size, batch CLI rates
cd "$(dirname "$0")/.."
H=libmodjpeg_amd/host
T=tests/tools
for mode in "" "MJX_DECODE_WAVE_CHUNKS=0" "MJX_DECODE_WAVE_CHUNKS=1" "MJX_DECODE_WAVE_CHUNKS=1 MJX_DECODE_LANE_ROUNDS=1"; do
  echo "=== parity, $mode"
  env $mode timeout -k 10 600 python -m pytest tests/test_gpu_huffdec.py tests/test_dropin_shim.py -m gpu -x -q -k "huff or decoder or entropy" 2>&1 | tail -2 || exit 1
done
for cfg in "1920 1080 85" "3840 2160 95" "7680 4320 85"; do
  set -- $cfg
  for mode in "MJX_DECODE_WAVE_CHUNKS=0" "MJX_DECODE_WAVE_CHUNKS=100000000" "MJX_DECODE_WAVE_CHUNKS=1 MJX_DECODE_LANE_ROUNDS=2" "MJX_DECODE_WAVE_CHUNKS=1 MJX_DECODE_LANE_ROUNDS=3" "MJX_DECODE_WAVE_CHUNKS=1 MJX_DECODE_LANE_ROUNDS=4"; do
    echo "-- $1x$2 q$3 $mode"
    env $mode BENCH_QUALITY=$3 MJX_HUFFDEC_DEBUG=1 timeout -k 10 120 $T/bench_dropin $H/libmodjpeg.so.1 $1 $2 256 256 8 2>&1 | grep -E "batch of" | tail -4 | grep -o "finished [0-9.]* ms" | paste -sd' '
  done
done
for mode in "MJX_DECODE_WAVE_CHUNKS=0" "" "MJX_DECODE_LANE_ROUNDS=2" "MJX_DECODE_LANE_ROUNDS=4"; do
  echo "=== batch CLI, $mode"
  for jobs in 4 16; do
    env $mode timeout -k 10 300 bash scripts/cli_batch_rate.sh 2048 $jobs 2>&1 | grep -E "^modjpeg:|differing"
  done
done
