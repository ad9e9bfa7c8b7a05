#!/bin/bash
# small files: read time through the GPU decoder (forced) against libjpeg, to place the cost model's crossover (gpudecode.c)
cd "$(dirname "$0")/.."
H=libmodjpeg_amd/host
T=tests/tools
for cfg in "320 240 85" "640 480 85" "800 600 85" "1280 720 85" "640 480 95" "1280 720 95" "640 480 100" "1280 720 100"; do
  set -- $cfg
  g=$(MJ_GPU_DECODE_MIN_BYTES=0 BENCH_QUALITY=$3 timeout -k 10 120 $T/bench_dropin $H/libmodjpeg.so.1 $1 $2 16 16 60 2>/dev/null | tail -1 | grep -o '"read_ms": [0-9.]*')
  c=$(MJ_GPU_DECODE=0 BENCH_QUALITY=$3 timeout -k 10 120 $T/bench_dropin $H/libmodjpeg.so.1 $1 $2 16 16 60 2>/dev/null | tail -1 | grep -o '"read_ms": [0-9.]*')
  m=$(BENCH_QUALITY=$3 timeout -k 10 120 $T/bench_dropin $H/libmodjpeg.so.1 $1 $2 16 16 60 2>/dev/null | tail -1 | grep -o '"gpu_decodes": [0-9]*, \|"read_ms": [0-9.]*' | paste -sd' ')
  b=$(BENCH_QUALITY=$3 BENCH_DUMP_INPUT=/tmp/x.jpg MJ_GPU_DECODE=0 MJ_GPU_ENCODE=0 $T/bench_dropin $H/libmodjpeg.so.1 $1 $2 16 16 1 >/dev/null 2>&1; stat -c %s /tmp/x.jpg)
  echo "$1x$2 q$3 ($b bytes): device $g | libjpeg $c | the model's choice: $m"
done
