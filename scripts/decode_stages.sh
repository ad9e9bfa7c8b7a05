#!/bin/bash
# stage times of the GPU read path (debug output of gpudecode.c and mjx_huff_decode_host)
cd "$(dirname "$0")/.."
H=libmodjpeg_amd/host
T=tests/tools
for cfg in "1920 1080 85" "1920 1080 95" "3840 2160 85" "3840 2160 95"; do
  set -- $cfg
  echo "== $1x$2 q$3"
  BENCH_QUALITY=$3 MJ_AMD=debug LD_PRELOAD=$PWD/libmodjpeg_amd/libmjx_hip_faultinj.so MJX_HUFFDEC_DEBUG=1 $T/bench_dropin $H/libmodjpeg.so.1 $1 $2 256 256 3 2>&1 | grep -E "gpudecode: unstuff|mjx_huff_decode_host" | tail -6
done
