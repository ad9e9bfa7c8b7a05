#!/bin/bash
# synchronisation rounds the GPU decoder needs against quality / sampling (GPU box)
cd "$(dirname "$0")/.."
H=libmodjpeg_amd/host
T=tests/tools
for samp in 420 444 422 gray; do for q in 50 75 85 92 95 98 100; do
  echo -n "sampling $samp quality $q: "
  BENCH_SAMPLING=$samp BENCH_QUALITY=$q LD_PRELOAD=$PWD/libmodjpeg_amd/libmjx_hip_faultinj.so MJX_HUFFDEC_DEBUG=1 $T/bench_dropin $H/libmodjpeg.so.1 1920 1080 16 16 2 2>&1 | grep "bits per block" | tail -1 | sed 's/mjx_huff_decode_host: //'
done; done
