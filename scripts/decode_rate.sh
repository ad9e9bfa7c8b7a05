#!/bin/bash
# read/compose/write stage times of the drop-in with the entropy decoder on the GPU and in libjpeg,
# and of the reference library, on the bench harness's synthetic JPEGs (run on the GPU box)
set -e
cd "$(dirname "$0")/.."
H=libmodjpeg_amd/host
T=tests/tools
for size in "1920 1080" "3840 2160"; do
  for q in 85 95; do
    echo "== $size quality $q"
    BENCH_QUALITY=$q $T/bench_dropin $H/libmodjpeg.so.1 $size 256 256 20
    BENCH_QUALITY=$q MJ_AMD=no-gpu-decode $T/bench_dropin $H/libmodjpeg.so.1 $size 256 256 20
    if [ -e oracle/_ref/libmodjpeg_ref.so ]; then BENCH_QUALITY=$q $T/bench_dropin oracle/_ref/libmodjpeg_ref.so $size 256 256 20; fi
  done
done
BENCH_DUMP_INPUT=gpurun_out/in1080.jpg MJ_AMD=no-gpu-decode $T/bench_dropin $H/libmodjpeg.so.1 1920 1080 16 16 1 > /dev/null
LD_PRELOAD=$PWD/libmodjpeg_amd/libmjx_hip_faultinj.so MJX_HUFFDEC_DEBUG=1 $T/mjhost_probe decodecmp gpurun_out/in1080.jpg
$T/mjhost_probe decodefuzz gpurun_out/in1080.jpg 200 3
