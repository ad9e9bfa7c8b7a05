#!/bin/bash
# how much of the read path is the idle shader clock?  the same reads with and without a one-workgroup
# spinner (scripts/ubench/warm.hip) keeping the device "busy" for the power management (GPU box)
cd "$(dirname "$0")/.."
H=libmodjpeg_amd/host
T=tests/tools
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o gpurun_variants/warm scripts/ubench/warm.hip || exit 1
for q in 85 95; do
  echo "== 1080p quality $q, idle device"
  BENCH_QUALITY=$q $T/bench_dropin $H/libmodjpeg.so.1 1920 1080 256 256 40 | sed -e 's/.*\("read_ms.*write_ms": [0-9.]*\).*/\1/'
  echo "== 1080p quality $q, with the spinner"
  gpurun_variants/warm 6 & W=$!
  sleep 1.5
  BENCH_QUALITY=$q $T/bench_dropin $H/libmodjpeg.so.1 1920 1080 256 256 40 | sed -e 's/.*\("read_ms.*write_ms": [0-9.]*\).*/\1/'
  wait $W
done
