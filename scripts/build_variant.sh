#!/bin/bash
# build a kernel variant for A/B experiments: scripts/build_variant.sh <name> [-DMACRO=1 ...]
# -> gpurun_variants/libmjx_<name>.so ; run with MJX_HIP_LIB=$PWD/gpurun_variants/libmjx_<name>.so python bench.py ...
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p gpurun_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -Wall "$@" \
  -Rpass-analysis=kernel-resource-usage \
  -o gpurun_variants/libmjx_$name.so libmodjpeg_amd/csrc/mjx_compose.hip libmodjpeg_amd/csrc/mjx_effects.hip libmodjpeg_amd/csrc/mjx_synth.hip libmodjpeg_amd/csrc/mjx_tileprep.hip libmodjpeg_amd/csrc/mjx_huffman.hip libmodjpeg_amd/csrc/mjx_huffdec.hip libmodjpeg_amd/csrc/mjx_runtime.hip 2>&1 \
  | grep -A9 'compose_kernel' | grep -E 'VGPRs:|Occupancy|Spill|LDS Size' | tr '\n' ' '
echo " <- $name"
