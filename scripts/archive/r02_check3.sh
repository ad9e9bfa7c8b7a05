#!/bin/bash
# round-2 GPU check 3: device tile preparation - parity tests, the whole drop-in / CLI suites (they now go
# through it), and what a tile-cache miss costs with and without it
set -o pipefail
O=gpurun_out/r02h; mkdir -p $O
python -m pytest tests/test_dropin_shim.py tests/test_cli.py tests/test_abi.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
H=libmodjpeg_amd/host
T=tests/tools
# end-to-end through the drop-in: bench_dropin <lib> <w> <h> <dropon w> <dropon h> <images>.  Default: one
# placement (tile cache hits); BENCH_VARY_POSITION=1: 35 placements over 8 cache slots (misses)
run() { name=$1; lib=$2; shift 2; echo "== $name"; env "$@" $T/bench_dropin $lib 1920 1080 256 256 200 2>&1 | tail -1 | cut -c1-420 | tee $O/dropin_$name.txt; }
run cache_hit $H/libmodjpeg.so.1 X=1
run miss_gpu_prep $H/libmodjpeg.so.1 BENCH_VARY_POSITION=1
run miss_host_prep $H/libmodjpeg.so.1 BENCH_VARY_POSITION=1 MJ_GPU_TILEPREP=0
run reference oracle/_ref/libmodjpeg_ref.so BENCH_VARY_POSITION=1
run miss_gpu_prep_4k $H/libmodjpeg.so.1 BENCH_VARY_POSITION=1
