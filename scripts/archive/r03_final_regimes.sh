#!/bin/bash
# round-3 closing lines of every regime with the final kernels (one call, one box) -> profiles/r03/final_bench/
set -o pipefail
O=gpurun_out/r03final; mkdir -p $O
q() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('   value %.4g %s  %.4f ms/step  frac %s (of the bytes the plan moves: %s B per block)  at the nominal bytes per block %s  %s' % (d['value'], d['unit'], d['ms_per_step'], r.get('frac'), r.get('bytes_per_block_of_this_plan'), r.get('frac_at_nominal_bytes_per_block'), d.get('parity')))"; }
echo "== default: python bench.py"; timeout -k 10 300 python bench.py 2>$O/default.err | tee $O/bench_default.json | q
for R in A1 A2 A3 A4 A5; do echo "== regime $R"; timeout -k 10 300 python bench.py --regime $R --cpu-sample 64 2>/dev/null | tee $O/bench_$R.json | q; done
for R in A1 A5; do echo "== regime $R without the promise that alpha is final (--keep-zero-alpha)"; timeout -k 10 300 python bench.py --regime $R --keep-zero-alpha --cpu-sample 16 2>/dev/null | tee $O/bench_${R}_alpha_not_final.json | q; done
for R in A4 A5; do echo "== regime $R consecutive octets (--no-group)"; timeout -k 10 300 python bench.py --regime $R --no-group --cpu-sample 16 2>/dev/null | tee $O/bench_${R}_consecutive.json | q; done
for R in A1 A2 A5; do echo "== regime $R, int16 tiles"; timeout -k 10 300 python bench.py --regime $R --tiles i16 --cpu-sample 64 2>/dev/null | tee $O/bench_${R}_i16.json | q; done
echo "== 512 images (the 8-GPU shard) and 4096, interleaved"; for rep in 1 2; do for IM in 512 4096; do timeout -k 10 200 python bench.py --images $IM --cpu-sample 8 --pristine-launches 0 --steps 100 2>/dev/null | q; done; done
echo "== 4 ranks on one GPU"; MJX_BENCH_DEVICES=0,0,0,0 timeout -k 10 400 python bench.py --gpus 4 --cpu-sample 4 2>$O/r4.err | tee $O/bench_4ranks_strong_on_one_gpu.json | q
H=libmodjpeg_amd/host; T=tests/tools
echo "== drop-in end to end, 1080p + 256x256 overlay: tile-cache hits / misses"
timeout -k 10 300 $T/bench_dropin $H/libmodjpeg.so.1 1920 1080 256 256 200 2>/dev/null | tail -1 | cut -c1-330
BENCH_VARY_POSITION=1 timeout -k 10 300 $T/bench_dropin $H/libmodjpeg.so.1 1920 1080 256 256 200 2>/dev/null | tail -1 | cut -c1-330
