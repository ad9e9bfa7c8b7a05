#!/bin/bash
# round-3 closing measurements (one GPU box, one call): the default bench line, the strong-split shard sizes
# interleaved, N-rank rehearsals on the one GPU (4 and 6 ranks: the pool's process guard stops at 6), every regime,
# the 4K workloads, effects, the drop-in end to end (tile cache hit / miss, 4K full frame) and the batch CLI with the
# DC effects folded into the encoder and as a pass of their own.
set -o pipefail
O=gpurun_out/r03final; mkdir -p $O
q() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('   value %.4g %s  %.4f ms/step  kern %s  frac %s  %s' % (d['value'], d['unit'], d['ms_per_step'], r.get('kernel_ms_mean'), r.get('frac'), d.get('parity')))"; }
echo "== default: python bench.py"; timeout -k 10 300 python bench.py 2>$O/default.err | tee $O/bench_default.json | q
echo "== shard sizes of the strong split, interleaved (same box, same call)"
for rep in 1 2 3; do for IM in 4096 512 2048 1024; do echo "images $IM"; timeout -k 10 200 python bench.py --images $IM --cpu-sample 8 --pristine-launches 0 --steps 100 2>/dev/null | tee $O/bench_images_${IM}_rep$rep.json | q; done; done
echo "== 4 ranks on one GPU"; MJX_BENCH_DEVICES=0,0,0,0 timeout -k 10 400 python bench.py --gpus 4 --cpu-sample 4 2>$O/r4.err | tee $O/bench_4ranks_strong_on_one_gpu.json | q
# (under torch.distributed.run the launcher itself holds the GPU open: 5 ranks + launcher = the guard's 6 processes)
echo "== 5 ranks on one GPU, launched by torch.distributed.run as the driver does"; MJX_BENCH_DEVICES=0,0,0,0,0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 5 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 5 --cpu-sample 4 2>$O/r5.err | tee $O/bench_5ranks_strong_on_one_gpu_torchrun.json | q
echo "== 6 ranks on one GPU, started by bench.py itself"; MJX_BENCH_DEVICES=0,0,0,0,0,0 timeout -k 10 400 python bench.py --gpus 6 --cpu-sample 4 2>$O/r6.err | tee $O/bench_6ranks_strong_on_one_gpu.json | q
for R in A1 A2 A3 A4 A5; do echo "== regime $R"; timeout -k 10 300 python bench.py --regime $R --cpu-sample 64 2>/dev/null | tee $O/bench_$R.json | q; done
for R in A2 A5; do echo "== regime $R, int16 tiles"; timeout -k 10 300 python bench.py --regime $R --tiles i16 --cpu-sample 64 2>/dev/null | tee $O/bench_${R}_i16.json | q; done
echo "== one 4K image, full-frame dropon"; timeout -k 10 300 python bench.py --workload compose_4k_fullframe --steps 500 2>/dev/null | tee $O/bench_4k_fullframe.json | q
echo "== 256 4K images, full-frame dropons"; timeout -k 10 400 python bench.py --workload compose_4k_batch --steps 20 2>/dev/null | tee $O/bench_4k_batch.json | q
echo "== effects over 4096 4K images"; timeout -k 10 500 python bench.py --workload effects_4k_batch --steps 10 2>/dev/null | tee $O/bench_effects.json | q
echo "== entropy coders"; timeout -k 10 300 python bench.py --workload entropy_1080p_batch 2>/dev/null | tee $O/bench_entropy.json | q
H=libmodjpeg_amd/host; T=tests/tools
echo "== drop-in end to end, 1080p + 256x256 overlay: tile-cache hits / misses / the reference library"
timeout -k 10 300 $T/bench_dropin $H/libmodjpeg.so.1 1920 1080 256 256 200 2>/dev/null | tail -1 | cut -c1-330
BENCH_VARY_POSITION=1 timeout -k 10 300 $T/bench_dropin $H/libmodjpeg.so.1 1920 1080 256 256 200 2>/dev/null | tail -1 | cut -c1-330
[ -e oracle/_ref/libmodjpeg_ref.so ] && timeout -k 10 300 $T/bench_dropin oracle/_ref/libmodjpeg_ref.so 1920 1080 256 256 40 2>/dev/null | tail -1 | cut -c1-330
echo "== drop-in end to end, 4K + full-frame overlay: hits / misses / reference"
timeout -k 10 300 $T/bench_dropin $H/libmodjpeg.so.1 3840 2160 3840 2160 12 2>/dev/null | tail -1 | cut -c1-330
BENCH_VARY_POSITION=1 timeout -k 10 300 $T/bench_dropin $H/libmodjpeg.so.1 3840 2160 3840 2160 12 2>/dev/null | tail -1 | cut -c1-330
[ -e oracle/_ref/libmodjpeg_ref.so ] && BENCH_VARY_POSITION=1 timeout -k 10 300 $T/bench_dropin oracle/_ref/libmodjpeg_ref.so 3840 2160 3840 2160 4 2>/dev/null | tail -1 | cut -c1-330
echo "== batch CLI, 2048 files, 16 threads: DC effects folded into the encoder / as their own pass"
timeout -k 10 500 bash scripts/cli_batch_rate.sh 2048 16 2>&1 | tail -6
MJ_FOLD_EFFECTS=0 timeout -k 10 500 bash scripts/cli_batch_rate.sh 2048 16 2>&1 | grep -A1 "drop-in" | tail -2
