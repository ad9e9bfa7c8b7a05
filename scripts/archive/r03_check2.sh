#!/bin/bash
# round-3 GPU check 2: the restructured run table (self-contained runs, quantisation rows by LDS-DMA, pattern schedules,
# zero-alpha runs, shrinking tail): parity first, then shard sizes against MJX_TAIL_MIN
set -o pipefail
O=gpurun_out/r03c; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
q() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('   %.4f ms/step  kern %.4f  frac %.3f  runs %s..%s %s' % (d['ms_per_step'], r['kernel_ms_mean'], r['frac'], d['config']['plan']['run_octets_min'], d['config']['plan']['run_octets_max'], d['parity']))"; }
for IM in 512 1024 4096; do for TM in 0 4 2 0 4; do echo "images $IM MJX_TAIL_MIN=$TM"; MJX_TAIL_MIN=$TM timeout -k 10 200 python bench.py --images $IM --cpu-sample 8 --pristine-launches 0 --steps 100 2>$O/err.txt | q; done; done
for R in A1 A3 A4 A5; do echo "regime $R"; timeout -k 10 200 python bench.py --regime $R --cpu-sample 8 --pristine-launches 0 2>$O/err.txt | tee $O/bench_$R.json | q; done
echo "A4 keep-zero-alpha"; timeout -k 10 200 python bench.py --regime A4 --keep-zero-alpha --cpu-sample 8 --pristine-launches 0 2>$O/err.txt | q
echo "A5 keep-zero-alpha"; timeout -k 10 200 python bench.py --regime A5 --keep-zero-alpha --cpu-sample 8 --pristine-launches 0 2>$O/err.txt | q
echo "4k fullframe"; timeout -k 10 200 python bench.py --workload compose_4k_fullframe --cpu-sample 1 --pristine-launches 0 --steps 500 2>$O/err.txt | q
