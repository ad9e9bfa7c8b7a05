#!/bin/bash
# round-3 GPU check 3: whole GPU suite after the tile cache learnt int16 tiles + schedules, then A4/A5 and shard sizes with the merged runs
set -o pipefail
O=gpurun_out/r03d; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -8 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
q() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; p=d['config']['plan']; print('   %.4f ms/step  kern %.4f  frac %.3f  runs %s (%s..%s octets) union/octet %s  terms/block %s  %s' % (d['ms_per_step'], r['kernel_ms_mean'], r['frac'], p['runs'], p['run_octets_min'], p['run_octets_max'], p['mean_union_terms_per_octet'], p['mean_terms_per_block'], d['parity']))"; }
for R in A4 A5 A4 A5; do echo "regime $R"; timeout -k 10 200 python bench.py --regime $R --cpu-sample 8 --pristine-launches 0 2>$O/err.txt | tee $O/bench_$R.json | q; done
for R in A4 A5; do echo "regime $R keep-zero-alpha"; timeout -k 10 200 python bench.py --regime $R --keep-zero-alpha --cpu-sample 8 --pristine-launches 0 2>$O/err.txt | q; done
for R in A4 A5; do echo "regime $R no-group"; timeout -k 10 200 python bench.py --regime $R --no-group --cpu-sample 8 --pristine-launches 0 2>$O/err.txt | q; done
for R in A4 A5; do echo "regime $R i16"; timeout -k 10 200 python bench.py --regime $R --tiles i16 --cpu-sample 8 --pristine-launches 0 2>$O/err.txt | q; done
for IM in 512 4096 512 4096; do echo "images $IM"; timeout -k 10 200 python bench.py --images $IM --cpu-sample 8 --pristine-launches 0 --steps 100 2>$O/err.txt | q; done
