#!/bin/bash
# round-3 GPU check 4: DC-only runs (blocks whose only weight is (0,0)): parity, then every regime with and without them
set -o pipefail
O=gpurun_out/r03i; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
q() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; p=d['config']['plan']; print('   %.4f ms/step  kern %.4f  frac %.3f  runs %s (%s..%s)  octets: dc-only %s zero %s of %s  %s' % (d['ms_per_step'], r['kernel_ms_mean'], r['frac'], p['runs'], p['run_octets_min'], p['run_octets_max'], p['dc_only_octets'], p['zero_alpha_octets'], p['octets'], d['parity']))"; }
for R in A1 A2 A4 A5 A1 A5; do echo "regime $R"; timeout -k 10 200 python bench.py --regime $R --cpu-sample 8 --pristine-launches 0 2>$O/err.txt | tee $O/bench_$R.json | q; done
for R in A1 A5; do echo "regime $R keep-zero-alpha (alpha not promised final: every block streams 768 B)"; timeout -k 10 200 python bench.py --regime $R --keep-zero-alpha --cpu-sample 8 --pristine-launches 0 2>$O/err.txt | q; done
for R in A1 A5; do echo "regime $R i16"; timeout -k 10 200 python bench.py --regime $R --tiles i16 --cpu-sample 8 --pristine-launches 0 2>$O/err.txt | q; done
