#!/bin/bash
# round-3 GPU check 6: the lean kernel for zero-alpha / DC-only runs: parity, then A1 / A4 / A5 against its run length
set -o pipefail
O=gpurun_out/r03k; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
q() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; p=d['config']['plan']; print('   %.4f ms/step  kern %.4f  frac %.3f  runs %s  octets: dc-only %s zero %s of %s  %s' % (d['ms_per_step'], r['kernel_ms_mean'], r['frac'], p['runs'], p['dc_only_octets'], p['zero_alpha_octets'], p['octets'], d['parity']))"; }
for R in A1 A5 A4; do for LR in 8 4 2 16 8; do echo "regime $R MJX_LIGHT_RUN=$LR"; MJX_LIGHT_RUN=$LR timeout -k 10 200 python bench.py --regime $R --cpu-sample 8 --pristine-launches 0 2>$O/err.txt | q; done; done
for R in A1 A5; do echo "regime $R i16"; timeout -k 10 200 python bench.py --regime $R --tiles i16 --cpu-sample 8 --pristine-launches 0 2>$O/err.txt | q; done
echo "regime A2"; timeout -k 10 200 python bench.py --cpu-sample 8 --pristine-launches 0 2>$O/err.txt | q
