#!/bin/bash
# round-3 GPU check 5: run length against the regime with the round-3 prologue (one descriptor read, quantisation rows by DMA)
set -o pipefail
q() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; p=d['config']['plan']; print('   %.4f ms/step  kern %.4f  frac %.3f  runs %s (%s..%s)' % (d['ms_per_step'], r['kernel_ms_mean'], r['frac'], p['runs'], p['run_octets_min'], p['run_octets_max']))"; }
for R in A2 A1; do for RO in 16 8 4 16 8 4 12 6; do echo "regime $R MJX_RUN_OCTETS=$RO"; MJX_RUN_OCTETS=$RO timeout -k 10 200 python bench.py --regime $R --keep-zero-alpha --cpu-sample 8 --pristine-launches 0 2>/dev/null | q; done; done
