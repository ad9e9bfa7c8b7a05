#!/bin/bash
# end-of-round checkpoint after the decoder work: smoke, default bench, the entropy workload, drop-in end to end
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r03final2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > gpurun_out/r03final2/bench_default.json 2> gpurun_out/r03final2/bench_default.err; tail -c 1500 gpurun_out/r03final2/bench_default.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['value'], d['unit'], d['ms_per_step'], 'frac', d['roofline']['frac'])"
python bench.py --workload entropy_1080p_batch > gpurun_out/r03final2/bench_entropy.json 2> gpurun_out/r03final2/bench_entropy.err; python3 -c "import json; d=json.loads(open('gpurun_out/r03final2/bench_entropy.json').read().strip().splitlines()[-1]); print('entropy', d['value'], d['unit'], d['ms_per_step'])"
H=libmodjpeg_amd/host
T=tests/tools
for i in 1 2; do timeout -k 10 300 $T/bench_dropin $H/libmodjpeg.so.1 1920 1080 256 256 200 2>/dev/null | tail -1 | cut -c1-330; done
timeout -k 10 300 $T/bench_dropin $H/libmodjpeg.so.1 3840 2160 3840 2160 12 2>/dev/null | tail -1 | cut -c1-330
