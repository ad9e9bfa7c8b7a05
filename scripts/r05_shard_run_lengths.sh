#!/bin/bash
# the shards of the 8-GPU strong split on one GPU against the run length (mjx_plan_options_t.run_octets)
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/r05
out=gpurun_out/r05/shard_run_lengths.txt; : > $out
one() { python bench.py --images $1 --steps 200 --also none --cpu-sample 4 --pristine-launches 0 $2 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); r=d['roofline']; p=d['config']['plan']
print('images %-5s %-18s step %.4f ms  frac %.3f  runs %6d  run_octets %d..%d  parity %s' % ('$1', '$2', r['kernel_ms_mean'], r['frac'], p['runs'], p['run_octets_min'], p['run_octets_max'], d['parity']))" >> $out; }
for round in 1 2; do
for ro in "" "--run-octets 4" "--run-octets 6" "--run-octets 8" "--run-octets 12" "--run-octets 16"; do one 512 "$ro"; done
done
for ro in "" "--run-octets 8" "--run-octets 12" "--run-octets 16"; do one 1024 "$ro"; one 2048 "$ro"; done
cat $out
