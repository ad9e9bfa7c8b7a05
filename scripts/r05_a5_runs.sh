#!/bin/bash
# round 5: A5 / A4 (few blended octets per rectangle): how the step time depends on how the blended stretches are cut into runs
for RG in A5 A4; do for L in serial auto; do for cfg in "0 0" "16 -1" "8 -1" "4 -1" "16 0" "8 0" "2 -1"; do set -- $cfg
  python bench.py --regime $RG --launch $L --also none --cpu-sample 0 --steps 40 --run-octets $1 --tail-min-octets $2 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); p=d['config']['plan']
print('$RG %-6s run_octets %2s tail_min %2s  step %.4f ms  runs %d (beside %d)  run octets %d..%d' % ('$L', '$1', '$2', d['ms_per_step'], p['runs'], p['runs_beside'], p['run_octets_min'], p['run_octets_max']))"
done; done; done
