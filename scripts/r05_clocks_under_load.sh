#!/bin/bash
# what the GPU's clocks, power and temperature are while the headline launch (A2), the dense launch (A3) and the effects pass run:
# boxes of the pool measure the headline between 0.74 and 0.83 ms with one library
for cfg in "--regime A2" "--regime A3" "--workload effects_4k_batch"; do
  echo "== bench.py $cfg"
  python bench.py $cfg --also none --cpu-sample 0 --steps $([ "$cfg" = "--workload effects_4k_batch" ] && echo 700 || echo 30000) --pristine-launches 0 > /tmp/clk_bench.json 2>/dev/null &
  pid=$!
  sleep 14
  for i in 1 2 3; do
    rocm-smi --showclocks --showpower --showtemp --showperflevel 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Power|Temperature \(Sensor (edge|junction|memory)|Performance" | sed 's/^/   /' | tr -s ' ' | head -12
    echo "   --"
    sleep 1
  done
  wait $pid
  python -c "
import json; d=json.loads(open('/tmp/clk_bench.json').read().strip().splitlines()[-1]); print('   ms_per_step %.4f' % d['ms_per_step'])"
done
