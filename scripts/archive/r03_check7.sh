#!/bin/bash
# after the wave-per-chunk decoder rounds: the whole GPU suite, then the batch CLI rate with the rounds both ways
cd "$(dirname "$0")/.."
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 || exit 1
for lim in 32768 0; do
  echo "=== MJX_DECODE_WAVE_CHUNKS=$lim"
  for jobs in 1 4 16; do
    MJX_DECODE_WAVE_CHUNKS=$lim timeout -k 10 300 bash scripts/cli_batch_rate.sh 2048 $jobs 2>&1 | grep -E "^modjpeg:|differing"
  done
done
