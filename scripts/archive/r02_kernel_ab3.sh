#!/bin/bash
# round-2 kernel experiments 3: what the memory system gives for this access pattern; r01 structure + increments
set -o pipefail
R=$PWD; O=$R/gpurun_out/r02e; mkdir -p $O
./scripts/ubench/stream_pattern 4096 2>&1 | tee $O/stream_pattern.txt
for R_ in A1 A2 A3; do
  bash scripts/ab_bench.sh "--regime $R_ --cpu-sample 16 --pristine-launches 0" r01base r01c r01hs d1w4 2>&1 | tee -a $O/ab_$R_.txt
done
echo "== dense: 256 x 4K full frame, A1"
bash scripts/ab_bench.sh "--workload compose_4k_fullframe --images 256 --regime A1 --cpu-sample 1 --pristine-launches 0" r01base r01hs 2>&1 | tee -a $O/ab_dense4k.txt
