#!/bin/bash
# round-2 kernel experiments 4: final structure (depth 1, 4 waves/workgroup, half-swapped landing, Y image,
# gather mode) against the round-1 kernel; density grouping on A4; memory ceiling for the read/write mix
set -o pipefail
R=$PWD; O=$R/gpurun_out/r02f; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
./scripts/ubench/stream_pattern 4096 2>&1 | tee $O/stream_pattern.txt
for R_ in A1 A2 A3; do
  bash scripts/ab_bench.sh "--regime $R_ --cpu-sample 16 --pristine-launches 0" r01base n1 n1sz n2 2>&1 | tee -a $O/ab_$R_.txt
done
bash scripts/ab_bench.sh "--regime A4 --cpu-sample 16 --pristine-launches 0" r01base n2 2>&1 | tee -a $O/ab_A4_grouped.txt
bash scripts/ab_bench.sh "--regime A4 --cpu-sample 16 --pristine-launches 0 --no-group" n2 2>&1 | tee -a $O/ab_A4_nogroup.txt
