#!/bin/bash
# round-2 kernel experiments 2: prefetch depth x workgroup size, against the round-1 kernel
set -o pipefail
R=$PWD; O=$R/gpurun_out/r02d; mkdir -p $O
# correctness of the new kernel first (default build = d2w1)
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for R_ in A1 A2; do
  bash scripts/ab_bench.sh "--regime $R_ --cpu-sample 16 --pristine-launches 0" r01base d2w1 d1w1 d3w1 d2w4 d2w2 2>&1 | tee -a $O/ab_$R_.txt
done
bash scripts/ab_bench.sh "--regime A3 --cpu-sample 16 --pristine-launches 0" r01base d2w1 d3w1 2>&1 | tee -a $O/ab_A3.txt
bash scripts/ab_bench.sh "--regime A4 --cpu-sample 16 --pristine-launches 0" r01base d2w1 2>&1 | tee -a $O/ab_A4.txt
bash scripts/ab_bench.sh "--regime A1 --cpu-sample 0 --pristine-launches 0" d2w1nb d3w1nb 2>&1 | tee -a $O/ab_noblend.txt
for RO in 8 32 64; do echo "MJX_RUN_OCTETS=$RO"; MJX_RUN_OCTETS=$RO bash scripts/ab_bench.sh "--regime A2 --cpu-sample 16 --pristine-launches 0" d2w1 2>&1 | tee -a $O/ab_runlen.txt; done
