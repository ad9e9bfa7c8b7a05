#!/bin/bash
# round-2 GPU check 2: the whole GPU suite on the final build, the host code under UBSan through its GPU
# paths, default bench line
set -o pipefail
O=gpurun_out/r02g; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
bash scripts/ubsan_host.sh > $O/ubsan.log 2>&1; echo "ubsan rc=$?"; tail -4 $O/ubsan.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-400 $O/bench_default.json
python -c "import __graft_entry__ as g; g.smoke()"
