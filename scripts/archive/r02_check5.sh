#!/bin/bash
# round-2 GPU check 5: Y image 16/120 with swapped halves - parity, LDS conflict counters per regime, timing
set -o pipefail
R=$PWD; O=$R/gpurun_out/r02j; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
for RG in A1 A2 A3; do
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d $O/sq_$RG -o p -- python3 $R/bench.py --cpu-sample 0 --regime $RG --steps 10 > $O/sq_$RG.json 2> $O/sq_$RG.err || echo "pmc $RG failed"
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for rg in ("A1","A2","A3"):
    acc=collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/r02j/sq_{rg}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "compose_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m={k:sum(v)/len(v) for k,v in acc.items()}
    print(rg, {k:"%.4g"%v for k,v in m.items()}, "conflict/active = %.3f" % (m.get("SQ_LDS_BANK_CONFLICT",0)/max(m.get("SQ_LDS_IDX_ACTIVE",1),1)))
PY
for RG in A1 A2 A3 A4; do python bench.py --regime $RG --cpu-sample 16 --pristine-launches 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$RG', '%.3f ms'%d['roofline']['kernel_ms_mean'], '%.3f'%d['roofline']['frac'], d['parity'])"; done
