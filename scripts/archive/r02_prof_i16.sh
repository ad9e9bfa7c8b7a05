cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_i16; mkdir -p $O
for T in i16 f32; do
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU --output-format csv -d $O/$T -o p -- python3 $R/bench.py --cpu-sample 0 --regime A2 --tiles $T --steps 10 > /dev/null 2> $O/$T.err
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/${T}b -o p -- python3 $R/bench.py --cpu-sample 0 --regime A2 --tiles $T --steps 10 > /dev/null 2>> $O/$T.err
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for t in ("i16","i16b","f32","f32b"):
    acc=collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/prof_i16/{t}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "compose_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(t, {k:"%.4g"%(sum(v)/len(v)) for k,v in sorted(acc.items())})
PY
