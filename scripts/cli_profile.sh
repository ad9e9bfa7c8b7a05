#!/bin/bash
# kernel trace of the batch CLI (GPU box): scripts/cli_profile.sh <n_images> <jobs>
cd "$(dirname "$0")/.."
R=$PWD
n=${1:-256}; jobs=${2:-8}
work=$(mktemp -d /tmp/mjprof.XXXX); mkdir -p $work/in $work/out
H=libmodjpeg_amd/host
T=tests/tools
for i in $(seq 1 $n); do
  f=$work/in/img$i.jpg
  if [ $i -le 8 ]; then BENCH_DUMP_INPUT=$f MJ_AMD=no-gpu-decode,no-gpu-encode $T/bench_dropin $H/libmodjpeg.so.1 1920 1080 16 16 1 > /dev/null; else cp $work/in/img$(( (i % 8) + 1 )).jpg $f; fi
  echo $f >> $work/list.txt
done
cp tests/golden/images/dropon.png $work/logo.png
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_cli -o cli -- $R/$H/modjpeg --batch $work/list.txt --outdir $work/out --jobs $jobs -p br -m -10,-10 -d $work/logo.png -y 10
cd $R && python3 scripts/kernel_overlap.py gpurun_out/prof_cli/cli_results.db && python3 scripts/kernel_times.py gpurun_out/prof_cli/cli_results.db
rm -rf $work
