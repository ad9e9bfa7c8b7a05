#!/bin/bash
# per-file stage times of the batch CLI (MJ_CLI_TIMING) at 1 and 16 threads on 512 copies of 8 synthetic 1080p files
cd "$(dirname "$0")/.."
R=$PWD
H=libmodjpeg_amd/host
T=tests/tools
work=$(mktemp -d /tmp/mjst.XXXX); mkdir -p $work/in $work/out
for i in $(seq 1 ${NFILES:-512}); do
  f=$work/in/img$i.jpg
  if [ $i -le 8 ]; then BENCH_DUMP_INPUT=$f MJ_GPU_DECODE=0 MJ_GPU_ENCODE=0 $T/bench_dropin $H/libmodjpeg.so.1 1920 1080 16 16 1 > /dev/null; else cp $work/in/img$(( (i % 8) + 1 )).jpg $f; fi
  echo $f >> $work/list.txt
done
cp tests/golden/images/dropon.png $work/logo.png
for jobs in ${JOBS:-1 16}; do
  echo "== $jobs threads ${ENVS}"
  env ${ENVS} MJ_CLI_TIMING=1 $H/modjpeg --batch $work/list.txt --outdir $work/out --jobs $jobs -p br -m -10,-10 -d $work/logo.png -y 10 2>&1 | tail -3
done
rm -rf $work
