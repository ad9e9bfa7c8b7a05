#!/bin/bash
# batch CLI throughput against hardware queues (GPU_MAX_HW_QUEUES), staging arenas (MJX_OPTIONS=arenas=N) and
# host threads; synthetic 1080p q85 inputs from the bench harness (GPU box)
cd "$(dirname "$0")/.."
n=${1:-4096}
work=$(mktemp -d /tmp/mjq.XXXX); mkdir -p $work/in $work/out
H=libmodjpeg_amd/host
T=tests/tools
# 16 distinct files, listed round-robin (the outputs overwrite each other: this is a timing run)
for i in $(seq 1 16); do BENCH_DUMP_INPUT=$work/in/img$i.jpg MJ_AMD=no-gpu-decode,no-gpu-encode $T/bench_dropin $H/libmodjpeg.so.1 1920 1080 16 16 1 > /dev/null; done
for i in $(seq 1 $n); do echo $work/in/img$(( (i % 16) + 1 )).jpg >> $work/list.txt; done
cp tests/golden/images/dropon.png $work/logo.png
for q in 4 8; do for a in 8 16; do for jobs in 8 16 32 64; do
  echo -n "hw_queues=$q arenas=$a jobs=$jobs : "
  GPU_MAX_HW_QUEUES=$q MJX_OPTIONS=arenas=$a $H/modjpeg --batch $work/list.txt --outdir $work/out --jobs $jobs -p br -m -10,-10 -d $work/logo.png -y 10 2>&1 | tail -1
done; done; done
echo -n "libjpeg entropy coding both ways, jobs=32 : "
MJ_AMD=no-gpu-decode,no-gpu-encode $H/modjpeg --batch $work/list.txt --outdir $work/out --jobs 32 -p br -m -10,-10 -d $work/logo.png -y 10 2>&1 | tail -1
echo -n "libjpeg decode, GPU encode, jobs=32 : "
MJ_AMD=no-gpu-decode $H/modjpeg --batch $work/list.txt --outdir $work/out --jobs 32 -p br -m -10,-10 -d $work/logo.png -y 10 2>&1 | tail -1
rm -rf $work
