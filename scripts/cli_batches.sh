#!/bin/bash
# batch CLI throughput against the number of decode batches in flight (GPU box)
cd "$(dirname "$0")/.."
n=${1:-4096}
work=$(mktemp -d /tmp/mjq.XXXX); mkdir -p $work/in $work/out
H=libmodjpeg_amd/host
T=tests/tools
for i in $(seq 1 16); do BENCH_DUMP_INPUT=$work/in/img$i.jpg MJ_AMD=no-gpu-decode,no-gpu-encode $T/bench_dropin $H/libmodjpeg.so.1 1920 1080 16 16 1 > /dev/null; done
for i in $(seq 1 $n); do echo $work/in/img$(( (i % 16) + 1 )).jpg >> $work/list.txt; done
cp tests/golden/images/dropon.png $work/logo.png
for b in ${BATCHES:-1 2 4 32}; do for jobs in 16 32; do
  echo -n "encode batches=$b jobs=$jobs : "
  LD_PRELOAD=$PWD/libmodjpeg_amd/libmjx_hip_faultinj.so MJX_ENCODE_BATCHES=$b $H/modjpeg --batch $work/list.txt --outdir $work/out --jobs $jobs -p br -m -10,-10 -d $work/logo.png -y 10 2>&1 | tail -1
done; done
rm -rf $work
