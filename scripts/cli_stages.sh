#!/bin/bash
# per-stage wall time inside the batch CLI's workers against the number of threads (GPU box)
cd "$(dirname "$0")/.."
n=${1:-2048}
work=$(mktemp -d /tmp/mjq.XXXX); mkdir -p $work/in $work/out
H=libmodjpeg_amd/host
T=tests/tools
for i in $(seq 1 16); do BENCH_DUMP_INPUT=$work/in/img$i.jpg MJ_AMD=no-gpu-decode,no-gpu-encode $T/bench_dropin $H/libmodjpeg.so.1 1920 1080 16 16 1 > /dev/null; done
for i in $(seq 1 $n); do echo $work/in/img$(( (i % 16) + 1 )).jpg >> $work/list.txt; done
cp tests/golden/images/dropon.png $work/logo.png
nproc
for jobs in 1 4 8 16 32; do
  MJ_AMD=cli-timing $H/modjpeg --batch $work/list.txt --outdir $work/out --jobs $jobs -p br -m -10,-10 -d $work/logo.png -y 10 2>&1 | tail -3
done
echo "== outputs to /dev/shm"
mkdir -p /dev/shm/mjq_out
for jobs in 8 16; do
  MJ_AMD=cli-timing $H/modjpeg --batch $work/list.txt --outdir /dev/shm/mjq_out --jobs $jobs -p br -m -10,-10 -d $work/logo.png -y 10 2>&1 | tail -3
done
rm -rf /dev/shm/mjq_out $work
