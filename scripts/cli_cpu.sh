#!/bin/bash
# CPU time against wall time of the batch CLI (is the host the limit?), with HIP's default spin-wait and with blocking waits
cd "$(dirname "$0")/.."
n=${1:-8192}
work=$(mktemp -d /tmp/mjq.XXXX); mkdir -p $work/in $work/out
H=libmodjpeg_amd/host
T=tests/tools
for i in $(seq 1 16); do BENCH_DUMP_INPUT=$work/in/img$i.jpg MJ_AMD=no-gpu-decode,no-gpu-encode $T/bench_dropin $H/libmodjpeg.so.1 1920 1080 16 16 1 > /dev/null; done
for i in $(seq 1 $n); do echo $work/in/img$(( (i % 16) + 1 )).jpg >> $work/list.txt; done
cp tests/golden/images/dropon.png $work/logo.png
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc
for jobs in 8 16 32; do
  for wait in spin block; do
    echo "== jobs=$jobs wait=$wait"
    ( time MJX_OPTIONS=wait=$wait $H/modjpeg --batch $work/list.txt --outdir $work/out --jobs $jobs -p br -m -10,-10 -d $work/logo.png -y 10 ) 2>&1 | grep -E "images/s|real|user|sys"
  done
done
rm -rf $work
