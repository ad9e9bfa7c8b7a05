#!/bin/bash
# batch CLI rate against the way the decoder's synchronisation rounds are walked (MJX_OPTIONS=decode-rounds=...) and the thread count
# usage (GPU box): scripts/cli_rounds.sh <n_images> "<jobs list>"
cd "$(dirname "$0")/.."
n=${1:-1024}; jobs_list=${2:-"16"}
work=$(mktemp -d /tmp/mjcli.XXXX); mkdir -p $work/in $work/out
H=libmodjpeg_amd/host; T=tests/tools
for i in $(seq 1 $n); do
  f=$work/in/img$i.jpg
  if [ $i -le 16 ]; then BENCH_SEED=$i BENCH_DUMP_INPUT=$f MJ_AMD=no-gpu-decode,no-gpu-encode $T/bench_dropin $H/libmodjpeg.so.1 1920 1080 16 16 1 > /dev/null; else cp $work/in/img$(( (i % 16) + 1 )).jpg $f; fi
  echo $f >> $work/list.txt
done
cp tests/golden/images/dropon.png $work/logo.png
for jobs in $jobs_list; do for mode in "" wave lane lane:3 lane:6; do for rep in 1 2; do
  printf "%-8s jobs %-3s " "${mode:-default}" $jobs
  MJX_OPTIONS=decode-rounds=$mode $H/modjpeg --batch $work/list.txt --outdir $work/out --jobs $jobs -p br -m -10,-10 -d $work/logo.png -y 10 2>&1 | tail -1
done; done; done
rm -rf $work
