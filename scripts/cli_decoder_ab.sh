#!/bin/bash
# batch CLI rate with the committed kernels against variant builds of libmjx_hip (gpurun_variants/libmjx_<name>.so): every variant
# gets a copy of the host tree with its library in place ($ORIGIN-relative rpaths), nothing is preloaded
# usage (GPU box): scripts/cli_decoder_ab.sh <n_images> <jobs> name...   ("none" = the library as built)
cd "$(dirname "$0")/.."
n=${1:-1024}; jobs=${2:-16}; shift 2
work=$(mktemp -d /tmp/mjcli.XXXX); mkdir -p $work/in $work/out
H=libmodjpeg_amd/host; T=tests/tools
for i in $(seq 1 $n); do
  f=$work/in/img$i.jpg
  if [ $i -le 16 ]; then BENCH_SEED=$i BENCH_DUMP_INPUT=$f MJ_AMD=no-gpu-decode,no-gpu-encode $T/bench_dropin $H/libmodjpeg.so.1 1920 1080 16 16 1 > /dev/null; else cp $work/in/img$(( (i % 16) + 1 )).jpg $f; fi
  echo $f >> $work/list.txt
done
cp tests/golden/images/dropon.png $work/logo.png
for v in "$@"; do
  mkdir -p $work/tree_$v/libmodjpeg_amd; cp -r $H $work/tree_$v/libmodjpeg_amd/host
  if [ $v = none ]; then cp libmodjpeg_amd/libmjx_hip.so $work/tree_$v/libmodjpeg_amd/; else cp gpurun_variants/libmjx_$v.so $work/tree_$v/libmodjpeg_amd/libmjx_hip.so; fi
done
echo "inputs ready"
for rep in 1 2 3; do for v in "$@"; do
  printf "%-8s jobs %-3s " $v $jobs
  timeout -k 10 120 $work/tree_$v/libmodjpeg_amd/host/modjpeg --batch $work/list.txt --outdir $work/out --jobs $jobs -p br -m -10,-10 -d $work/logo.png -y 10 2>&1 | tail -1
  echo
done; done
rm -rf $work
