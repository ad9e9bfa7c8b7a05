#!/bin/bash
# mj_write_jpeg_to_memory(MJ_OPTION_OPTIMIZE): GPU (statistics + tables + encode) against libjpeg, and
# the batch CLI with -O (GPU box)
cd "$(dirname "$0")/.."
H=libmodjpeg_amd/host
T=tests/tools
for size in "1920 1080" "3840 2160"; do
  echo "== $size, write with optimised tables"
  BENCH_WRITE_OPTIONS=1 $T/bench_dropin $H/libmodjpeg.so.1 $size 256 256 20 | sed -e 's/.*\("write_ms": [0-9.]*\).*\("out_bytes.*\)/drop-in        \1 \2/'
  BENCH_WRITE_OPTIONS=1 MJ_AMD=no-gpu-encode $T/bench_dropin $H/libmodjpeg.so.1 $size 256 256 20 | sed -e 's/.*\("write_ms": [0-9.]*\).*\("out_bytes.*\)/libjpeg writes \1 \2/'
  if [ -e oracle/_ref/libmodjpeg_ref.so ]; then BENCH_WRITE_OPTIONS=1 $T/bench_dropin oracle/_ref/libmodjpeg_ref.so $size 256 256 20 | sed -e 's/.*\("write_ms": [0-9.]*\).*\("out_bytes.*\)/reference      \1 \2/'; fi
done
work=$(mktemp -d /tmp/mjq.XXXX); mkdir -p $work/in $work/out
for i in $(seq 1 16); do BENCH_DUMP_INPUT=$work/in/img$i.jpg MJ_AMD=no-gpu-decode,no-gpu-encode $T/bench_dropin $H/libmodjpeg.so.1 1920 1080 16 16 1 > /dev/null; done
for i in $(seq 1 4096); do echo $work/in/img$(( (i % 16) + 1 )).jpg >> $work/list.txt; done
cp tests/golden/images/dropon.png $work/logo.png
echo "== batch CLI, 4096 x 1080p, 16 threads, -O"
$H/modjpeg --batch $work/list.txt --outdir $work/out --jobs 16 -p br -m -10,-10 -d $work/logo.png -y 10 -O 2>&1 | tail -1
MJ_AMD=no-gpu-encode $H/modjpeg --batch $work/list.txt --outdir $work/out --jobs 16 -p br -m -10,-10 -d $work/logo.png -y 10 -O 2>&1 | tail -1 | sed -e 's/^/libjpeg writes: /'
rm -rf $work
