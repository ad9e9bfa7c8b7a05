#!/bin/bash
# the drop-in's host C code under UBSan (-fsanitize=undefined, trapping) through its GPU paths:
# build a sanitised libmodjpeg.so.1 + probe + CLI next to the normal ones and run the concurrency
# stress, the decoder comparison/fuzz and a device-resident batch.  (ASan cannot map its shadow
# next to the HIP runtime on this image; GPU-side sanitizers are not available on the pool.)
set -e
cd "$(dirname "$0")/.."
R=$PWD; H=$R/libmodjpeg_amd/host; O=$R/gpurun_variants/ubsan; mkdir -p $O
SAN="-O1 -g -fsanitize=undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer"
( cd $H && gcc $SAN -fPIC -std=gnu99 -I/opt/conda/include -DWITH_LIBPNG -I/opt/conda/include/libpng16 -shared -Wl,-soname,libmodjpeg.so.1 -o $O/libmodjpeg.so.1 \
    jpegio.c dropon.c tiles.c tilecache.c compose.c effects.c gpuencode.c gpudecode.c shadow.c options.c -L/opt/conda/lib -Wl,-rpath,/opt/conda/lib -ljpeg -L$R/libmodjpeg_amd -Wl,-rpath,$R/libmodjpeg_amd -lmjx_hip -lm -lpthread -lpng16
  ln -sf libmodjpeg.so.1 $O/libmodjpeg.so
  for p in $R/tests/tools/probe.c modjpeg.c $R/tests/tools/bench_dropin.c; do
    gcc $SAN -std=gnu99 -I$R/include -I/opt/conda/include -o $O/$(basename $p .c) $p -L$O -Wl,-rpath,$O -Wl,--allow-shlib-undefined -lmodjpeg -lpthread -ldl -lm -L/opt/conda/lib -Wl,-rpath,/opt/conda/lib -ljpeg
  done )
I=tests/golden/images
BENCH_DUMP_INPUT=$O/in.jpg MJ_AMD=no-gpu-decode,no-gpu-encode $O/bench_dropin $O/libmodjpeg.so.1 800 600 16 16 1 > /dev/null
export MJ_AMD=gpu-decode-min-bytes=0
$O/probe stress $O/in.jpg $I/dropon.png 16 20
MJ_AMD=gpu-decode-min-bytes=0,device-resident $O/probe stress $O/in.jpg $I/dropon.png 16 20
$O/probe decodecmp $O/in.jpg
$O/probe decodefuzz $O/in.jpg 150 5
for i in $(seq 1 64); do echo $O/in.jpg; done > $O/list.txt; mkdir -p $O/out
$O/modjpeg --batch $O/list.txt --outdir $O/out --jobs 8 -p br -m -10,-10 -d $I/dropon.png -y 10 -x
$O/modjpeg --batch $O/list.txt --outdir $O/out --jobs 8 -g -d $I/dropon.jpg,$I/mask.jpg -O
$O/bench_dropin $O/libmodjpeg.so.1 641 479 64 64 4 5 3 -2 | cut -c1-120
echo "ubsan: all clean"
