#!/bin/bash
# AddressSanitizer + UBSan over the CPU-only code: runs in the build container, no GPU.
#   1. the oracle (oracle/mjx_oracle.c, mjx_synth.c) under the whole CPU oracle test file, with the
#      sanitizer runtime preloaded into python
#   2. the drop-in's host C code (tile preparation, JPEG in/out, overlay readers, tile cache, CLI
#      option parsing) linked against a stub of include/mjx.h that answers "no device" — the
#      library's own behaviour on a machine without a GPU — so that every path that does not
#      need the device runs to completion and every path that does stops at MJ_ERR_MEMORY
#   3. the same host code on SIMULATED devices (MJX_STUB_DEVICES=2/3): batch CLI sharded over two
#      devices, tile cache and buffer pools under a 3-device stress, every cross-device hand-over aborts
# GPU-side sanitizers are not available on the pool, and ASan cannot map its shadow next to the
# HIP runtime on the GPU box; the GPU paths of the host code run under UBSan there (ubsan_host.sh).
# (libjpeg/libpng are linked by path: a -L/opt/conda/lib would pick up conda's older libasan)
set -e
cd "$(dirname "$0")/.."
R=$PWD; H=$R/libmodjpeg_amd/host; O=$R/gpurun_variants/asan; mkdir -p $O/out
SAN="-O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer"
ASAN_RT=$(gcc -print-file-name=libasan.so)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1

echo "== 1. oracle under ASan+UBSan: tests/test_oracle_golden.py"
gcc $SAN -fPIC -ffp-contract=off -std=c99 -pthread -shared -o $O/libmjx_oracle.so oracle/mjx_oracle.c oracle/mjx_synth.c -lm
MJX_ORACLE_SO=$O/libmjx_oracle.so LD_PRELOAD=$ASAN_RT python -m pytest tests/test_oracle_golden.py -x -q -p no:cacheprovider 2>&1 | tail -2

echo "== 2. host C code against a no-device stub of mjx.h"
# tests/tools/mjx_stub.c (tracked): with MJX_STUB_DEVICES unset it answers "no device" to everything
gcc $SAN -fPIC -std=gnu99 -I$R/include -shared -o $O/libmjx_hip.so tests/tools/mjx_stub.c -Wno-unused-parameter
( cd $H && gcc $SAN -fPIC -std=gnu99 -Wall -I/opt/conda/include -DWITH_LIBPNG -I/opt/conda/include/libpng16 -shared -Wl,-soname,libmodjpeg.so.1 -o $O/libmodjpeg.so.1 \
    jpegio.c dropon.c tiles.c tilecache.c compose.c effects.c gpuencode.c gpudecode.c shadow.c options.c /opt/conda/lib/libjpeg.so /opt/conda/lib/libpng16.so -Wl,-rpath,/opt/conda/lib -L$O -Wl,-rpath,$O -lmjx_hip -lm -lpthread
  ln -sf libmodjpeg.so.1 $O/libmodjpeg.so
  for p in $R/tests/tools/probe.c modjpeg.c example.c; do
    gcc $SAN -std=gnu99 -I$R/include -I/opt/conda/include -o $O/$(basename $p .c) $p -L$O -Wl,-rpath,$O -Wl,--allow-shlib-undefined -lmodjpeg -lpthread -ldl -lm /opt/conda/lib/libjpeg.so -Wl,-rpath,/opt/conda/lib
  done )
I=tests/golden/images
export MJ_AMD=no-gpu-decode,no-gpu-encode
# tile preparation for every colour space / offset / crop the parity test uses, plus odd sizes
python3 - > $O/raw.bin <<'PY'
import sys, random
random.seed(5); sys.stdout.buffer.write(bytes(random.randrange(256) for _ in range(4 * 67 * 45)))
PY
for cs in 1 2 5 6; do for off in "0 0" "5 3" "15 15"; do   # RGB, RGBA, YCC, YCCA
  $O/probe tiles $I/image.jpg $O/raw.bin 67 45 $cs 255 $off 0 0 67 45 $O/out/tiles.bin
  $O/probe tiles $I/image.jpg $O/raw.bin 67 45 $cs 128 $off 10 7 40 30 $O/out/tiles.bin
done; done
for cs in 3 4; do  # a grayscale overlay on a colour image is refused by tile preparation, as in the reference: exit code 30 + MJ_ERR_UNSUPPORTED_COLORSPACE
  $O/probe tiles $I/image.jpg $O/raw.bin 67 45 $cs 255 0 0 0 0 67 45 $O/out/tiles.bin && exit 1 || test $? -eq 36
done
$O/probe roundtrip $I/image.jpg $O/out/rt.bin
$O/probe roundtrip $I/image_dropon.jpg $O/out/rt.bin
$O/probe nodevice $I/image.jpg
# CLI: everything that needs no device runs; compose/effects stop with the library's error
$O/modjpeg -i $I/image.jpg -O -o $O/out/a.jpg
$O/modjpeg -i $I/image.jpg -P -o $O/out/b.jpg -A -o $O/out/c.jpg
$O/modjpeg -i $I/image_dropon.jpg -p xx -m 3 -o $O/out/d.jpg
$O/modjpeg -i $I/image.jpg -p br -m -10,-10 -d $I/dropon.png -o $O/out/e.jpg 2>&1 | tail -1 || true
$O/modjpeg -i $I/image.jpg -d $I/dropon.jpg,$I/mask.jpg -o $O/out/f.jpg 2>&1 | tail -1 || true
$O/modjpeg -i /nonexistent.jpg -o $O/out/g.jpg 2>&1 | tail -1 || true
$O/modjpeg -i $I/dropon.png -o $O/out/h.jpg 2>&1 | tail -1 || true
printf '%s\n' $I/image.jpg $I/image_dropon.jpg /nonexistent.jpg > $O/list.txt
$O/modjpeg --batch $O/list.txt --outdir $O/out --jobs 3 --devices 2 -O 2>&1 | tail -1 || true
echo "== 3. in-process multi-device operation on simulated devices (tests/test_multidevice_stub.py) under ASan+UBSan"
MJX_STUB_SANITIZE=1 python -m pytest tests/test_multidevice_stub.py -x -q -p no:cacheprovider 2>&1 | tail -2
echo "asan+ubsan: all clean"
