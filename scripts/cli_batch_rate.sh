#!/bin/bash
# images/s of the batch CLI of the drop-in on synthetic 1080p JPEGs, next to the reference CLI run
# as one process per file with the same parallelism (GPU box).  Also compares the outputs.
# usage: scripts/cli_batch_rate.sh <n_images> <jobs>
set -e
cd "$(dirname "$0")/.."
n=${1:-256}; jobs=${2:-32}
work=$(mktemp -d /tmp/mjcli.XXXX); mkdir -p $work/in $work/out $work/refout
python3 - "$work" "$n" <<'PY'
import os, sys
import numpy as np
from PIL import Image
work, n = sys.argv[1], int(sys.argv[2])
rng = np.random.default_rng(1)
y, x = np.mgrid[0:1080, 0:1920]
for i in range(n):
    img = np.stack([(x * 255 // 1920 + i) % 256, (y * 255 // 1080 + 2 * i) % 256, ((x + y) // 16 * 8 + i) % 256], -1).astype(np.int16)
    img = (img + rng.integers(-12, 12, img.shape[:2] + (1,))).clip(0, 255).astype(np.uint8)
    Image.fromarray(img).save(os.path.join(work, "in", "img%04d.jpg" % i), quality=85, subsampling="4:2:0")
yy, xx = np.mgrid[0:256, 0:256]
r = np.hypot(xx - 128, yy - 128) / 128
alpha = (np.clip((1.0 - r) / 0.4, 0, 1) * 255).astype(np.uint8)
logo = np.stack([255 - xx, ((xx // 8 + yy // 8) % 2) * 190 + 40, yy, alpha], -1).astype(np.uint8)
Image.fromarray(logo, "RGBA").save(os.path.join(work, "logo.png"))
open(os.path.join(work, "list.txt"), "w").write("\n".join(os.path.join(work, "in", "img%04d.jpg" % i) for i in range(n)) + "\n")
PY
echo "== drop-in, batch mode, $jobs threads"
libmodjpeg_amd/host/modjpeg --batch $work/list.txt --outdir $work/out --jobs $jobs -p br -m -10,-10 -d $work/logo.png -y 10
echo "== reference CLI, one process per file, $jobs in parallel"
t0=$(date +%s.%N)
ls $work/in | xargs -P $jobs -I{} oracle/_ref/modjpeg_ref -i $work/in/{} -p br -m -10,-10 -d $work/logo.png -y 10 -o $work/refout/{}
t1=$(date +%s.%N)
python3 -c "print('reference: %d files, %.3f s, %.1f images/s' % ($n, $t1-$t0, $n/($t1-$t0)))"
bad=0; for f in $(ls $work/in); do cmp -s $work/out/$f $work/refout/$f || bad=$((bad+1)); done
echo "outputs differing from the reference: $bad of $n"
rm -rf $work
