#!/bin/bash
# where does the batch CLI's time go?  read+write only / +effect / +compose, for several thread counts
set -e
cd "$(dirname "$0")/.."
n=${1:-512}
work=$(mktemp -d /tmp/mjcli.XXXX); mkdir -p $work/in $work/out
python3 - "$work" "$n" <<'PY'
import os, sys
import numpy as np
from PIL import Image
work, n = sys.argv[1], int(sys.argv[2])
rng = np.random.default_rng(1)
y, x = np.mgrid[0:1080, 0:1920]
base = np.stack([(x * 255 // 1920) % 256, (y * 255 // 1080) % 256, ((x + y) // 16 * 8) % 256], -1).astype(np.int16)
for i in range(n):
    img = (base + i + rng.integers(-12, 12, base.shape[:2] + (1,))).clip(0, 255).astype(np.uint8)
    Image.fromarray(img).save(os.path.join(work, "in", "img%04d.jpg" % i), quality=85, subsampling="4:2:0")
yy, xx = np.mgrid[0:256, 0:256]
r = np.hypot(xx - 128, yy - 128) / 128
alpha = (np.clip((1.0 - r) / 0.4, 0, 1) * 255).astype(np.uint8)
logo = np.stack([255 - xx, ((xx // 8 + yy // 8) % 2) * 190 + 40, yy, alpha], -1).astype(np.uint8)
Image.fromarray(logo, "RGBA").save(os.path.join(work, "logo.png"))
open(os.path.join(work, "list.txt"), "w").write("\n".join(os.path.join(work, "in", "img%04d.jpg" % i) for i in range(n)) + "\n")
PY
for jobs in 1 8 32 64 128; do
  for script in "" "-y 10" "-p br -m -10,-10 -d $work/logo.png"; do
    echo -n "jobs=$jobs script='$script' : "
    libmodjpeg_amd/host/modjpeg --batch $work/list.txt --outdir $work/out --jobs $jobs $script 2>&1 | tail -1
  done
done
rm -rf $work
