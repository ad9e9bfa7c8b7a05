cd /root/repo
work=$(mktemp -d /tmp/mjdbg.XXXX); mkdir -p $work/out
python3 - "$work" <<'PY'
import os, sys
import numpy as np
from PIL import Image
work = sys.argv[1]
rng = np.random.default_rng(1)
y, x = np.mgrid[0:1080, 0:1920]
img = np.stack([(x * 255 // 1920) % 256, (y * 255 // 1080) % 256, ((x + y) // 16 * 8) % 256], -1).astype(np.int16)
img = (img + rng.integers(-12, 12, img.shape[:2] + (1,))).clip(0, 255).astype(np.uint8)
Image.fromarray(img).save(os.path.join(work, "a.jpg"), quality=85, subsampling="4:2:0")
yy, xx = np.mgrid[0:256, 0:256]
r = np.hypot(xx - 128, yy - 128) / 128
alpha = (np.clip((1.0 - r) / 0.4, 0, 1) * 255).astype(np.uint8)
logo = np.stack([255 - xx, ((xx // 8 + yy // 8) % 2) * 190 + 40, yy, alpha], -1).astype(np.uint8)
Image.fromarray(logo, "RGBA").save(os.path.join(work, "logo.png"))
PY
echo $work/a.jpg > $work/list.txt
H=libmodjpeg_amd/host
MJ_DEBUG=1 MJX_HUFFDEC_DEBUG=1 $H/modjpeg --batch $work/list.txt --outdir $work/out --jobs 1 -p br -m -10,-10 -d $work/logo.png -y 10
echo "rc=$?"
MJ_DEBUG=1 $H/modjpeg -i $work/a.jpg -p br -m -10,-10 -d $work/logo.png -y 10 -o $work/out/b.jpg
echo "rc=$?"
MJ_DEVICE_RESIDENT=0 MJ_DEBUG=1 $H/modjpeg --batch $work/list.txt --outdir $work/out --jobs 1 -p br -m -10,-10 -d $work/logo.png -y 10
echo "rc=$?"
