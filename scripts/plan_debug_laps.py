import sys, os, torch
sys.path.insert(0, os.getcwd())
from libmodjpeg_amd.batch import ComposeBatch
for regime in (2, 2, 5):
    b = ComposeBatch(4096, 1920, 1080, 256, 256, 52, 26, regime)
    b.fill_synthetic()
    torch.cuda.synchronize()
    print("regime", regime, file=sys.stderr)
    b.build_plan()
    print(b.plan_stats()["build_ms"], file=sys.stderr)
    del b
