"""per-step kernel times of the default workload (HIP events), to look at DVFS / variance"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libmodjpeg_amd.batch import ComposeBatch
regime = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
b = ComposeBatch(4096, 1920, 1080, 256, 256, 52, 26, regime)
b.fill_synthetic(); b.build_plan()
for _ in range(3): b.run()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
for k in range(steps):
    ev[k][0].record(); b.run(); ev[k][1].record()
torch.cuda.synchronize()
t = [a.elapsed_time(c) for a, c in ev]
print(" ".join("%.3f" % v for v in t))
print("mean %.4f min %.4f max %.4f median %.4f" % (sum(t)/len(t), min(t), max(t), sorted(t)[len(t)//2]))
