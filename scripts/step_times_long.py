"""per-step time of the headline workload over a long run (HIP events every 50 steps): does the device hold its pace?
usage: python scripts/step_times_long.py [steps=3000] [general]  (MJX_HIP_LIB selects the library; `general`: MJX_PLAN_NO_LOW_COLUMN_BUILD)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libmodjpeg_amd.batch import ComposeBatch
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
b = ComposeBatch(4096, 1920, 1080, 256, 256, 52, 26, 2)
b.fill_synthetic(); b.build_plan(low_column_build=not (len(sys.argv) > 2 and sys.argv[2] == "general"))
print(b.plan_stats().get("runs_low_columns"), "low-column runs of", b.plan_stats().get("runs"))
torch.cuda.synchronize()
time.sleep(2.0)  # idle first: what a fresh process sees
marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps // 50 + 1)]
marks[0].record()
for k in range(steps):
    b.run()
    if (k + 1) % 50 == 0:
        marks[(k + 1) // 50].record()
torch.cuda.synchronize()
t = [marks[i].elapsed_time(marks[i + 1]) / 50 for i in range(len(marks) - 1)]
print("ms per step, mean of every 50 steps:", " ".join("%.3f" % v for v in t))
