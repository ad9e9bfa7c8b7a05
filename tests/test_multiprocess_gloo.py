"""N > 1 path on CPU: two gloo ranks run bench.py's multi-rank protocol (image sharding by rank,
barrier + sync bracket, MAX-over-ranks of the elapsed time, SUM of the units, rank-0-only report)
with the CPU oracle standing in for the device step.  The data path has no collective
(SURVEY.md 8e); what is tested here is that the shards are disjoint and complete, that every rank
composes exactly its own images, and that the reductions are what bench.py reports."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.environ["MJX_ROOT"])
import torch, torch.distributed as dist
import bench
from tests import oracle_lib as O
from libmodjpeg_amd.batch import quant_table, shard_range

rank, world, _ = bench.rank_env()
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["MASTER_PORT"], rank=rank, world_size=world)
images = 3                                     # images per rank (weak scaling)
image0 = bench.shard_image0(rank, images)
q = quant_table(False)
rows, cols = 4, 5
state = []
for i in range(images):
    g = image0 + i
    state.append((O.synth_coef(g, 0, 0, rows * cols, q), O.synth_tile(g, 0, 0, rows * cols), O.synth_alpha(g, 0, 0, rows * cols, 2)))

def step():
    for plane, tile, alpha in state:
        O.compose_rect(plane, cols, 0, 0, rows, cols, q, tile, alpha)
    time.sleep(0.01 * (rank + 1))              # rank 1 is slower: MAX must pick it up

elapsed, _ = bench.timed_steps(step, lambda: None, 4, world, "cpu")
total = bench.sum_over_ranks(images * rows * cols, world, "cpu")
digest = [int(np.bitwise_xor.reduce(p.view(np.uint16).astype(np.int64) * 31 + 7)) for p, _, _ in state]
out = {"rank": rank, "image0": image0, "elapsed": elapsed, "total": total, "digest": digest, "strong": shard_range(10, rank, world)}
gathered = [None] * world
dist.all_gather_object(gathered, out)
if rank == 0:
    print(json.dumps(gathered))
dist.destroy_process_group()
'''


def test_two_rank_protocol_on_gloo(tmp_path):
    port = 29000 + (os.getpid() % 2000)
    env = dict(os.environ, MJX_ROOT=ROOT, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT))
    outs = [p.communicate(timeout=240) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    def reports(text):  # gloo itself may print "[Gloo] Rank ..." lines
        return [ln for ln in text.splitlines() if ln.startswith("[{")]
    assert reports(outs[1][0]) == []  # only rank 0 reports
    res = json.loads(reports(outs[0][0])[-1])
    assert [r["image0"] for r in res] == [0, 3]  # disjoint, contiguous shards
    assert res[0]["total"] == res[1]["total"] == 2 * 3 * 20  # SUM over ranks
    assert res[0]["elapsed"] == res[1]["elapsed"] >= 4 * 0.02  # MAX over ranks (rank 1 sleeps 20 ms per step)
    assert res[0]["strong"] == [0, 5] and res[1]["strong"] == [5, 10]
    # each rank composed its own images: the result equals a single-process run over all six
    from tests import oracle_lib as O
    from libmodjpeg_amd.batch import quant_table
    q = quant_table(False)
    want = []
    for g in range(6):
        plane = O.synth_coef(g, 0, 0, 20, q)
        tile, alpha = O.synth_tile(g, 0, 0, 20), O.synth_alpha(g, 0, 0, 20, 2)
        for _ in range(4):
            O.compose_rect(plane, 5, 0, 0, 4, 5, q, tile, alpha)
        want.append(int(np.bitwise_xor.reduce(plane.view(np.uint16).astype(np.int64) * 31 + 7)))
    assert res[0]["digest"] + res[1]["digest"] == want
