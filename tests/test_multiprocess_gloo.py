"""N > 1 path on CPU: two gloo ranks run bench.py's multi-rank protocol (the ONE batch cut by image
index = strong scaling, barrier + sync bracket, MAX-over-ranks of the elapsed time, SUM of the
units, per-rank reports gathered on rank 0) with the CPU oracle standing in for the device step.
The data path has no collective; what is tested here is that the shards are disjoint
and complete (also for rank counts that do not divide the batch), that every rank composes exactly
its own images, and that the reductions are what bench.py reports.  bench.py itself uses this same
gloo group on the GPU box: no RCCL communicator is involved in the measurement."""
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
bench.init_ranks(world)                        # gloo, env:// rendezvous on 127.0.0.1 (what bench.py does on the GPU box)
TOTAL = 7                                      # images of the whole job: strong scaling, 4 + 3
image0, images = bench.shard_of(TOTAL, rank, world, "strong")
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

elapsed, _ = bench.timed_steps(step, lambda: None, 4, world)
total = bench.sum_over_ranks(images * rows * cols, world)
digest = [int(np.bitwise_xor.reduce(p.view(np.uint16).astype(np.int64) * 31 + 7)) for p, _, _ in state]
out = {"rank": rank, "image0": image0, "images": images, "elapsed": elapsed, "total": total, "digest": digest,
       "weak": bench.shard_of(TOTAL, rank, world, "weak"), "backend": dist.get_backend()}
gathered = bench.gather_objects(out, world)
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
    def reports(text):
        return [ln for ln in text.splitlines() if ln.startswith("[{")]
    assert reports(outs[1][0]) == []  # only rank 0 reports
    # stdout carries nothing but that one line: gloo's "[Gloo] Rank 0 is connected to ..." chatter goes to
    # stderr (bench.init_ranks points file descriptor 1 there while the group is set up)
    assert outs[1][0].strip() == "" and len(outs[0][0].strip().splitlines()) == 1, (outs[0][0][:300], outs[1][0][:300])
    res = json.loads(reports(outs[0][0])[-1])
    assert [(r["image0"], r["images"]) for r in res] == [(0, 4), (4, 3)]  # the one batch of 7, cut by image index
    assert res[0]["total"] == res[1]["total"] == 7 * 20  # SUM over ranks = the whole job
    assert res[0]["elapsed"] == res[1]["elapsed"] >= 4 * 0.02  # MAX over ranks (rank 1 sleeps 20 ms per step)
    assert [tuple(r["weak"]) for r in res] == [(0, 7), (7, 7)]
    assert {r["backend"] for r in res} == {"gloo"}  # no RCCL in the measurement protocol
    # each rank composed its own images: the result equals a single-process run over all seven
    from tests import oracle_lib as O
    from libmodjpeg_amd.batch import quant_table
    q = quant_table(False)
    want = []
    for g in range(7):
        plane = O.synth_coef(g, 0, 0, 20, q)
        tile, alpha = O.synth_tile(g, 0, 0, 20), O.synth_alpha(g, 0, 0, 20, 2)
        for _ in range(4):
            O.compose_rect(plane, 5, 0, 0, 4, 5, q, tile, alpha)
        want.append(int(np.bitwise_xor.reduce(plane.view(np.uint16).astype(np.int64) * 31 + 7)))
    assert res[0]["digest"] + res[1]["digest"] == want


def test_strong_split_is_disjoint_and_complete():
    """BASELINE configs[3]: the same 4096-image batch over N GPUs, image i -> rank floor(i*N/4096)
    (SURVEY.md 8e); also rank counts that do not divide the batch"""
    import bench
    for total in (4096, 4095, 7, 1):
        for world in (1, 2, 3, 4, 5, 6, 7, 8):
            seen = []
            for r in range(world):
                lo, n = bench.shard_of(total, r, world, "strong")
                assert n >= 0
                seen.extend(range(lo, lo + n))
            assert seen == list(range(total)), (total, world)
            sizes = [bench.shard_of(total, r, world, "strong")[1] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    assert bench.shard_of(4096, 3, 8, "strong") == (1536, 512)
    assert bench.shard_of(4096, 3, 8, "weak") == (3 * 4096, 4096)


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, cwd=ROOT)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr


def test_eight_rank_rendezvous_and_reductions_on_gloo():
    """the rank count the driver's scaling run ends with: 8 processes rendezvous over gloo on the loopback
    interface, cut a 4096-image job by image index, barrier, reduce and gather as bench.py does"""
    worker = r'''
import json, os, sys
sys.path.insert(0, os.environ["MJX_ROOT"])
import bench
rank, world, _ = bench.rank_env()
bench.init_ranks(world)
image0, images = bench.shard_of(4096, rank, world, "strong")
elapsed, _ = bench.timed_steps(lambda: None, lambda: None, 3, world)
total = bench.sum_over_ranks(images, world)
worst = bench.reduce_scalar(float(rank), world, "max")
got = bench.gather_objects({"rank": rank, "image0": image0, "images": images}, world)
if rank == 0:
    print(json.dumps({"total": total, "worst": worst, "shards": got}))
import torch.distributed as dist
dist.destroy_process_group()
'''
    port = 31000 + (os.getpid() % 2000)
    env = dict(os.environ, MJX_ROOT=ROOT, WORLD_SIZE="8", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, "-c", worker], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT)
             for r in range(8)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-1500:]
    assert all(o[0].strip() == "" for o in outs[1:])
    res = json.loads(outs[0][0].strip())
    assert res["total"] == 4096 and res["worst"] == 7.0
    assert [(s["rank"], s["image0"], s["images"]) for s in res["shards"]] == [(r, 512 * r, 512) for r in range(8)]


def test_a_dead_rank_takes_the_group_down_instead_of_hanging_it():
    """bench.py starts its own ranks when no launcher did; a rank that dies before a collective must
    not leave the others waiting in a gloo barrier (holding their GPUs) until the timeout"""
    import subprocess
    import sys
    import time

    import bench
    procs = [subprocess.Popen([sys.executable, "-c", "import time; time.sleep(120)"]) for _ in range(3)]
    procs.insert(1, subprocess.Popen([sys.executable, "-c", "import sys, time; time.sleep(0.3); sys.exit(7)"]))
    t0 = time.perf_counter()
    assert bench.wait_ranks(procs, poll_s=0.05, grace_s=3.0) == 7
    assert time.perf_counter() - t0 < 30
    assert all(p.poll() is not None for p in procs), "survivors were left running"
    ok = [subprocess.Popen([sys.executable, "-c", "pass"]) for _ in range(3)]
    assert bench.wait_ranks(ok, poll_s=0.05) == 0
    hw, phys = bench.cpu_topology()
    assert 1 <= phys <= hw


def test_traffic_figures_exist_for_every_shard_of_the_strong_split():
    """roofline.traffic of an N > 1 line (it printed null): profiles/traffic.json holds a
    measured rocprofv3 figure for the full batch and for the 512-image shard of the 8-GPU split (round 4: 656 bytes
    per block, the second line of the smooth regime's alpha blocks is not fetched), and any other shard size gets the
    nearest one scaled by its block count, labelled as scaled"""
    import os

    import bench
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
    for images in (4096, 512):
        got, src = bench.traffic_from_profiles(path, f"compose_1080p_batch:A2:{images}")
        assert got is not None and 0.99 < got / (images * 1536 * 656) < 1.02, (images, got)
        assert "rocprofv3" in src
    for images in (2048, 1024, 683):  # the 2- and 4-GPU shards, the 6-rank rehearsal's
        got, src = bench.traffic_scaled(path, "compose_1080p_batch:A2:", images * 1536)
        assert got is not None and 0.99 < got / (images * 1536 * 656) < 1.02 and src.startswith("scaled by block count"), (images, got, src)
