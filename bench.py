#!/usr/bin/env python3
"""bench.py — throughput of the masked-compose hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of mj_compose_with_mask (reference compose.c:237) over one device-resident
batch of synthetic images: by default BASELINE.json configs[2]/[3], 4096 x 1920x1080 4:2:0 JPEGs,
one 256x256 masked dropon each (1536 blocks per image, per-image tiles), smooth alpha regime A2
(10 non-zero alpha-DCT coefficients per block).  Inputs are in HBM before the timed region.

Multi-GPU (BASELINE configs[3]): the block loop has no cross-block state (reference
compose.c:256-338), so the ONE batch is cut by image index: rank r of N owns the images
shard_range(4096, r, N) (512 per GPU at N = 8) -> "scaling": "strong".  `--scaling weak` keeps the
workload's image count PER rank instead.  There is no data-path collective and no RCCL: the ranks
only meet at the two barriers of the timed region and to reduce a few scalars, over gloo (TCP on
127.0.0.1).  Started without torchrun, `--gpus N` (N > 1) spawns the N ranks itself.

Also printed on the same JSON line:
  roofline      algorithmic bytes (768 B per composed block) / mean kernel duration from HIP events
                recorded on the launch stream inside the timed region, against the 8 TB/s HBM peak;
                `kernel_ms_pristine` = the same kernel on planes restored from a pristine copy
                before every launch (untimed restore; compose is in place and not idempotent)
  per_rank      kernel ms, blocks and the parity verdict of every rank
  cpu_baseline  the UNMODIFIED reference's mj_compose_with_mask (oracle/_ref/ref_bench: compiled from the
                reference's own sources, "kind": "reference") on a bounded sample of the same workload on
                this box's cores, with the CPU oracle's figures beside it (oracle/mjx_oracle.c, "port": the
                restatement, compared with the reference row by row in the same run; it alone when
                oracle/_ref is absent); rank 0, N == 1 only.  The oracle's output on the sample is what
                the GPU output is bit-compared with.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
BYTES_PER_BLOCK = 768  # 128 JCOEF read + 256 tile + 256 alpha + 128 JCOEF write
# what the effects pass moves per DC: the DC's whole 128-byte block in and out (effect_dc_lines_kernel: eight lanes per block, non-temporal).  A
# read of ANY part of a block is one request for its whole line anyway (TCC_EA0_RDREQ = 1 per DC, none of them 32-byte ones, whether 2, 16, 64 or
# 128 bytes are read: profiles/r05/dc_stride_counters.txt), and whole-line writes are what the memory system takes fastest
EFFECT_MOVED_BYTES = 256

WORKLOADS = {
    # name: (width, height, dropon_w, dropon_h, mcu_x, mcu_y, images of the whole job)
    "compose_1080p_batch": (1920, 1080, 256, 256, 52, 26, 4096),  # BASELINE configs[2]/[3]
    "compose_4k_fullframe": (3840, 2160, 3840, 2160, 0, 0, 1),    # BASELINE configs[1]: one image, launch-bound (24 300 octets)
    # configs[1]'s kernel on a launch that fills the device: 256 independent 4K images, each with its own full-frame dropon
    "compose_4k_batch": (3840, 2160, 3840, 2160, 0, 0, 256),
}
REGIMES = {"A1": 1, "A2": 2, "A3": 3, "A4": 4, "A5": 5}  # A5: tiles of a real overlay (hard-edged disc), prepared on the device
METRIC_1080P = "8×8 DCT blocks/s masked-compose, 1080p batch; achieved HBM GB/s vs roofline"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="strong (default): the workload's images are cut over the ranks (configs[3]: the same 4096 images, 512 per GPU "
                         "at N = 8); weak: every rank gets the workload's image count")
    ap.add_argument("--spinup-ms", type=float, default=400.0,
                    help="untimed steps run for this long before the W warm-up steps: the MI355X clock governor needs "
                         "~25 ms of sustained load to leave its idle state (per-step time falls from 1.29 to 0.92 ms)")
    ap.add_argument("--workload", default="compose_1080p_batch", choices=list(WORKLOADS) + ["effects_4k_batch", "entropy_1080p_batch"])
    ap.add_argument("--threads", type=int, default=16, help="entropy_1080p_batch: host threads calling the coders concurrently")
    ap.add_argument("--regime", default="A2", choices=list(REGIMES))
    ap.add_argument("--images", type=int, default=0, help="images of the whole job (strong) / per GPU (weak); default: the workload's")
    ap.add_argument("--cpu-sample", type=int, default=256, help="images in the CPU baseline / parity sample (0 = skip)")
    ap.add_argument("--pristine-launches", type=int, default=10, help="extra launches from restored planes for roofline.kernel_ms_pristine (0 = skip)")
    ap.add_argument("--tiles", default="f32", choices=["f32", "i16"],
                    help="f32 (default, the workload as SURVEY 8d defines it: 768 B per block); i16: the prepared tiles kept as the int16 "
                         "coefficients they are made from, converted in the kernel (512 B per block; secondary, labelled)")
    ap.add_argument("--no-group", action="store_true", help="compose: keep every octet 8 consecutive blocks (no grouping of blocks by what they need)")
    ap.add_argument("--launch", choices=["auto", "serial", "side-by-side"], default="auto",
                    help="compose, plans with blocks that are blended AND blocks that need no blend (regimes A4, A5): the two kernels of a step one after "
                         "the other, side by side, or as the library decides (default; side by side when both fill the device)")
    ap.add_argument("--beside-share", type=float, default=0.0, help="compose, side-by-side launches: this share of the stream runs goes beside the blend kernel (mjx_plan_options_t; 0 = the library's estimate)")
    ap.add_argument("--run-octets", type=int, default=0, help="compose: octets per wave of the blend kernel (mjx_plan_options_t.run_octets; 0 = by the size of the launch)")
    ap.add_argument("--tail-min-octets", type=int, default=0, help="compose: mjx_plan_options_t.tail_min_octets (0 = the library's choice, -1 = runs never shrink towards the end of a launch)")
    ap.add_argument("--stream-run-octets", type=int, default=0, help="compose: octets per run of the stream kernel (mjx_plan_options_t.stream_run_octets; 0 = 4)")
    ap.add_argument("--no-low-column-build", action="store_true", help="compose: MJX_PLAN_NO_LOW_COLUMN_BUILD (smooth runs take the general build of the blend kernel)")
    ap.add_argument("--keep-zero-alpha", action="store_true",
                    help="compose: blocks without any weight still stream their tile and alpha blocks (no MJX_PLAN_ALPHA_IS_FINAL)")
    ap.add_argument("--shared-dropon", action="store_true",
                    help="compose: ONE set of tiles for every image of the batch (the reference's own use: one overlay, many images; SURVEY 8d's secondary variant - "
                         "256 B of JCOEF per block cross HBM, the tiles are read from cache)")
    ap.add_argument("--tiles-not-final", action="store_true",
                    help="compose: no MJX_PLAN_TILES_ARE_FINAL - opaque blocks of float tiles (single weight 0.25, whole-number tile) read the image's block like the other "
                         "single-weight blocks")
    ap.add_argument("--also", default="auto", choices=["auto", "none"],
                    help="auto (default): the headline run (default workload, regime, tiles and launch on one GPU) also measures, inside the same JSON line, what "
                         "making its plan costs (`plan_ms`), the same launch planned WITHOUT the promise that alpha is final (`roofline.frac_without_promise`) and, "
                         "under `also`, the other regimes of the same batch (A5 real mask, A4 mixed, A3 dense, A1 opaque; A2 and A5 with ONE overlay shared by every image), the 4K full-frame config and the effects batch, each "
                         "with its step time, its parity verdict and its fraction of the HBM peak on the bytes its plan moves; none: the headline only")
    ap.add_argument("--also-steps", type=int, default=20, help="timed steps of every `also` measurement (effects: a quarter of it)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic.json"))
    return ap.parse_args(argv)


# ---- ranks: process per GPU; gloo carries the barriers and the scalar reductions ---------------

def rank_env():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_ranks(world):
    """gloo process group over 127.0.0.1 (env:// rendezvous of torchrun or of spawn_ranks).  The
    data path needs no communicator, so none is created on the GPUs."""
    if world == 1:
        return
    import datetime
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    if os.path.isdir("/sys/class/net/lo"):  # one node: the ranks talk over the loopback interface, whatever the host name resolves to
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    # gloo announces its connections on stdout ("[Gloo] Rank 0 is connected to ..."), and stdout is for the
    # ONE JSON line: file descriptor 1 points at stderr while the group is set up
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=600))
        dist.barrier()
    except Exception as e:  # noqa: BLE001 - only the one case is handled, everything else is re-raised
        if "EADDRINUSE" in str(e) or "address already in use" in str(e).lower():
            print(f"bench.py: rendezvous port {os.environ.get('MASTER_PORT')} is in use: {e}", file=sys.stderr)
            sys.exit(EXIT_PORT_IN_USE)
        raise
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def shard_of(total, rank, world, scaling):
    """(first global image index, image count) of `rank`.  strong: image i -> rank floor(i*world/total)
    (contiguous shards); weak: `total` images per rank."""
    from libmodjpeg_amd.batch import shard_range
    if scaling == "weak":
        return rank * total, total
    lo, hi = shard_range(total, rank, world)
    return lo, hi - lo


def shard_image0(rank, images_per_gpu):
    """weak scaling: rank r owns the images [r*images_per_gpu, (r+1)*images_per_gpu)"""
    return rank * images_per_gpu


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def reduce_scalar(value, world, op):
    """MAX / SUM of one host scalar over the ranks (CPU tensor over gloo)"""
    if world == 1:
        return value
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
    return float(t.item())


def sum_over_ranks(value, world, device=None):
    return int(round(reduce_scalar(value, world, "sum")))


def gather_objects(obj, world):
    if world == 1:
        return [obj]
    import torch.distributed as dist
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out


def timed_steps(step, sync, steps, world, device=None, make_event=None):
    """Exactly `steps` calls of `step()` bracketed by barrier + sync on both sides.  The clock runs
    from after the opening barrier + sync to after the closing sync; the closing barrier follows.
    Two HIP events (make_event) bracket the K launches on the launch stream: the mean launch duration
    is their distance / K.  (Round 2 recorded a pair of events around EVERY step: two more packets in
    the queue between consecutive kernels, 3 us of the 5.5 us a step took beyond its kernel - 2.5 % of
    the step at the 8-GPU shard size.)  Returns (elapsed seconds = MAX over ranks, event ms or None)."""
    ev = (make_event(), make_event()) if make_event else None
    barrier(world)
    sync()
    t0 = time.perf_counter()
    if ev:
        ev[0].record()  # on the current stream == the stream the kernels are launched on
    for _ in range(steps):
        step()
    if ev:
        ev[1].record()
    sync()
    elapsed = time.perf_counter() - t0
    barrier(world)
    return reduce_scalar(elapsed, world, "max"), (ev[0].elapsed_time(ev[1]) if ev else None)


def spin_up(step, sync, spinup_ms):
    """untimed: keep the device busy until its clocks have ramped (DVFS), in bursts of 8 steps"""
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < spinup_ms:
        for _ in range(8):
            step()
        sync()


EXIT_PORT_IN_USE = 98  # rank 0 could not bind the rendezvous port (somebody took it between the probe and the bind)


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args_list, world, attempts=3):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes (this
    process has not touched the GPU) and return the first non-zero exit code, or 0.  The children are
    polled: when one dies (out of memory, a failed assertion on its shard) the others would sit in a
    gloo barrier until its timeout while holding their GPUs, so they are terminated at once.  The
    rendezvous port is probed and released before rank 0 binds it; if it was taken in between, rank 0
    exits with EXIT_PORT_IN_USE and the group is started again on another port."""
    code = 0
    for _ in range(attempts):
        port = free_port()
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + args_list, env=env))
        code = wait_ranks(procs)
        if code != EXIT_PORT_IN_USE:
            break
    return code


def wait_ranks(procs, poll_s=0.2, grace_s=5.0):
    """first non-zero exit code of the children (0 if all succeed); survivors of a failed group are
    terminated (then killed) instead of being left in a collective"""
    code = 0
    live = list(procs)
    while live and code == 0:
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0 and code == 0:
                code = rc
        if live and code == 0:
            time.sleep(poll_s)
    if live:
        for p in live:
            p.terminate()
        t0 = time.perf_counter()
        for p in live:
            try:
                p.wait(timeout=max(0.1, grace_s - (time.perf_counter() - t0)))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return code


# ---- CPU oracle legs ----------------------------------------------------------------------------

def cpu_baseline_and_expected(batch_cfg, regime, nsample, image0, timing=True, tiles=None):
    """Oracle compose on `nsample` images of the workload (only the dropon rectangles are
    generated: nothing else is read by the path).  `tiles`: per component (tile, alpha) arrays used
    for every image instead of the synthetic ones (regime A5).  Returns (info dict, expected rect arrays)."""
    import numpy as np
    from libmodjpeg_amd.batch import Geometry420, quant_table
    from tests import oracle_lib as O

    width, height, dw, dh, mcu_x, mcu_y = batch_cfg
    geo = Geometry420(width, height)
    quant = [quant_table(False), quant_table(True), quant_table(True)]
    cores = len(os.sched_getaffinity(0))

    def make_jobs():
        jobs, keep = [], []
        for i in range(nsample):
            for c, (h, v) in enumerate(geo.samp):
                bpr, _ = geo.array_dims[c]
                row0, col0, rows, cols = mcu_y * v, mcu_x * h, dh // 16 * v, dw // 16 * h
                rect = np.empty((rows, cols * 64), np.int16)
                for l in range(rows):  # the rectangle's blocks keep their plane block index
                    rect[l] = O.synth_coef(image0 + i, c, (row0 + l) * bpr + col0, cols, quant[c])
                rect = rect.reshape(-1)
                if tiles is not None:
                    tile, alpha = tiles[c]
                else:
                    tile = O.synth_tile(image0 + i, c, 0, rows * cols)
                    alpha = O.synth_alpha(image0 + i, c, 0, rows * cols, regime)
                jobs.append((rect, cols, 0, 0, rows, cols, quant[c], tile, alpha))
                keep.append(rect)
        return jobs, keep

    jobs, rects = make_jobs()
    nblocks = sum(j[4] * j[5] for j in jobs)
    t0 = time.perf_counter()
    O.compose_jobs(jobs, cores)
    t_all = time.perf_counter() - t0
    expected = [r.copy() for r in rects]  # one composition from pristine: the parity reference
    if not timing:
        return None, expected

    # single thread (the reference is single-threaded) on a quarter of the sample, again in place:
    # the work is data-independent
    sub = jobs[: max(3, (len(jobs) // 4) // 3 * 3)]
    t0 = time.perf_counter()
    O.compose_jobs(sub, 1)
    t_one = time.perf_counter() - t0
    sub_blocks = sum(j[4] * j[5] for j in sub)
    # repeat the all-core run for a steadier number
    reps, t_rep = 0, 0.0
    while t_rep < 2.0 and reps < 20:
        t0 = time.perf_counter()
        O.compose_jobs(jobs, cores)
        t_rep += time.perf_counter() - t0
        reps += 1
    rate_all = nblocks * reps / t_rep if reps else nblocks / t_all
    hw_threads, physical = cpu_topology()
    info = {
        # "cores" = the threads the run used (one pthread per hardware thread of the affinity mask), as the bench contract defines it
        "value": rate_all, "unit": "blocks/s", "cores": cores, "hw_threads": hw_threads, "physical_cores": physical, "kind": "port",
        "sample": f"{nsample} images x {nblocks // max(nsample, 1)} blocks of the same synthetic workload, oracle/mjx_oracle.c "
                  f"(-O2 -ffp-contract=off), {cores} pthreads on {hw_threads} hardware threads = {physical} physical cores, {reps} repeats",
        "value_1thread": sub_blocks / t_one, "gbs_equiv": rate_all * BYTES_PER_BLOCK / 1e9, "cpu_model": cpu_model(),
    }
    ref = reference_cpu_baseline(batch_cfg, regime, nsample, image0, cores) if tiles is None else None
    if ref is not None:
        # the UNMODIFIED reference's mj_compose_with_mask, compiled from its own sources (oracle/Makefile -> oracle/_ref), on the same
        # sample: the baseline north_star names; the port's figures stay beside it
        info = dict(info, kind="reference", value=ref["blocks_per_s"], value_1thread=ref["blocks_per_s_1thread"], gbs_equiv=ref["blocks_per_s"] * BYTES_PER_BLOCK / 1e9,
                    sample=f"{nsample} images x {ref['blocks_per_image']} blocks of the same synthetic workload through the unmodified reference's mj_compose_with_mask "
                           f"(compose.c:237; oracle/_ref/ref_bench, -O2 -ffp-contract=off, IJG libjpeg 9d arrays), {cores} pthreads on {hw_threads} hardware threads = {physical} physical cores, "
                           f"whole passes for at least 2 s; timed: copying the rectangle's coefficients into libjpeg's arrays + the call",
                    reference_rows_compared_with_the_port=ref["rows_compared_with_the_restatement"], reference_rows_that_differ=ref["rows_that_differ"],
                    port={"value": rate_all, "value_1thread": sub_blocks / t_one, "what": "oracle/mjx_oracle.c on the same sample: the checker of the GPU output"})
    return info, expected


def reference_cpu_baseline(batch_cfg, regime, nsample, image0, threads):
    """oracle/_ref/ref_bench (oracle/ref_bench.c: our harness around the reference compiled from its own sources by oracle/Makefile; the
    binaries are built in the build container and travel with the repository snapshot - nothing is read from the reference's tree at
    run time).  Returns its JSON, or None when it is absent or fails (the port is the baseline then)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_bench")
    if not os.path.exists(exe):
        return None
    width, height, dw, dh, mcu_x, mcu_y = batch_cfg
    try:
        out = subprocess.run([exe, str(width), str(height), str(dw), str(dh), str(mcu_x), str(mcu_y), str(regime), str(image0), str(nsample), str(threads), "2.0"],
                             capture_output=True, text=True, timeout=300)
        if out.returncode != 0:
            print(f"bench.py: oracle/_ref/ref_bench failed ({out.returncode}): {out.stderr[-300:]}", file=sys.stderr)
            return None
        return json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001 - a missing libjpeg, a timeout: the port stays the baseline
        print(f"bench.py: oracle/_ref/ref_bench not usable: {e}", file=sys.stderr)
        return None


def cpu_topology():
    """(hardware threads this process may run on, distinct physical cores among them)"""
    cpus = sorted(os.sched_getaffinity(0))
    cores = set()
    for c in cpus:
        try:
            base = f"/sys/devices/system/cpu/cpu{c}/topology/"
            cores.add((open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip()))
        except OSError:
            cores.add(("?", str(c)))
    return len(cpus), len(cores)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def traffic_from_profiles(path, key):
    """HBM bytes per launch from the rocprofv3 --pmc passes committed under profiles/ (FETCH_SIZE
    doubled + WRITE_SIZE, as MI355X_MICROARCH.md prescribes for gfx950).  PMC counters cannot be
    collected from inside the process being timed, so this is the figure of the profiled run of the
    same command, labelled with its source; None when there is no such pass."""
    try:
        tj = json.load(open(path))
        if key in tj:
            return tj[key]["hbm_bytes_per_launch"], tj[key].get("source", os.path.relpath(path, ROOT))
    except Exception:
        pass
    return None, None


def traffic_scaled(path, key_prefix, blocks):
    """No PMC pass of exactly this launch size: the counters of the nearest measured size of the same
    workload and regime, scaled by the block count (every block of a launch moves the same bytes:
    the measured passes of 512 .. 4096 images agree to 0.1 % per block), labelled as scaled."""
    try:
        tj = json.load(open(path))
        best = None
        for k, v in tj.items():
            if k.startswith(key_prefix) and not k.endswith(":i16") and isinstance(v, dict) and v.get("blocks_per_launch"):
                d = abs(v["blocks_per_launch"] - blocks)
                if best is None or d < best[0]:
                    best = (d, k, v)
        if best:
            _, k, v = best
            return int(round(v["hbm_bytes_per_launch"] * blocks / v["blocks_per_launch"])), f"scaled by block count from {k}: {v.get('source', '')}"
    except Exception:
        pass
    return None, None


def plan_launch_bytes(plan_stats, blocks, tiles_i16, shared_dropon=False):
    """Bytes one launch of a plan asks the memory system for (the counters agree: profiles/traffic.json).
    SURVEY 8d prices a composed block at 768 B (512 with int16 tiles); a plan made with the promise that alpha is final moves
    less for four kinds of block: no weight at all -> the JCOEF block in + its table entry, written back only if it changed
    (132 B); the single weight (0,0) -> JCOEF in and out, the tile block and a table entry with the weight (520 B, 392 with
    int16 tiles); float tiles, no weight in a row k >= 4 -> the second 128-byte line of the alpha block stays where it is
    (eight first lines and one second line per octet in the low-column build: 656 B per block; two in the general and lean
    builds: 672 B); an opaque block (MJX_PLAN_TILES_ARE_FINAL: that weight is 0.25 and the tile holds whole numbers) -> the tile
    block in, a table entry, JCOEF out (388 B, 260 with int16 tiles)."""
    if shared_dropon:  # one tile set for the whole batch (a megabyte, read from cache): what crosses HBM is JCOEF in and out, and the tables
        special_dc, special_zero, special_opq = 8 * plan_stats["dc_only_octets"], 8 * plan_stats["zero_alpha_octets"], 8 * plan_stats.get("opaque_octets", 0)
        return (blocks - special_dc - special_zero - special_opq) * 256 + special_dc * (256 + 8) + special_zero * 132 + special_opq * 132
    per_block = 512 if tiles_i16 else BYTES_PER_BLOCK
    special_dc, special_zero = 8 * plan_stats["dc_only_octets"], 8 * plan_stats["zero_alpha_octets"]
    dc_bytes = 128 + (128 if tiles_i16 else 256) + 8 + 128
    special_opq = 8 * plan_stats.get("opaque_octets", 0)
    opq_bytes = (128 if tiles_i16 else 256) + 4 + 128
    lowl_only = plan_stats.get("runs_low_columns") and plan_stats.get("low_column_octets") == plan_stats.get("low_row_octets") and not plan_stats.get("runs_beside")
    skipped = 8 * plan_stats.get("low_row_octets", 0) * (112 if lowl_only else 96) if not tiles_i16 else 0
    launch_bytes = (blocks - special_dc - special_zero - special_opq) * per_block - skipped + special_dc * dc_bytes + special_zero * 132 + special_opq * opq_bytes
    return max(min(launch_bytes, blocks * per_block), 0)


def also_compose(batch, batch_cfg, regime_name, steps, nsample, alpha_is_final=True, spinup_ms=150.0, tiles_are_final=None):
    """One more measurement on an existing batch (same images, same rectangles): tiles and alpha of `regime_name` and the planes
    are generated afresh (deterministic: the planes are pristine again), a plan is made, the first pass is compared with the CPU
    oracle on `nsample` images, then `steps` launches are timed between two HIP events.  Returns the entry of the JSON line's
    `also` block."""
    import numpy as np
    import torch
    t_wall = time.perf_counter()
    regime = REGIMES[regime_name]
    batch.regime = regime
    batch.fill_synthetic()
    batch.build_plan(alpha_is_final=alpha_is_final, tiles_are_final=tiles_are_final)
    st = batch.plan_stats()
    tiles = [batch.tiles_as_float(0, c) for c in range(3)] if regime == 5 or batch.shared_dropon else None
    _, expected = cpu_baseline_and_expected(batch_cfg, regime, nsample, batch.image0, timing=False, tiles=tiles)
    batch.run()
    torch.cuda.synchronize()
    bad = sum(not np.array_equal(batch.rect_numpy(i, c).reshape(-1), expected[3 * i + c]) for i in range(nsample) for c in range(3))
    spin_up(batch.run, torch.cuda.synchronize, spinup_ms)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        batch.run()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / steps
    blocks = batch.total_blocks()
    moved = plan_launch_bytes(st, blocks, False, batch.shared_dropon)
    return {"alpha_regime": regime_name, "shared_dropon": batch.shared_dropon, "images": batch.n, "blocks_per_step": blocks, "steps": steps, "ms_per_step": ms, "blocks_per_s": blocks / (ms * 1e-3),
            "parity": "bit-exact" if bad == 0 else f"MISMATCH in {bad} of {3 * nsample} rectangles", "parity_sample_images": nsample,
            "alpha_is_final": alpha_is_final, "bytes_moved_per_block": moved / blocks, "achieved_on_bytes_moved": moved / (ms * 1e-3) / 1e9,
            "frac_on_bytes_moved": moved / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "frac_at_nominal_bytes_per_block": blocks * BYTES_PER_BLOCK / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "launch": "side by side (%d stream runs beside the lean blend kernel)" % st["runs_beside"] if st.get("runs_beside") else "one kernel after the other",
            "mean_terms_per_block": st.get("mean_terms_per_block"), "mean_union_terms_per_octet": st.get("mean_union_terms_per_octet"),
            "plan_ms": st["build_ms"], "wall_s": time.perf_counter() - t_wall}


def also_effects(steps, image0=0, images=4096):
    """BASELINE configs[4] at its size on this GPU (4096 x 4K resident = 102 GB; luminance(+7) then tint(+5,-3), two launches per step):
    first pass compared with the oracle on two whole images, then `steps` steps between two HIP events."""
    import numpy as np
    import torch
    from libmodjpeg_amd.batch import EffectBatch
    t_wall = time.perf_counter()
    lum, cb, cr = 7, 5, -3
    eb = EffectBatch(images, 3840, 2160, image0=image0)
    eb.fill_synthetic()
    eb.build_plans(lum, cb, cr, fused=False)
    dcs = eb.dc_blocks_per_image() * images
    _, expected = effects_cpu_baseline_and_expected(eb.geo, eb.quant, image0, 2, lum, cb, cr, timing=False)
    eb.run()
    torch.cuda.synchronize()
    bad = sum(not np.array_equal(eb.coef[c][i].cpu().numpy().reshape(-1), expected[3 * i + c]) for i in range(2) for c in range(3))
    spin_up(eb.run, torch.cuda.synchronize, 150.0)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        eb.run()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / steps
    out = {"workload": "effects_4k_batch", "images": images, "dc_blocks_per_step": dcs, "steps": steps, "ms_per_step": ms, "blocks_per_s": dcs / (ms * 1e-3),
           "parity": "bit-exact" if bad == 0 else f"MISMATCH in {bad} of 6 planes", "parity_sample_images": 2,
           "frac_at_4_bytes_per_dc": dcs * 4 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "bytes_moved_per_dc": EFFECT_MOVED_BYTES, "frac_on_bytes_moved": dcs * EFFECT_MOVED_BYTES / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "read_only_pass_would_take_ms": dcs / 50e9 * 1e3,  # 49-56 G lines/s: what reading every DC costs alone (scripts/ubench/dc_stride.hip rd2 .. rd128)
           "wall_s": None}
    eb.destroy_plans()
    del eb
    torch.cuda.empty_cache()
    out["wall_s"] = time.perf_counter() - t_wall
    return out


# ---- masked compose -------------------------------------------------------------------------------

def main():
    args = parse()
    rank, world, local_rank = rank_env()
    if world == 1 and args.gpus > 1:
        sys.exit(spawn_ranks(sys.argv[1:], args.gpus))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"

    import torch
    # MJX_BENCH_DEVICES="0,0": rehearsal of the N-rank protocol on a box with fewer GPUs than ranks
    devmap = [int(d) for d in os.environ.get("MJX_BENCH_DEVICES", "").split(",") if d != ""]
    torch.cuda.set_device(devmap[local_rank] if devmap else local_rank)
    init_ranks(world)
    try:
        if args.workload == "effects_4k_batch":
            bench_effects(args, rank, world)
        elif args.workload == "entropy_1080p_batch":
            bench_entropy(args, rank, world)
        else:
            bench_compose(args, rank, world)
    finally:
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()


def bench_compose(args, rank, world):
    import numpy as np
    import torch
    from libmodjpeg_amd.batch import ComposeBatch

    width, height, dw, dh, mcu_x, mcu_y, default_images = WORKLOADS[args.workload]
    total_images = args.images or default_images
    regime = REGIMES[args.regime]
    image0, images = shard_of(total_images, rank, world, args.scaling)
    assert images > 0, f"rank {rank}: empty shard ({total_images} images over {world} ranks)"
    batch = ComposeBatch(images, width, height, dw, dh, mcu_x, mcu_y, regime, image0=image0, tiles_i16=(args.tiles == "i16"), shared_dropon=args.shared_dropon)
    bytes_per_block = 512 if args.tiles == "i16" else BYTES_PER_BLOCK  # int16 tiles: 128 + 128 + 128 read, 128 written
    batch.fill_synthetic()
    batch.build_plan(group_by_density=not args.no_group, alpha_is_final=not args.keep_zero_alpha, side_by_side={"auto": None, "serial": False, "side-by-side": True}[args.launch],
                     run_octets=args.run_octets, stream_run_octets=args.stream_run_octets, tail_min_octets=args.tail_min_octets, beside_share=args.beside_share, low_column_build=not args.no_low_column_build,
                     tiles_are_final=False if args.tiles_not_final else None)
    plan_stats = batch.plan_stats()
    blocks_per_step = batch.total_blocks()
    sync = torch.cuda.synchronize

    # ---- parity (+ CPU baseline) on a bounded sample: first pass from pristine inputs ----
    # N == 1: rank 0 times the CPU baseline on `--cpu-sample` images and checks them; N > 1: every
    # rank checks a few images of its own shard (no CPU timing); every rank's verdict is reported
    cpu_info, parity, bad, nsample = None, "skipped", 0, 0
    if args.cpu_sample > 0:
        nsample = min(args.cpu_sample, images) if world == 1 else min(args.cpu_sample, images, 4)
        # bounded CPU work whatever the image size: about as many blocks as 256 images of the 1080p workload
        nsample = min(nsample, max(1, (256 * 1536) // batch.blocks_per_image))
    pristine = batch.rect_clone() if args.pristine_launches > 0 else None
    expected = None
    if nsample > 0:
        # the sample: the first images of this rank's shard
        tiles = None
        if regime == 5 or args.shared_dropon:  # the oracle gets the device-prepared tiles as its input (their own parity: tests/test_dropin_shim.py)
            tiles = [batch.tiles_as_float(0, c) for c in range(3)]
        info, expected = cpu_baseline_and_expected((width, height, dw, dh, mcu_x, mcu_y), regime, nsample, image0, timing=(world == 1), tiles=tiles)
        cpu_info = info if world == 1 else None
    batch.run()
    sync()
    if expected is not None:
        for i in range(nsample):
            for c in range(3):
                if not np.array_equal(batch.rect_numpy(i, c).reshape(-1), expected[3 * i + c]):
                    bad += 1
        parity = "bit-exact" if bad == 0 else f"MISMATCH in {bad} of {3 * nsample} rectangles"
    bad_all = sum_over_ranks(bad, world)
    if bad_all:
        print(f"bench.py: rank {rank}: GPU output differs from the oracle: {parity}", file=sys.stderr)
        sys.exit(1)

    # ---- clock spin-up (untimed, see --spinup-ms), then the W warm-up steps ----
    spin_up(batch.run, sync, args.spinup_ms)
    for _ in range(args.warmup):
        batch.run()
    sync()

    # ---- timed region: exactly K steps ----
    elapsed, ev_ms = timed_steps(batch.run, sync, args.steps, world, make_event=lambda: torch.cuda.Event(enable_timing=True))
    mean_kern_ms = ev_ms / args.steps  # K launches back to back between two events on the launch stream
    tot_blocks = sum_over_ranks(blocks_per_step, world)

    # ---- the same launch on planes restored from the pristine copy before every launch (outside
    # the contract's timed region; the restore is not between the events) ----
    pristine_ms, ms = None, []
    if pristine is not None:
        for _ in range(args.pristine_launches):
            batch.rect_restore(pristine)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            batch.run()
            b.record()
            sync()
            ms.append(a.elapsed_time(b))
        pristine_ms = sum(ms) / len(ms)

    mine = {"rank": rank, "image0": image0, "images": images, "blocks_per_step": blocks_per_step, "kernel_ms_mean": mean_kern_ms,
            "kernel_ms_pristine": pristine_ms, "kernel_ms_pristine_min": min(ms) if ms else None, "parity": parity, "parity_sample_images": nsample,
            "device": torch.cuda.get_device_name()}
    per_rank = gather_objects(mine, world)

    if rank == 0:
        # Algorithmic bytes of THIS launch.  SURVEY 8d prices a composed block at 768 B (512 with int16 tiles); a plan made
        # with the promise that alpha is final moves less for two kinds of block: no weight at all -> the
        # JCOEF block in + its table entry; the single weight (0,0) -> JCOEF in and out, the tile block and
        # a table entry with the weight (520 B, 392 with int16 tiles).  The headline regime A2 has neither kind.
        launch_bytes = plan_launch_bytes(plan_stats, blocks_per_step, args.tiles == "i16", args.shared_dropon)
        achieved = launch_bytes / (mean_kern_ms * 1e-3) / 1e9
        achieved_nominal = blocks_per_step * bytes_per_block / (mean_kern_ms * 1e-3) / 1e9
        moved_bytes, achieved_moved = launch_bytes, achieved  # (the same figures under the names the round-4 profiles use)
        # every rank's own launch against the peak of its own GPU: the figures above describe rank 0's shard
        rank_gbs = [r["blocks_per_step"] * (launch_bytes / blocks_per_step) / (r["kernel_ms_mean"] * 1e-3) / 1e9 for r in per_rank]
        key = f"{args.workload}:{args.regime}:"
        traffic, traffic_src = traffic_from_profiles(args.traffic_json, f"{args.workload}:{args.regime}:{images}" + (":i16" if args.tiles == "i16" else "") + (":shared_dropon" if args.shared_dropon else ""))
        if traffic is None and args.tiles == "f32" and not args.shared_dropon:
            traffic, traffic_src = traffic_scaled(args.traffic_json, key, blocks_per_step)
        verdicts = {r["parity"] for r in per_rank}
        out = {
            # BASELINE.json's metric string, verbatim, for the workload it is quoted on
            "metric": METRIC_1080P if args.workload == "compose_1080p_batch" else "8×8 DCT blocks/s masked-compose",
            "value": tot_blocks * args.steps / elapsed, "unit": "blocks/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "s16 JCOEF, f32/f64 blend (reference typing)", "data": "synthetic",
            "config": {"workload": args.workload, "images_total": tot_blocks // batch.blocks_per_image, "images_per_gpu": images, "image": f"{width}x{height} 4:2:0",
                       "dropon": f"{dw}x{dh} alpha-masked, " + ("ONE tile set shared by every image (read from cache)" if args.shared_dropon else "per-image tiles") + f", MCU ({mcu_x},{mcu_y})",
                       "alpha_regime": args.regime, "tiles": "float (768 B per block)" if args.tiles == "f32" else "int16 coefficients, converted in the kernel (512 B per block)",
                       "density_grouping": (not args.no_group), "gather_octets_per_gpu": int(batch.gather_octets), "plan": plan_stats,
                       "launch": ("side by side: %d light runs streamed beside the lean blend kernel, the rest after it" % plan_stats["runs_beside"]) if plan_stats.get("runs_beside") else "one kernel after the other",
                       "blocks_per_step_total": tot_blocks, "blocks_per_step_per_gpu": blocks_per_step, "spinup_ms": args.spinup_ms,
                       "parallelism": f"images sharded x{world} ({args.scaling}), no collective, no RCCL"},
            "parity": verdicts.pop() if len(verdicts) == 1 else "mixed: see per_rank",
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": ("mjx::compose_lowl_kernel (the blend kernel's build for weights in columns 0..3: four workgroups per CU; weights in rows 0..3 as well: the second line of the alpha blocks is not fetched)" if plan_stats.get("runs_low_columns") == plan_stats.get("runs") else "mjx::compose_kernel") if not plan_stats.get("zero_alpha_octets") and not plan_stats.get("dc_only_octets") and not plan_stats.get("opaque_octets") else "one step = mjx::compose_kernel (or its lean build) + mjx::compose_light_kernel (+ mjx::compose_light_shared_kernel beside it)",
                         "kernel_ms_mean": mean_kern_ms,
                         "kernel_ms_mean_is": "distance of two HIP events around the K timed launches on the launch stream / K",
                         "kernel_ms_pristine": pristine_ms, "kernel_ms_pristine_min": min(ms) if ms else None,
                         "kernel_ms_pristine_median": sorted(ms)[len(ms) // 2] if ms else None,
                         "algorithmic_bytes_per_launch": launch_bytes, "bytes_per_block": bytes_per_block,
                         "bytes_per_block_of_this_plan": launch_bytes / blocks_per_step,
                         "bytes_moved_per_launch": moved_bytes, "bytes_moved_per_block": moved_bytes / blocks_per_step,
                         "achieved_on_bytes_moved": achieved_moved, "frac_on_bytes_moved": achieved_moved / HBM_PEAK_GBS,
                         "achieved_at_nominal_bytes_per_block": achieved_nominal, "frac_at_nominal_bytes_per_block": achieved_nominal / HBM_PEAK_GBS,
                         "scope": "rank 0" if world > 1 else "the one rank", "rank": 0,
                         "achieved_min_over_ranks": min(rank_gbs), "achieved_max_over_ranks": max(rank_gbs),
                         "frac_min_over_ranks": min(rank_gbs) / HBM_PEAK_GBS, "frac_max_over_ranks": max(rank_gbs) / HBM_PEAK_GBS},
            "per_rank": per_rank,
            "cpu_baseline": cpu_info,
        }
        # what making the plan cost (the reference pays tile compilation and its zero tests inside every call, compose.c:155-157, convolve.c:32;
        # a plan pays once, outside the timed region): mjx_plan_stats_t.build_us
        out["plan_ms"] = dict(plan_stats["build_ms"], note="mjx_compose_plan_create_opts of the headline batch, host wall clock: scan of the alpha tiles on the device, "
                              "its results coming down, the host's schedule (patterns -> octets -> runs), tables going up; once per plan, not inside the timed region; "
                              "in steps of the headline launch: %.1f" % (plan_stats["build_ms"]["total"] / mean_kern_ms))
        headline = (args.workload == "compose_1080p_batch" and args.regime == "A2" and args.tiles == "f32" and args.launch == "auto" and not args.no_group
                    and not args.keep_zero_alpha and not args.images and world == 1 and not (args.beside_share or args.run_octets or args.stream_run_octets or args.no_low_column_build or args.shared_dropon or args.tiles_not_final or args.tail_min_octets))
        if args.also == "auto" and headline:
            cfg = (width, height, dw, dh, mcu_x, mcu_y)
            t_also = time.perf_counter()
            # the same launch planned WITHOUT the promise that alpha is final: whole alpha blocks travel (768 B per block), general or low-column build
            wp = also_compose(batch, cfg, "A2", args.also_steps, 8, alpha_is_final=False)
            out["roofline"]["frac_without_promise"] = wp["frac_on_bytes_moved"]
            out["roofline"]["kernel_ms_without_promise"] = wp["ms_per_step"]
            out["roofline"]["without_promise"] = wp
            also = {"compose_1080p_batch:" + r: also_compose(batch, cfg, r, args.also_steps, 8) for r in ("A5", "A4", "A3", "A1")}
            del batch
            # SURVEY 8d's secondary variant, and what the reference's callers do: ONE overlay for the whole batch (smooth mask; the real disc mask)
            sb = ComposeBatch(images, width, height, dw, dh, mcu_x, mcu_y, REGIMES["A2"], image0=image0, shared_dropon=True)
            for r in ("A2", "A5"):
                also["compose_1080p_batch:%s:shared_dropon" % r] = also_compose(sb, cfg, r, args.also_steps, 8)
            del sb
            if pristine is not None:
                del pristine
            torch.cuda.empty_cache()
            w4, h4 = WORKLOADS["compose_4k_fullframe"][0:2]
            one = ComposeBatch(1, w4, h4, w4, h4, 0, 0, REGIMES["A2"])
            also["compose_4k_fullframe:A2"] = also_compose(one, (w4, h4, w4, h4, 0, 0), "A2", 10 * args.also_steps, 1, spinup_ms=50.0)
            del one
            torch.cuda.empty_cache()
            also["effects_4k_batch"] = also_effects(max(2, args.also_steps // 4))
            also["wall_s"] = time.perf_counter() - t_also
            also["note"] = ("secondary measurements of the same run, each from pristine synthetic inputs, first pass checked against the CPU oracle; "
                            "frac_on_bytes_moved = the bytes the plan asks the memory system for / step time / 8 TB/s")
            out["also"] = also
            mism = [k for k, v in also.items() if isinstance(v, dict) and v.get("parity") not in (None, "bit-exact")] + ([] if wp["parity"] == "bit-exact" else ["without_promise"])
            if mism:
                print(json.dumps(out))
                print(f"bench.py: GPU output differs from the oracle in {mism}", file=sys.stderr)
                sys.exit(1)
        print(json.dumps(out))
        sys.stdout.flush()


# ---- DC effects (BASELINE configs[4]) ---------------------------------------------------------

def effects_cpu_baseline_and_expected(geo, quant, image0, nsample, lum, cb, cr, timing):
    """oracle orc_effect_dc (reference effect.c:185-222, :116-183) on `nsample` whole 4K images of
    the same synthetic batch.  Returns (info, expected DC planes per (image, component))."""
    import numpy as np
    from tests import oracle_lib as O
    planes = []
    for i in range(nsample):
        for c in range(3):
            bpr, prow = geo.array_dims[c]
            planes.append(O.synth_coef(image0 + i, c, 0, bpr * prow, quant[c]))

    def run():
        k = 0
        for i in range(nsample):
            for c, value in enumerate((lum, cb, cr)):
                bpr, _ = geo.array_dims[c]
                tw, th = geo.true_dims[c]
                if not (c > 0 and value == 0):  # effect.c:134/:158
                    O.effect_dc(planes[k], bpr, th, tw, int(quant[c][0]), value)
                k += 1

    t0 = time.perf_counter()
    run()
    t_first = time.perf_counter() - t0
    expected = [p.copy() for p in planes]
    if not timing:
        return None, expected
    reps, t_rep = 1, t_first
    while t_rep < 2.0 and reps < 50:  # again in place: the work is data-independent
        t0 = time.perf_counter()
        run()
        t_rep += time.perf_counter() - t0
        reps += 1
    dcs = nsample * sum(w * h for w, h in geo.true_dims)
    info = {"value": dcs * reps / t_rep, "unit": "blocks/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(),
            "sample": f"{nsample} images x {dcs // nsample} DC blocks (3840x2160 4:2:0), oracle/mjx_oracle.c orc_effect_dc, 1 thread "
                      f"(the reference is single-threaded), {reps} repeats"}
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_bench")
    if os.path.exists(exe):  # the unmodified reference's mj_effect_luminance + mj_effect_tint on libjpeg's arrays (oracle/ref_bench.c, effects mode)
        try:
            out = subprocess.run([exe, "effects", str(geo.width), str(geo.height), str(image0), str(nsample), str(lum), str(cb), str(cr), "2.0"], capture_output=True, text=True, timeout=300)
            if out.returncode == 0:
                ref = json.loads(out.stdout.strip().splitlines()[-1])
                info = dict(info, kind="reference", value=ref["dc_blocks_per_s"],
                            sample=f"{nsample} images x {ref['dc_blocks_per_image']} DC blocks ({geo.width}x{geo.height} 4:2:0) through the unmodified reference's mj_effect_luminance + mj_effect_tint "
                                   f"(effect.c:185, :116; oracle/_ref/ref_bench), 1 thread (the reference is single-threaded), {ref['passes']} passes over the same "
                                   f"{nsample * geo.blocks_per_image() * 128 // 2 ** 20} MiB of planes, which a large last-level cache keeps: an upper bound for the CPU",
                            reference_rows_compared_with_the_port=ref["rows_compared_with_the_restatement"], reference_rows_that_differ=ref["rows_that_differ"],
                            port={"value": dcs * reps / t_rep, "what": "oracle/mjx_oracle.c orc_effect_dc on the same sample"})
            else:
                print(f"bench.py: oracle/_ref/ref_bench effects failed ({out.returncode}): {out.stderr[-300:]}", file=sys.stderr)
        except Exception as e:  # noqa: BLE001
            print(f"bench.py: oracle/_ref/ref_bench not usable: {e}", file=sys.stderr)
    return info, expected


def bench_effects(args, rank, world):
    """BASELINE configs[4]: mj_effect_luminance(+7) then mj_effect_tint(+5, -3) over 4096 4K 4:2:0
    images (two launches, as the reference makes two calls), 1 -> 8 GPUs by image index."""
    import numpy as np
    import torch
    from libmodjpeg_amd.batch import EffectBatch

    total_images = args.images or 4096
    image0, images = shard_of(total_images, rank, world, args.scaling)
    lum, cb, cr = 7, 5, -3
    eb = EffectBatch(images, 3840, 2160, image0=image0)
    eb.fill_synthetic()
    eb.build_plans(lum, cb, cr, fused=False)
    dcs = eb.dc_blocks_per_image() * images
    sync = torch.cuda.synchronize

    # parity (+ CPU baseline at N == 1) on sampled images, first pass from pristine inputs
    nsample = 0 if args.cpu_sample <= 0 else min(images, 2)
    cpu_info, parity, bad = None, "skipped", 0
    expected = None
    if nsample:
        cpu_info, expected = effects_cpu_baseline_and_expected(eb.geo, eb.quant, image0, nsample, lum, cb, cr, timing=(world == 1))
    eb.run()
    sync()
    if expected is not None:
        for i in range(nsample):
            for c in range(3):
                if not np.array_equal(eb.coef[c][i].cpu().numpy().reshape(-1), expected[3 * i + c]):
                    bad += 1
        parity = "bit-exact" if bad == 0 else f"MISMATCH in {bad} of {3 * nsample} planes"
    if sum_over_ranks(bad, world):
        print(f"bench.py: rank {rank}: GPU effects differ from the oracle: {parity}", file=sys.stderr)
        sys.exit(1)

    spin_up(eb.run, sync, args.spinup_ms)
    for _ in range(max(args.warmup, 1)):
        eb.run()
    sync()
    elapsed, ev_ms = timed_steps(eb.run, sync, args.steps, world, make_event=lambda: torch.cuda.Event(enable_timing=True))
    mean_ms = ev_ms / args.steps
    tot = sum_over_ranks(dcs, world)
    per_rank = gather_objects({"rank": rank, "image0": image0, "images": images, "dc_blocks_per_step": dcs, "kernel_ms_mean": mean_ms, "parity": parity}, world)
    if rank == 0:
        achieved = dcs * 4 / (mean_ms * 1e-3) / 1e9          # 2 B read + 2 B written per DC
        traffic, traffic_src = traffic_from_profiles(args.traffic_json, f"effects_4k_batch:{images}")
        verdicts = {r["parity"] for r in per_rank}
        print(json.dumps({
            "metric": "DC blocks/s mj_effect_luminance+mj_effect_tint, 4K batch", "value": tot * args.steps / elapsed, "unit": "blocks/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "s16", "data": "synthetic",
            "config": {"workload": "effects_4k_batch", "images_total": total_images if args.scaling == "strong" else total_images * world, "images_per_gpu": images,
                       "image": "3840x2160 4:2:0", "calls": "luminance(+7); tint(+5,-3)", "parallelism": f"images sharded x{world} ({args.scaling}), no collective, no RCCL"},
            "parity": verdicts.pop() if len(verdicts) == 1 else "mixed: see per_rank",
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "kernel": "mjx::effect_dc_lines_kernel (two launches per step)", "kernel_ms_mean": mean_ms,
                         "algorithmic_bytes_per_launch": dcs * 4,
                         # "also report the sector-realistic figure": what the memory system moves per DC
                         "bytes_moved": {"achieved": dcs * EFFECT_MOVED_BYTES / (mean_ms * 1e-3) / 1e9, "frac": dcs * EFFECT_MOVED_BYTES / (mean_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                         "bytes_per_dc": EFFECT_MOVED_BYTES, "note": "the whole 128-byte block read and written back per 2-byte DC: a copy of the planes onto themselves"}},
            "per_rank": per_rank,
            "cpu_baseline": cpu_info}))
        sys.stdout.flush()


def bench_entropy(args, rank, world):
    """Secondary workload: the baseline Huffman coders on composed 1080p images
    whose planes stay on the device.  One step = every image of the batch encoded to scan data and
    decoded back into scratch planes, by `--threads` host threads calling the C ABI concurrently (the
    library combines concurrent callers into one launch set).  Parity: decoded planes == originals."""
    import threading
    import torch
    from libmodjpeg_amd import entropy as E
    from libmodjpeg_amd.batch import ComposeBatch

    images = args.images or 256
    b = ComposeBatch(images, 1920, 1080, 256, 256, 52, 26, REGIMES[args.regime], image0=shard_image0(rank, images))
    b.fill_synthetic()
    b.build_plan().run()
    torch.cuda.synchronize()
    g = b.geo
    scratch = [[torch.empty_like(b.coef[c][0]) for c in range(3)] for _ in range(args.threads)]
    nbytes = [0] * args.threads
    # every worker thread has a stream of its own, as a threaded torch program has: the coders order their first access to the device
    # planes after the CALLER's stream (mjx.h, ordering contract; nothing is pending there, so the library's look at the stream is all it costs)
    own = [torch.cuda.Stream() for _ in range(args.threads)]

    def step():
        def work(tid):
            total = 0
            with torch.cuda.stream(own[tid]):
                for i in range(tid, images, args.threads):
                    planes = [b.coef[c][i] for c in range(3)]
                    data = E.encode_scan(planes, g.true_dims, g.samp, g.mcus_x, g.mcus_y)
                    E.decode_scan(data + E.EOI, scratch[tid], g.samp, g.mcus_x, g.mcus_y)
                    total += len(data)
            nbytes[tid] = total
        ts = [threading.Thread(target=work, args=(t,)) for t in range(args.threads)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()

    step()  # warm-up: arenas, first launches
    ok = True
    for tid in range(min(args.threads, images)):  # what each thread decoded last
        i = tid + ((images - 1 - tid) // args.threads) * args.threads
        for c in range(3):
            w, h = g.true_dims[c]
            ok = ok and bool(torch.equal(scratch[tid][c][:h, :w], b.coef[c][i][:h, :w]))
    steps = max(1, min(args.steps, 10))
    elapsed, _ = timed_steps(step, torch.cuda.synchronize, steps, world)
    tot = sum_over_ranks(images, world)
    if rank == 0:
        print(json.dumps({
            "metric": "images/s baseline Huffman encode + decode, 1080p 4:2:0, planes resident on the device", "value": tot * steps / elapsed, "unit": "images/s",
            "n_gpus": world, "steps": steps, "warmup": 1, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "s16 JCOEF <-> Huffman bitstream", "data": "synthetic",
            "config": {"workload": "entropy_1080p_batch", "images_per_gpu": images, "threads": args.threads, "scan_bytes_per_image": sum(nbytes) // images},
            "parity": "decoded == original" if ok else "MISMATCH",
            "roofline": None, "note": "latency-bound chains of small kernels (DESIGN.md 7): no HBM or MFMA roofline applies", "cpu_baseline": None}))


if __name__ == "__main__":
    main()
