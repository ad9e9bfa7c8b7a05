"""In-process multi-device operation on REAL devices: skipped on the one-GPU boxes of the pool, runs on
any host with two or more GPUs (the driver's 8-GPU node).  The same checks on simulated devices run on
every machine: tests/test_multidevice_stub.py."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "libmodjpeg_amd", "host")
TOOLS = os.path.join(ROOT, "tests", "tools")  # mjhost_probe, bench_dropin: test harnesses, built by tests/tools/Makefile
IMAGES = os.path.join(ROOT, "tests", "golden", "images")

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def two_devices():
    import torch
    if torch.cuda.device_count() < 2:  # counting devices does not initialise the GPU
        pytest.skip("fewer than two GPUs")
    import __graft_entry__ as g
    g.build_hip()
    g.build_host_shim()
    return min(torch.cuda.device_count(), 4)


def test_batch_cli_on_two_devices_writes_what_one_device_writes(two_devices, tmp_path):
    """modjpeg --batch --jobs 8 --devices G (worker t on GPU t mod G; images resident on their device,
    tiles and schedules cached per device) == the same command on one device, file by file"""
    files = []
    for i in range(48):
        f = tmp_path / f"in{i:02d}.jpg"
        shutil.copy(os.path.join(IMAGES, "image.jpg" if i % 2 else "image_dropon.jpg"), f)
        files.append(str(f))
    lst = tmp_path / "list.txt"
    lst.write_text("\n".join(files) + "\n")
    script = ["-p", "br", "-m", "-10,-10", "-d", os.path.join(IMAGES, "dropon.png"), "-y", "20", "-b", "-5"]
    outs = []
    for devices in (two_devices, 1):
        out = tmp_path / f"out{devices}"
        out.mkdir()
        subprocess.run([os.path.join(HOST, "modjpeg"), "--batch", str(lst), "--outdir", str(out), "--jobs", "8", "--devices", str(devices), *script], check=True, capture_output=True)
        outs.append(out)
    for name in sorted(os.listdir(outs[1])):
        assert (outs[0] / name).read_bytes() == (outs[1] / name).read_bytes(), name
    assert len(os.listdir(outs[0])) == 48


def test_tile_cache_and_buffer_pools_across_devices(two_devices):
    """the probe's concurrency stress with its threads sharded over the GPUs (every result must equal the
    single-thread result for the same placement), and the recycled image buffers stay on their device"""
    probe = os.path.join(TOOLS, "mjhost_probe")
    res = subprocess.run([probe, "stress", os.path.join(IMAGES, "image.jpg"), os.path.join(IMAGES, "dropon.png"), "12", "30"],
                         env=dict(os.environ, MJ_STRESS_DEVICES=str(two_devices)), capture_output=True, text=True)
    assert res.returncode == 0 and res.stdout.strip() == "ok", (res.stdout[-300:], res.stderr[-800:])
    res = subprocess.run([probe, "shadowpool", os.path.join(IMAGES, "image.jpg")], capture_output=True, text=True)
    assert res.returncode == 0 and res.stdout.strip() == "ok", (res.returncode, res.stdout, res.stderr[-800:])


def test_a_plan_runs_on_the_device_it_was_made_on(two_devices):
    """mjx.h (105): a plan's tables, its side stream and its events belong to the device it was made on; mjx_compose_run from a
    thread whose current device is another one is refused (MJX_ERR_BAD_ARG) instead of launching with foreign pointers"""
    import ctypes
    import numpy as np
    import torch
    from libmodjpeg_amd._lib import MJX_ERR_BAD_ARG, RectJob, check, lib
    torch.cuda.set_device(0)
    coef = torch.zeros((4, 4, 64), dtype=torch.int16, device="cuda:0")
    tile = torch.zeros((4, 64), dtype=torch.float32, device="cuda:0")
    q = np.ones(64, np.uint16)
    job = RectJob(coef.data_ptr(), tile.data_ptr(), tile.data_ptr(), 4, 4, 1, 1, 2, 2, 0, 0)
    plan = ctypes.c_void_p()
    check(lib().mjx_compose_plan_create(ctypes.byref(plan), ctypes.pointer(job), 1, q.ctypes.data_as(ctypes.c_void_p), 1), "plan")
    try:
        check(lib().mjx_compose_run(plan, None), "run on device 0")
        torch.cuda.synchronize()
        torch.cuda.set_device(1)
        assert lib().mjx_compose_run(plan, None) == MJX_ERR_BAD_ARG
    finally:
        torch.cuda.set_device(0)
        lib().mjx_compose_plan_destroy(plan)
