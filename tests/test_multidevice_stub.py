"""In-process multi-device operation of the drop-in's host code, on simulated devices (CPU, no GPU).

tests/tools/mjx_stub.c stands in for libmjx_hip.so: MJX_STUB_DEVICES=n devices, per-thread current
device, "device memory" tagged with the device it was allocated on, and an abort whenever memory, a
tile schedule or an image plane of one device reaches an entry point called by a thread of another
(the drop-in sets up no peer access: SURVEY.md 8e, image i -> device, nothing crosses).  The host
library, the CLI and the probe are built against it from the product's own sources.
MJX_STUB_SANITIZE=1 builds everything with ASan + UBSan (scripts/asan_cpu.sh does).
Reference caller: /root/reference/src/contrib/modjpeg.c:52-179 (one process, one device there)."""
import collections
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "libmodjpeg_amd", "host")
IMAGES = os.path.join(ROOT, "tests", "golden", "images")
JPEG = "/opt/conda"
HOST_SRCS = ["jpegio.c", "dropon.c", "tiles.c", "tilecache.c", "compose.c", "effects.c", "gpuencode.c", "gpudecode.c", "shadow.c", "options.c"]

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(JPEG, "include", "jpeglib.h")), reason="no jpeglib.h in this image")


@pytest.fixture(scope="module")
def stubbed(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("stub"))
    san = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"] if os.environ.get("MJX_STUB_SANITIZE") == "1" else ["-O1", "-g"]
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(JPEG, "include")]
    png = ["-DWITH_LIBPNG", "-I" + os.path.join(JPEG, "include", "libpng16")] if os.path.exists(os.path.join(JPEG, "include", "libpng16", "png.h")) else []
    libs = [os.path.join(JPEG, "lib", "libjpeg.so")] + ([os.path.join(JPEG, "lib", "libpng16.so")] if png else [])
    run = lambda cmd, **kw: subprocess.run(cmd, check=True, capture_output=True, text=True, **kw)
    run(["gcc", *san, "-fPIC", "-std=gnu99", "-Wall", *inc, "-shared", "-o", os.path.join(d, "libmjx_hip.so"), os.path.join(ROOT, "tests", "tools", "mjx_stub.c")])
    run(["gcc", *san, "-fPIC", "-std=gnu99", "-Wall", *inc, *png, "-shared", "-Wl,-soname,libmodjpeg.so.1", "-o", os.path.join(d, "libmodjpeg.so.1"),
         *HOST_SRCS, *libs, "-Wl,-rpath," + os.path.join(JPEG, "lib"), "-L" + d, "-Wl,-rpath," + d, "-lmjx_hip", "-lm", "-lpthread"], cwd=HOST)
    os.symlink("libmodjpeg.so.1", os.path.join(d, "libmodjpeg.so"))
    for prog, src in (("modjpeg", "modjpeg.c"), ("probe", os.path.join(ROOT, "tests", "tools", "probe.c"))):
        run(["gcc", *san, "-std=gnu99", *inc, "-o", os.path.join(d, prog), src, "-L" + d, "-Wl,-rpath," + d, "-Wl,--allow-shlib-undefined", "-lmodjpeg", "-lpthread", "-ldl", "-lm",
             *libs, "-Wl,-rpath," + os.path.join(JPEG, "lib")], cwd=HOST)
    env = dict(os.environ, MJ_AMD="no-gpu-decode,no-gpu-encode", ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    return d, env


def _trace(path):
    rows = []
    for line in open(path):
        what, *kv = line.split()
        rows.append((what, dict(f.split("=", 1) for f in kv)))
    return rows


def test_stub_defines_the_whole_abi(stubbed):
    """every function include/mjx.h declares exists in the stand-in (else the host library would not link)"""
    import re
    d, _ = stubbed
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "mjx.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(mjx_[a-z0-9_]+)\s*\(", hdr))
    have = {l.split()[-1] for l in subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(d, "libmjx_hip.so")], text=True).splitlines()}
    assert declared <= have, declared - have


def test_batch_cli_shards_its_workers_over_two_devices(stubbed, tmp_path):
    """modjpeg --batch --jobs 8 --devices 2: worker t works on device t mod 2 for its whole life, the tile
    cache hands every worker tiles (and a schedule) of ITS device, prepared once per device"""
    d, env = stubbed
    files = []
    for i in range(32):
        f = tmp_path / f"in{i:02d}.jpg"
        shutil.copy(os.path.join(IMAGES, "image.jpg"), f)
        files.append(str(f))
    lst = tmp_path / "list.txt"
    lst.write_text("\n".join(files) + "\n")
    out = tmp_path / "out"
    out.mkdir()
    trace = tmp_path / "trace.txt"
    res = subprocess.run([os.path.join(d, "modjpeg"), "--batch", str(lst), "--outdir", str(out), "--jobs", "8", "--devices", "2", "-d", os.path.join(IMAGES, "dropon.png"), "-y", "10"],
                         env=dict(env, MJX_STUB_DEVICES="2", MJX_STUB_TRACE=str(trace)), capture_output=True, text=True)
    assert res.returncode == 0, (res.returncode, res.stdout[-500:], res.stderr[-1500:])
    assert len(os.listdir(out)) == 32
    rows = _trace(trace)
    devs_of_thread = collections.defaultdict(set)
    for what, kv in rows:
        devs_of_thread[kv["thread"]].add(kv["device"])
    assert all(len(v) == 1 for v in devs_of_thread.values()), "a worker changed its device"
    per_dev = collections.Counter(next(iter(v)) for v in devs_of_thread.values())
    assert set(per_dev) == {"0", "1"} and max(per_dev.values()) <= 4, per_dev  # 8 workers: t mod 2
    composes = [kv for what, kv in rows if what == "compose"]
    assert len(composes) == 32 * 3
    # One overlay, one target format, two devices.  Only what holds for EVERY interleaving of the eight workers is asserted
    # (a lower bound on the scheduled composes is a matter of timing - a worker's next image can arrive while another
    # worker is still making the schedule):
    #  - an entry's tiles are scheduled when they are used the second time, so on each device the first use of every component
    #    has no schedule;
    #  - exactly one caller makes an entry's schedules (sched_state 0 -> 1 under the lock), one per component; a device has one
    #    entry for this overlay, or one more per worker that missed at the same moment as another (4 workers per device): a
    #    schedule that was rebuilt on every use would show as dozens of rows here;
    #  - each device has at least 16 uses of the entry, so the second use happens and the schedules ARE made, and whatever is
    #    composed after they are published uses them: at least the very last compose of the run does.
    first_seen = set()
    for kv in composes:
        key = (kv["device"], kv["rect"])
        if key not in first_seen:
            first_seen.add(key)
            assert kv["schedule"] == "0", ("the first use of an entry is never scheduled", kv)
    scheds = collections.Counter(kv["device"] for what, kv in rows if what == "schedule")
    assert set(scheds) == {"0", "1"} and all(3 <= n <= 4 * 3 for n in scheds.values()), scheds
    assert sum(kv["schedule"] == "1" for kv in composes) >= 1
    # the same command on ONE device must write the same files (nothing depends on the sharding)
    out1 = tmp_path / "out1"
    out1.mkdir()
    subprocess.run([os.path.join(d, "modjpeg"), "--batch", str(lst), "--outdir", str(out1), "--jobs", "3", "-d", os.path.join(IMAGES, "dropon.png"), "-y", "10"],
                   env=dict(env, MJX_STUB_DEVICES="1"), check=True, capture_output=True)
    for name in os.listdir(out):
        assert open(out / name, "rb").read() == open(out1 / name, "rb").read(), name


def test_tile_cache_never_hands_out_another_devices_tiles(stubbed):
    """16 threads on 3 devices, 12 placements of one overlay, more (device, placement) pairs than the cache
    has slots: hits, misses and evictions of all devices interleave; the stand-in aborts on the first
    tile, schedule or staging buffer that reaches a thread of another device"""
    d, env = stubbed
    res = subprocess.run([os.path.join(d, "probe"), "stress", os.path.join(IMAGES, "image.jpg"), os.path.join(IMAGES, "dropon.png"), "16", "40"],
                         env=dict(env, MJX_STUB_DEVICES="3", MJ_STRESS_DEVICES="3"), capture_output=True, text=True)
    assert res.returncode == 0 and res.stdout.strip() == "ok", (res.returncode, res.stdout[-300:], res.stderr[-1500:])


def test_recycled_image_buffers_stay_on_their_device(stubbed):
    """device-resident images (shadow.c): a plane buffer released on GPU 0 is not handed to an image of
    GPU 1, and is reused on GPU 0"""
    d, env = stubbed
    res = subprocess.run([os.path.join(d, "probe"), "shadowpool", os.path.join(IMAGES, "image.jpg")], env=dict(env, MJX_STUB_DEVICES="2"), capture_output=True, text=True)
    assert res.returncode == 0 and res.stdout.strip() == "ok", (res.returncode, res.stdout, res.stderr[-1500:])


def test_no_simulated_device_is_no_device(stubbed):
    """MJX_STUB_DEVICES unset: the stand-in answers like a machine without a GPU and the library says so
    (MJ_ERR_MEMORY + a diagnostic, no CPU fallback)"""
    d, env = stubbed
    res = subprocess.run([os.path.join(d, "probe"), "nodevice", os.path.join(IMAGES, "image.jpg")], env=env, capture_output=True, text=True)
    assert res.returncode == 0 and res.stdout.split()[:2] == ["1", "1"], (res.stdout, res.stderr[-500:])


def test_the_drop_in_reads_one_environment_variable(stubbed):
    """host/options.c: MJ_AMD is a comma-separated list of tokens, parsed once; an unknown token is reported once on stderr and
    ignored, the known ones beside it still count (here `debug`: the probe's read of a progressive file says why it fell back)"""
    d, env = stubbed
    res = subprocess.run([os.path.join(d, "probe"), "nodevice", os.path.join(IMAGES, "image.jpg")], env=dict(env, MJ_AMD="bogus-token,no-gpu-decode,,no-gpu-encode"), capture_output=True, text=True)
    assert res.returncode == 0, (res.stdout, res.stderr[-500:])
    assert res.stderr.count('MJ_AMD: unknown option "bogus-token" ignored') == 1, res.stderr[-800:]
    # the round-4 names are not read any more: setting them changes nothing and warns about nothing
    res = subprocess.run([os.path.join(d, "probe"), "nodevice", os.path.join(IMAGES, "image.jpg")], env=dict({k: v for k, v in env.items() if k != "MJ_AMD"}, MJ_GPU_DECODE="0"), capture_output=True,
                         text=True)
    assert "unknown option" not in res.stderr
