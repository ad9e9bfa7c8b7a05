"""modjpeg command line tool of the drop-in vs the reference's own CLI (oracle/_ref/modjpeg_ref,
built from the reference sources where they lie): the same argv must write the same bytes.
Exercises the order-dependent option semantics (reference src/contrib/modjpeg.c:63-173) and the
batch mode."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "libmodjpeg_amd", "host")
TOOLS = os.path.join(ROOT, "tests", "tools")  # mjhost_probe, bench_dropin: test harnesses, built by tests/tools/Makefile
IMAGES = os.path.join(ROOT, "tests", "golden", "images")
REF_CLI = os.path.join(ROOT, "oracle", "_ref", "modjpeg_ref")

pytestmark = pytest.mark.gpu

I = os.path.join(IMAGES, "image.jpg")
PNG = os.path.join(IMAGES, "dropon.png")
JPG = os.path.join(IMAGES, "dropon.jpg") + "," + os.path.join(IMAGES, "mask.jpg")

SCRIPTS = [
    ["-i", I, "-p", "tl", "-d", PNG, "-o", "OUT"],                                   # the README example
    ["-i", I, "-p", "br", "-m", "-10,-10", "-d", JPG, "-O", "-o", "OUT"],
    ["-i", I, "-x", "-p", "tr", "-d", PNG, "-o", "OUT"],                             # pixelate, then logo
    ["-i", I, "-p", "tr", "-d", PNG, "-x", "-o", "OUT"],                             # logo, then pixelate both
    ["-i", I, "-y", "40", "-b", "-25", "-r", "15", "-P", "-o", "OUT"],
    ["-i", I, "-g", "-p", "cc", "-m", "3", "-d", PNG, "-d", JPG, "-o", "OUT"],       # offset without y, two dropons
    ["-i", I, "-p", "xyz", "-m", "-300,-300", "-d", PNG, "-p", "bl", "-m", "7,9", "-d", os.path.join(IMAGES, "dropon.jpg"), "-o", "OUT"],
]


@pytest.fixture(scope="module")
def cli():
    if not os.path.exists(REF_CLI):
        pytest.skip("oracle/_ref/modjpeg_ref not present")
    import torch  # noqa: F401
    import __graft_entry__ as g
    g.build_hip()
    g.build_host_shim()
    return os.path.join(HOST, "modjpeg")


@pytest.mark.parametrize("script", SCRIPTS, ids=[str(i) for i in range(len(SCRIPTS))])
def test_same_argv_same_bytes(cli, tmp_path, script):
    outs = []
    for exe, name in ((REF_CLI, "ref.jpg"), (cli, "new.jpg")):
        out = tmp_path / name
        argv = [str(out) if a == "OUT" else a for a in script]
        subprocess.run([exe] + argv, check=True, capture_output=True)
        outs.append(out.read_bytes())
    assert len(outs[0]) > 1000 and outs[0] == outs[1]


@pytest.mark.parametrize("sampling", ["1x4,1x1,1x1", "2x4,1x1,1x1", "3x1,1x1,1x1", "1x3,1x1,1x1", "3x2,1x1,1x1", "2x2,2x1,1x2", "4x1,1x4,1x1", "1x1,2x2,2x2"])
def test_unusual_samplings_same_bytes(cli, tmp_path, sampling):
    """targets whose components are subsampled 3:1 or 4:1, vertically too, or whose luma lies BELOW the chroma resolution: the overlay's
    tiles are prepared on the device (libjpeg's box filters in front of its scaled transforms, mjx_tileprep.hip), composed, written:
    the same bytes as the reference CLI, which prepares them through libjpeg (dropon.c:325-428)"""
    src = tmp_path / "in.jpg"
    env = dict(os.environ, BENCH_SAMPLING=sampling, BENCH_DUMP_INPUT=str(src), MJ_AMD="no-gpu-decode,no-gpu-encode")
    subprocess.check_output([os.path.join(TOOLS, "bench_dropin"), os.path.join(HOST, "libmodjpeg.so.1"), "403", "299", "16", "16", "1"], env=env)
    for script in (["-p", "cc", "-m", "5,-3", "-d", PNG, "-y", "12"], ["-p", "br", "-m", "-7,-9", "-d", JPG, "-b", "-8", "-O"]):
        outs = []
        for exe, name in ((REF_CLI, "ref.jpg"), (cli, "new.jpg")):
            out = tmp_path / name
            res = subprocess.run([exe, "-i", str(src)] + script + ["-o", str(out)], capture_output=True, text=True, env=dict(os.environ, MJ_AMD="debug-tiles"))
            assert res.returncode == 0, (exe, res.stderr[-400:])
            outs.append(out.read_bytes())
            if exe == cli:  # the tile cache's trace says where the tiles were made
                misses = [l for l in res.stderr.splitlines() if l.startswith("tilecache: miss")]
                assert misses and all(", device," in l for l in misses), res.stderr[-600:]
        assert len(outs[0]) > 1000 and outs[0] == outs[1], (sampling, script)


def test_error_exit_codes(cli, tmp_path):
    for exe in (REF_CLI, cli):
        assert subprocess.run([exe, "-i", "/nonexistent.jpg"], capture_output=True).returncode == 1
        assert subprocess.run([exe, "-i", I, "-d", "/nonexistent.png"], capture_output=True).returncode == 1


def test_batch_mode_equals_one_by_one(cli, tmp_path):
    src = tmp_path / "in"
    src.mkdir()
    names = []
    for k in range(12):  # same picture under 12 names is enough to exercise the thread pool
        p = src / f"img{k:02d}.jpg"
        shutil.copy(I, p)
        names.append(str(p))
    (tmp_path / "list.txt").write_text("\n".join(names) + "\n")
    out = tmp_path / "out"
    out.mkdir()
    script = ["-p", "br", "-m", "-10,-10", "-d", JPG, "-y", "20", "-O"]
    subprocess.run([cli, "--batch", str(tmp_path / "list.txt"), "--outdir", str(out), "--jobs", "4"] + script, check=True, capture_output=True)
    ref = tmp_path / "one.jpg"
    subprocess.run([REF_CLI, "-i", I] + script + ["-o", str(ref)], check=True, capture_output=True)
    want = ref.read_bytes()
    for k in range(12):
        assert (out / f"img{k:02d}.jpg").read_bytes() == want, k


BATCH_SCRIPTS = [
    ["-p", "br", "-m", "-10,-10", "-d", PNG, "-y", "20"],        # compose + luminance, GPU encoder
    ["-x", "-p", "tr", "-d", PNG],                                # pixelate, then logo
    ["-g", "-b", "-25", "-r", "15", "-y", "-30"],                 # grayscale, tint (no-ops on zeroed chroma DCs aside), luminance
    ["-p", "cc", "-d", JPG, "-O"],                                # optimised output: statistics and tables for the image, still on the device
    ["-y", "10", "-P"],                                           # progressive output
    ["-p", "cc", "-m", "5,-3", "-d", os.path.join(IMAGES, "image_dropon.jpg"), "-d", PNG],  # a 256x256 dropon (thousands of blocks per launch), then a small one
    # DC effects on a device-resident image are only noted and applied by the encoder as it reads the blocks (effects.c):
    ["-y", "10", "-y", "-20", "-y", "35", "-y", "5", "-y", "-60", "-b", "7", "-r", "-9"],   # more pending values than the encoder takes: the oldest are run first
    ["-y", "30", "-b", "-12", "-p", "tl", "-d", PNG, "-y", "-10", "-r", "4"],               # pending, then a compose that has to see them, then pending again
    ["-b", "20", "-g", "-r", "5", "-x", "-y", "2000", "-y", "-3000"],                       # grayscale drops what was pending on chroma; pixelate keeps; clamps at +-2047
    ["-y", "25", "-b", "3", "-P"],                                                          # libjpeg has to write: the pending effects run before the download
]


@pytest.mark.parametrize("sampling", ["420", "444", "gray", "411"])
@pytest.mark.parametrize("script", BATCH_SCRIPTS, ids=[str(i) for i in range(len(BATCH_SCRIPTS))])
def test_batch_pipeline_keeps_images_on_the_device_and_matches_the_reference(cli, tmp_path, script, sampling):
    """batch mode decodes on the GPU and leaves the planes there (shadow.c): compose, effects and the
    encoder work on the device copy, libjpeg's host arrays are only filled when libjpeg itself has
    to write.  Every output must equal what the reference CLI writes for the same file and script."""
    src = tmp_path / "in"
    src.mkdir()
    names = []
    for k in range(6):
        p = src / f"img{k}.jpg"
        env = dict(os.environ, BENCH_SAMPLING=sampling, BENCH_DUMP_INPUT=str(p), BENCH_QUALITY=str(60 + 7 * k), MJ_AMD="no-gpu-decode,no-gpu-encode")
        subprocess.check_output([os.path.join(TOOLS, "bench_dropin"), os.path.join(HOST, "libmodjpeg.so.1"), str(400 + 33 * k), str(300 + 17 * k), "16", "16", "1"], env=env)
        names.append(str(p))
    (tmp_path / "list.txt").write_text("\n".join(names) + "\n")
    out = tmp_path / "out"
    out.mkdir()
    res = subprocess.run([cli, "--batch", str(tmp_path / "list.txt"), "--outdir", str(out), "--jobs", "3"] + script, capture_output=True, text=True,
                         env=dict(os.environ, MJ_AMD="cli-timing,gpu-decode-min-bytes=0"))
    assert res.returncode == 0, res.stderr[-500:]
    import re
    kept = re.search(r"(\d+) kept on the device", res.stderr)
    assert kept and int(kept.group(1)) >= 6, res.stderr[-500:]  # (the dropon's own tiles are read back through the same path)
    for k, name in enumerate(names):
        ref = tmp_path / f"ref{k}.jpg"
        subprocess.run([REF_CLI, "-i", name] + script + ["-o", str(ref)], check=True, capture_output=True)
        assert (out / f"img{k}.jpg").read_bytes() == ref.read_bytes(), k
