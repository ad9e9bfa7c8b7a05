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
