"""The libmodjpeg.h drop-in (libmodjpeg_amd/host/libmodjpeg.so.1: C host on libjpeg + GPU block
loops).  CPU tests cover what runs on the host (JPEG round trip, tile preparation against the tiles
the reference prepared, error behaviour without a device); GPU tests run the README-style example
program end to end and compare its coefficient planes with the vectors dumped from the reference.
Needs jpeglib.h (IJG 9, as in the build container and the GPU box image); skipped otherwise."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "libmodjpeg_amd", "host")
TOOLS = os.path.join(ROOT, "tests", "tools")  # mjhost_probe, bench_dropin: test harnesses, built by tests/tools/Makefile
IMAGES = os.path.join(ROOT, "tests", "golden", "images")

PUBLIC_API = ["mj_init_dropon", "mj_read_dropon_from_raw", "mj_read_dropon_from_memory", "mj_read_dropon_from_file", "mj_init_jpeg",
              "mj_read_jpeg_from_memory", "mj_read_jpeg_from_file", "mj_compose", "mj_write_jpeg_to_memory", "mj_write_jpeg_to_file",
              "mj_free_jpeg", "mj_free_dropon", "mj_effect_grayscale", "mj_effect_pixelate", "mj_effect_tint", "mj_effect_luminance"]
INTERNAL_EXPORTS = ["mj_compose_with_mask", "mj_convolve", "mj_compile_dropon", "mj_free_compileddropon", "mj_free_component",
                    "mj_read_droponimage_from_memory", "mj_read_droponalpha_from_memory"]


@pytest.fixture(scope="module")
def shim():
    if not os.path.exists("/opt/conda/include/jpeglib.h"):
        pytest.skip("no jpeglib.h in this image")
    # torch first: it carries its own HIP runtime, and a process must not end up with two of them
    # (libmjx_hip.so then binds to the one torch has already loaded, as in every other test)
    import torch  # noqa: F401
    import __graft_entry__ as g
    g.build_hip()
    g.build_host_shim()
    so = os.path.join(HOST, "libmodjpeg.so.1")
    assert os.path.exists(so)
    return so


def _planes(path, meta, count=1):
    raw = np.fromfile(path, dtype=np.int16)
    dims = [(int(meta[5 + 4 * c]), int(meta[4 + 4 * c])) for c in range(3)]
    per = sum(h * w * 64 for h, w in dims)
    assert raw.size == per * count
    out = []
    for k in range(count):
        off, planes = k * per, []
        for h, w in dims:
            planes.append(raw[off:off + h * w * 64].reshape(h, w, 64))
            off += h * w * 64
        out.append(planes)
    return out


def test_exports(shim):
    L = ctypes.CDLL(shim)
    for name in PUBLIC_API + INTERNAL_EXPORTS:
        assert hasattr(L, name), name


def test_jpeg_roundtrip_on_host(shim, golden_dir, tmp_path):
    g = np.load(os.path.join(golden_dir, "compose_kat.npz"))
    out = tmp_path / "rt.bin"
    subprocess.check_call([os.path.join(TOOLS, "mjhost_probe"), "roundtrip", os.path.join(IMAGES, "image.jpg"), str(out)])
    a, b = _planes(out, g["meta"], 2)
    for c in range(3):
        assert np.array_equal(a[c].reshape(-1), g[f"before_c{c}"])  # same coefficients as the reference read
        assert np.array_equal(a[c], b[c])  # write (optimised + progressive) -> read is lossless


def _placement(g, case):
    """overlay placement as compose.c:33-172 defines it (test-side restatement)"""
    align, ox, oy = [int(v) for v in g[f"{case}_call"]]
    w, h = int(g[f"{case}_dropon_meta"][0]), int(g[f"{case}_dropon_meta"][1])
    W, H = int(g["meta"][0]), int(g["meta"][1])

    def axis(L, l, near, far, off):
        pos = 0 if near else (L - l if far else L // 2 - l // 2)
        pos += off
        crop = -pos if pos < 0 else 0
        n = l - crop
        if pos + crop + n > L:
            n = L - crop - pos
        return pos, crop, n

    px, cx, cw = axis(W, w, align & 1, align & 2, ox)
    py, cy, ch = axis(H, h, align & 4, align & 8, oy)
    bx = px % 16 if px >= 0 else 0
    by = py % 16 if py >= 0 else 0
    return bx, by, cx, cy, cw, ch


@pytest.mark.parametrize("case", ["tl", "br", "crop"])
def test_tile_preparation_matches_reference(shim, golden_dir, tmp_path, case):
    """mj_compile_dropon of the drop-in == tiles prepared by the reference's mj_compile_dropon
    (same libjpeg): contiguous tiles, alpha DC offset and scaling, padding"""
    g = np.load(os.path.join(golden_dir, "compose_kat.npz"))
    w, h, cs, blend = [int(v) for v in g[f"{case}_dropon_meta"]]
    img = g[f"{case}_dropon_image"].reshape(h, w, 3)
    alp = g[f"{case}_dropon_alpha"].reshape(h, w, 3)
    assert cs == 1  # stored as RGB
    rgba = np.concatenate([img, alp[:, :, :1]], axis=2).astype(np.uint8)
    raw = tmp_path / "raw.bin"
    rgba.tofile(raw)
    bx, by, cx, cy, cw, ch = _placement(g, case)
    out = tmp_path / "tiles.bin"
    subprocess.check_call([os.path.join(TOOLS, "mjhost_probe"), "tiles", os.path.join(IMAGES, "image.jpg"), str(raw), str(w), str(h), "2", "255",
                           str(bx), str(by), str(cx), str(cy), str(cw), str(ch), str(out)])
    buf = open(out, "rb").read()
    off = 0
    for c in range(3):
        wib, hib = np.frombuffer(buf, np.int32, 2, off)
        off += 8
        n = int(wib) * int(hib) * 64
        tile = np.frombuffer(buf, np.float32, n, off)
        off += n * 4
        alpha = np.frombuffer(buf, np.float32, n, off)
        off += n * 4
        rows, cols = int(g[f"{case}_rect_c{c}"][2]), int(g[f"{case}_rect_c{c}"][3])
        assert (hib, wib) == (rows, cols)
        assert np.array_equal(tile.view(np.uint32), g[f"{case}_tile_c{c}"].reshape(-1).view(np.uint32)), (case, c)
        assert np.array_equal(alpha.view(np.uint32), g[f"{case}_alpha_c{c}"].reshape(-1).view(np.uint32)), (case, c)
    assert off == len(buf)


def _read_tiles(path):
    buf = open(path, "rb").read()
    off, out = 0, []
    while off < len(buf):
        wib, hib = np.frombuffer(buf, np.int32, 2, off)
        off += 8
        n = int(wib) * int(hib) * 64
        tile = np.frombuffer(buf, np.float32, n, off)
        off += n * 4
        alpha = np.frombuffer(buf, np.float32, n, off)
        off += n * 4
        out.append((int(hib), int(wib), tile, alpha))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("tiles", ["i16", "f32"])
@pytest.mark.parametrize("case", ["tl", "br", "crop"])
def test_gpu_tile_preparation_matches_reference(shim, golden_dir, tmp_path, case, tiles):
    """tiles prepared ON THE DEVICE on a tile-cache miss (mjx_tiles_prepare: libjpeg 9d's forward DCTs
    as exact integer forms, its colour conversion, DCT-scaled chroma) == the tiles the reference's
    mj_compile_dropon prepared through libjpeg, as float bit patterns: image tiles, alpha DC offset
    and scaling, padding, in-MCU offsets.  The cache keeps them as int16 coefficients by default
    (converted with the reference's expressions by the probe, as the kernel does) or, with
    MJ_AMD=float-tiles, as the floats themselves."""
    g = np.load(os.path.join(golden_dir, "compose_kat.npz"))
    w, h, cs, blend = [int(v) for v in g[f"{case}_dropon_meta"]]
    img = g[f"{case}_dropon_image"].reshape(h, w, 3)
    alp = g[f"{case}_dropon_alpha"].reshape(h, w, 3)
    rgba = np.concatenate([img, alp[:, :, :1]], axis=2).astype(np.uint8)
    raw = tmp_path / "raw.bin"
    rgba.tofile(raw)
    bx, by, cx, cy, cw, ch = _placement(g, case)
    out = tmp_path / "tiles.bin"
    res = subprocess.run([os.path.join(TOOLS, "mjhost_probe"), "tilesgpu", os.path.join(IMAGES, "image.jpg"), str(raw), str(w), str(h), "2", "255",
                          str(bx), str(by), str(cx), str(cy), str(cw), str(ch), str(out)], capture_output=True, text=True,
                         env=dict(os.environ, MJ_AMD="" if tiles == "i16" else "float-tiles"))
    assert res.returncode == 0 and res.stdout.strip() == "gpu", (res.returncode, res.stdout, res.stderr[-300:])
    for c, (hib, wib, tile, alpha) in enumerate(_read_tiles(out)):
        rows, cols = int(g[f"{case}_rect_c{c}"][2]), int(g[f"{case}_rect_c{c}"][3])
        assert (hib, wib) == (rows, cols)
        assert np.array_equal(tile.view(np.uint32), g[f"{case}_tile_c{c}"].reshape(-1).view(np.uint32)), (case, c)
        assert np.array_equal(alpha.view(np.uint32), g[f"{case}_alpha_c{c}"].reshape(-1).view(np.uint32)), (case, c)


@pytest.mark.gpu
@pytest.mark.parametrize("sampling,expect", [("420", "gpu"), ("444", "gpu"), ("422", "gpu"), ("440", "gpu"), ("rgb", "gpu"), ("gray", "gpu"), ("411", "gpu"), ("410", "gpu"),
                                             # round 5: every whole-number ratio libjpeg compresses (<= 10 blocks per MCU): vertical 4 (8x16 transform + its 1x2 box filter),
                                             # 3 in either direction (box filter alone), luma BELOW the chroma resolution, and 4 in both directions (h2v2 + 16x16)
                                             ("1x4,1x1,1x1", "gpu"), ("2x4,1x1,1x1", "gpu"), ("3x1,1x1,1x1", "gpu"), ("3x2,1x1,1x1", "gpu"), ("1x3,1x1,1x1", "gpu"),
                                             ("2x3,1x1,1x1", "gpu"), ("2x2,2x1,1x2", "gpu"), ("4x1,1x4,1x1", "gpu"), ("1x1,2x2,2x2", "gpu")])
def test_gpu_tile_preparation_equals_libjpeg_on_random_overlays(shim, tmp_path, sampling, expect):
    """device-prepared tiles == libjpeg-prepared tiles (the drop-in's host path, itself pinned to the
    reference) on random overlays: RGB / RGBA / YCC / YCCA pixels, odd sizes, every kind of in-MCU
    offset and crop, noise / smooth / saturated contents, translucent blends; 4:2:0, 4:4:4, 4:2:2 and
    4:4:0 targets (libjpeg's 8x8, 16x16, 16x8 and 8x16 transforms), grayscale targets and 4:1:1 / 4:1:0 targets
    (its h2v1 box filter in front of a 16-wide transform) are all prepared on the device"""
    rng = np.random.default_rng(int(sampling) if sampling.isdigit() else (7 if sampling == "rgb" else (9 if sampling == "gray" else sum(map(ord, sampling)))))
    target = str(tmp_path / "target.jpg")
    _make_input(target, ("200", "120"), sampling)
    probe = os.path.join(TOOLS, "mjhost_probe")
    for trial in range(int(os.environ.get("MJX_TILEPREP_TRIALS", "6" if "x" in sampling else "9"))):  # a soak run sets more (profiles/r05/tileprep_soak_40_overlays_17_samplings.log: 40)
        w, h = int(rng.integers(1, 90)), int(rng.integers(1, 70))
        cs = [1, 2, 5, 6][trial % 4] if sampling != "rgb" else [1, 2][trial % 2]  # (libjpeg cannot turn YCbCr pixels into an RGB JPEG)
        bpp = 4 if cs in (2, 6) else 3
        kind = trial % 3
        if kind == 0:
            pix = rng.integers(0, 256, (h, w, bpp), dtype=np.uint8)
        elif kind == 1:
            ramp = np.add.outer(np.arange(h) * int(rng.integers(0, 5)), np.arange(w) * int(rng.integers(0, 5)))[:, :, None] + rng.integers(0, 40, (1, 1, bpp))
            pix = np.clip(ramp, 0, 255).astype(np.uint8)
        else:
            pix = rng.choice(np.array([0, 255], np.uint8), (h, w, bpp))
        raw = tmp_path / f"raw{trial}.bin"
        pix.tofile(raw)
        cx, cy = int(rng.integers(0, w)), int(rng.integers(0, h))
        cw, ch = int(rng.integers(1, w - cx + 1)), int(rng.integers(1, h - cy + 1))
        bx, by = int(rng.integers(0, 32)), int(rng.integers(0, 32))  # (MCUs are up to 32 pixels wide / high with the round-5 samplings; a larger offset only pads more, in both paths)
        blend = [255, 128, 1][trial % 3]
        args = [target, str(raw), str(w), str(h), str(cs), str(blend), str(bx), str(by), str(cx), str(cy), str(cw), str(ch)]
        a, b = tmp_path / f"host{trial}.bin", tmp_path / f"gpu{trial}.bin"
        subprocess.check_call([probe, "tiles", *args, str(a)])
        res = subprocess.run([probe, "tilesgpu", *args, str(b)], capture_output=True, text=True)
        assert res.returncode == 0 and res.stdout.strip() == expect, (trial, res.returncode, res.stdout, res.stderr[-300:])
        ta, tb = _read_tiles(a), _read_tiles(b)
        assert len(ta) == len(tb) == (1 if sampling == "gray" else 3)
        for c in range(len(ta)):
            assert ta[c][:2] == tb[c][:2], (trial, c)
            assert np.array_equal(ta[c][2].view(np.uint32), tb[c][2].view(np.uint32)), (trial, c, "image", args)
            assert np.array_equal(ta[c][3].view(np.uint32), tb[c][3].view(np.uint32)), (trial, c, "alpha", args)


def test_block_loops_fail_loudly_without_device(shim):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    res = subprocess.run([os.path.join(TOOLS, "mjhost_probe"), "nodevice", os.path.join(IMAGES, "image.jpg")], capture_output=True, text=True, check=True)
    lum, compose, null, pix = [int(v) for v in res.stdout.split()]
    assert (lum, compose, null, pix) == (1, 1, 2, 1)  # MJ_ERR_MEMORY, MJ_ERR_MEMORY, MJ_ERR_NULL_DATA, MJ_ERR_MEMORY
    assert "no CPU fallback" in res.stderr


CASES = {"tl": ("dropon.png", "-"), "br": ("dropon.jpg", "mask.jpg"), "crop": ("dropon.png", "-")}


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["tl", "br", "crop"])
def test_example_program_matches_reference(shim, golden_dir, tmp_path, case):
    """the reference README flow (read, read dropon, mj_compose, write) compiled against the
    drop-in: coefficient planes identical to the reference's own output; `tl` is the reference
    repository's image_dropon.jpg"""
    g = np.load(os.path.join(golden_dir, "compose_kat.npz"))
    align, ox, oy = [int(v) for v in g[f"{case}_call"]]
    dropon, mask = CASES[case]
    out_jpg, out_bin = tmp_path / "out.jpg", tmp_path / "planes.bin"
    subprocess.check_call([os.path.join(HOST, "modjpeg_example"), os.path.join(IMAGES, "image.jpg"), os.path.join(IMAGES, dropon),
                           mask if mask == "-" else os.path.join(IMAGES, mask), str(align), str(ox), str(oy), "0", "0", "0", str(out_jpg), str(out_bin)])
    got = _planes(out_bin, g["meta"])[0]
    for c in range(3):
        row0, col0, rows, cols = [int(v) for v in g[f"{case}_rect_c{c}"][:4]]
        want = g[f"before_c{c}"].reshape(got[c].shape).copy()
        want[row0:row0 + rows, col0:col0 + cols] = g[f"{case}_after_rect_c{c}"]
        assert np.array_equal(got[c], want), (case, c)
    assert os.path.getsize(out_jpg) > 1000


@pytest.mark.gpu
def test_example_program_effects(shim, golden_dir, tmp_path):
    g = np.load(os.path.join(golden_dir, "effect_kat.npz"))
    out_jpg, out_bin = tmp_path / "out.jpg", tmp_path / "planes.bin"
    subprocess.check_call([os.path.join(HOST, "modjpeg_example"), os.path.join(IMAGES, "image.jpg"), "-", "-", "0", "0", "0", "100", "50", "0",
                           str(out_jpg), str(out_bin)])
    got = _planes(out_bin, g["meta"])[0]
    for c in range(3):
        assert np.array_equal(got[c].reshape(-1), g[f"lum100_tint50_0_c{c}"]), c


@pytest.mark.gpu
def test_mj_convolve_entry_point(shim, golden_dir):
    g = np.load(os.path.join(golden_dir, "convolve_kat.npz"))
    L = ctypes.CDLL(shim)
    fp = ctypes.POINTER(ctypes.c_float)
    L.mj_convolve.argtypes = [fp, fp, ctypes.c_float, ctypes.c_int, ctypes.c_int]
    L.mj_convolve.restype = None
    for i in range(0, g["x"].shape[0], 7):
        x, y = np.ascontiguousarray(g["x"][i]), g["yin"][i].copy()
        L.mj_convolve(x.ctypes.data_as(fp), y.ctypes.data_as(fp), float(g["w"][i]), int(g["kl"][i, 0]), int(g["kl"][i, 1]))
        assert np.array_equal(y.view(np.uint32), g["yout"][i].view(np.uint32)), i


@pytest.mark.gpu
@pytest.mark.parametrize("sampling", ["420", "422", "440", "444", "gray"])
@pytest.mark.parametrize("size", [("641", "479"), ("200", "100"), ("33", "17"), ("1000", "33")])
def test_gpu_entropy_encoder_byte_identical(shim, size, sampling):
    """mj_write_jpeg_to_memory(MJ_OPTION_NONE): the scan data comes from the GPU Huffman encoder
    (row f-4) and must equal libjpeg's byte for byte - odd sizes exercise the dummy blocks libjpeg
    inserts at the right/bottom edge, the samplings the MCU layouts, gray the non-interleaved scan"""
    import json
    exe = os.path.join(TOOLS, "bench_dropin")
    args = [*size, "16", "16", "3", "9", "-4", "6"]
    env = dict(os.environ, BENCH_SAMPLING=sampling, BENCH_VARY_POSITION="1")
    gpu = json.loads(subprocess.check_output([exe, shim, *args], text=True, env=env))
    cpu = json.loads(subprocess.check_output([exe, shim, *args], text=True, env=dict(env, MJ_AMD="no-gpu-encode")))
    assert gpu["gpu_encodes"] == 4 and cpu["gpu_encodes"] == 0  # 3 images + the warm-up pass
    assert gpu["out_bytes"] == cpu["out_bytes"] and gpu["fnv1a"] == cpu["fnv1a"], (gpu, cpu)
    ref = os.path.join(ROOT, "oracle", "_ref", "libmodjpeg_ref.so")
    if os.path.exists(ref) and sampling != "gray":  # (a colour dropon on a gray image fails in the reference's tile prep)
        r = json.loads(subprocess.check_output([exe, ref, *args], text=True, env=env))
        assert gpu["fnv1a"] == r["fnv1a"]


@pytest.mark.parametrize("mode", ["0", "1", "2"])
def test_optimal_huffman_tables_equal_libjpegs(shim, mode):
    """MJ_OPTION_OPTIMIZE: the tables mjx_huff_optimal_table() builds from a symbol histogram are the
    ones libjpeg 9d writes for the same coefficients (DHT segments compared) - Fibonacci-like counts
    that push code lengths past 16 bits, equal counts (ties everywhere), random counts.  No GPU."""
    for seed in ("1", "2", "3"):
        res = subprocess.run([os.path.join(TOOLS, "mjhost_probe"), "opttables", os.path.join(IMAGES, "image.jpg"), mode, seed], capture_output=True, text=True)
        assert res.returncode == 0 and "DIFFERENT" not in res.stdout and res.stdout.count("same") == 4, (res.stdout, res.stderr[-300:])


@pytest.mark.gpu
@pytest.mark.parametrize("sampling", ["420", "422", "440", "444", "gray"])
@pytest.mark.parametrize("size", [("641", "479"), ("33", "17"), ("1000", "33"), ("1920", "1080")])
def test_gpu_entropy_encoder_with_optimised_tables_byte_identical(shim, size, sampling):
    """mj_write_jpeg_to_memory(MJ_OPTION_OPTIMIZE): symbol statistics gathered on the device, tables
    built as libjpeg's optimize_coding builds them, DHT segments rewritten, scan encoded on the GPU:
    the same bytes as libjpeg writing the file, and as the reference"""
    import json
    exe = os.path.join(TOOLS, "bench_dropin")
    args = [*size, "16", "16", "2", "9", "-4", "6"]
    env = dict(os.environ, BENCH_SAMPLING=sampling, BENCH_WRITE_OPTIONS="1")
    gpu = json.loads(subprocess.check_output([exe, shim, *args], text=True, env=env))
    cpu = json.loads(subprocess.check_output([exe, shim, *args], text=True, env=dict(env, MJ_AMD="no-gpu-encode")))
    assert gpu["gpu_encodes"] == 3 and cpu["gpu_encodes"] == 0
    assert gpu["out_bytes"] == cpu["out_bytes"] and gpu["fnv1a"] == cpu["fnv1a"], (gpu, cpu)
    ref = os.path.join(ROOT, "oracle", "_ref", "libmodjpeg_ref.so")
    if os.path.exists(ref) and sampling != "gray":
        r = json.loads(subprocess.check_output([exe, ref, *args], text=True, env=env))
        assert gpu["fnv1a"] == r["fnv1a"]


@pytest.mark.gpu
@pytest.mark.parametrize("args", [("640", "480", "160", "96", "3"), ("1920", "1080", "256", "256", "2", "7", "5", "-3")])
def test_written_jpegs_byte_identical_to_reference(shim, args):
    """the same program (read -> mj_compose at an unaligned position -> effects -> write) run
    against the unmodified reference library and against the drop-in writes identical files.
    Needs oracle/_ref (built in the container, travels with gpurun)."""
    import json
    ref = os.path.join(ROOT, "oracle", "_ref", "libmodjpeg_ref.so")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref not present")
    exe = os.path.join(TOOLS, "bench_dropin")
    for env in (dict(os.environ, BENCH_VARY_POSITION="1"), dict(os.environ), dict(os.environ, MJ_AMD="no-tile-cache")):
        a = json.loads(subprocess.check_output([exe, ref, *args], text=True, env=env))
        b = json.loads(subprocess.check_output([exe, shim, *args], text=True, env=env))  # tile cache: misses, hits, disabled
        assert a["out_bytes"] == b["out_bytes"] and a["fnv1a"] == b["fnv1a"], (a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("sampling,dropon", [("420", "rgb"), ("420", "ycca"), ("444", "rgb"), ("422", "rgb"), ("440", "ycca"), ("rgb", "rgb"), ("rgb", "rgba"), ("rgb", "ycca"),
                                             ("gray", "gray"), ("gray", "graya"), ("gray", "rgba"), ("420", "gray")])
def test_dropon_colour_spaces_match_reference(shim, sampling, dropon):
    """raw dropon colour spaces x target colour spaces: same files, or the same failure, as the
    reference (a gray dropon is fed to libjpeg with the reference's 3-bytes-as-1 quirk; colour on
    gray and gray on colour fail in libjpeg's colour conversion for both libraries)"""
    import json
    ref = os.path.join(ROOT, "oracle", "_ref", "libmodjpeg_ref.so")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref not present")
    exe = os.path.join(TOOLS, "bench_dropin")
    args = ["200", "120", "48", "40", "2"]
    env = dict(os.environ, BENCH_SAMPLING=sampling, BENCH_DROPON=dropon)
    a = subprocess.run([exe, ref, *args], capture_output=True, text=True, env=env)
    b = subprocess.run([exe, shim, *args], capture_output=True, text=True, env=env)
    assert a.returncode == b.returncode, (a.returncode, b.returncode, b.stderr[-500:])
    if a.returncode == 0:
        ja, jb = json.loads(a.stdout), json.loads(b.stdout)
        assert ja["out_bytes"] == jb["out_bytes"] and ja["fnv1a"] == jb["fnv1a"], (ja, jb)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["host", "gpu-decode", "device-resident"])
def test_concurrent_compose_with_cache_evictions(shim, tmp_path, mode):
    """16 threads, 12 overlay placements (> 8 cache slots), shared arenas: every result equals the
    single-threaded one for the same placement.  gpu-decode: every read goes through the batched GPU
    entropy decoder (concurrent callers share one launch set); device-resident: the images also stay
    on the device between read and write."""
    image = os.path.join(IMAGES, "image.jpg")
    env = dict(os.environ)
    if mode != "host":
        image = str(tmp_path / "big.jpg")
        _make_input(image, ("800", "600"), "420")
        env["MJ_AMD"] = "gpu-decode-min-bytes=0" + (",device-resident" if mode == "device-resident" else "")
    res = subprocess.run([os.path.join(TOOLS, "mjhost_probe"), "stress", image, os.path.join(IMAGES, "dropon.png"), "16", "40"], capture_output=True, text=True,
                         env=env)
    assert res.returncode == 0 and res.stdout.strip() == "ok", (res.stdout, res.stderr[-500:])


def _make_input(path, size, sampling, **knobs):
    """one synthetic JPEG from the bench harness's generator (libjpeg encoder)"""
    env = dict(os.environ, BENCH_SAMPLING=sampling, BENCH_DUMP_INPUT=str(path), MJ_AMD="no-gpu-decode,no-gpu-encode")
    for k, v in knobs.items():
        env["BENCH_" + k.upper()] = str(v)
    subprocess.check_output([os.path.join(TOOLS, "bench_dropin"), os.path.join(HOST, "libmodjpeg.so.1"), *size, "16", "16", "1"], env=env)
    assert os.path.getsize(path) > 100


DECODE_CASES = [  # size, generator knobs
    (("641", "479"), {}),
    (("33", "17"), {}),
    (("1000", "33"), {"restart": 1}),
    (("200", "100"), {"restart": 5, "optimize": 1}),
    (("641", "479"), {"restart": 41, "quality": 100}),
    (("640", "480"), {"quality": 8, "optimize": 1}),
    (("1920", "1080"), {"quality": 95}),
]


@pytest.mark.gpu
@pytest.mark.parametrize("sampling", ["420", "422", "440", "444", "gray"])
@pytest.mark.parametrize("case", range(len(DECODE_CASES)))
def test_gpu_entropy_decoder_matches_libjpeg(shim, tmp_path, sampling, case):
    """mj_read_jpeg_from_memory: the scan data is decoded on the GPU (row f-4, read side) into arrays
    that must equal jpeg_read_coefficients()'s block for block, MCU padding included, together with
    the latched quantisation tables and the dimensions the writer copies - for every MCU layout,
    odd sizes, restart segments, per-image Huffman tables and both ends of the quality scale"""
    size, knobs = DECODE_CASES[case]
    src = tmp_path / "in.jpg"
    _make_input(src, size, sampling, **knobs)
    res = subprocess.run([os.path.join(TOOLS, "mjhost_probe"), "decodecmp", str(src)], capture_output=True, text=True,
                         env=dict(os.environ, MJ_AMD="gpu-decode-min-bytes=0"))
    assert res.returncode == 0 and res.stdout.strip() == "same gpu", (res.stdout, res.stderr[-500:])


ROUND_MODES = {  # how the decoder's synchronisation rounds are walked (MJX_OPTIONS=decode-rounds=...)
    "default": {},
    "a-lane-per-chunk": {"MJX_OPTIONS": "decode-rounds=lane"},
    "a-wavefront-per-chunk": {"MJX_OPTIONS": "decode-rounds=wave"},
    "one-lane-round-then-wavefronts": {"MJX_OPTIONS": "decode-rounds=lane:1"},
    "four-lane-rounds-then-wavefronts": {"MJX_OPTIONS": "decode-rounds=lane:4"},
}


@pytest.mark.gpu
@pytest.mark.parametrize("mode", list(ROUND_MODES))
def test_gpu_entropy_decoder_rounds_walked_either_way(shim, tmp_path, mode):
    """the two kinds of synchronisation round (one lane / one wavefront per chunk) and their mixtures leave the same states
    behind: every choice decodes the same arrays as libjpeg (4:2:0 at two qualities, 4:4:4 with restart markers, grayscale)"""
    env = dict(os.environ, MJ_AMD="gpu-decode-min-bytes=0", **ROUND_MODES[mode])
    for i, (size, sampling, knobs) in enumerate([(("1280", "720"), "420", {}), (("800", "600"), "420", {"quality": 97}), (("641", "479"), "444", {"restart": 7}),
                                                  (("1000", "700"), "gray", {"optimize": 1})]):
        src = tmp_path / ("in%d.jpg" % i)
        _make_input(src, size, sampling, **knobs)
        res = subprocess.run([os.path.join(TOOLS, "mjhost_probe"), "decodecmp", str(src)], capture_output=True, text=True, env=env)
        assert res.returncode == 0 and res.stdout.strip() == "same gpu", (mode, size, sampling, res.stdout, res.stderr[-500:])


@pytest.mark.gpu
def test_gpu_entropy_decoder_declines_what_it_does_not_decode(shim, tmp_path):
    """progressive, arithmetic-coded and small files are read by libjpeg exactly as in the reference"""
    src = tmp_path / "in.jpg"
    _make_input(src, ("320", "240"), "420")
    cli, probe = os.path.join(HOST, "modjpeg"), os.path.join(TOOLS, "mjhost_probe")
    for flag in ("-P", "-A"):
        out = tmp_path / ("out%s.jpg" % flag)
        subprocess.check_call([cli, flag, "-i", str(src), "-o", str(out)])
        res = subprocess.run([probe, "decodecmp", str(out)], capture_output=True, text=True, env=dict(os.environ, MJ_AMD="gpu-decode-min-bytes=0"))
        assert res.stdout.strip() == "same fallback", (flag, res.stdout, res.stderr[-300:])
    res = subprocess.run([probe, "decodecmp", str(src)], capture_output=True, text=True, env=dict(os.environ, MJ_AMD="gpu-decode-min-bytes=100000000"))
    assert res.stdout.strip() == "same fallback"


@pytest.mark.gpu
@pytest.mark.parametrize("sampling,knobs", [("420", {}), ("444", {"restart": 7}), ("gray", {"optimize": 1})])
def test_gpu_entropy_decoder_on_damaged_files(shim, tmp_path, sampling, knobs):
    """flipped bits, random runs, truncation and stray restart markers in the scan data: whatever
    libjpeg makes of the broken stream (it recovers from almost everything), the drop-in returns the
    same arrays or the same error - the GPU decoder refuses anything that is not a clean stream"""
    src = tmp_path / "in.jpg"
    _make_input(src, ("400", "300"), sampling, **knobs)
    res = subprocess.run([os.path.join(TOOLS, "mjhost_probe"), "decodefuzz", str(src), "150", "7"], capture_output=True, text=True,
                         env=dict(os.environ, MJ_AMD="gpu-decode-min-bytes=0"))
    assert res.returncode == 0 and res.stdout.strip().endswith(" 0 differ"), (res.stdout, res.stderr[-500:])


@pytest.mark.gpu
def test_read_compose_write_identical_with_both_entropy_paths_on_the_gpu(shim):
    """end to end with decoder and encoder on the GPU: same bytes as with libjpeg doing both, and
    as the reference library"""
    import json
    exe = os.path.join(TOOLS, "bench_dropin")
    args = ["1920", "1080", "256", "256", "3", "7", "5", "-3"]
    env = dict(os.environ, BENCH_RESTART="120")
    gpu = json.loads(subprocess.check_output([exe, shim, *args], text=True, env=env))
    cpu = json.loads(subprocess.check_output([exe, shim, *args], text=True, env=dict(env, MJ_AMD="no-gpu-decode,no-gpu-encode")))
    assert gpu["gpu_decodes"] >= 4 and cpu["gpu_decodes"] == 0
    assert gpu["out_bytes"] == cpu["out_bytes"] and gpu["fnv1a"] == cpu["fnv1a"], (gpu, cpu)
    ref = os.path.join(ROOT, "oracle", "_ref", "libmodjpeg_ref.so")
    if os.path.exists(ref):
        r = json.loads(subprocess.check_output([exe, ref, *args], text=True, env=env))
        assert gpu["fnv1a"] == r["fnv1a"]


@pytest.mark.gpu
def test_files_from_another_encoder(shim, tmp_path):
    """JPEGs written by PIL (libjpeg-turbo: its own marker layout, all Huffman tables in one DHT segment,
    optimised tables, comments and application segments): the GPU decoder gives libjpeg's arrays, and
    the drop-in's CLI writes what the reference's CLI writes"""
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(3)
    y, x = np.mgrid[0:413, 0:655]
    base = np.stack([(x * 255 // 655), (y * 255 // 413), ((x + y) // 8 * 8) % 256], -1).astype(np.int16)
    img = PIL.fromarray((base + rng.integers(-20, 20, base.shape)).clip(0, 255).astype(np.uint8))
    probe, cli = os.path.join(TOOLS, "mjhost_probe"), os.path.join(HOST, "modjpeg")
    ref_cli = os.path.join(ROOT, "oracle", "_ref", "modjpeg_ref")
    env = dict(os.environ, MJ_AMD="gpu-decode-min-bytes=0")
    k = 0
    for quality in (20, 75, 96):
        for subsampling in (0, 1, 2):
            for optimize in (False, True):
                k += 1
                src = tmp_path / f"pil{k}.jpg"
                img.save(src, quality=quality, subsampling=subsampling, optimize=optimize, comment=b"made by PIL " * 50, dpi=(300, 300))
                res = subprocess.run([probe, "decodecmp", str(src)], capture_output=True, text=True, env=env)
                assert res.stdout.strip() == "same gpu", (quality, subsampling, optimize, res.stdout, res.stderr[-300:])
                if os.path.exists(ref_cli) and k % 3 == 0:
                    a, b = tmp_path / "a.jpg", tmp_path / "b.jpg"
                    argv = ["-i", str(src), "-p", "br", "-m", "-7,-5", "-d", os.path.join(IMAGES, "dropon.png"), "-y", "15"]
                    subprocess.run([ref_cli] + argv + ["-o", str(a)], check=True, capture_output=True)
                    subprocess.run([cli] + argv + ["-o", str(b)], check=True, capture_output=True, env=env)
                    assert a.read_bytes() == b.read_bytes(), (quality, subsampling, optimize)
    gray = tmp_path / "gray.jpg"
    img.convert("L").save(gray, quality=88, optimize=True)
    res = subprocess.run([probe, "decodecmp", str(gray)], capture_output=True, text=True, env=env)
    assert res.stdout.strip() == "same gpu", res.stdout
    prog = tmp_path / "prog.jpg"
    img.save(prog, quality=80, progressive=True)
    res = subprocess.run([probe, "decodecmp", str(prog)], capture_output=True, text=True, env=env)
    assert res.stdout.strip() == "same fallback", res.stdout


@pytest.mark.gpu
def test_mj_compose_takes_the_pattern_schedule_and_int16_tiles(shim):
    """the public API runs the kernel paths the batch plans do: an mj_compose of
    the reference's own logo overlay (tests/golden/images/dropon.png: text, i.e. blocks with very
    different zero patterns) reaches the compose kernel with the tiles as int16 coefficients and, from
    the second use of the cached tiles on (the schedule is made at an entry's first hit, not for tiles
    that are used once), with a schedule that picks the blocks of an octet by pattern; MJ_AMD=float-tiles /
    MJ_AMD=no-tile-schedule switch each off.  (That the files stay byte-identical to the reference's is tests/test_cli.py and
    test_example_program_matches_reference, which run through this default.)"""
    probe = os.path.join(TOOLS, "mjhost_probe")
    args = [probe, "composepath", os.path.join(IMAGES, "image.jpg"), os.path.join(IMAGES, "dropon.png"), "4"]
    rects, i16, sched, gpu = [int(v) for v in subprocess.check_output(args, text=True).split()]
    assert rects == 12 and i16 == 12 and gpu == 1, (rects, i16, sched, gpu)
    assert 3 <= sched <= 9, "the luma rectangle of a text overlay has blocks of very different patterns: calls 2..4 carry its schedule, call 1 (the miss) none"
    rects, i16, sched, gpu = [int(v) for v in subprocess.check_output(args, text=True, env=dict(os.environ, MJ_AMD="float-tiles")).split()]
    assert rects == 12 and i16 == 0 and 3 <= sched <= 9
    rects, i16, sched, gpu = [int(v) for v in subprocess.check_output(args, text=True, env=dict(os.environ, MJ_AMD="no-tile-schedule")).split()]
    assert rects == 12 and i16 == 12 and sched == 0
