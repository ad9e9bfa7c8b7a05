"""GPU parity tests (-m gpu): every golden vector dumped from the reference, and randomised
differential tests against the CPU oracle, replayed through the C ABI of include/mjx.h
(libmjx_hip.so).  Bar: bit-exact JCOEF / float bit patterns."""
import ctypes
import os

import numpy as np
import pytest

from tests import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mj():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import libmodjpeg_amd
    return libmodjpeg_amd


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_convolve_kat(mj, golden_dir):
    g = _load(golden_dir, "convolve_kat.npz")
    for i in range(g["x"].shape[0]):
        x, y = _t(g["x"][i:i + 1]), _t(g["yin"][i:i + 1])
        mj.mj_convolve(x, y, g["w"][i], int(g["kl"][i, 0]), int(g["kl"][i, 1]))
        assert np.array_equal(y.cpu().numpy().view(np.uint32), g["yout"][i:i + 1].view(np.uint32)), (i, g["kl"][i])


def test_convolve_batched_random_vs_oracle(mj):
    rng = np.random.default_rng(7)
    for k in range(8):
        for l in range(8):
            n = int(rng.integers(1, 70))  # ragged: not a multiple of the 8-block octet
            x = (rng.standard_normal((n, 64)) * 100).astype(np.float32)
            y0 = (rng.standard_normal((n, 64)) * 100).astype(np.float32)
            w = np.float32(rng.standard_normal() * 0.1)
            yd = _t(y0)
            mj.mj_convolve(_t(x), yd, w, k, l)
            want = y0.copy()
            for b in range(n):
                O.convolve(x[b], want[b], w, k, l)
            assert np.array_equal(yd.cpu().numpy().view(np.uint32), want.view(np.uint32)), (k, l)


def test_block_kat(mj, golden_dir):
    """reference mj_compose_with_mask on fabricated blocks: smooth/dense alpha, exact blends,
    int16 wrap, near-integer Y, sparse, NaN/Inf/out-of-int-range"""
    g = _load(golden_dir, "block_kat.npz")
    nfam = g["q"].shape[0]
    per_fam = g["coef_in"].shape[0] // nfam
    for fam in range(nfam):
        geom = g["geom"][fam]
        off = fam * per_fam
        planes, tiles, alphas, want = [], [], [], []
        for c in range(3):
            row0, col0, rows, cols, wib, hib = [int(v) for v in geom[c][:6]]
            n = rows * cols
            cin = g["coef_in"][off:off + n].reshape(rows, cols, 64)
            plane = np.full((hib, wib, 64), 77, np.int16)  # sentinel outside the rectangle
            plane[row0:row0 + rows, col0:col0 + cols] = cin
            exp = plane.copy()
            exp[row0:row0 + rows, col0:col0 + cols] = g["coef_out"][off:off + n].reshape(rows, cols, 64)
            planes.append(plane)
            want.append(exp)
            tiles.append(mj.Component(_t(g["tile"][off:off + n]), cols, rows))
            alphas.append(mj.Component(_t(g["alpha"][off:off + n]), cols, rows))
            off += n
        m = mj.Jpeg([_t(p) for p in planes], [g["q"][fam, c] for c in range(3)])
        assert mj.mj_compose_with_mask(m, mj.CompiledDropon(tiles, alphas), int(geom[0][6]), int(geom[0][7])) == mj.MJ_OK
        got = m.planes_numpy()
        for c in range(3):
            assert np.array_equal(got[c], want[c]), (fam, c, int((got[c] != want[c]).sum()))


@pytest.mark.parametrize("case", ["tl", "br", "crop"])
def test_compose_rect_kat(mj, golden_dir, case):
    """mj_compose end-to-end dumps; `tl` == the reference repo's image_dropon.jpg"""
    g = _load(golden_dir, "compose_kat.npz")
    planes, tiles, alphas, quant = [], [], [], []
    for c in range(3):
        wib, hib = int(g["meta"][4 + 4 * c]), int(g["meta"][5 + 4 * c])
        rows, cols = int(g[f"{case}_rect_c{c}"][2]), int(g[f"{case}_rect_c{c}"][3])
        planes.append(g[f"before_c{c}"].reshape(hib, wib, 64))
        tiles.append(mj.Component(_t(g[f"{case}_tile_c{c}"]), cols, rows))
        alphas.append(mj.Component(_t(g[f"{case}_alpha_c{c}"]), cols, rows))
        quant.append(g[f"q_c{c}"])
    m = mj.Jpeg([_t(p) for p in planes], quant)
    bx, by = int(g[f"{case}_rect_c0"][4]), int(g[f"{case}_rect_c0"][5])
    assert mj.mj_compose_with_mask(m, mj.CompiledDropon(tiles, alphas), bx, by) == mj.MJ_OK
    got = m.planes_numpy()
    for c in range(3):
        row0, col0, rows, cols = [int(v) for v in g[f"{case}_rect_c{c}"][:4]]
        want = planes[c].copy()
        want[row0:row0 + rows, col0:col0 + cols] = g[f"{case}_after_rect_c{c}"]
        assert np.array_equal(got[c], want), (case, c)


def test_effect_kat(mj, golden_dir):
    g = _load(golden_dir, "effect_kat.npz")
    dims = [(int(g["meta"][5 + 4 * c]), int(g["meta"][4 + 4 * c])) for c in range(3)]
    m = mj.Jpeg([_t(g[f"before_c{c}"].reshape(dims[c][0], dims[c][1], 64)) for c in range(3)], [g[f"q_c{c}"] for c in range(3)])

    def same(tag):
        got = m.planes_numpy()
        for c in range(3):
            assert np.array_equal(got[c].reshape(-1), g[f"{tag}_c{c}"]), (tag, c)

    assert mj.mj_effect_luminance(m, 100) == mj.MJ_OK
    assert mj.mj_effect_tint(m, 50, 0) == mj.MJ_OK
    same("lum100_tint50_0")
    mj.mj_effect_tint(m, -300, 77)
    mj.mj_effect_luminance(m, -1900)
    same("clamp")
    mj.mj_effect_luminance(m, 0)
    mj.mj_effect_tint(m, 0, 0)
    same("zero")
    mj.mj_effect_pixelate(m)
    same("pixelate")
    mj.mj_effect_grayscale(m)
    same("grayscale")


def test_effect_wrap_kat_and_non_ycc_noop(mj, golden_dir):
    g = _load(golden_dir, "effect_kat.npz")
    dims = [(int(g["w_meta"][5 + 4 * c]), int(g["w_meta"][4 + 4 * c])) for c in range(3)]
    planes = [g[f"w_before_c{c}"].reshape(dims[c][0], dims[c][1], 64) for c in range(3)]
    m = mj.Jpeg([_t(p) for p in planes], [g[f"w_q_c{c}"] for c in range(3)])
    mj.mj_effect_luminance(m, 40000)
    mj.mj_effect_tint(m, -70000, 123456789)
    got = m.planes_numpy()
    for c in range(3):
        assert np.array_equal(got[c].reshape(-1), g[f"w_after_c{c}"]), c
    # effects on a non-YCbCr image are silent no-ops (effect.c:39/:126/:195); pixelate is not
    rgb = mj.Jpeg([_t(p) for p in planes], [g[f"w_q_c{c}"] for c in range(3)], colorspace=mj.JCS_RGB)
    assert mj.mj_effect_luminance(rgb, 5) == mj.MJ_OK and mj.mj_effect_tint(rgb, 5, 5) == mj.MJ_OK and mj.mj_effect_grayscale(rgb) == mj.MJ_OK
    for c in range(3):
        assert np.array_equal(rgb.planes_numpy()[c], planes[c])
    assert mj.mj_compose_with_mask(None, None, 0, 0) == mj.MJ_ERR_NULL_DATA  # compose.c:238


@pytest.mark.parametrize("regime", [1, 2, 3])
def test_random_rects_vs_oracle(mj, regime):
    """ragged rectangles (cols not a multiple of the 8-block octet, 1xN, Nx1), offsets, all
    three alpha regimes of the synthetic workload, vs the oracle"""
    rng = np.random.default_rng(100 + regime)
    for trial in range(6):
        wib, hib = int(rng.integers(8, 60)), int(rng.integers(8, 40))
        cols, rows = int(rng.integers(1, wib + 1)), int(rng.integers(1, hib + 1))
        if trial == 0:
            wib, hib = max(wib, 20), max(hib, 19)
            cols, rows = 17, 17
        if trial == 1:
            cols, rows = 1, hib
        if trial == 2:
            cols, rows = wib, 1
        col0, row0 = int(rng.integers(0, wib - cols + 1)), int(rng.integers(0, hib - rows + 1))
        q = O.quant_table(trial % 2 == 1)
        plane = O.synth_coef(trial, 0, 0, wib * hib, q)
        tile = O.synth_tile(trial, 0, 0, rows * cols)
        alpha = O.synth_alpha(trial, 0, 0, rows * cols, regime)
        m = mj.Jpeg([_t(plane.reshape(hib, wib, 64))], [q], samp=[(1, 1)], colorspace=mj.JCS_GRAYSCALE)
        cd = mj.CompiledDropon([mj.Component(_t(tile), cols, rows)], [mj.Component(_t(alpha), cols, rows)])
        # block_x/block_y are MCU units; with 1x1 sampling they are block units
        assert mj.mj_compose_with_mask(m, cd, col0, row0) == mj.MJ_OK
        want = plane.copy()
        O.compose_rect(want, wib, row0, col0, rows, cols, q, tile, alpha)
        assert np.array_equal(m.planes_numpy()[0].reshape(-1), want), (regime, trial, wib, hib, rows, cols)


def test_synth_generators_match_cpu(mj):
    import torch
    from libmodjpeg_amd._lib import lib, check
    q = O.quant_table(False)
    n, first = 301, 12345
    d16 = torch.empty(n * 64, dtype=torch.int16, device="cuda")
    df = torch.empty(n * 64, dtype=torch.float32, device="cuda")
    check(lib().mjx_synth_coef(d16.data_ptr(), 9, 1, first, n, q.ctypes.data_as(ctypes.c_void_p), None), "synth_coef")
    torch.cuda.synchronize()
    assert np.array_equal(d16.cpu().numpy(), O.synth_coef(9, 1, first, n, q))
    check(lib().mjx_synth_tile(df.data_ptr(), 9, 1, first, n, None), "synth_tile")
    torch.cuda.synchronize()
    assert np.array_equal(df.cpu().numpy().view(np.uint32), O.synth_tile(9, 1, first, n).view(np.uint32))
    for regime in (1, 2, 3):
        check(lib().mjx_synth_alpha(df.data_ptr(), 9, 1, first, n, regime, None), "synth_alpha")
        torch.cuda.synchronize()
        assert np.array_equal(df.cpu().numpy().view(np.uint32), O.synth_alpha(9, 1, first, n, regime).view(np.uint32))


def test_host_buffer_entry_point(mj):
    """mjx_compose_rect_host: the per-image call the libmodjpeg.h shim uses (host buffers)"""
    from libmodjpeg_amd._lib import lib, check
    wib, hib, rows, cols, row0, col0 = 30, 20, 7, 11, 5, 13
    q = O.quant_table(False)
    plane = O.synth_coef(1, 0, 0, wib * hib, q)
    tile, alpha = O.synth_tile(1, 0, 0, rows * cols), O.synth_alpha(1, 0, 0, rows * cols, 3)
    got = plane.copy()
    vp = ctypes.c_void_p
    check(lib().mjx_compose_rect_host(got.ctypes.data_as(vp), wib, hib, row0, col0, rows, cols, q.ctypes.data_as(vp),
                                      tile.ctypes.data_as(vp), alpha.ctypes.data_as(vp)), "mjx_compose_rect_host")
    want = plane.copy()
    O.compose_rect(want, wib, row0, col0, rows, cols, q, tile, alpha)
    assert np.array_equal(got, want)


def test_host_buffer_call_mixes_narrow_and_wide_rectangles(mj):
    """mjx_compose_host with three rectangles in one call: a narrow one (travels alone through pinned
    memory), a wide one and a full-width one (whole rows straight from the caller's buffer); blocks
    outside the rectangles must come back untouched"""
    from libmodjpeg_amd._lib import HostRect, check, lib
    vp = ctypes.c_void_p
    wib, hib = 40, 24
    shapes = [(3, 2, 5, 9), (1, 10, 6, 25), (15, 0, 4, 40)]  # row0, col0, rows, cols: 9*3 <= 40 < 25*3
    rects = (HostRect * 3)()
    keep, planes, want = [], [], []
    for c, (row0, col0, rows, cols) in enumerate(shapes):
        q = O.quant_table(c > 0)
        plane = O.synth_coef(7, c, 0, wib * hib, q)
        tile, alpha = O.synth_tile(7, c, 0, rows * cols), O.synth_alpha(7, c, 0, rows * cols, 2 + (c & 1))
        ref = plane.copy()
        O.compose_rect(ref, wib, row0, col0, rows, cols, q, tile, alpha)
        got = plane.copy()
        rects[c] = HostRect(got.ctypes.data + row0 * wib * 128, tile.ctypes.data, alpha.ctypes.data, q.ctypes.data, wib, col0, rows, cols, 0, 0)
        keep += [q, tile, alpha]
        planes.append(got)
        want.append(ref)
    check(lib().mjx_compose_host(rects, 3), "mjx_compose_host")
    for c in range(3):
        assert np.array_equal(planes[c], want[c]), c


@pytest.mark.parametrize("family", ["bitpattern", "denormal", "huge_q", "mixed_sparse"])
def test_fuzz_blocks_vs_oracle(mj, family):
    """large randomised differential test on adversarial inputs: raw random float bit patterns
    (NaN, Inf, denormals, huge magnitudes) for tiles and weights, full-range int16 coefficients,
    quantisers up to 65535, random sparsity patterns (each block its own) -- GPU == oracle"""
    import zlib
    seed = zlib.crc32(family.encode())  # stable across runs (str hashes change with PYTHONHASHSEED)
    rng = np.random.default_rng(seed)
    rows, cols = 96, 100  # 9600 blocks per plane, cols not a multiple of 8
    n = rows * cols
    q = rng.integers(1, 256, 64).astype(np.uint16)
    coef = rng.integers(-2048, 2048, (n, 64)).astype(np.int16)
    tile = np.round(rng.standard_normal((n, 64)) * 200).astype(np.float32)
    alpha = (rng.standard_normal((n, 64)) * 0.05).astype(np.float32)
    if family == "bitpattern":
        tile = rng.integers(0, 2**32, (n, 64), dtype=np.uint64).astype(np.uint32).view(np.float32)
        alpha = np.where(rng.random((n, 64)) < 0.3, rng.integers(0, 2**32, (n, 64), dtype=np.uint64).astype(np.uint32).view(np.float32), np.float32(0))
        coef = rng.integers(-32768, 32768, (n, 64)).astype(np.int16)
    elif family == "denormal":
        alpha = (rng.standard_normal((n, 64)) * 1e-40).astype(np.float32)  # denormal weights are not skipped
        alpha[:, 0] = 0.25
        tile = (rng.standard_normal((n, 64)) * 1e-38).astype(np.float32)
    elif family == "huge_q":
        q = rng.integers(1, 65536, 64).astype(np.uint16)
        coef = rng.integers(-32768, 32768, (n, 64)).astype(np.int16)
    else:  # every block its own sparsity pattern, signed zeros included
        keep = rng.random((n, 64)) < rng.random((n, 1)) * 0.5
        alpha = np.where(keep, alpha, np.float32(0))
        alpha[rng.random((n, 64)) < 0.05] = np.float32(-0.0)
    tile, alpha = np.ascontiguousarray(tile, np.float32), np.ascontiguousarray(alpha, np.float32)
    m = mj.Jpeg([_t(coef.reshape(rows, cols, 64))], [q], samp=[(1, 1)], colorspace=mj.JCS_GRAYSCALE)
    cd = mj.CompiledDropon([mj.Component(_t(tile), cols, rows)], [mj.Component(_t(alpha), cols, rows)])
    assert mj.mj_compose_with_mask(m, cd, 0, 0) == mj.MJ_OK
    want = coef.reshape(-1).copy()
    import os
    nthreads = min(16, os.cpu_count() or 1)
    step = rows // nthreads + 1
    jobs = []
    for r0 in range(0, rows, step):
        r1 = min(rows, r0 + step)
        jobs.append((want[r0 * cols * 64:r1 * cols * 64], cols, 0, 0, r1 - r0, cols, q, tile[r0 * cols:r1 * cols].reshape(-1), alpha[r0 * cols:r1 * cols].reshape(-1)))
    O.compose_jobs(jobs, nthreads)
    got = m.planes_numpy()[0].reshape(-1)
    assert np.array_equal(got, want), (family, "seed", seed, "differing coefficients", int((got != want).sum()))


@pytest.mark.parametrize("where", ["hipMemcpyAsync(D2H)", "HostToDevice, s):2", "compose_kernel launch"])
def test_host_buffer_call_survives_an_injected_failure(where, tmp_path):
    """mjx_compose_host bails out early when a HIP call fails (MJX_INJECT_FAILURE reports one
    successful call as failed): it must return the error with nothing of its own still queued on
    the staging arena it hands back - the very next call reuses that arena and must be bit-exact -
    and it must not have corrupted the caller's planes beyond the rectangle."""
    import subprocess
    import sys
    code = r"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.environ["MJX_ROOT"])
from libmodjpeg_amd._lib import lib, MJX_OK, MJX_ERR_LAUNCH
from tests import oracle_lib as O
L = lib()
rows, cols, bpr, prow = 24, 40, 120, 68
q = O.quant_table(False)
def run(seed):
    plane = O.synth_coef(seed, 0, 0, bpr * prow, q)
    tile, alpha = O.synth_tile(seed, 0, 0, rows * cols), O.synth_alpha(seed, 0, 0, rows * cols, 2)
    want = plane.copy()
    O.compose_rect(want, bpr, 5, 7, rows, cols, q, tile, alpha)
    before = plane.copy()
    rc = L.mjx_compose_rect_host(plane.ctypes.data, bpr, prow, 5, 7, rows, cols, q.ctypes.data, tile.ctypes.data, alpha.ctypes.data)
    return rc, plane, want, before
rc, plane, want, before = run(1)
assert rc == MJX_ERR_LAUNCH, rc
view = lambda p: p.reshape(prow, bpr, 64)
mask = np.ones((prow, bpr), bool); mask[5:5 + rows, 7:7 + cols] = False
assert np.array_equal(view(plane)[mask], view(before)[mask])      # nothing outside the rectangle was touched
for seed in (2, 3, 4):                                            # the arena is reused at once
    rc, plane, want, _ = run(seed)
    assert rc == MJX_OK and np.array_equal(plane, want), (seed, rc)
print("ok")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    faulty = os.path.join(root, "libmodjpeg_amd", "libmjx_hip_faultinj.so")  # the product library carries no injector (-DMJX_FAULT_INJECTION, __graft_entry__.build_hip)
    if not os.path.exists(faulty):
        import __graft_entry__ as g
        g.build_hip()
    env = dict(os.environ, MJX_INJECT_FAILURE=where, MJX_ROOT=root, MJX_OPTIONS="arenas=1", MJX_HIP_LIB=faulty)
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert res.returncode == 0 and res.stdout.strip().endswith("ok"), (res.stdout[-300:], res.stderr[-800:])
    assert "failure injected" in res.stderr


_EXTRA_SEEDS = [int(v) for v in os.environ.get("MJX_FUZZ_SEEDS", "").split(",") if v]  # soak runs: MJX_FUZZ_SEEDS=100,101,...


@pytest.mark.parametrize("seed", [11, 12] + _EXTRA_SEEDS)
def test_plans_grouped_by_density_vs_oracle_on_random_patterns(mj, seed):
    """mjx_compose_plan_create_ex(MJX_PLAN_GROUP_BY_DENSITY) on rectangles whose blocks have every
    density from 0 to 64 non-zero weights (signed zeros, denormals and NaN weights included), ragged
    sizes, several jobs per plan sharing quant tables: grouped plan == plain plan == oracle"""
    import torch
    from libmodjpeg_amd._lib import MJX_PLAN_ALPHA_IS_FINAL, MJX_PLAN_GROUP_BY_DENSITY, MJX_PLAN_SERIAL, MJX_PLAN_SIDE_BY_SIDE, PlanStats, RectJob, check, lib
    rng = np.random.default_rng(seed)
    shapes = [(40, 50, 3, 5, 13, 21), (20, 25, 0, 0, 20, 25), (9, 33, 2, 30, 7, 3), (64, 16, 10, 4, 37, 9), (8, 8, 1, 1, 1, 1)]  # prow, bpr, row0, col0, rows, cols
    quants = np.stack([rng.integers(1, 256, 64), rng.integers(1, 65536, 64)]).astype(np.uint16)
    host, dev, jobs = [], [], (RectJob * len(shapes))()
    for j, (prow, bpr, row0, col0, rows, cols) in enumerate(shapes):
        n = rows * cols
        coef = rng.integers(-1024, 1024, (prow, bpr, 64)).astype(np.int16)
        tile = np.round(rng.standard_normal((n, 64)) * 200).astype(np.float32)
        alpha = (rng.standard_normal((n, 64)) * 0.05).astype(np.float32)
        dens = rng.choice([0.0, 0.02, 0.1, 0.3, 0.6, 1.0], n)[:, None]
        alpha = np.where(rng.random((n, 64)) < dens, alpha, np.float32(0))
        alpha[rng.random((n, 64)) < 0.02] = np.float32(-0.0)
        alpha[rng.random((n, 64)) < 0.002] = np.float32(1e-41)
        if seed % 2 == 0:
            alpha[rng.random((n, 64)) < 0.001] = np.float32("nan")
        # blocks whose only weight is (0,0) (what an opaque or evenly translucent part of an overlay gives):
        # exactly 0.25 (opaque), ordinary, denormal, negative, infinite and NaN weights
        dc = rng.random(n) < 0.2
        alpha[dc] = 0
        alpha[dc, 0] = rng.choice(np.array([0.25, 0.1, 1e-41, -0.3, np.inf, np.nan, 3e38], np.float32), int(dc.sum()))
        alpha = np.ascontiguousarray(alpha, np.float32)
        host.append((coef, tile, alpha))
    results = []
    # the zero pattern the device scan must find: a weight counts when `w != 0` (convolve.c:32): -0.0 does not, denormals and NaN do
    terms = sum(int(np.count_nonzero(a != 0)) for _, _, a in host)
    # (the last two: the blend kernel and the streams of the blocks that need no blend side by side - the lean build of the one, the
    # long-lived stream kernel for the other, forked from and joined to the caller's stream - and strictly one after the other)
    final = MJX_PLAN_GROUP_BY_DENSITY | MJX_PLAN_ALPHA_IS_FINAL
    for flags in (final, MJX_PLAN_GROUP_BY_DENSITY, 0, final | MJX_PLAN_SIDE_BY_SIDE, final | MJX_PLAN_SERIAL):
        dev = [(_t(c), _t(t), _t(a)) for c, t, a in host]
        for j, (prow, bpr, row0, col0, rows, cols) in enumerate(shapes):
            jobs[j] = RectJob(dev[j][0].data_ptr(), dev[j][1].data_ptr(), dev[j][2].data_ptr(), bpr, prow, row0, col0, rows, cols, j % 2)
        plan = ctypes.c_void_p()
        check(lib().mjx_compose_plan_create_ex(ctypes.byref(plan), jobs, len(shapes), quants.ctypes.data_as(ctypes.c_void_p), 2, flags), "plan")
        st = PlanStats()
        check(lib().mjx_compose_plan_stats(plan, ctypes.byref(st)), "stats")
        assert st.blocks == sum(s[4] * s[5] for s in shapes)
        if flags:
            assert lib().mjx_compose_plan_gather_octets(plan) > 0
            assert st.block_terms == terms, "zero patterns of the alpha blocks"
            assert st.union_terms >= (st.block_terms + 7) // 8
            assert (st.zero_alpha_octets > 0) == bool(flags & MJX_PLAN_ALPHA_IS_FINAL)
            assert (st.dc_only_octets > 0) == bool(flags & MJX_PLAN_ALPHA_IS_FINAL)
        assert (st.runs_beside > 0) == bool(flags & MJX_PLAN_SIDE_BY_SIDE), "small plans run their kernels one after the other unless told otherwise"
        check(lib().mjx_compose_run(plan, None), "run")
        torch.cuda.synchronize()
        lib().mjx_compose_plan_destroy(plan)
        results.append([d[0].cpu().numpy() for d in dev])
    for j, (prow, bpr, row0, col0, rows, cols) in enumerate(shapes):
        want = host[j][0].reshape(-1).copy()
        O.compose_rect(want, bpr, row0, col0, rows, cols, np.ascontiguousarray(quants[j % 2]), host[j][1].reshape(-1), host[j][2].reshape(-1))
        assert np.array_equal(results[2][j].reshape(-1), want), ("plain plan", j)
        assert np.array_equal(results[1][j].reshape(-1), want), ("grouped plan", j)
        assert np.array_equal(results[0][j].reshape(-1), want), ("grouped plan, zero-alpha blocks without their tiles", j)
        assert np.array_equal(results[3][j].reshape(-1), want), ("the two kernels side by side", j)
        assert np.array_equal(results[4][j].reshape(-1), want), ("the two kernels one after the other", j)
    assert lib().mjx_compose_plan_create_ex(ctypes.byref(plan), jobs, len(shapes), quants.ctypes.data_as(ctypes.c_void_p), 2, final | MJX_PLAN_SERIAL | MJX_PLAN_SIDE_BY_SIDE) == mj._lib.MJX_ERR_BAD_ARG


@pytest.mark.parametrize("i16", [False, True])
@pytest.mark.parametrize("seed", [77, 78] + _EXTRA_SEEDS)
def test_low_column_and_low_row_runs_vs_oracle(mj, seed, i16):
    """The low-column build of the blend kernel (compose_lowl_kernel: runs none of whose weights lies in a column l >= 4) and its
    low-row form (none in a row k >= 4 either: rows 4..7 of the alpha blocks are not fetched, their lanes read row 4 of the octet's
    first block instead).  Jobs made of: smooth blocks (k + l <= 3), blocks with weights anywhere in columns 0..3, blocks with
    weights anywhere in rows 0..3 (not low), mixtures of them with empty and single-weight blocks, a smooth job with ONE block that
    has a weight in row 5 (its octet must fetch the second line, its neighbours need not), negative zeros where the schedule sees
    no weight, ragged block counts; consecutive and grouped plans under the promise that alpha is final == the same without it
    == oracle.  With int16 tiles (compose_lowl_i16_kernel: the lane that holds a row of alpha coefficients converts the weights
    of columns 0..3): the same jobs, oracle fed with the converted floats."""
    import torch
    from libmodjpeg_amd._lib import MJX_PLAN_ALPHA_IS_FINAL, MJX_PLAN_GROUP_BY_DENSITY, MJX_PLAN_SERIAL, MJX_PLAN_TILES_I16, PlanStats, RectJob, check, lib
    rng = np.random.default_rng(seed + 100 * i16)
    c0, c1 = 0.3535534, 0.5
    kscale = np.array([(c0 if j // 8 == 0 else c1) * (c0 if j % 8 == 0 else c1) / 1020.0 for j in range(64)], np.float64)
    kk8, ll8 = np.mgrid[0:8, 0:8]
    smooth, low_cols, low_rows_only = (kk8 + ll8) <= 3, ll8 < 4, kk8 < 4
    # prow, bpr, row0, col0, rows, cols, kind
    shapes = [(20, 40, 1, 2, 16, 32, "smooth"), (12, 30, 0, 3, 9, 21, "cols"), (10, 26, 2, 1, 7, 23, "rows"), (40, 40, 3, 3, 33, 29, "mix"), (16, 24, 0, 0, 16, 24, "one"),
              (6, 9, 1, 1, 3, 5, "smooth")]
    quants = np.stack([rng.integers(1, 256, 64), rng.integers(1, 4000, 64)]).astype(np.uint16)
    host = []
    for prow, bpr, row0, col0, rows, cols, kind in shapes:
        n = rows * cols
        coef = rng.integers(-1024, 1024, (prow, bpr, 64)).astype(np.int16)
        tile = np.round(rng.standard_normal((n, 64)) * 200).astype(np.float32)
        w = (rng.random((n, 8, 8)) * 0.2 + 0.001).astype(np.float32) * rng.choice(np.array([-1, 1], np.float32), (n, 8, 8))
        if kind == "smooth":
            mask = np.broadcast_to(smooth, (n, 8, 8)).copy()
        elif kind == "cols":
            mask = np.broadcast_to(low_cols, (n, 8, 8)) & (rng.random((n, 8, 8)) < 0.5)
        elif kind == "rows":
            mask = np.broadcast_to(low_rows_only, (n, 8, 8)) & (rng.random((n, 8, 8)) < 0.5)
        elif kind == "one":
            mask = np.broadcast_to(smooth, (n, 8, 8)).copy()
            mask[n // 2 + 3, 5, 0] = True
        else:
            which = rng.integers(0, 6, n)
            mask = np.zeros((n, 8, 8), bool)
            mask[which == 1, 0, 0] = True
            mask[which == 2] = smooth
            mask[which == 3] = low_cols & (rng.random((8, 8)) < 0.6)
            mask[which == 4] = low_rows_only
            mask[which == 5] = rng.random((int((which == 5).sum()), 8, 8)) < 0.7
        if i16:  # int16 coefficients; what the oracle sees: tile = (float)coef, weight = (float)((double)(float)coef * K) (dropon.c:476-485, :548-566)
            ti = rng.integers(-2000, 2000, (n, 64)).astype(np.int16)
            ai = np.where(mask.reshape(n, 64), rng.integers(1, 2040, (n, 64)) * rng.choice(np.array([-1, 1]), (n, 64)), 0).astype(np.int16)
            host.append((coef, ti, ai, ti.astype(np.float32), (ai.astype(np.float32).astype(np.float64) * kscale).astype(np.float32)))
            continue
        alpha = np.where(mask, w, np.float32(0))
        alpha[(~mask) & (rng.random((n, 8, 8)) < 0.1)] = np.float32(-0.0)  # not a weight (convolve.c:32): the schedule must not count it, the kernel may fetch it
        alpha = np.ascontiguousarray(alpha.reshape(n, 64), np.float32)
        host.append((coef, tile, alpha, tile, alpha))
    jobs = (RectJob * len(shapes))()
    results, stats = [], []
    for flags in (MJX_PLAN_ALPHA_IS_FINAL | MJX_PLAN_SERIAL, MJX_PLAN_GROUP_BY_DENSITY | MJX_PLAN_ALPHA_IS_FINAL | MJX_PLAN_SERIAL, MJX_PLAN_GROUP_BY_DENSITY, 0):
        flags |= MJX_PLAN_TILES_I16 if i16 else 0
        dev = [(_t(c), _t(t), _t(a)) for c, t, a, _, _ in host]
        for j, (prow, bpr, row0, col0, rows, cols, _) in enumerate(shapes):
            jobs[j] = RectJob(dev[j][0].data_ptr(), dev[j][1].data_ptr(), dev[j][2].data_ptr(), bpr, prow, row0, col0, rows, cols, j % 2)
        plan = ctypes.c_void_p()
        check(lib().mjx_compose_plan_create_ex(ctypes.byref(plan), jobs, len(shapes), quants.ctypes.data_as(ctypes.c_void_p), 2, flags), "plan")
        st = PlanStats()
        check(lib().mjx_compose_plan_stats(plan, ctypes.byref(st)), "stats")
        stats.append((st.low_column_octets, st.runs_low_columns))
        if flags & MJX_PLAN_GROUP_BY_DENSITY and flags & MJX_PLAN_ALPHA_IS_FINAL:
            # both smooth jobs whole (64 + 2 octets) at least; the "rows" job counts as low-row only, the "cols" job as low-column only
            assert st.low_row_octets >= 64 + 2 + 7 * 23 // 8 and st.low_column_octets >= 64 + 2 + 9 * 21 // 8
        check(lib().mjx_compose_run(plan, None), "run")
        torch.cuda.synchronize()
        lib().mjx_compose_plan_destroy(plan)
        results.append([d[0].cpu().numpy() for d in dev])
    assert stats[1][0] > 0 and stats[1][1] > 0, "the low-column build took the smooth runs (the alpha tiles are scanned for MJX_PLAN_GROUP_BY_DENSITY)"
    assert stats[0] == (0, 0) and stats[2] == (0, 0) and stats[3] == (0, 0), "without the scan, or without the promise that alpha is final, nothing is relied on"
    for j, (prow, bpr, row0, col0, rows, cols, kind) in enumerate(shapes):
        want = host[j][0].reshape(-1).copy()
        O.compose_rect(want, bpr, row0, col0, rows, cols, np.ascontiguousarray(quants[j % 2]), host[j][3].reshape(-1), host[j][4].reshape(-1))
        for which, name in enumerate(("plain, alpha final", "grouped, alpha final", "grouped", "plain")):
            assert np.array_equal(results[which][j].reshape(-1), want), (name, j, kind)


@pytest.mark.parametrize("seed", [2027] + _EXTRA_SEEDS)
def test_int16_tile_plans_vs_oracle_on_full_range_values(mj, seed):
    """MJX_JOB_TILES_I16 on arbitrary int16 contents (tile and alpha coefficients over the whole
    range, every density, ragged rectangles, grouped by density): the kernel's conversions
    tile = (float)coef and weight = (float)((double)(float)coef * K) equal the host's, so the planes
    equal the oracle's fed with the converted float tiles"""
    import torch
    from libmodjpeg_amd._lib import MJX_JOB_TILES_I16, MJX_PLAN_ALPHA_IS_FINAL, MJX_PLAN_GROUP_BY_DENSITY, MJX_PLAN_SIDE_BY_SIDE, RectJob, check, lib
    rng = np.random.default_rng(seed)
    c0, c1 = 0.3535534, 0.5
    kk = np.array([(c0 if j // 8 == 0 else c1) * (c0 if j % 8 == 0 else c1) / 1020.0 for j in range(64)], np.float64)
    shapes = [(40, 50, 3, 5, 13, 21), (9, 33, 2, 30, 7, 3), (8, 8, 1, 1, 1, 1), (30, 30, 0, 0, 30, 30)]
    quants = np.stack([rng.integers(1, 256, 64), rng.integers(1, 2000, 64)]).astype(np.uint16)
    host, jobs = [], (RectJob * len(shapes))()
    for prow, bpr, row0, col0, rows, cols in shapes:
        n = rows * cols
        coef = rng.integers(-1024, 1024, (prow, bpr, 64)).astype(np.int16)
        tile = rng.integers(-32768, 32768, (n, 64)).astype(np.int16)
        alpha = rng.integers(-32768, 32768, (n, 64)).astype(np.int16)
        alpha = np.where(rng.random((n, 64)) < rng.choice([0.0, 0.05, 0.3, 1.0], n)[:, None], alpha, np.int16(0)).astype(np.int16)
        dc = rng.random(n) < 0.25  # blocks whose only weight is (0,0): the whole int16 range, 2040 (opaque) among them
        alpha[dc] = 0
        alpha[dc, 0] = rng.choice(np.array([2040, 1, -1, 32767, -32768, 777], np.int16), int(dc.sum()))
        host.append((coef, np.ascontiguousarray(tile), np.ascontiguousarray(alpha)))
    for flags in (0, MJX_PLAN_GROUP_BY_DENSITY, MJX_PLAN_GROUP_BY_DENSITY | MJX_PLAN_ALPHA_IS_FINAL, MJX_PLAN_GROUP_BY_DENSITY | MJX_PLAN_ALPHA_IS_FINAL | MJX_PLAN_SIDE_BY_SIDE):
        dev = [(_t(c), _t(t), _t(a)) for c, t, a in host]
        for j, (prow, bpr, row0, col0, rows, cols) in enumerate(shapes):
            jobs[j] = RectJob(dev[j][0].data_ptr(), dev[j][1].data_ptr(), dev[j][2].data_ptr(), bpr, prow, row0, col0, rows, cols, j % 2, MJX_JOB_TILES_I16)
        plan = ctypes.c_void_p()
        check(lib().mjx_compose_plan_create_ex(ctypes.byref(plan), jobs, len(shapes), quants.ctypes.data_as(ctypes.c_void_p), 2, flags), "plan")
        check(lib().mjx_compose_run(plan, None), "run")
        torch.cuda.synchronize()
        lib().mjx_compose_plan_destroy(plan)
        for j, (prow, bpr, row0, col0, rows, cols) in enumerate(shapes):
            want = host[j][0].reshape(-1).copy()
            tf = host[j][1].astype(np.float32).reshape(-1)
            af = (host[j][2].astype(np.float32).astype(np.float64) * kk).astype(np.float32).reshape(-1)
            O.compose_rect(want, bpr, row0, col0, rows, cols, np.ascontiguousarray(quants[j % 2]), tf, af)
            assert np.array_equal(dev[j][0].cpu().numpy().reshape(-1), want), (flags, j)
    # a plan must not mix the two tile formats
    jobs[0].flags = 0
    plan = ctypes.c_void_p()
    assert lib().mjx_compose_plan_create_ex(ctypes.byref(plan), jobs, len(shapes), quants.ctypes.data_as(ctypes.c_void_p), 2, 0) == mj._lib.MJX_ERR_BAD_ARG


@pytest.mark.parametrize("i16", [False, True])
def test_tile_schedules_travel_with_resident_tiles_through_the_host_buffer_call(mj, i16):
    """What the drop-in's tile cache does (reference compose.c:155-177: one overlay, many images):
    tiles stay on the device, mjx_tile_schedule_create reads the zero patterns of their alpha blocks
    once, and every mjx_compose_host call hands tiles + schedule over.  Masks with structure (rows of
    weights, columns of weights, dense, a single weight, none at all), three rectangles per call (narrow,
    wide, ragged), two different images per schedule, quantisers that make coef * q wrap in the
    blocks without weights: == oracle, and == the same call without the schedules."""
    from libmodjpeg_amd._lib import (MJX_JOB_TILES_I16, MJX_RECT_TILES_I16, MJX_RECT_TILES_ON_DEVICE, HostRect, PlanStats, check, lib)
    vp = ctypes.c_void_p
    rng = np.random.default_rng(5 + i16)
    c0, c1 = 0.3535534, 0.5
    kk = np.array([(c0 if j // 8 == 0 else c1) * (c0 if j % 8 == 0 else c1) / 1020.0 for j in range(64)], np.float64)
    wib, hib = 60, 40
    shapes = [(3, 2, 19, 17), (1, 10, 30, 45), (33, 7, 3, 5)]  # row0, col0, rows, cols
    quants = [rng.integers(1, 256, 64).astype(np.uint16), rng.integers(1, 65536, 64).astype(np.uint16), rng.integers(1, 40, 64).astype(np.uint16)]
    tiles, scheds, dev = [], [], []
    for (row0, col0, rows, cols) in shapes:
        n = rows * cols
        kind = rng.integers(0, 6, n)  # 0 none, 1 DC only, 2 one column, 3 one row, 4 low-frequency triangle, 5 dense
        mask = np.zeros((n, 8, 8), bool)
        mask[kind == 1, 0, 0] = True
        mask[kind == 2, :, 0] = True
        mask[kind == 3, 0, :] = True
        kk8, ll8 = np.mgrid[0:8, 0:8]
        mask[kind == 4] = (kk8 + ll8) <= 3
        mask[kind == 5] = True
        mask = mask.reshape(n, 64)
        if i16:
            tile = rng.integers(-2000, 2000, (n, 64)).astype(np.int16)
            alpha = np.where(mask, rng.integers(1, 2040, (n, 64)), 0).astype(np.int16)
            # opaque blocks (the single weight 2040 -> 0.25): written from the tile alone, whatever it holds (mjx.h, MJX_PLAN_TILES_ARE_FINAL)
            opq = (kind == 1) & (rng.random(n) < 0.6)
            alpha[opq, 0] = 2040
            tile[opq] = rng.integers(-32768, 32768, (int(opq.sum()), 64)).astype(np.int16)
            tf = tile.astype(np.float32)
            af = (alpha.astype(np.float32).astype(np.float64) * kk).astype(np.float32)
        else:
            tile = np.round(rng.standard_normal((n, 64)) * 200).astype(np.float32)
            alpha = np.where(mask, (rng.random((n, 64)) * 0.2 + 0.001), 0).astype(np.float32)
            tf, af = tile, alpha
        d_t, d_a = _t(tile), _t(alpha)
        sc = vp()
        check(lib().mjx_tile_schedule_create(ctypes.byref(sc), vp(d_a.data_ptr()), rows, cols, MJX_JOB_TILES_I16 if i16 else 0, mj._lib.caller_stream()), "mjx_tile_schedule_create")
        st = PlanStats()
        check(lib().mjx_tile_schedule_stats(sc, ctypes.byref(st)), "mjx_tile_schedule_stats")
        assert st.blocks == n and st.block_terms == int(mask.sum())
        assert st.opaque_octets == ((int(opq.sum()) + 7) // 8 if i16 and opq.sum() >= 8 else 0), "float tiles: a schedule has not seen the tiles"
        if n >= 64:
            assert lib().mjx_tile_schedule_gather_octets(sc) > 0 and st.zero_alpha_octets > 0
            assert st.union_terms < 2 * st.block_terms / 8 + 64 * 8, "octets of like blocks: the unions stay close to the blocks' own patterns"
        tiles.append((tf.reshape(-1), af.reshape(-1)))
        scheds.append(sc)
        dev.append((d_t, d_a))
    flag = MJX_RECT_TILES_ON_DEVICE | (MJX_RECT_TILES_I16 if i16 else 0)
    for image in (1, 2):
        for use_sched in (True, False):
            rects = (HostRect * 3)()
            planes, want = [], []
            for c, (row0, col0, rows, cols) in enumerate(shapes):
                plane = rng.integers(-1024, 1024, wib * hib * 64).astype(np.int16)
                ref = plane.copy()
                O.compose_rect(ref, wib, row0, col0, rows, cols, quants[c], tiles[c][0], tiles[c][1])
                rects[c] = HostRect(plane.ctypes.data + row0 * wib * 128, dev[c][0].data_ptr(), dev[c][1].data_ptr(), quants[c].ctypes.data, wib, col0, rows, cols,
                                    flag, 0, scheds[c] if use_sched else None)
                planes.append(plane)
                want.append(ref)
            check(lib().mjx_compose_host(rects, 3), "mjx_compose_host")
            for c in range(3):
                assert np.array_equal(planes[c], want[c]), (image, use_sched, c)
    # a schedule belongs to the tiles it was made from
    bad = (HostRect * 1)()
    bad[0] = HostRect(planes[0].ctypes.data, dev[0][0].data_ptr(), dev[0][1].data_ptr(), quants[0].ctypes.data, wib, 0, shapes[0][2] + 1, shapes[0][3], flag, 0, scheds[0])
    assert lib().mjx_compose_host(bad, 1) == mj._lib.MJX_ERR_BAD_ARG
    for sc in scheds:
        lib().mjx_tile_schedule_destroy(sc)


@pytest.mark.parametrize("i16", [False, True])
def test_rectangles_made_of_blocks_that_need_no_blend(mj, i16):
    """compose_light_kernel on its own: rectangles whose alpha blocks are ALL without weights, ALL with the single
    weight (0,0), or an exact mixture at the 8-block threshold, ragged block counts (35, 33, 1 x 19), a rectangle
    one block wide, quantisers that make coef * q wrap; float and int16 tiles; plan with the promise that alpha is
    final == oracle == plan without it"""
    import torch
    from libmodjpeg_amd._lib import (MJX_JOB_TILES_I16, MJX_PLAN_ALPHA_IS_FINAL, MJX_PLAN_GROUP_BY_DENSITY, PlanStats, RectJob, check, lib)
    rng = np.random.default_rng(31 + i16)
    c0, c1 = 0.3535534, 0.5
    kk = np.array([(c0 if j // 8 == 0 else c1) * (c0 if j % 8 == 0 else c1) / 1020.0 for j in range(64)], np.float64)
    # prow, bpr, row0, col0, rows, cols, kind of alpha
    shapes = [(12, 20, 2, 3, 5, 7, "zero"), (9, 16, 1, 2, 3, 11, "dc"), (30, 4, 5, 2, 19, 1, "dc"), (8, 30, 0, 0, 4, 6, "mix"), (6, 6, 1, 1, 1, 1, "zero"), (10, 12, 0, 1, 2, 9, "zero")]
    quants = np.stack([rng.integers(1, 256, 64), rng.integers(200, 65536, 64)]).astype(np.uint16)
    host = []
    for prow, bpr, row0, col0, rows, cols, kind in shapes:
        n = rows * cols
        coef = rng.integers(-1024, 1024, (prow, bpr, 64)).astype(np.int16)
        if i16:
            tile = rng.integers(-3000, 3000, (n, 64)).astype(np.int16)
            alpha = np.zeros((n, 64), np.int16)
            dcv = rng.choice(np.array([2040, 1020, 1, -7, 32767], np.int16), n)
            dense = rng.integers(1, 2040, (n, 64)).astype(np.int16)
        else:
            tile = np.round(rng.standard_normal((n, 64)) * 300).astype(np.float32)
            alpha = np.zeros((n, 64), np.float32)
            dcv = rng.choice(np.array([0.25, 0.125, 1e-41, -0.4, 7.0], np.float32), n)
            dense = (rng.random((n, 64)) * 0.1 + 0.01).astype(np.float32)
        if kind == "dc":
            alpha[:, 0] = dcv
        elif kind == "mix":  # 8 without weights, 8 with the single weight, 8 dense
            alpha[8:16, 0] = dcv[8:16]
            alpha[16:24] = dense[16:24]
        host.append((coef, np.ascontiguousarray(tile), np.ascontiguousarray(alpha)))
    jobs = (RectJob * len(shapes))()
    results = []
    for flags in (MJX_PLAN_GROUP_BY_DENSITY | MJX_PLAN_ALPHA_IS_FINAL, 0):
        dev = [(_t(c), _t(t), _t(a)) for c, t, a in host]
        for j, (prow, bpr, row0, col0, rows, cols, kind) in enumerate(shapes):
            jobs[j] = RectJob(dev[j][0].data_ptr(), dev[j][1].data_ptr(), dev[j][2].data_ptr(), bpr, prow, row0, col0, rows, cols, j % 2, MJX_JOB_TILES_I16 if i16 else 0)
        plan = ctypes.c_void_p()
        check(lib().mjx_compose_plan_create_ex(ctypes.byref(plan), jobs, len(shapes), quants.ctypes.data_as(ctypes.c_void_p), 2, flags), "plan")
        if flags:
            st = PlanStats()
            check(lib().mjx_compose_plan_stats(plan, ctypes.byref(st)), "stats")
            # the 35- and 18-block all-zero rectangles: 5 + 3 octets; the mix: 1; the 33- and 19-block DC-only ones: 5 + 3; the mix: 1;
            # the single all-zero block stays below the 8-block threshold and goes the general way.  int16 tiles: the blocks of a
            # rectangle whose single weight is 2040 (-> 0.25) are opaque blocks and leave for a stream of their own when there are eight
            want_dc = want_opq = 0
            for (_, _, alpha), shape in zip(host, shapes):
                single = (np.count_nonzero(alpha, axis=1) == 1) & (alpha[:, 0] != 0)
                nopq = int((single & (alpha[:, 0] == 2040)).sum()) if i16 else 0
                nopq = nopq if nopq >= 8 else 0
                ndc = int(single.sum()) - nopq
                want_opq += (nopq + 7) // 8
                want_dc += (ndc + 7) // 8 if ndc >= 8 else 0
            assert want_dc + want_opq >= 5 + 3 + 1
            assert st.zero_alpha_octets == 5 + 3 + 1 and st.dc_only_octets == want_dc and st.opaque_octets == want_opq, (st.zero_alpha_octets, st.dc_only_octets, st.opaque_octets)
        check(lib().mjx_compose_run(plan, None), "run")
        check(lib().mjx_compose_run(plan, None), "run")  # twice: the second pass starts from composed planes
        torch.cuda.synchronize()
        lib().mjx_compose_plan_destroy(plan)
        results.append([d[0].cpu().numpy() for d in dev])
    for j, (prow, bpr, row0, col0, rows, cols, kind) in enumerate(shapes):
        want = host[j][0].reshape(-1).copy()
        if i16:
            tf = host[j][1].astype(np.float32).reshape(-1)
            af = (host[j][2].astype(np.float32).astype(np.float64) * kk).astype(np.float32).reshape(-1)
        else:
            tf, af = host[j][1].reshape(-1), host[j][2].reshape(-1)
        for _ in range(2):
            O.compose_rect(want, bpr, row0, col0, rows, cols, np.ascontiguousarray(quants[j % 2]), tf, af)
        assert np.array_equal(results[1][j].reshape(-1), want), ("plain plan", j)
        assert np.array_equal(results[0][j].reshape(-1), want), ("lean kernel", j, kind)


@pytest.mark.parametrize("i16", [False, True])
@pytest.mark.parametrize("seed", [404] + _EXTRA_SEEDS)
def test_opaque_blocks_are_written_from_their_tiles_alone(mj, seed, i16):
    """MJX_PLAN_TILES_ARE_FINAL (float tiles) / int16 tiles under MJX_PLAN_ALPHA_IS_FINAL: a block whose only weight is (0,0) = 0.25
    and whose tile holds whole numbers below 2^23 is written as (short)((short)(int)tile / q) without a read of the image
    (compose.c:277-336 is exact for it).  Here: such blocks with tile values up to +-(2^23 - 1), -0.0, and images that hold
    +-32767 under quantisers that make coef * q wrap; and every near miss - one tile value with a fraction, +-2^23, NaN, an infinity;
    the weight one ulp beside 0.25; a second weight elsewhere in the block - which must stay with the single-weight blocks.
    Plans with and without the promises, kernels side by side and one after the other: all == oracle."""
    import torch
    from libmodjpeg_amd._lib import (MJX_JOB_TILES_I16, MJX_PLAN_ALPHA_IS_FINAL, MJX_PLAN_GROUP_BY_DENSITY, MJX_PLAN_SERIAL, MJX_PLAN_SIDE_BY_SIDE, MJX_PLAN_TILES_ARE_FINAL, PlanStats,
                                     RectJob, check, lib)
    rng = np.random.default_rng(seed + i16)
    c0, c1 = 0.3535534, 0.5
    kk = np.array([(c0 if j // 8 == 0 else c1) * (c0 if j % 8 == 0 else c1) / 1020.0 for j in range(64)], np.float64)
    shapes = [(30, 40, 2, 3, 21, 33), (12, 9, 0, 0, 12, 9), (5, 70, 1, 4, 3, 61), (6, 6, 2, 2, 2, 3)]  # prow, bpr, row0, col0, rows, cols
    quants = np.stack([rng.integers(1, 256, 64), rng.integers(1, 65536, 64), np.ones(64)]).astype(np.uint16)
    host, want_opaque = [], 0
    for prow, bpr, row0, col0, rows, cols in shapes:
        n = rows * cols
        coef = rng.integers(-1024, 1024, (prow, bpr, 64)).astype(np.int16)
        coef[rng.random((prow, bpr, 64)) < 0.01] = 32767
        coef[rng.random((prow, bpr, 64)) < 0.01] = -32768
        kind = rng.choice(np.arange(8), n, p=[0.4, 0.1, 0.1, 0.05, 0.05, 0.1, 0.1, 0.1])  # 0 opaque, 1-4 near misses, 5 no weight, 6 single weight, 7 dense
        if i16:
            tile = rng.integers(-32768, 32768, (n, 64)).astype(np.int16)
            alpha = np.zeros((n, 64), np.int16)
            alpha[kind <= 4, 0] = 2040
            alpha[kind == 1, 0] = 2039
            alpha[kind == 2, 0] = 2041
            alpha[kind == 3, 63] = 1
            alpha[kind == 4, rng.integers(1, 64, int((kind == 4).sum()))] = -1
            alpha[kind == 6, 0] = rng.integers(1, 2040, int((kind == 6).sum()))
            alpha[kind == 7] = rng.integers(1, 2040, (int((kind == 7).sum()), 64))
            exact = kind == 0
        else:
            big = rng.random(n) < 0.3
            tile = np.where(big[:, None], rng.integers(-(2**23) + 1, 2**23, (n, 64)), rng.integers(-3000, 3000, (n, 64))).astype(np.float32)
            tile[rng.random((n, 64)) < 0.01] = np.float32(-0.0)
            alpha = np.zeros((n, 64), np.float32)
            alpha[kind <= 4, 0] = 0.25
            at = rng.integers(0, 64, n)
            miss = np.array([0.5, 8388608.0, -8388608.0, np.nan, np.inf, -np.inf, 1e-3, 8388607.5], np.float32)
            for k in np.flatnonzero(kind == 1):  # one value of the tile is not a whole number below 2^23
                tile[k, at[k]] = miss[rng.integers(0, len(miss))]
            alpha[kind == 2, 0] = np.nextafter(np.float32(0.25), np.float32(rng.choice([0.0, 1.0])))
            # (the NaN whose bit pattern is the scan's tag for an opaque block: a weight like any other NaN)
            alpha[kind == 2, 0] = np.where(rng.random(int((kind == 2).sum())) < 0.5, np.array([0xFFFFFFFF], np.uint32).view(np.float32)[0], alpha[kind == 2, 0])
            alpha[kind == 3, 63] = 1e-41
            alpha[kind == 4, rng.integers(1, 64, int((kind == 4).sum()))] = np.float32("nan")
            alpha[kind == 6, 0] = (rng.random(int((kind == 6).sum())) * 0.3 + 0.01).astype(np.float32)
            alpha[kind == 7] = (rng.standard_normal((int((kind == 7).sum()), 64)) * 0.05).astype(np.float32)
            exact = (kind == 0) | ((kind == 1) & np.array([miss_ok(tile[k]) for k in range(n)]))
        ne = int(exact.sum())
        want_opaque += (ne + 7) // 8 if ne >= 8 else 0
        host.append((coef, np.ascontiguousarray(tile), np.ascontiguousarray(alpha)))
    jobs = (RectJob * len(shapes))()
    final = MJX_PLAN_GROUP_BY_DENSITY | MJX_PLAN_ALPHA_IS_FINAL
    results, variants = [], (final | MJX_PLAN_TILES_ARE_FINAL, final, 0, final | MJX_PLAN_TILES_ARE_FINAL | MJX_PLAN_SIDE_BY_SIDE, final | MJX_PLAN_TILES_ARE_FINAL | MJX_PLAN_SERIAL,
                         MJX_PLAN_GROUP_BY_DENSITY | MJX_PLAN_TILES_ARE_FINAL)
    for flags in variants:
        dev = [(_t(c), _t(t), _t(a)) for c, t, a in host]
        for j, (prow, bpr, row0, col0, rows, cols) in enumerate(shapes):
            jobs[j] = RectJob(dev[j][0].data_ptr(), dev[j][1].data_ptr(), dev[j][2].data_ptr(), bpr, prow, row0, col0, rows, cols, j % 3, MJX_JOB_TILES_I16 if i16 else 0)
        plan = ctypes.c_void_p()
        check(lib().mjx_compose_plan_create_ex(ctypes.byref(plan), jobs, len(shapes), quants.ctypes.data_as(ctypes.c_void_p), 3, flags), "plan")
        st = PlanStats()
        check(lib().mjx_compose_plan_stats(plan, ctypes.byref(st)), "stats")
        promised = bool(flags & MJX_PLAN_ALPHA_IS_FINAL) and (i16 or bool(flags & MJX_PLAN_TILES_ARE_FINAL))
        assert st.opaque_octets == (want_opaque if promised else 0), (hex(flags), st.opaque_octets, want_opaque)
        check(lib().mjx_compose_run(plan, None), "run")
        check(lib().mjx_compose_run(plan, None), "run")  # twice: an opaque block comes out the same, the others start from composed planes
        torch.cuda.synchronize()
        lib().mjx_compose_plan_destroy(plan)
        results.append([d[0].cpu().numpy() for d in dev])
    for j, (prow, bpr, row0, col0, rows, cols) in enumerate(shapes):
        want = host[j][0].reshape(-1).copy()
        if i16:
            tf = host[j][1].astype(np.float32).reshape(-1)
            af = (host[j][2].astype(np.float32).astype(np.float64) * kk).astype(np.float32).reshape(-1)
        else:
            tf, af = host[j][1].reshape(-1), host[j][2].reshape(-1)
        for _ in range(2):
            O.compose_rect(want, bpr, row0, col0, rows, cols, np.ascontiguousarray(quants[j % 3]), tf, af)
        for v, flags in enumerate(variants):
            assert np.array_equal(results[v][j].reshape(-1), want), (hex(flags), j)


def miss_ok(row):
    """all 64 values whole numbers below 2^23 (what the scan of MJX_PLAN_TILES_ARE_FINAL asks of a tile block)"""
    with np.errstate(invalid="ignore"):
        return bool(np.all(np.isfinite(row)) and np.all(row == np.rint(row)) and np.all(np.abs(row) < 8388608.0))


def test_host_paths_in_place_and_staged_give_the_same_bytes(mj):
    """mjx_compose_host / mjx_effect_dc_host work IN PLACE on pinned memory by default (spans up to 1 MB) and through staged
    copies with MJX_OPTIONS=compose-copies,effect-copies or above 1 MB: both forms == oracle and == each other
    (tests/tools/host_paths_check.py, one process per setting because the library reads the switches once)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = []
    for copies in (False, True):
        env = {k: v for k, v in os.environ.items() if k != "MJX_OPTIONS"}
        if copies:
            env.update(MJX_OPTIONS="compose-copies,effect-copies")
        out = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "host_paths_check.py")], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout + out.stderr
        last = out.stdout.strip().splitlines()[-1].split()
        assert last[:3] == ["host", "paths", "ok"] and last[3] == ("compose-copies,effect-copies" if copies else "-")
        digests.append(last[-1])
    assert digests[0] == digests[1]


def test_grouped_plan_without_the_promise_follows_later_changes_of_alpha(mj):
    """MJX_PLAN_GROUP_BY_DENSITY alone is "purely a schedule: the results are identical, whatever alpha holds later" (mjx.h): the
    alpha tiles are scanned when the plan is made - here every weight lies in columns 0..3 and half of the blocks have none or
    only (0,0) - then REPLACED by dense weights.  Nothing the scan saw may have been baked in: no low-column build of the
    blend kernel, no zero-alpha or DC-only streams (those need MJX_PLAN_ALPHA_IS_FINAL)."""
    import torch
    from libmodjpeg_amd._lib import MJX_PLAN_GROUP_BY_DENSITY, PlanStats, RectJob, check, lib
    rng = np.random.default_rng(77)
    prow, bpr, row0, col0, rows, cols = 40, 64, 2, 7, 24, 40
    n = rows * cols
    q = rng.integers(1, 200, 64).astype(np.uint16)
    coef = rng.integers(-1024, 1024, (prow, bpr, 64)).astype(np.int16)
    tile = np.round(rng.standard_normal((n, 64)) * 200).astype(np.float32)
    smooth = np.zeros((n, 8, 8), np.float32)
    smooth[:, :, :4] = (rng.standard_normal((n, 8, 4)) * 0.05).astype(np.float32)
    kind = rng.integers(0, 3, n)
    smooth[kind == 0] = 0
    smooth[kind == 1] = 0
    smooth[kind == 1, 0, 0] = 0.25
    dense = (rng.standard_normal((n, 64)) * 0.05).astype(np.float32)
    d_c, d_t, d_a = _t(coef), _t(tile), _t(smooth.reshape(n, 64))
    jobs = (RectJob * 1)()
    jobs[0] = RectJob(d_c.data_ptr(), d_t.data_ptr(), d_a.data_ptr(), bpr, prow, row0, col0, rows, cols, 0)
    plan = ctypes.c_void_p()
    check(lib().mjx_compose_plan_create_ex(ctypes.byref(plan), jobs, 1, q.ctypes.data_as(ctypes.c_void_p), 1, MJX_PLAN_GROUP_BY_DENSITY), "plan")
    st = PlanStats()
    check(lib().mjx_compose_plan_stats(plan, ctypes.byref(st)), "stats")
    assert st.zero_alpha_octets == 0 and st.dc_only_octets == 0 and st.runs_low_columns == 0 and st.low_column_octets == 0
    d_a.copy_(torch.from_numpy(dense))  # same stream as the run below
    check(lib().mjx_compose_run(plan, None), "run")
    torch.cuda.synchronize()
    lib().mjx_compose_plan_destroy(plan)
    want = coef.reshape(-1).copy()
    O.compose_rect(want, bpr, row0, col0, rows, cols, q, tile.reshape(-1), dense.reshape(-1))
    assert np.array_equal(d_c.cpu().numpy().reshape(-1), want)


def test_the_kernel_library_reads_one_environment_variable():
    """mjx::option(): MJX_OPTIONS is parsed once; an unknown token is reported once on stderr and ignored, known tokens beside
    it count (arenas=1 here; decode-rounds / compose-copies / effect-copies are forced by the tests that compare their forms)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, numpy as np, ctypes; sys.path.insert(0, %r); from libmodjpeg_amd._lib import lib; vp = ctypes.c_void_p\n"
            "p = np.zeros(4 * 4 * 64, np.int16); q = np.ones(64, np.uint16); t = np.zeros(64, np.float32)\n"
            "for _ in range(3): assert lib().mjx_compose_rect_host(p.ctypes.data_as(vp), 4, 4, 1, 1, 1, 1, q.ctypes.data_as(vp), t.ctypes.data_as(vp), t.ctypes.data_as(vp)) == 0\n"
            "print('ok')\n") % root
    res = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MJX_OPTIONS="bogus=3,arenas=1,,wait=block"), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and res.stdout.strip().endswith("ok"), res.stdout + res.stderr
    assert res.stderr.count('MJX_OPTIONS: unknown option "bogus" ignored') == 1, res.stderr[-800:]
