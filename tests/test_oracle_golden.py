"""CPU tests (no GPU): pin the oracle restatement against vectors dumped from the unmodified
reference (tests/golden/*.npz, generator tests/golden/make_golden.py + oracle/ref_dump.c) and,
when oracle/_ref/libmodjpeg_ref.so exists (build container), against the reference live."""
import ctypes
import os

import numpy as np
import pytest

from tests import oracle_lib as O


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_convolve_kat(golden_dir):
    g = _load(golden_dir, "convolve_kat.npz")
    n = g["x"].shape[0]
    assert n == 768
    for i in range(n):
        y = g["yin"][i].copy()
        O.convolve(np.ascontiguousarray(g["x"][i]), y, g["w"][i], g["kl"][i, 0], g["kl"][i, 1])
        assert np.array_equal(y.view(np.uint32), g["yout"][i].view(np.uint32)), (i, g["kl"][i])


def test_block_kat(golden_dir):
    """every fabricated family (smooth, dense, exact blends, int16 wrap, near-integer, sparse,
    NaN/Inf/out-of-range) through orc_compose_block == reference mj_compose_with_mask"""
    g = _load(golden_dir, "block_kat.npz")
    nfam = g["q"].shape[0]
    per_fam = g["coef_in"].shape[0] // nfam
    bad = 0
    for b in range(g["coef_in"].shape[0]):
        fam, r = divmod(b, per_fam)
        geom = g["geom"][fam]
        n0 = geom[0, 2] * geom[0, 3]
        n1 = geom[1, 2] * geom[1, 3]
        c = 0 if r < n0 else (1 if r < n0 + n1 else 2)
        coef = g["coef_in"][b].copy()
        O.compose_block(coef, np.ascontiguousarray(g["q"][fam, c]), np.ascontiguousarray(g["tile"][b]), np.ascontiguousarray(g["alpha"][b]))
        if not np.array_equal(coef, g["coef_out"][b]):
            bad += 1
    assert bad == 0


@pytest.mark.parametrize("case", ["tl", "br", "crop"])
def test_compose_rect_kat(golden_dir, case):
    """whole rectangles with tiles prepared by the reference's mj_compile_dropon; `tl` is the
    reference repository's own image.jpg + dropon.png -> image_dropon.jpg example"""
    g = _load(golden_dir, "compose_kat.npz")
    for c in range(3):
        wib, hib = int(g["meta"][4 + 4 * c]), int(g["meta"][5 + 4 * c])
        row0, col0, rows, cols = [int(v) for v in g[f"{case}_rect_c{c}"][:4]]
        plane = g[f"before_c{c}"].copy()
        O.compose_rect(plane, wib, row0, col0, rows, cols, g[f"q_c{c}"], np.ascontiguousarray(g[f"{case}_tile_c{c}"]).ravel(),
                       np.ascontiguousarray(g[f"{case}_alpha_c{c}"]).ravel())
        got = plane.reshape(hib, wib, 64)
        assert np.array_equal(got[row0:row0 + rows, col0:col0 + cols], g[f"{case}_after_rect_c{c}"])
        mask = np.ones((hib, wib), bool)
        mask[row0:row0 + rows, col0:col0 + cols] = False
        assert np.array_equal(got[mask], g[f"before_c{c}"].reshape(hib, wib, 64)[mask])


def _apply_lum_tint(g, planes, prefix, lum, cb, cr):
    for c, val in ((0, lum), (1, cb), (2, cr)):
        if val is None:
            continue
        if c > 0 and val == 0:
            continue  # effect.c:134/:158 skip a zero channel; luminance(0) still runs (effect.c:185)
        meta = g[prefix + "meta"]
        wib, hib = int(meta[4 + 4 * c]), int(meta[5 + 4 * c])
        O.effect_dc(planes[c], wib, hib, wib, int(g[f"{prefix}q_c{c}"][0]), val)


def test_effect_kat(golden_dir):
    g = _load(golden_dir, "effect_kat.npz")
    planes = [g[f"before_c{c}"].copy() for c in range(3)]
    _apply_lum_tint(g, planes, "", 100, None, None)
    _apply_lum_tint(g, planes, "", None, 50, 0)
    for c in range(3):
        assert np.array_equal(planes[c], g[f"lum100_tint50_0_c{c}"]), c
    _apply_lum_tint(g, planes, "", None, -300, 77)
    _apply_lum_tint(g, planes, "", -1900, None, None)
    for c in range(3):
        assert np.array_equal(planes[c], g[f"clamp_c{c}"]), c
    _apply_lum_tint(g, planes, "", 0, None, None)
    for c in range(3):
        assert np.array_equal(planes[c], g[f"zero_c{c}"]), c
    for c in range(3):
        wib, hib = int(g["meta"][4 + 4 * c]), int(g["meta"][5 + 4 * c])
        O.effect_zero(planes[c], wib, hib, wib, 1)
        assert np.array_equal(planes[c], g[f"pixelate_c{c}"]), c
    for c in (1, 2):
        wib, hib = int(g["meta"][4 + 4 * c]), int(g["meta"][5 + 4 * c])
        O.effect_zero(planes[c], wib, hib, wib, 0)
    for c in range(3):
        assert np.array_equal(planes[c], g[f"grayscale_c{c}"]), c


def test_effect_wrap_kat(golden_dir):
    """int16 wrap of `short *= q`, `short += value` with huge values (value is an int in the API)"""
    g = _load(golden_dir, "effect_kat.npz")
    planes = [g[f"w_before_c{c}"].copy() for c in range(3)]
    _apply_lum_tint(g, planes, "w_", 40000, None, None)
    _apply_lum_tint(g, planes, "w_", None, -70000, 123456789)
    for c in range(3):
        assert np.array_equal(planes[c], g[f"w_after_c{c}"]), c


@pytest.mark.skipif(O.ref_lib() is None, reason="oracle/_ref not built (no /root/reference here)")
def test_convolve_live_vs_reference():
    """fresh random inputs: restatement vs the compiled reference's own mj_convolve, all (k,l)"""
    R = O.ref_lib()
    fp = ctypes.POINTER(ctypes.c_float)
    rng = np.random.default_rng(20240607)
    for trial in range(60):
        scale = 10.0 ** rng.integers(-3, 6)
        x = (rng.standard_normal(64) * scale).astype(np.float32)
        if trial % 2:
            x = np.round(x)
        y0 = (rng.standard_normal(64) * scale).astype(np.float32)
        for k in range(8):
            for l in range(8):
                w = np.float32(rng.standard_normal() * 0.1)
                ya, yb = y0.copy(), y0.copy()
                R.mj_convolve(x.ctypes.data_as(fp), ya.ctypes.data_as(fp), w, k, l)
                O.convolve(x, yb, w, k, l)
                assert np.array_equal(ya.view(np.uint32), yb.view(np.uint32)), (trial, k, l)


def test_synth_is_deterministic_and_in_range():
    q = O.quant_table(False)
    a = O.synth_coef(3, 0, 1000, 64, q)
    b = O.synth_coef(3, 0, 1000, 64, q)
    assert np.array_equal(a, b)
    deq = a.reshape(-1, 64).astype(np.int32) * q.astype(np.int32)
    assert np.abs(deq).max() <= 1000
    t = O.synth_tile(3, 0, 0, 64).reshape(-1, 64)
    assert np.array_equal(t, np.round(t)) and np.abs(t[:, 0]).max() <= 1016 and np.abs(t[:, 1:]).max() <= 255
    for regime, nnz in ((1, 1), (2, 10), (3, 64)):
        al = O.synth_alpha(3, 0, 0, 256, regime).reshape(-1, 64)
        cnt = (al != 0).sum(axis=1)
        assert cnt.max() <= nnz and np.median(cnt) >= nnz - 1
    # A1: opaque, w00 == 0.25 exactly
    assert np.all(O.synth_alpha(0, 0, 0, 4, 1).reshape(-1, 64)[:, 0] == np.float32(0.25))
    # A4 mixed: the non-zero pattern differs from block to block (0 / 1 / ~10 / ~64 per block), ~8 on average,
    # and most octets of 8 consecutive blocks hold blocks of different classes
    cnt = (O.synth_alpha(5, 1, 0, 4096, 4).reshape(-1, 64) != 0).sum(axis=1)
    frac = [np.mean(cnt == 0), np.mean(cnt == 1), np.mean((cnt >= 8) & (cnt <= 10)), np.mean(cnt >= 60)]
    assert abs(frac[0] - 8 / 64) < 0.03 and abs(frac[1] - 40 / 64) < 0.04 and abs(frac[2] - 10 / 64) < 0.03 and abs(frac[3] - 6 / 64) < 0.03, frac
    assert 6.5 < cnt.mean() < 10
    octets = cnt.reshape(-1, 8)
    assert np.mean(octets.max(axis=1) >= 60) > 0.4 and np.mean((octets.max(axis=1) >= 60) & (octets.min(axis=1) <= 1)) > 0.4


@pytest.mark.skipif(not os.path.exists(os.path.join(O.ORACLE_DIR, "_ref", "ref_bench")), reason="oracle/_ref/ref_bench not built (no /root/reference here)")
@pytest.mark.parametrize("regime", [1, 2, 3, 4])
def test_block_loop_live_vs_reference_on_the_bench_workload(regime):
    """oracle/_ref/ref_bench: the UNMODIFIED reference's mj_compose_with_mask (compose.c:237, 64 mj_convolve calls per block) on
    libjpeg's coefficient arrays against the restatement, on the synthetic workload bench.py measures (a 64x48 dropon here, every
    alpha regime): every row of every rectangle identical.  The same binary is bench.py's cpu_baseline ("kind": "reference")."""
    import json
    import subprocess
    exe = os.path.join(O.ORACLE_DIR, "_ref", "ref_bench")
    out = subprocess.run([exe, "640", "480", "64", "48", "5", "7", str(regime), "1000", "4", "2", "0.05"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-500:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["rows_compared_with_the_restatement"] == 4 * (6 + 3 + 3) and d["rows_that_differ"] == 0, d
    assert d["blocks_per_s"] > 0 and d["blocks_per_s_1thread"] > 0


@pytest.mark.skipif(not os.path.exists(os.path.join(O.ORACLE_DIR, "_ref", "ref_bench")), reason="oracle/_ref/ref_bench not built (no /root/reference here)")
def test_dc_effects_live_vs_reference_on_the_bench_workload():
    """oracle/_ref/ref_bench effects: the reference's mj_effect_luminance + mj_effect_tint (effect.c:185, :116) on libjpeg's arrays
    holding the synthetic planes of bench.py's effects workload == the restatement's orc_effect_dc, every row of every plane"""
    import json
    import subprocess
    exe = os.path.join(O.ORACLE_DIR, "_ref", "ref_bench")
    out = subprocess.run([exe, "effects", "330", "250", "77", "2", "7", "5", "-3", "0.02"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-500:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["rows_that_differ"] == 0 and d["rows_compared_with_the_restatement"] == 2 * (32 + 16 + 16), d
