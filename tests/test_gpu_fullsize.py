"""GPU tests at BASELINE.json's full sizes (-m gpu): 4096 x 1080p 4:2:0 with a 256x256 masked
dropon per image (6 291 456 blocks per pass), one 3840x2160 full-frame compose, and the DC
effects over a 4K batch.  At these sizes the oracle is too slow to replay everything, so the
checks are size-independent properties evaluated exactly over the WHOLE batch on the device, plus
the oracle on sampled images:

  * opaque overlay (regime A1: w00 = 0.25 exactly => Y == X): every composed block equals
    trunc(tile / q) -- closed form of reference compose.c:277-336 for integer tiles
  * zero alpha: the batch comes back unchanged (dequantise/requantise is the identity when
    coef*q does not wrap)
  * schedule independence: the result does not depend on how the launch is cut into per-wave
    runs (run length 1, 5, 16 octets) -- the LDS-DMA / vmcnt bookkeeping of the pipelined kernel
    would show up here as a data race
  * sampled images (first, last, middle of the batch) bit-equal to the oracle
  * entropy coders (row f-4): composed 1080p / 4K images -> scan data -> planes, everything on
    the device, is the identity on the real blocks, and libjpeg's edge dummy blocks come back in
    the MCU padding
"""
import ctypes
import os

import numpy as np
import pytest

from tests import oracle_lib as O

pytestmark = pytest.mark.gpu

N_IMAGES = 4096


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available()
    import libmodjpeg_amd  # noqa: F401
    return torch


def _batch(regime, n=N_IMAGES):
    from libmodjpeg_amd.batch import ComposeBatch
    b = ComposeBatch(n, 1920, 1080, 256, 256, 52, 26, regime)
    b.fill_synthetic()
    return b


def _rect_view(b, c):
    row0, col0, rows, cols = b.rect[c]
    return b.coef[c][:, row0:row0 + rows, col0:col0 + cols]


def test_full_batch_opaque_overlay_closed_form(torch_mod):
    torch = torch_mod
    b = _batch(1).build_plan()
    assert b.total_blocks() == 6291456
    before_outside = [b.coef[c][:, 0:4].clone() for c in range(3)]  # rows above the rectangle
    b.run()
    torch.cuda.synchronize()
    for c in range(3):
        rows, cols = b.rect[c][2], b.rect[c][3]
        tile = b.tile[c].view(b.n, rows, cols, 64).to(torch.int32)  # integer-valued floats
        q = torch.from_numpy(b.quant[c].astype(np.int32)).cuda()
        want = torch.div(tile, q, rounding_mode="trunc").to(torch.int16)
        assert torch.equal(_rect_view(b, c), want), c
        assert torch.equal(b.coef[c][:, 0:4], before_outside[c])


def test_full_batch_zero_alpha_is_identity(torch_mod):
    torch = torch_mod
    b = _batch(2)
    for a in b.alpha:
        a.zero_()
    before = [p.clone() for p in b.coef]
    b.build_plan().run()
    torch.cuda.synchronize()
    for c in range(3):
        assert torch.equal(b.coef[c], before[c]), c


@pytest.mark.parametrize("regime", [2, 3, 4])
def test_full_batch_schedule_independence_and_oracle_samples(torch_mod, regime):
    """regime 4 (mixed): every block has its own non-zero pattern, so octets mix blocks that skip
    almost every term with blocks that need all 64.  The run length is a real parameter of the plan
    (mjx_plan_options_t.run_octets; rounds 3-4 set an environment variable that the product library
    stopped reading): the test asserts through mjx_compose_plan_stats that the three plans really are cut differently,
    once with the low-column build of the blend kernel and (regime 2) once with the general build doing the same runs."""
    torch = torch_mod
    n = N_IMAGES if regime == 2 else 1024
    ref, seen = None, []
    cases = [(16, True), (1, True), (5, True)] + ([(5, False), (1, False)] if regime == 2 else [])
    for run_octets, low_column_build in cases:
        b = _batch(regime, n).build_plan(run_octets=run_octets, stream_run_octets=(0 if run_octets == 16 else run_octets), low_column_build=low_column_build)
        st = b.plan_stats()
        if run_octets == 16:  # (a rest shorter than half a run joins the run before it: up to 16 + 7)
            assert 16 <= st["run_octets_max"] <= 23, st
        elif run_octets == 1:
            assert st["run_octets_max"] == 1 and st["run_octets_min"] == 1, st
        else:  # a rest shorter than half a run joins the run before it
            assert 5 <= st["run_octets_max"] <= 7 and st["run_octets_min"] >= 2, st
        assert (st["runs_low_columns"] > 0) == (low_column_build and regime == 2), st
        seen.append((st["runs"], st["run_octets_max"], st["runs_low_columns"] > 0))
        b.run()
        torch.cuda.synchronize()
        rects = [_rect_view(b, c).clone() for c in range(3)]
        if ref is None:
            ref = rects
            # oracle on sampled images
            for i in (0, n // 2 + 3, n - 1):
                for c, (h, v) in enumerate(b.geo.samp):
                    bpr, _ = b.geo.array_dims[c]
                    row0, col0, rows, cols = b.rect[c]
                    want = np.empty((rows, cols * 64), np.int16)
                    for l in range(rows):
                        want[l] = O.synth_coef(i, c, (row0 + l) * bpr + col0, cols, b.quant[c])
                    want = want.reshape(-1)
                    O.compose_rect(want, cols, 0, 0, rows, cols, b.quant[c], O.synth_tile(i, c, 0, rows * cols), O.synth_alpha(i, c, 0, rows * cols, regime))
                    assert np.array_equal(rects[c][i].cpu().numpy().reshape(-1), want), (regime, i, c)
        else:
            for c in range(3):
                assert torch.equal(rects[c], ref[c]), (regime, run_octets, low_column_build, c)
        del b
    assert len(set(seen)) == len(seen), f"the plans do not differ: {seen}"


@pytest.mark.parametrize("dims", [(256, 256, 512), (48, 80, 64), (16, 16, 32)])
def test_density_grouping_is_only_a_schedule(torch_mod, dims):
    """mjx_compose_plan_create_ex(MJX_PLAN_GROUP_BY_DENSITY): with the mixed regime A4 (every block its
    own pattern) most octets are formed by density, through the gather table; with the uniform regimes
    none is.  Either way the result equals the plan of consecutive octets and the oracle.  The small
    rectangles are ragged (60 and 15 blocks; 4 and 1)."""
    torch = torch_mod
    from libmodjpeg_amd.batch import ComposeBatch
    dw, dh, n = dims
    for regime in (4, 2):
        out = []
        for group in (True, False):
            b = ComposeBatch(n, 1920, 1080, dw, dh, 52, 26, regime)
            b.fill_synthetic()
            b.build_plan(group_by_density=group)
            if not group or regime != 4:
                assert b.gather_octets == 0
            elif dw * dh >= 48 * 80:
                assert b.gather_octets > 0.5 * b.total_blocks() / 8
            b.run()
            torch.cuda.synchronize()
            out.append([_rect_view(b, c).clone() for c in range(3)])
            outside = [b.coef[c][:, 0:4].clone() for c in range(3)]
            fresh = ComposeBatch(n, 1920, 1080, dw, dh, 52, 26, regime)
            fresh.fill_synthetic()
            for c in range(3):
                assert torch.equal(outside[c], fresh.coef[c][:, 0:4])  # nothing outside the rectangle moved
            del fresh
        for c in range(3):
            assert torch.equal(out[0][c], out[1][c]), (regime, c)
        for i in (0, n - 1):
            for c in range(3):
                bpr, _ = b.geo.array_dims[c]
                row0, col0, rows, cols = b.rect[c]
                want = np.empty((rows, cols * 64), np.int16)
                for l in range(rows):
                    want[l] = O.synth_coef(i, c, (row0 + l) * bpr + col0, cols, b.quant[c])
                want = want.reshape(-1)
                O.compose_rect(want, cols, 0, 0, rows, cols, b.quant[c], O.synth_tile(i, c, 0, rows * cols), O.synth_alpha(i, c, 0, rows * cols, regime))
                assert np.array_equal(out[0][c][i].cpu().numpy().reshape(-1), want), (regime, i, c)


@pytest.mark.parametrize("regime,group", [(1, True), (2, True), (3, True), (4, True), (4, False), (5, True)])
def test_int16_tiles_equal_float_tiles(torch_mod, regime, group):
    """MJX_JOB_TILES_I16: the tiles kept as the int16 coefficients they are made from and converted in
    the kernel (tile = (float)coef; weight = (float)((float)coef * K) with the reference's double K) give
    the same planes as the float tiles, in every regime, with and without the gather table, for ragged
    rectangles too; and the float form derived from the int16 arrays on the host equals the float arrays"""
    torch = torch_mod
    from libmodjpeg_amd.batch import ComposeBatch
    for dw, dh, n in ((256, 256, 128), (48, 80, 16)):
        out = {}
        for i16 in (False, True):
            b = ComposeBatch(n, 1920, 1080, dw, dh, 52, 26, regime, tiles_i16=i16)
            b.fill_synthetic()
            b.build_plan(group_by_density=group)
            if i16:
                for c in range(3):
                    tf, af = b.tiles_as_float(3, c)
                    assert np.array_equal(tf.view(np.uint32), keep[c][0].view(np.uint32)) and np.array_equal(af.view(np.uint32), keep[c][1].view(np.uint32)), (regime, c)
            else:
                keep = [b.tiles_as_float(3, c) for c in range(3)]
            b.run()
            b.run()  # twice: the second pass starts from composed planes
            torch.cuda.synchronize()
            out[i16] = [_rect_view(b, c).clone() for c in range(3)]
            del b
        for c in range(3):
            assert torch.equal(out[False][c], out[True][c]), (regime, group, dw, c)


def test_real_mask_regime_a5(torch_mod):
    """regime A5: the tiles of a real overlay (hard-edged disc) prepared from pixels on the device;
    the alpha tiles have the density profile the survey measured for such a mask (bimodal, ~7 non-zero
    weights per block on average), the grouped plan uses the gather table and equals the plain plan
    and the oracle fed with the same tiles"""
    torch = torch_mod
    from libmodjpeg_amd.batch import ComposeBatch
    n, out = 64, []
    for group in (True, False):
        b = ComposeBatch(n, 1920, 1080, 256, 256, 52, 26, 5)
        b.fill_synthetic()
        b.build_plan(group_by_density=group)
        if group:
            nnz = (b.alpha[0][0] != 0).sum(dim=1).cpu().numpy()
            assert 4.0 < nnz.mean() < 12.0 and (nnz <= 1).mean() > 0.6 and (nnz >= 32).mean() > 0.04, (nnz.mean(), np.bincount(nnz, minlength=65))
            assert b.gather_octets > 0
            tiles = [(b.tile[c][0].cpu().numpy().reshape(-1).copy(), b.alpha[c][0].cpu().numpy().reshape(-1).copy()) for c in range(3)]
        b.run()
        torch.cuda.synchronize()
        out.append([_rect_view(b, c).clone() for c in range(3)])
    for c in range(3):
        assert torch.equal(out[0][c], out[1][c]), c
    for i in (0, n - 1):
        for c in range(3):
            bpr, _ = b.geo.array_dims[c]
            row0, col0, rows, cols = b.rect[c]
            want = np.empty((rows, cols * 64), np.int16)
            for l in range(rows):
                want[l] = O.synth_coef(i, c, (row0 + l) * bpr + col0, cols, b.quant[c])
            want = want.reshape(-1)
            O.compose_rect(want, cols, 0, 0, rows, cols, b.quant[c], tiles[c][0], tiles[c][1])
            assert np.array_equal(out[0][c][i].cpu().numpy().reshape(-1), want), (i, c)


@pytest.mark.parametrize("regime,i16", [(2, False), (5, False), (4, True)])
def test_one_overlay_shared_by_every_image(torch_mod, regime, i16):
    """SURVEY 8d's secondary variant and the reference's own use (one dropon, many images: compose.c:155-177): every rectangle of
    the plan points at the SAME tile arrays.  Smooth masks, the real disc mask (opaque blocks written from the shared tile, rim
    blocks gathered from it), the mixed regime with int16 tiles; == the oracle fed with that one tile set, image by image"""
    torch = torch_mod
    from libmodjpeg_amd.batch import ComposeBatch
    n = 96
    b = ComposeBatch(n, 1920, 1080, 256, 256, 52, 26, regime, tiles_i16=i16, shared_dropon=True)
    b.fill_synthetic()
    assert b.tile[0].shape[0] == 1
    b.build_plan()
    st = b.plan_stats()
    if regime != 2:
        assert st["opaque_octets"] > 0 and st["zero_alpha_octets"] > 0
    b.run()
    torch.cuda.synchronize()
    tiles = [b.tiles_as_float(0, c) for c in range(3)]
    for i in (0, 1, n // 2, n - 1):
        for c in range(3):
            bpr, _ = b.geo.array_dims[c]
            row0, col0, rows, cols = b.rect[c]
            want = np.empty((rows, cols * 64), np.int16)
            for l in range(rows):
                want[l] = O.synth_coef(i, c, (row0 + l) * bpr + col0, cols, b.quant[c])
            want = want.reshape(-1)
            O.compose_rect(want, cols, 0, 0, rows, cols, b.quant[c], tiles[c][0], tiles[c][1])
            assert np.array_equal(b.rect_numpy(i, c).reshape(-1), want), (i, c)


def test_4k_fullframe_vs_oracle(torch_mod):
    """BASELINE configs[1]: one 3840x2160 4:2:0 image, full-frame alpha-masked dropon (194 400 blocks)"""
    torch = torch_mod
    from libmodjpeg_amd.batch import ComposeBatch
    b = ComposeBatch(1, 3840, 2160, 3840, 2160, 0, 0, 3)
    b.fill_synthetic()
    b.build_plan()
    assert b.total_blocks() == 194400
    jobs = []
    for c in range(3):
        bpr, prow = b.geo.array_dims[c]
        _, _, rows, cols = b.rect[c]
        assert (rows, cols) == (prow, bpr)
        jobs.append((O.synth_coef(0, c, 0, rows * cols, b.quant[c]), cols, 0, 0, rows, cols, b.quant[c], O.synth_tile(0, c, 0, rows * cols),
                     O.synth_alpha(0, c, 0, rows * cols, 3)))
    O.compose_jobs(jobs, min(16, os.cpu_count() or 1))
    b.run()
    torch.cuda.synchronize()
    for c in range(3):
        assert np.array_equal(b.coef[c][0].cpu().numpy().reshape(-1), jobs[c][0]), c


def _dc_expect(torch, dc, q0, value):
    """reference effect.c:207-217 on a tensor of DCs: every step on a short"""
    d = (dc.to(torch.int32) * q0).to(torch.int16).to(torch.int32)  # short *= q
    d = (d + value).to(torch.int16).to(torch.int32)                # short += value
    d = d.clamp(-2047, 2047)
    return torch.div(d, q0, rounding_mode="trunc").to(torch.int16)


def test_effects_4k_batch_closed_form(torch_mod):
    """BASELINE configs[4] on 512 images (12.7 GB), compared element for element: luminance(+7),
    tint(+5,-3), then values that clamp; the DC arithmetic of reference effect.c:207-217 evaluated
    with torch int ops over the whole batch, everything but the DCs untouched"""
    torch = torch_mod
    from libmodjpeg_amd.batch import EffectBatch
    eb = EffectBatch(512, 3840, 2160)
    eb.fill_synthetic()

    for lum, cb, cr, fused in ((7, 5, -3, False), (-1900, 300, 0, True)):
        before = [p.clone() for p in eb.coef]
        eb.build_plans(lum, cb, cr, fused=fused).run()
        torch.cuda.synchronize()
        for c, value in enumerate((lum, cb, cr)):
            tw, th = eb.geo.true_dims[c]
            want = before[c].clone()
            if not (c > 0 and value == 0):
                want[:, :th, :tw, 0] = _dc_expect(torch, before[c][:, :th, :tw, 0], int(eb.quant[c][0]), value)
            assert torch.equal(eb.coef[c], want), (lum, cb, cr, c)
    # the padded block row of the Y array (row 270.. none for 4K; for 1080p row 135) is never touched
    eb2 = EffectBatch(8, 1920, 1080)
    eb2.fill_synthetic()
    pad = eb2.coef[0][:, 135].clone()
    eb2.build_plans(100, 0, 0).run()
    torch.cuda.synchronize()
    assert torch.equal(eb2.coef[0][:, 135], pad)


def test_effects_4k_batch_full_size(torch_mod):
    """BASELINE configs[4] at its size: 4096 x 3840x2160 4:2:0 = 102 GB of JCOEF resident on one
    device, mj_effect_luminance(+7) then mj_effect_tint(+5,-3) as two launches.  Every DC of the
    batch is compared with the closed form of reference effect.c:207-217; the other 63 coefficients
    of every block are covered by two checksums taken before and after (a copy of the batch would
    not fit); sampled images are replayed through the oracle."""
    torch = torch_mod
    from libmodjpeg_amd.batch import EffectBatch
    n = N_IMAGES
    eb = EffectBatch(n, 3840, 2160)
    assert sum(p.numel() * 2 for p in eb.coef) == n * 194400 * 128 == 101921587200
    eb.fill_synthetic()
    wts = torch.arange(1, 64, device="cuda", dtype=torch.int32) * 2654435 % 1009  # position weights of the AC checksum

    def ac_sums(p):
        s1 = s2 = 0
        for lo in range(0, n, 128):  # chunks: the int32 temporaries stay small
            ac = p[lo:lo + 128, :, :, 1:].to(torch.int32)
            s1 += int(ac.sum(dtype=torch.int64))
            s2 += int((ac * wts).sum(dtype=torch.int64))
        return s1, s2

    dc_before = [p[..., 0].clone() for p in eb.coef]
    sums_before = [ac_sums(p) for p in eb.coef]
    lum, cb, cr = 7, 5, -3
    eb.build_plans(lum, cb, cr, fused=False).run()
    torch.cuda.synchronize()
    for c, value in enumerate((lum, cb, cr)):
        tw, th = eb.geo.true_dims[c]
        want = dc_before[c].clone()
        want[:, :th, :tw] = _dc_expect(torch, dc_before[c][:, :th, :tw], int(eb.quant[c][0]), value)
        assert torch.equal(eb.coef[c][..., 0], want), c
        assert ac_sums(eb.coef[c]) == sums_before[c], c
    for i in (0, n // 2 + 1, n - 1):
        for c, value in enumerate((lum, cb, cr)):
            bpr, prow = eb.geo.array_dims[c]
            tw, th = eb.geo.true_dims[c]
            want = O.synth_coef(i, c, 0, bpr * prow, eb.quant[c])
            O.effect_dc(want, bpr, th, tw, int(eb.quant[c][0]), value)
            assert np.array_equal(eb.coef[c][i].cpu().numpy().reshape(-1), want), (i, c)


def test_compose_step_time_does_not_depend_on_the_plane_contents(torch_mod):
    """bench.py composes the same planes in place step after step (compose.c:237 is in place and
    not idempotent).  The kernel has one data-dependent path, (int)Y for Y >= 2^31
    (mjx_compose.hip octet_step): with |tile|, |coef*q| <= 2^15 and sum |w| < 64 it cannot be
    reached from this workload, before or after any number of compositions.  Timing the launch on
    pristine planes and on planes composed 200 times must therefore agree."""
    torch = torch_mod
    b = _batch(2).build_plan()
    pristine = b.rect_clone()

    for _ in range(200):  # clocks up; the planes have then been composed 200 times
        b.run()
    torch.cuda.synchronize()
    deep = b.rect_clone()

    def launch_ms(planes):
        b.rect_restore(planes)  # the same copy before either case: same cache state at launch
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        b.run()
        e.record()
        torch.cuda.synchronize()
        return a.elapsed_time(e)

    # alternate the two cases launch by launch: the device's own drift (several per cent over
    # seconds on these boxes) then hits both alike
    t_pristine, t_deep = [], []
    for _ in range(20):
        t_pristine.append(launch_ms(pristine))
        t_deep.append(launch_ms(deep))
    t_pristine, t_deep = float(np.median(t_pristine)), float(np.median(t_deep))
    assert abs(t_pristine - t_deep) <= 0.06 * t_deep, (t_pristine, t_deep)
    b.rect_restore(deep)
    # and the values stay inside the range the bound assumes
    for c in range(3):
        assert int(_rect_view(b, c).to(torch.int32).abs().max()) * int(b.quant[c].max()) < 2 ** 31


@pytest.mark.parametrize("side_by_side", [False, True])
def test_compose_inside_a_hip_graph(torch_mod, side_by_side):
    """mjx_compose_run does no allocation / host sync: it can be captured and replayed - also when the step's two kernels run
    side by side (regime A4: blocks of every kind), i.e. when the capture forks to the plan's side stream and joins again"""
    torch = torch_mod
    regime = 4 if side_by_side else 2
    a, b = _batch(regime, 64).build_plan(side_by_side=False), _batch(regime, 64).build_plan(side_by_side=side_by_side)
    assert (b.plan_stats()["runs_beside"] > 0) == side_by_side
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        b.run()  # warm-up on the capture stream
        torch.cuda.current_stream().synchronize()
        b.fill_synthetic()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            b.run()
            b.run()
        g.replay()
    torch.cuda.synchronize()
    b.fill_synthetic()  # the capture pass itself did not execute; run the graph once from pristine
    g.replay()
    a.run()
    a.run()
    torch.cuda.synchronize()
    for c in range(3):
        assert torch.equal(a.coef[c], b.coef[c]), c


@pytest.mark.parametrize("size,n", [((1920, 1080), 48), ((3840, 2160), 4)])
def test_entropy_round_trip_of_composed_images_on_the_device(torch_mod, size, n):
    """encode -> decode of whole composed images with the planes resident on the device
    (MJX_HUFF_PLANES_ON_DEVICE): what comes back equals what went in on every real block; in the MCU
    padding (1080 = 67.5 MCU rows) it is what libjpeg's encoder put there: DC of the block before it
    in the MCU, AC zero (jctrans.c compress_output)"""
    torch = torch_mod
    from libmodjpeg_amd import entropy as E
    from libmodjpeg_amd.batch import ComposeBatch
    b = ComposeBatch(n, size[0], size[1], 256, 256, 52, 26, 2)
    b.fill_synthetic()
    b.build_plan().run()
    torch.cuda.synchronize()
    g = b.geo
    total = 0
    for i in range(n):
        planes = [b.coef[c][i] for c in range(3)]
        data = E.encode_scan(planes, g.true_dims, g.samp, g.mcus_x, g.mcus_y)
        total += len(data)
        back = [torch.full_like(p, 0x5555) for p in planes]
        E.decode_scan(data + E.EOI, back, g.samp, g.mcus_x, g.mcus_y)
        for c in range(3):
            w, h = g.true_dims[c]
            assert torch.equal(back[c][:h, :w], planes[c][:h, :w]), (i, c)
            if h < planes[c].shape[0]:  # dummy rows below the image: AC zero, DC carried inside the MCU
                pad = back[c][h:, :w]
                assert int(pad[..., 1:].abs().max()) == 0
                assert g.samp[c] == (2, 2)  # (chroma arrays of these sizes have no padding rows)
                # 2x2 blocks per MCU: slot 2 (row h, even column) follows slot 1 (row h-1, odd column), slot 3 follows slot 2
                assert torch.equal(pad[0, 0::2, 0], planes[c][h - 1, 1:w:2, 0])
                assert torch.equal(pad[0, 1::2, 0], pad[0, 0::2, 0])
    assert total > n * 100000  # real amounts of scan data


@pytest.mark.parametrize("i16", [False, True])
def test_rectangles_beyond_4_gib_of_a_plane(torch_mod, i16):
    """One component plane of 4.5 GiB (8 192 blocks per row x 4 608 rows x 128 bytes: a 65 536 pixel wide image; JPEG allows
    65 535 x 65 535, reference compose.c:264-272 addresses rows and columns without any such limit) with two rectangles whose
    blocks need no blend - empty alpha blocks and blocks whose only weight is (0,0), a few blended ones among them: one that
    ends below 4 GiB of the plane and one in its last rows, past block 2^25.  Round 4's stream kernel addressed the plane with
    32-bit byte offsets in both launch modes; now the kernel of the serial launch uses 64-bit offsets and a plan whose rectangles
    reach beyond 4 GiB never takes the kernel that cannot (mjx_compose_plan_stats: runs_beside == 0 even when side by side is
    asked for).  Quantisers up to 4 000 make coef * q leave int16, so the blocks without weights ARE written back.  Checked:
    both rectangles equal the oracle, and every other byte of the 4.5 GiB is what it was (a wrapped offset would land there)."""
    torch = torch_mod
    from libmodjpeg_amd._lib import (MJX_PLAN_ALPHA_IS_FINAL, MJX_PLAN_GROUP_BY_DENSITY, MJX_PLAN_SERIAL, MJX_PLAN_SIDE_BY_SIDE, MJX_PLAN_TILES_I16, PlanStats, RectJob,
                                     check, lib)
    bpr, prow = 8192, 4608
    assert bpr * prow * 128 >= int(4.5 * 2 ** 30)
    rng = np.random.default_rng(4500 + i16)
    gen = torch.Generator(device="cuda").manual_seed(4500 + i16)
    pristine = torch.randint(-1024, 1024, (prow, bpr, 64), dtype=torch.int16, device="cuda", generator=gen)
    # (row0, col0, rows, cols): the first ends at byte 3.9 GiB of the plane, the second starts past block 2^25 (row 4096)
    rects = [(3000, 17, 40, 48), (4560, 8100, 48, 64)]
    assert (rects[0][0] + rects[0][2]) * bpr * 128 < 2 ** 32 and rects[1][0] * bpr >= 2 ** 25
    quant = np.stack([rng.integers(1, 256, 64), rng.integers(1, 4000, 64)]).astype(np.uint16)
    c0, c1 = 0.3535534, 0.5
    kscale = np.array([(c0 if j // 8 == 0 else c1) * (c0 if j % 8 == 0 else c1) / 1020.0 for j in range(64)], np.float64)
    host = []
    for (row0, col0, rows, cols) in rects:
        n = rows * cols
        kind = rng.choice(3, n, p=[0.45, 0.45, 0.10])  # no weight / only (0,0) / a handful anywhere
        if i16:
            ti = rng.integers(-2000, 2000, (n, 64)).astype(np.int16)
            ai = np.zeros((n, 64), np.int16)
            ai[kind == 1, 0] = rng.choice(np.array([2040, 1020, 1, -7], np.int16), int((kind == 1).sum()))
            dense = rng.integers(-2040, 2040, (n, 64)).astype(np.int16) * (rng.random((n, 64)) < 0.2)
            ai[kind == 2] = dense[kind == 2]
            host.append((ti, ai, ti.astype(np.float32), (ai.astype(np.float32).astype(np.float64) * kscale).astype(np.float32)))
        else:
            t = np.round(rng.standard_normal((n, 64)) * 200).astype(np.float32)
            a = np.zeros((n, 64), np.float32)
            a[kind == 1, 0] = rng.choice(np.array([0.25, 0.125, 1e-3, -0.01], np.float32), int((kind == 1).sum()))
            dense = (rng.random((n, 64)) * 0.1).astype(np.float32) * (rng.random((n, 64)) < 0.2)
            a[kind == 2] = dense[kind == 2]
            host.append((t, a, t, a))
    want = []
    for j, (row0, col0, rows, cols) in enumerate(rects):  # the oracle on the rows the rectangle touches
        rowsv = pristine[row0:row0 + rows].cpu().numpy().reshape(-1).copy()
        O.compose_rect(rowsv, bpr, 0, col0, rows, cols, np.ascontiguousarray(quant[j]), host[j][2].reshape(-1), host[j][3].reshape(-1))
        want.append(torch.from_numpy(rowsv.reshape(rows, bpr, 64)).cuda())
    dev = [(torch.from_numpy(np.ascontiguousarray(h[0])).cuda(), torch.from_numpy(np.ascontiguousarray(h[1])).cuda()) for h in host]
    final = MJX_PLAN_GROUP_BY_DENSITY | MJX_PLAN_ALPHA_IS_FINAL | (MJX_PLAN_TILES_I16 if i16 else 0)
    # (which rectangles, launch flag, is the co-resident stream kernel allowed)
    for which, how, beside in (((0, 1), MJX_PLAN_SERIAL, False), ((0, 1), MJX_PLAN_SIDE_BY_SIDE, False), ((0,), MJX_PLAN_SIDE_BY_SIDE, True), ((1,), MJX_PLAN_SIDE_BY_SIDE, False)):
        plane = pristine.clone()
        jobs = (RectJob * len(which))()
        for k, j in enumerate(which):
            row0, col0, rows, cols = rects[j]
            # plane_rows = 0: "not given" (it is optional) - the guard must not depend on it
            jobs[k] = RectJob(plane.data_ptr(), dev[j][0].data_ptr(), dev[j][1].data_ptr(), bpr, 0 if j == 1 else prow, row0, col0, rows, cols, j, 0)
        plan = ctypes.c_void_p()
        torch.cuda.synchronize()
        check(lib().mjx_compose_plan_create_ex(ctypes.byref(plan), jobs, len(which), quant.ctypes.data_as(ctypes.c_void_p), 2, final | how), "plan")
        st = PlanStats()
        check(lib().mjx_compose_plan_stats(plan, ctypes.byref(st)), "stats")
        assert st.zero_alpha_octets > 0 and st.dc_only_octets > 0, "the rectangles have blocks of both stream kinds"
        assert (st.runs_beside > 0) == beside, (which, how, st.runs_beside)
        check(lib().mjx_compose_run(plan, None), "run")
        torch.cuda.synchronize()
        lib().mjx_compose_plan_destroy(plan)
        expect = pristine.clone()
        for j in which:
            expect[rects[j][0]:rects[j][0] + rects[j][2]] = want[j]
        if not torch.equal(plane, expect):
            diff = (plane != expect).any(dim=2).nonzero()
            raise AssertionError(f"rectangles {which}, flags {how}: {diff.shape[0]} blocks differ, first at (row, col) {diff[0].tolist()}")
        del plane, expect
