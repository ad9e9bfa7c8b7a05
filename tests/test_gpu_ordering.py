"""The ordering contract of include/mjx.h for the synchronous entry points that are handed DEVICE memory.

Those calls replace blocking calls of the reference (image.c:94 jpeg_read_coefficients, image.c:194
jpeg_write_coefficients, compose.c:175 mj_compose_with_mask, effect.c) and work on a private stream of the
library's pool.  The contract: their first device operation is ordered after everything queued, when the call
is made, on the caller's stream (`stream` member / `after` argument), and they return after their own work.

Every test here makes the race certain instead of likely: a kernel that spins for tens of milliseconds is
queued on the caller's stream IN FRONT OF the operation that produces the call's input (or overwrites its
output), and the call follows without any host synchronisation.  Round 3's red test had exactly this shape by
accident (torch.full_like on torch's stream, the decoder's zeroing on the library's stream)."""
import ctypes

import numpy as np
import pytest

from tests import oracle_lib as O

pytestmark = pytest.mark.gpu

GEOM, MCUS_X, MCUS_Y = [(2, 2), (1, 1), (1, 1)], 40, 24


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch


@pytest.fixture(scope="module")
def spin(torch_mod):
    """spin(ms): queue a kernel on torch's current stream that keeps it busy for about `ms` milliseconds"""
    torch = torch_mod
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1000)
    torch.cuda.synchronize()
    a.record()
    torch.cuda._sleep(20_000_000)
    b.record()
    torch.cuda.synchronize()
    per_ms = 20_000_000 / max(a.elapsed_time(b), 1e-3)

    def go(ms):
        torch.cuda._sleep(int(per_ms * ms))
    go(1)
    return go


def _streams(torch):
    """the legacy default stream (stream = NULL) and a non-blocking side stream of torch's"""
    return [None, torch.cuda.Stream()]


def _planes(rng):
    planes = []
    for (hs, vs) in GEOM:
        p = np.zeros((MCUS_Y * vs, MCUS_X * hs, 64), np.int16)
        mask = rng.random(p.shape) < 0.1
        p[mask] = rng.integers(-300, 301, int(mask.sum()), dtype=np.int16)
        p[..., 0] = rng.integers(-900, 901, p.shape[:2], dtype=np.int16)
        planes.append(p)
    return planes


@pytest.mark.parametrize("side", [False, True])
def test_decoder_waits_for_a_pending_fill_of_its_output_planes(torch_mod, spin, side):
    """the planes are filled with 0x5555 by a kernel that is still waiting behind a 40 ms spin when
    mjx_huff_decode_host is called: the decoder's zeroing and stores must come AFTER that fill"""
    torch = torch_mod
    from libmodjpeg_amd import entropy as E
    rng = np.random.default_rng(31)
    planes = _planes(rng)
    true_dims = [(p.shape[1], p.shape[0]) for p in planes]
    data = E.encode_scan(planes, true_dims, GEOM, MCUS_X, MCUS_Y)
    back = [torch.zeros(p.shape, dtype=torch.int16, device="cuda") for p in planes]
    torch.cuda.synchronize()
    stream = torch.cuda.Stream() if side else torch.cuda.current_stream()
    with torch.cuda.stream(stream):
        spin(40)
        for t in back:
            t.fill_(0x5555)
        E.decode_scan(data + E.EOI, back, GEOM, MCUS_X, MCUS_Y)  # no host synchronisation in between
    # the call returns after its work: a plain copy on ANY stream sees the result
    for c, p in enumerate(planes):
        assert np.array_equal(back[c].cpu().numpy(), p), c


def test_the_spin_really_opens_the_window(torch_mod, spin):
    """control of the tests in this file: the same sequence with MJX_STREAM_IDLE (the caller's promise that nothing
    is pending, here a lie) lets the decoder run before the fill, and the fill then overwrites what it stored"""
    torch = torch_mod
    from libmodjpeg_amd import _lib
    from libmodjpeg_amd import entropy as E
    rng = np.random.default_rng(32)
    planes = _planes(rng)
    true_dims = [(p.shape[1], p.shape[0]) for p in planes]
    data = E.encode_scan(planes, true_dims, GEOM, MCUS_X, MCUS_Y)
    back = [torch.zeros(p.shape, dtype=torch.int16, device="cuda") for p in planes]
    torch.cuda.synchronize()
    real = _lib.caller_stream
    try:
        E.caller_stream = lambda: _lib.MJX_STREAM_IDLE
        spin(150)
        for t in back:
            t.fill_(0x5555)
        E.decode_scan(data + E.EOI, back, GEOM, MCUS_X, MCUS_Y)
    finally:
        E.caller_stream = real
    torch.cuda.synchronize()
    late_fill_won = all(int((t == 0x5555).sum()) == t.numel() for t in back)
    if not late_fill_won:
        # the decoder (a fraction of a millisecond of device work) did not get in front of a fill that sits behind 150 ms of
        # spinning: this box stalled the call itself for that long.  A control that depends on timing must not turn the suite red.
        pytest.skip("the unordered call was not faster than the spin on this box: the control proves nothing here")
    assert late_fill_won, "the late fill won: the window the other tests close is real"


@pytest.mark.parametrize("side", [False, True])
def test_encoder_waits_for_a_pending_write_of_its_input_planes(torch_mod, spin, side):
    torch = torch_mod
    from libmodjpeg_amd import entropy as E
    rng = np.random.default_rng(33)
    planes = _planes(rng)
    true_dims = [(p.shape[1], p.shape[0]) for p in planes]
    want = E.encode_scan(planes, true_dims, GEOM, MCUS_X, MCUS_Y)  # host planes
    src = [torch.from_numpy(p).cuda() for p in planes]
    dev = [torch.full(p.shape, 0x5555, dtype=torch.int16, device="cuda") for p in planes]
    torch.cuda.synchronize()
    stream = torch.cuda.Stream() if side else torch.cuda.current_stream()
    with torch.cuda.stream(stream):
        spin(40)
        for d, s in zip(dev, src):
            d.copy_(s)
        got = E.encode_scan(dev, true_dims, GEOM, MCUS_X, MCUS_Y)
    assert got == want


@pytest.mark.parametrize("side", [False, True])
def test_host_buffer_compose_waits_for_pending_rows_and_tiles_on_the_device(torch_mod, spin, side):
    """mjx_compose_host with MJX_RECT_ROWS_ON_DEVICE | MJX_RECT_TILES_ON_DEVICE: plane rows, tiles and alpha are all
    still being written behind a spin; also mjx_tile_schedule_create on the same pending alpha tiles"""
    torch = torch_mod
    from libmodjpeg_amd._lib import MJX_RECT_ROWS_ON_DEVICE, MJX_RECT_TILES_ON_DEVICE, HostRect, caller_stream, check, lib
    rng = np.random.default_rng(34)
    wib, hib, row0, col0, rows, cols = 40, 30, 3, 5, 16, 20
    q = O.quant_table(False)
    plane = O.synth_coef(3, 0, 0, wib * hib, q)
    tile = O.synth_tile(3, 0, 0, rows * cols)
    alpha = O.synth_alpha(3, 0, 0, rows * cols, 4)  # the mixed regime: blocks without weights, with one, with many
    want = plane.copy()
    O.compose_rect(want, wib, row0, col0, rows, cols, q, tile, alpha)
    src = [torch.from_numpy(a).cuda() for a in (plane, tile, alpha)]
    d_plane = torch.full((wib * hib * 64,), 0x5555, dtype=torch.int16, device="cuda")
    d_tile = torch.zeros(rows * cols * 64, dtype=torch.float32, device="cuda")
    d_alpha = torch.zeros(rows * cols * 64, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    stream = torch.cuda.Stream() if side else torch.cuda.current_stream()
    sc = ctypes.c_void_p()
    with torch.cuda.stream(stream):
        spin(40)
        d_plane.copy_(src[0].reshape(-1))
        d_tile.copy_(src[1].reshape(-1))
        d_alpha.copy_(src[2].reshape(-1))
        check(lib().mjx_tile_schedule_create(ctypes.byref(sc), ctypes.c_void_p(d_alpha.data_ptr()), rows, cols, 0, caller_stream()), "mjx_tile_schedule_create")
        rects = (HostRect * 1)()
        rects[0] = HostRect(d_plane.data_ptr() + row0 * wib * 128, d_tile.data_ptr(), d_alpha.data_ptr(), q.ctypes.data, wib, col0, rows, cols,
                            MJX_RECT_ROWS_ON_DEVICE | MJX_RECT_TILES_ON_DEVICE, 0, sc, caller_stream())
        # a second spin + rewrite in front of the compose itself: the schedule call above drained the first one
        spin(40)
        d_plane.copy_(src[0].reshape(-1))
        check(lib().mjx_compose_host(rects, 1), "mjx_compose_host")
    lib().mjx_tile_schedule_destroy(sc)
    assert np.array_equal(d_plane.cpu().numpy(), want.reshape(-1))


@pytest.mark.parametrize("side", [False, True])
def test_device_plane_effects_wait_for_a_pending_write(torch_mod, spin, side):
    torch = torch_mod
    from libmodjpeg_amd._lib import caller_stream, check, lib
    bpr, rows, cols = 33, 20, 31
    q = O.quant_table(False)
    plane = O.synth_coef(5, 0, 0, bpr * rows, q)
    want = plane.copy()
    O.effect_dc(want, bpr, rows, cols, int(q[0]), 37)
    O.effect_zero(want, bpr, rows, cols, 1)
    src = torch.from_numpy(plane).cuda()
    dev = torch.full((bpr * rows * 64,), 0x5555, dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    stream = torch.cuda.Stream() if side else torch.cuda.current_stream()
    with torch.cuda.stream(stream):
        spin(40)
        dev.copy_(src.reshape(-1))
        check(lib().mjx_effect_dc_device(dev.data_ptr(), bpr, rows, cols, int(q[0]), 37, caller_stream()), "mjx_effect_dc_device")
        check(lib().mjx_effect_zero_device(dev.data_ptr(), bpr, rows, cols, 1, caller_stream()), "mjx_effect_zero_device")
    assert np.array_equal(dev.cpu().numpy(), want.reshape(-1))


def test_null_stream_caller_waits_for_a_producer_on_a_blocking_stream(torch_mod, spin):
    """`stream` = NULL stands for the legacy default stream, which waits for every BLOCKING stream of the device: a producer
    (spin + fill) on a stream made by hipStreamCreate with default flags, then the decoder with stream = NULL.  A query of the
    NULL stream says nothing about such a producer, so the library must record its marker there unconditionally."""
    torch = torch_mod
    from libmodjpeg_amd import _lib
    from libmodjpeg_amd import entropy as E
    hip = ctypes.CDLL("libamdhip64.so")
    handle = ctypes.c_void_p()
    assert hip.hipStreamCreate(ctypes.byref(handle)) == 0  # default flags: a blocking stream
    try:
        rng = np.random.default_rng(35)
        planes = _planes(rng)
        true_dims = [(p.shape[1], p.shape[0]) for p in planes]
        data = E.encode_scan(planes, true_dims, GEOM, MCUS_X, MCUS_Y)
        back = [torch.zeros(p.shape, dtype=torch.int16, device="cuda") for p in planes]
        torch.cuda.synchronize()
        with torch.cuda.stream(torch.cuda.ExternalStream(handle.value)):
            spin(40)
            for t in back:
                t.fill_(0x5555)
        real = E.caller_stream
        try:
            E.caller_stream = lambda: 0  # NULL: the legacy default stream
            E.decode_scan(data + E.EOI, back, GEOM, MCUS_X, MCUS_Y)
        finally:
            E.caller_stream = real
        for c, p in enumerate(planes):
            assert np.array_equal(back[c].cpu().numpy(), p), c
    finally:
        torch.cuda.synchronize()
        hip.hipStreamDestroy(handle)


@pytest.mark.parametrize("side", [False, True])
def test_tile_preparation_waits_for_pending_work_on_its_output_tiles(torch_mod, spin, side):
    """mjx_tiles_prepare overwrites tile[] / alpha_tile[]: a fill of those arrays that is still queued behind a spin on the
    caller's stream (`stream` member) must not land on top of the prepared tiles"""
    torch = torch_mod
    from libmodjpeg_amd._lib import MJX_PREP_SOURCE_RGB, MJX_PREP_TARGET_YCC, TilePrep, caller_stream, check, lib
    from libmodjpeg_amd.batch import disc_overlay
    w, h = 64, 48
    rgba = disc_overlay(w, h)
    image = np.ascontiguousarray(rgba[..., :3])
    alpha = np.ascontiguousarray(np.repeat(rgba[..., 3:4], 3, axis=2))
    samp = [(2, 2), (1, 1), (1, 1)]

    def prepare(bufs, stream_handle):
        p = TilePrep(image.ctypes.data, alpha.ctypes.data, w, h, 0, 0, w, h, 0, 0, MJX_PREP_SOURCE_RGB, MJX_PREP_TARGET_YCC, 3)
        for c, (hs, vs) in enumerate(samp):
            p.h_samp[c], p.v_samp[c] = hs, vs
            p.tile[c], p.alpha_tile[c] = bufs[c][0].data_ptr(), bufs[c][1].data_ptr()
        p.stream = stream_handle
        check(lib().mjx_tiles_prepare(ctypes.byref(p)), "mjx_tiles_prepare")

    def buffers():
        return [torch.zeros((2, (h // 16 * vs) * (w // 16 * hs), 64), dtype=torch.float32, device="cuda") for hs, vs in samp]

    want = buffers()
    torch.cuda.synchronize()
    prepare(want, caller_stream())
    got = buffers()
    torch.cuda.synchronize()
    stream = torch.cuda.Stream() if side else torch.cuda.current_stream()
    with torch.cuda.stream(stream):
        spin(40)
        for t in got:
            t.fill_(12345.0)
        prepare(got, caller_stream())
    for c in range(3):
        assert torch.equal(got[c], want[c]), c
