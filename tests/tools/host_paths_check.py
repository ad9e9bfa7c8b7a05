"""Run by tests/test_gpu_parity.py::test_host_paths_in_place_and_staged_give_the_same_bytes, once per setting of
MJX_OPTIONS=compose-copies,effect-copies (the library reads it once per process): the per-image host-buffer calls
(mjx_compose_host with gathered rectangles, mjx_effect_dc_host) through their in-place form (the kernels work on the
pinned staging buffer, which is mapped into the device) and through their staged form (copies up and down), against the
oracle, and a digest of every output on stdout so that the caller can compare the two processes byte for byte.

Rectangles: a gathered span just below 1 MB (8190 blocks: blended in place by default), one just above (8280 blocks:
one-span copies either way), three small components at once, and a call that mixes a gathered host rectangle with
rows that live on the device (MJX_RECT_ROWS_ON_DEVICE)."""
import ctypes
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from libmodjpeg_amd._lib import MJX_RECT_ROWS_ON_DEVICE, MJX_RECT_TILES_ON_DEVICE, HostRect, check, lib  # noqa: E402
from tests import oracle_lib as O  # noqa: E402

digest = hashlib.sha256()
rng = np.random.default_rng(77)
L = lib()


def tiles_for(image, rows, cols, regime):
    tile, alpha = O.synth_tile(image, 0, 0, rows * cols), O.synth_alpha(image, 0, 0, rows * cols, regime)
    return tile, alpha, torch.from_numpy(tile).cuda(), torch.from_numpy(alpha).cuda()


def compose_case(name, bpr, plane_rows, rects, device_rows=()):
    """rects: (row0, col0, rows, cols, regime); indices in device_rows keep their plane on the device"""
    arr = (HostRect * len(rects))()
    keep, planes, wants = [], [], []
    for i, (row0, col0, rows, cols, regime) in enumerate(rects):
        q = rng.integers(1, 200, 64).astype(np.uint16)
        plane = O.synth_coef(100 + i, 0, 0, bpr * plane_rows, q)
        tile, alpha, d_t, d_a = tiles_for(200 + i, rows, cols, regime)
        want = plane.copy()
        O.compose_rect(want, bpr, row0, col0, rows, cols, q, tile, alpha)
        if i in device_rows:
            d_p = torch.from_numpy(plane).cuda()
            arr[i] = HostRect(d_p.data_ptr() + row0 * bpr * 128, d_t.data_ptr(), d_a.data_ptr(), q.ctypes.data, bpr, col0, rows, cols,
                              MJX_RECT_TILES_ON_DEVICE | MJX_RECT_ROWS_ON_DEVICE, 0, None, None)
            planes.append(d_p)
        else:
            arr[i] = HostRect(plane.ctypes.data + row0 * bpr * 128, d_t.data_ptr(), d_a.data_ptr(), q.ctypes.data, bpr, col0, rows, cols,
                              MJX_RECT_TILES_ON_DEVICE, 0, None, None)
            planes.append(plane)
        keep.append((q, d_t, d_a))
        wants.append(want)
    torch.cuda.synchronize()
    check(L.mjx_compose_host(arr, len(rects)), "mjx_compose_host")
    for i, (p, w) in enumerate(zip(planes, wants)):
        got = p.cpu().numpy() if isinstance(p, torch.Tensor) else p
        assert np.array_equal(got, w), (name, i)
        digest.update(got.tobytes())


compose_case("just below 1 MB", 300, 100, [(5, 17, 90, 91, 2)])
compose_case("just above 1 MB", 300, 100, [(5, 17, 90, 92, 2)])
compose_case("three small", 240, 40, [(1, 100, 32, 32, 4), (3, 7, 16, 16, 1), (20, 200, 16, 17, 3)])
compose_case("gathered + rows on the device", 240, 40, [(1, 100, 32, 32, 4), (3, 7, 16, 16, 2)], device_rows=(1,))

for (bpr, rows, cols, value) in [(240, 136, 240, 7), (120, 68, 120, -2047), (33, 5, 31, 4000)]:
    q = O.quant_table(False)
    plane = O.synth_coef(300, 0, 0, bpr * rows, q)
    want = plane.copy()
    O.effect_dc(want, bpr, rows, cols, int(q[0]), value)
    check(L.mjx_effect_dc_host(plane.ctypes.data, bpr, rows, cols, int(q[0]), value), "mjx_effect_dc_host")
    assert np.array_equal(plane, want), (bpr, rows, cols, value)
    digest.update(plane.tobytes())

print("host paths ok", os.environ.get("MJX_OPTIONS", "-"), digest.hexdigest())
