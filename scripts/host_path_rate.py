"""PCIe-inclusive rate of the per-image host-buffer entry point (mjx_compose_host): what the
libmodjpeg.h shim pays per mj_compose call.  1080p 4:2:0 image, 256x256 dropon, regime A2.
Reported in DESIGN.md next to the device-resident kernel rate (never as bench.py's `value`)."""
import ctypes
import sys
import time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (HIP runtime first)
from libmodjpeg_amd._lib import HostRect, lib, check
from libmodjpeg_amd.batch import Geometry420, quant_table
from tests import oracle_lib as O

geo = Geometry420(1920, 1080)
quant = [quant_table(False), quant_table(True), quant_table(True)]
rects, keep = (HostRect * 3)(), []
blocks = 0
for c, (h, v) in enumerate(geo.samp):
    bpr, prow = geo.array_dims[c]
    rows, cols, row0, col0 = 16 * v, 16 * h, 26 * v, 52 * h
    plane = O.synth_coef(0, c, 0, bpr * prow, quant[c])
    tile, alpha = O.synth_tile(0, c, 0, rows * cols), O.synth_alpha(0, c, 0, rows * cols, 2)
    keep += [plane, tile, alpha]
    vp = ctypes.c_void_p
    rects[c] = HostRect(plane.ctypes.data + row0 * bpr * 128, tile.ctypes.data, alpha.ctypes.data, quant[c].ctypes.data, bpr, col0, rows, cols)
    blocks += rows * cols
L = lib()
for _ in range(3):
    check(L.mjx_compose_host(rects, 3), "mjx_compose_host")
n = 50
t0 = time.perf_counter()
for _ in range(n):
    check(L.mjx_compose_host(rects, 3), "mjx_compose_host")
dt = (time.perf_counter() - t0) / n
# rows of a rectangle at most a third as wide as the plane travel as the bare rectangle, else as whole rows; tiles from the host
bytes_pcie = sum(r.rows * (r.cols if 3 * r.cols <= r.blocks_per_row else r.blocks_per_row) * 128 * 2 + r.rows * r.cols * 512 for r in rects)
print("mjx_compose_host: %.3f ms per call, %d blocks -> %.3f Mblock/s, %.2f MB over PCIe per call (%.2f GB/s)" %
      (dt * 1e3, blocks, blocks / dt / 1e6, bytes_pcie / 1e6, bytes_pcie / dt / 1e9))
