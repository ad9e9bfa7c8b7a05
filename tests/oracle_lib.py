"""ctypes access to the CPU oracle (oracle/libmjx_oracle.so) and, when it was built in the
container, to the unmodified reference (oracle/_ref/libmodjpeg_ref.so).

TEST INFRASTRUCTURE: imported by tests/, bench.py (cpu_baseline leg) and
__graft_entry__.smoke() only.  The product package never imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.environ.get("MJX_ORACLE_SO") or os.path.join(ORACLE_DIR, "libmjx_oracle.so")  # override: sanitizer builds (scripts/asan_cpu.sh)
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libmodjpeg_ref.so")

_f32p = ctypes.POINTER(ctypes.c_float)
_i16p = ctypes.POINTER(ctypes.c_int16)
_u16p = ctypes.POINTER(ctypes.c_uint16)


class RectJob(ctypes.Structure):
    _fields_ = [("plane", _i16p), ("blocks_per_row", ctypes.c_int), ("row0", ctypes.c_int), ("col0", ctypes.c_int),
                ("rows", ctypes.c_int), ("cols", ctypes.c_int), ("q", _u16p), ("tile", _f32p), ("alpha", _f32p)]


def build_oracle():
    """Compile oracle/libmjx_oracle.so (and oracle/_ref when /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        L = ctypes.CDLL(ORACLE_SO)
        L.orc_convolve.argtypes = [_f32p, _f32p, ctypes.c_float, ctypes.c_int, ctypes.c_int]
        L.orc_convolve.restype = None
        L.orc_compose_block.argtypes = [_i16p, _u16p, _f32p, _f32p]
        L.orc_compose_block.restype = None
        L.orc_compose_rect.argtypes = [_i16p] + [ctypes.c_int] * 5 + [_u16p, _f32p, _f32p]
        L.orc_compose_rect.restype = None
        L.orc_effect_dc.argtypes = [_i16p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint16, ctypes.c_int]
        L.orc_effect_dc.restype = None
        L.orc_effect_zero.argtypes = [_i16p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.orc_effect_zero.restype = None
        L.orc_compose_jobs.argtypes = [ctypes.POINTER(RectJob), ctypes.c_int, ctypes.c_int]
        L.orc_compose_jobs.restype = ctypes.c_int
        L.orc_synth_coef.argtypes = [_i16p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_size_t, _u16p]
        L.orc_synth_coef.restype = None
        L.orc_synth_tile.argtypes = [_f32p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_size_t]
        L.orc_synth_tile.restype = None
        L.orc_synth_alpha.argtypes = [_f32p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_size_t, ctypes.c_int]
        L.orc_synth_alpha.restype = None
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


def _chk(a, dtype):
    assert a.dtype == dtype and a.flags["C_CONTIGUOUS"], (a.dtype, dtype)
    return a


def convolve(x, y, w, k, l):
    """y (float32[64], updated in place) += V_k(H_l(x)) * w — reference mj_convolve."""
    lib().orc_convolve(_p(_chk(x, np.float32), _f32p), _p(_chk(y, np.float32), _f32p), float(w), int(k), int(l))
    return y


def compose_block(coef, q, tile, alpha):
    lib().orc_compose_block(_p(_chk(coef, np.int16), _i16p), _p(_chk(q, np.uint16), _u16p),
                            _p(_chk(tile, np.float32), _f32p), _p(_chk(alpha, np.float32), _f32p))
    return coef


def compose_rect(plane, blocks_per_row, row0, col0, rows, cols, q, tile, alpha):
    """plane: int16[rows_total*blocks_per_row*64] updated in place."""
    assert tile.size == rows * cols * 64 and alpha.size == rows * cols * 64
    assert (row0 + rows) * blocks_per_row * 64 <= plane.size and col0 + cols <= blocks_per_row
    lib().orc_compose_rect(_p(_chk(plane, np.int16), _i16p), blocks_per_row, row0, col0, rows, cols,
                           _p(_chk(q, np.uint16), _u16p), _p(_chk(tile, np.float32), _f32p), _p(_chk(alpha, np.float32), _f32p))
    return plane


def effect_dc(plane, blocks_per_row, rows, cols, q0, value):
    value = ((int(value) + 2**31) % 2**32) - 2**31
    lib().orc_effect_dc(_p(_chk(plane, np.int16), _i16p), blocks_per_row, rows, cols, int(q0), value)
    return plane


def effect_zero(plane, blocks_per_row, rows, cols, keep_dc):
    lib().orc_effect_zero(_p(_chk(plane, np.int16), _i16p), blocks_per_row, rows, cols, int(keep_dc))
    return plane


def compose_jobs(jobs, nthreads):
    """jobs: list of (plane, bpr, row0, col0, rows, cols, q, tile, alpha) numpy views."""
    arr = (RectJob * len(jobs))()
    for i, (plane, bpr, row0, col0, rows, cols, q, tile, alpha) in enumerate(jobs):
        arr[i] = RectJob(_p(_chk(plane, np.int16), _i16p), bpr, row0, col0, rows, cols,
                         _p(_chk(q, np.uint16), _u16p), _p(_chk(tile, np.float32), _f32p), _p(_chk(alpha, np.float32), _f32p))
    return lib().orc_compose_jobs(arr, len(jobs), int(nthreads))


def synth_coef(image, comp, first_block, nblocks, q):
    out = np.empty(nblocks * 64, np.int16)
    lib().orc_synth_coef(_p(out, _i16p), image, comp, first_block, nblocks, _p(_chk(q, np.uint16), _u16p))
    return out


def synth_tile(image, comp, first_block, nblocks):
    out = np.empty(nblocks * 64, np.float32)
    lib().orc_synth_tile(_p(out, _f32p), image, comp, first_block, nblocks)
    return out


def synth_alpha(image, comp, first_block, nblocks, regime):
    out = np.empty(nblocks * 64, np.float32)
    lib().orc_synth_alpha(_p(out, _f32p), image, comp, first_block, nblocks, regime)
    return out


# ---- the real reference, when available (build container only) ----
_ref = None


def ref_lib():
    global _ref
    if _ref is None and os.path.exists(REF_SO):
        try:
            R = ctypes.CDLL(REF_SO)
        except OSError:
            return None
        R.mj_convolve.argtypes = [_f32p, _f32p, ctypes.c_float, ctypes.c_int, ctypes.c_int]
        R.mj_convolve.restype = None
        _ref = R
    return _ref


# JPEG Annex K tables in natural order, libjpeg quality scaling (jcparam.c jpeg_quality_scaling)
_BASE_LUMA = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
                       14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
                       49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99])
_BASE_CHROMA = np.array([17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99,
                         47, 66, 99, 99, 99, 99, 99, 99] + [99] * 32)


def quant_table(chroma, quality=85):
    scale = 5000 // quality if quality < 50 else 200 - 2 * quality
    base = _BASE_CHROMA if chroma else _BASE_LUMA
    return np.clip((base * scale + 50) // 100, 1, 255).astype(np.uint16)
