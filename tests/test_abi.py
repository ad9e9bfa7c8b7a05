"""CPU tests of the boundary: libmjx_hip.so builds for gfx950, loads without a GPU, exports every
symbol include/mjx.h declares, and refuses to compute without a device (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hiplib():
    import __graft_entry__ as g
    g.build_hip()
    from libmodjpeg_amd import _lib
    return _lib


def test_header_symbols_all_exported(hiplib):
    hdr = open(os.path.join(ROOT, "include", "mjx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mjx_[a-z0-9_]+)\s*\(", hdr))
    bound = {name for name, _, _ in hiplib.SYMBOLS}
    assert declared == bound, (declared - bound, bound - declared)
    L = hiplib.lib()
    for name in declared:
        assert hasattr(L, name), name


def test_struct_layouts_match_header(hiplib):
    # mjx_rect_job: 3 pointers + 7 int32 (padded to 8) ; mjx_dc_job: pointer + 4 int32 + 2 uint16
    assert ctypes.sizeof(hiplib.RectJob) == 56
    assert ctypes.sizeof(hiplib.DcJob) == 32


def test_no_cpu_fallback_without_device(hiplib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = hiplib.lib()
    assert L.mjx_device_count() == 0
    plan = ctypes.c_void_p()
    job = hiplib.RectJob(16, 16, 16, 4, 4, 0, 0, 1, 1, 0)
    q = np.ones(64, np.uint16)
    rc = L.mjx_compose_plan_create(ctypes.byref(plan), ctypes.pointer(job), 1, q.ctypes.data_as(ctypes.c_void_p), 1)
    assert rc == hiplib.MJX_ERR_NO_DEVICE
    buf = np.zeros(64, np.int16)
    t = np.zeros(64, np.float32)
    vp = ctypes.c_void_p
    rc = L.mjx_compose_rect_host(buf.ctypes.data_as(vp), 1, 1, 0, 0, 1, 1, q.ctypes.data_as(vp), t.ctypes.data_as(vp), t.ctypes.data_as(vp))
    assert rc == hiplib.MJX_ERR_NO_DEVICE
    assert L.mjx_set_device(0) == hiplib.MJX_ERR_NO_DEVICE


def test_argument_validation_is_host_side(hiplib):
    L = hiplib.lib()
    vp = ctypes.c_void_p
    assert L.mjx_convolve(None, None, 1.0, 0, 0, 1, None) == hiplib.MJX_ERR_NULL_DATA
    assert L.mjx_convolve(vp(16), vp(16), 1.0, 8, 0, 1, None) == hiplib.MJX_ERR_BAD_ARG
    assert L.mjx_convolve(vp(16), vp(20), 1.0, 1, 0, 1, None) == hiplib.MJX_ERR_BAD_ARG  # misaligned
    assert L.mjx_effect_zero(None, 4, 4, 4, 0, None) == hiplib.MJX_ERR_NULL_DATA
    assert L.mjx_effect_zero(vp(16), 2, 4, 4, 0, None) == hiplib.MJX_ERR_BAD_ARG  # cols > pitch
    plan = ctypes.c_void_p()
    assert L.mjx_compose_plan_create(ctypes.byref(plan), None, 1, None, 0) == hiplib.MJX_ERR_NULL_DATA
