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


def test_library_and_mirror_agree_on_the_header_version(hiplib):
    """MJX_VERSION of include/mjx.h == mjx_version() of the built library == the Python mirror's constant (structs grow between
    versions; _lib.lib() refuses a library of another version)"""
    hdr = open(os.path.join(ROOT, "include", "mjx.h")).read()
    v = int(re.search(r"#define\s+MJX_VERSION\s+(\d+)", hdr).group(1))
    assert hiplib.lib().mjx_version() == v == hiplib.MJX_VERSION


def test_plan_options_are_validated_on_the_host(hiplib):
    L = hiplib.lib()
    plan = ctypes.c_void_p()
    job = hiplib.RectJob(16, 16, 16, 4, 4, 0, 0, 1, 1, 0)
    q = np.ones(64, np.uint16)
    qp = q.ctypes.data_as(ctypes.c_void_p)
    assert L.mjx_compose_plan_create_opts(ctypes.byref(plan), ctypes.pointer(job), 1, qp, 1, None) == hiplib.MJX_ERR_NULL_DATA
    for field, value in (("run_octets", 17), ("run_octets", -1), ("stream_run_octets", 25), ("tail_min_octets", -2), ("beside_share", 1.5), ("beside_share", -0.1), ("flags", 128)):
        opt = hiplib.PlanOptions()
        setattr(opt, field, value)
        assert L.mjx_compose_plan_create_opts(ctypes.byref(plan), ctypes.pointer(job), 1, qp, 1, ctypes.byref(opt)) == hiplib.MJX_ERR_BAD_ARG, (field, value)
    opt = hiplib.PlanOptions()
    opt.reserved[3] = 1
    assert L.mjx_compose_plan_create_opts(ctypes.byref(plan), ctypes.pointer(job), 1, qp, 1, ctypes.byref(opt)) == hiplib.MJX_ERR_BAD_ARG
    # a pitch or a row count the kernels' 24-bit block arithmetic cannot hold
    wide = hiplib.RectJob(16, 16, 16, 1 << 24, 4, 0, 0, 1, 1, 0)
    ok = hiplib.PlanOptions()
    assert L.mjx_compose_plan_create_opts(ctypes.byref(plan), ctypes.pointer(wide), 1, qp, 1, ctypes.byref(ok)) in (hiplib.MJX_ERR_BAD_ARG, hiplib.MJX_ERR_NO_DEVICE)


def test_struct_layouts_match_header(hiplib):
    # mjx_rect_job: 3 pointers + 7 int32 (padded to 8) ; mjx_dc_job: pointer + 4 int32 + 2 uint16
    assert ctypes.sizeof(hiplib.RectJob) == 56
    assert ctypes.sizeof(hiplib.DcJob) == 32


def test_ctypes_mirrors_have_the_c_layout(hiplib, tmp_path):
    """sizeof / offsetof of every struct of include/mjx.h as gcc sees them == the ctypes mirrors"""
    import os
    import subprocess
    from libmodjpeg_amd import entropy as E
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mirrors = {
        "mjx_rect_job_t": (hiplib.RectJob, ["coef", "blocks_per_row", "qtable", "flags"]),
        "mjx_host_rect_t": (hiplib.HostRect, ["rows_ptr", "q", "blocks_per_row", "flags", "schedule", "stream"]),
        "mjx_plan_stats_t": (hiplib.PlanStats, ["blocks", "runs", "union_terms", "dc_only_octets", "resident_waves", "run_octets_min", "low_row_octets", "build_us", "reserved", "opaque_octets"]),
        "mjx_plan_options_t": (hiplib.PlanOptions, ["flags", "run_octets", "stream_run_octets", "tail_min_octets", "beside_share", "reserved"]),
        "mjx_dc_job_t": (hiplib.DcJob, ["coef", "value", "q0"]),
        "mjx_tileprep_t": (hiplib.TilePrep, ["image", "crop_w", "off_y", "source", "ncomp", "h_samp", "v_samp", "tile", "alpha_tile", "flags", "stream"]),
        "mjx_huff_table_t": (E.HuffTable, ["code", "size"]),
        "mjx_huff_component_t": (E.HuffComponent, ["plane", "blocks_per_row", "h_samp", "ac_tbl", "dc_fx_n", "dc_fx_value", "dc_fx_q0"]),
        "mjx_huff_scan_t": (E.HuffScan, ["ncomp", "comp", "dc", "ac", "max_coef_bits", "flags", "optimal_dc", "optimal_ac", "stream"]),
        "mjx_huff_spec_t": (E.HuffSpec, ["bits", "vals"]),
        "mjx_huff_dcomponent_t": (E.HuffDComponent, ["plane", "blocks_per_row", "ac_tbl"]),
        "mjx_huff_dscan_t": (E.HuffDScan, ["ncomp", "comp", "dc", "ac", "data", "nbytes", "restart_interval", "flags", "stream"]),
    }
    lines = ['#include <stddef.h>', '#include <stdio.h>', '#include "mjx.h"', 'int main(void) {']
    for name, (_, fields) in mirrors.items():
        lines.append(f'printf("{name} %zu\\n", sizeof({name}));')
        for f in fields:
            lines.append(f'printf("{name}.{f} %zu\\n", offsetof({name}, {f}));')
    lines += ['return 0; }']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), "-o", str(exe), str(src)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for name, (mirror, fields) in mirrors.items():
        assert int(got[name]) == ctypes.sizeof(mirror), name
        for f in fields:
            assert int(got[f"{name}.{f}"]) == getattr(mirror, f).offset, (name, f)


def test_entropy_coders_validate_on_the_host_and_have_no_cpu_path(hiplib):
    """argument errors are reported before the device is touched; valid input without a device is
    MJX_ERR_NO_DEVICE (there is no CPU implementation of either coder in the library)"""
    import torch
    from libmodjpeg_amd import entropy as E
    L = hiplib.lib()
    assert L.mjx_huff_decode_host(None) == hiplib.MJX_ERR_NULL_DATA
    assert L.mjx_huff_encode_host(None, None, None) == hiplib.MJX_ERR_NULL_DATA
    plane = np.zeros((2, 2, 64), np.int16)
    data = np.frombuffer(b"\x00" * 64 + E.EOI, np.uint8).copy()
    sc = E.HuffDScan()
    sc.ncomp, sc.mcus_x, sc.mcus_y = 1, 2, 2
    sc.comp[0] = E.HuffDComponent(plane.ctypes.data, 2, 1, 1, 0, 0)
    dc, ac = E.spec(E.STD_DC_LUMA), E.spec(E.STD_AC_LUMA)
    sc.dc[0], sc.ac[0] = ctypes.pointer(dc), ctypes.pointer(ac)
    sc.data, sc.nbytes = data.ctypes.data, data.size
    sc.comp[0].blocks_per_row = 1  # narrower than the MCU grid
    assert L.mjx_huff_decode_host(ctypes.byref(sc)) == hiplib.MJX_ERR_BAD_ARG
    sc.comp[0].blocks_per_row = 2
    sc.comp[0].ac_tbl = 2  # no such table
    assert L.mjx_huff_decode_host(ctypes.byref(sc)) == hiplib.MJX_ERR_BAD_ARG
    sc.comp[0].ac_tbl = 0
    bad = E.spec(([0, 5] + [0] * 14, [0, 1, 2, 3, 4]))  # five codes of length 2
    sc.dc[0] = ctypes.pointer(bad)
    assert L.mjx_huff_decode_host(ctypes.byref(sc)) == hiplib.MJX_ERR_BAD_ARG
    sc.dc[0] = ctypes.pointer(dc)
    if not torch.cuda.is_available():
        assert L.mjx_huff_decode_host(ctypes.byref(sc)) == hiplib.MJX_ERR_NO_DEVICE
        with pytest.raises(hiplib.MjxError):
            E.encode_scan([plane], [(2, 2)], [(1, 1)], 2, 2)


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
