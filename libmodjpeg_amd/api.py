"""Python mirror of the reference's operator interface for the hot path.

Same names, argument meaning and return codes as the reference so that the parity tests read
like calls into libmodjpeg:

    mj_compose_with_mask(m, cd, block_x, block_y)   reference src/compose.c:237
    mj_convolve(x, y, w, k, l)                      reference src/convolve.c:29
    mj_effect_grayscale(m) / mj_effect_pixelate(m)  reference src/effect.c:28 / :70
    mj_effect_tint(m, cb, cr)                       reference src/effect.c:116
    mj_effect_luminance(m, value)                   reference src/effect.c:185

`Jpeg` stands for the coefficient side of mj_jpeg_t (reference libmodjpeg.h:99-107) without
libjpeg: per component a device-resident JCOEF plane and its quant table; `CompiledDropon` for
mj_compileddropon_t (libmodjpeg.h:120-127) with the per-block mallocs made contiguous.
All compute goes through the C ABI of include/mjx.h; torch is used for device memory only.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import DcJob, RectJob, check, lib

# return codes, reference libmodjpeg.h:60-69
MJ_OK = 0
MJ_ERR_MEMORY = 1
MJ_ERR_NULL_DATA = 2
MJ_ERR_DROPON_DIMENSIONS = 3
MJ_ERR_UNSUPPORTED_COLORSPACE = 4

# J_COLOR_SPACE values of jpeglib.h used by the reference (image.c:84-92)
JCS_GRAYSCALE = 1
JCS_RGB = 2
JCS_YCbCr = 3

_MJX_TO_MJ = {_lib.MJX_OK: MJ_OK, _lib.MJX_ERR_MEMORY: MJ_ERR_MEMORY, _lib.MJX_ERR_NULL_DATA: MJ_ERR_NULL_DATA}


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, dtype):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise TypeError(f"expected a contiguous cuda tensor of {dtype}")
    return ctypes.c_void_p(t.data_ptr())


class Component:
    """mj_component_t (reference libmodjpeg.h:88-97) with `blocks` contiguous on the device:
    float32[height_in_blocks*width_in_blocks, 64]"""

    def __init__(self, blocks, width_in_blocks, height_in_blocks, h_samp_factor=1, v_samp_factor=1):
        blocks = torch.as_tensor(blocks)
        assert blocks.numel() == width_in_blocks * height_in_blocks * 64
        self.blocks = blocks.to(device="cuda", dtype=torch.float32).contiguous().view(-1, 64)
        self.width_in_blocks = int(width_in_blocks)
        self.height_in_blocks = int(height_in_blocks)
        self.h_samp_factor = int(h_samp_factor)
        self.v_samp_factor = int(v_samp_factor)
        self.nblocks = self.width_in_blocks * self.height_in_blocks


class CompiledDropon:
    """mj_compileddropon_t (reference libmodjpeg.h:120-127)"""

    def __init__(self, image, alpha, image_colorspace=JCS_YCbCr):
        self.image = list(image)
        self.alpha = list(alpha)
        self.image_ncomponents = len(self.image)
        self.alpha_ncomponents = len(self.alpha)
        self.image_colorspace = image_colorspace


class Jpeg:
    """Coefficient side of mj_jpeg_t: `coef[c]` int16 cuda tensor [height_in_blocks,
    width_in_blocks, 64] (the virtual array of component c), `quant[c]` uint16[64] natural
    order, sampling factors and colour space as libjpeg reports them."""

    def __init__(self, planes, quant, samp=None, colorspace=JCS_YCbCr, width=None, height=None):
        self.coef = []
        for p in planes:
            p = torch.as_tensor(p)
            assert p.dim() == 3 and p.shape[2] == 64
            self.coef.append(p.to(device="cuda", dtype=torch.int16).contiguous())
        self.quant = [np.ascontiguousarray(np.asarray(q, dtype=np.uint16).reshape(64)) for q in quant]
        self.num_components = len(self.coef)
        self.samp = list(samp) if samp is not None else ([(2, 2), (1, 1), (1, 1)][: self.num_components] if self.num_components == 3 else [(1, 1)] * self.num_components)
        self.jpeg_color_space = colorspace
        self.width = width
        self.height = height

    def planes_numpy(self):
        return [p.cpu().numpy() for p in self.coef]


def mj_compose_with_mask(m, cd, block_x, block_y):
    """Blend the compiled dropon into the image at MCU block (block_x, block_y), all components,
    in place (reference compose.c:237-342)."""
    if m is None or cd is None:
        return MJ_ERR_NULL_DATA  # compose.c:238
    jobs = (RectJob * cd.image_ncomponents)()
    qt = np.zeros((cd.image_ncomponents, 64), np.uint16)
    for c in range(cd.image_ncomponents):  # compose.c:256
        ic, ac = cd.image[c], cd.alpha[c]
        plane = m.coef[c]
        h_samp, v_samp = m.samp[c]
        qt[c] = m.quant[c]
        jobs[c] = RectJob(_dev(plane, torch.int16), _dev(ic.blocks, torch.float32), _dev(ac.blocks, torch.float32),
                          plane.shape[1], plane.shape[0],
                          block_y * v_samp, block_x * h_samp,  # compose.c:264-265
                          ic.height_in_blocks, ic.width_in_blocks, c)
    plan = ctypes.c_void_p()
    rc = lib().mjx_compose_plan_create(ctypes.byref(plan), jobs, cd.image_ncomponents, qt.ctypes.data_as(ctypes.c_void_p), cd.image_ncomponents)
    check(rc, "mjx_compose_plan_create")
    try:
        check(lib().mjx_compose_run(plan, _stream()), "mjx_compose_run")
        torch.cuda.current_stream().synchronize()
    finally:
        lib().mjx_compose_plan_destroy(plan)
    return MJ_OK


def mj_convolve(x, y, w, k, l):
    """y += V_k(H_l(x)) * w on float32 cuda tensors [nblocks, 64] (reference convolve.c:29)."""
    n = x.numel() // 64
    check(lib().mjx_convolve(_dev(x, torch.float32), _dev(y, torch.float32), ctypes.c_float(float(w)), int(k), int(l), n, _stream()), "mjx_convolve")
    torch.cuda.current_stream().synchronize()


def _dc(m, comps_values):
    jobs = []
    for c, value in comps_values:
        p = m.coef[c]
        v = ((int(value) + 2**31) % 2**32) - 2**31
        jobs.append(DcJob(_dev(p, torch.int16), p.shape[1], p.shape[0], p.shape[1], v, int(m.quant[c][0]), 0))
    if jobs:
        arr = (DcJob * len(jobs))(*jobs)
        check(lib().mjx_effect_dc(arr, len(jobs), _stream()), "mjx_effect_dc")
    return MJ_OK


def mj_effect_luminance(m, value):
    if m is None or m.coef is None:
        return MJ_ERR_NULL_DATA  # effect.c:191
    if m.jpeg_color_space != JCS_YCbCr:
        return MJ_OK  # effect.c:195
    return _dc(m, [(0, value)])


def mj_effect_tint(m, cb_value, cr_value):
    if m is None or m.coef is None:
        return MJ_ERR_NULL_DATA  # effect.c:122
    if m.jpeg_color_space != JCS_YCbCr:
        return MJ_OK  # effect.c:126
    if cb_value == 0 and cr_value == 0:
        return MJ_OK  # effect.c:130
    return _dc(m, [(c, v) for c, v in ((1, cb_value), (2, cr_value)) if v != 0])  # effect.c:134/:158


def _zero(m, comps, keep_dc):
    for c in comps:
        p = m.coef[c]
        check(lib().mjx_effect_zero(_dev(p, torch.int16), p.shape[1], p.shape[0], p.shape[1], keep_dc, _stream()), "mjx_effect_zero")
    torch.cuda.current_stream().synchronize()
    return MJ_OK


def mj_effect_grayscale(m):
    if m is None or m.coef is None:
        return MJ_ERR_NULL_DATA  # effect.c:35
    if m.jpeg_color_space != JCS_YCbCr:
        return MJ_OK  # effect.c:39
    return _zero(m, range(1, m.num_components), 0)


def mj_effect_pixelate(m):
    if m is None or m.coef is None:
        return MJ_ERR_NULL_DATA  # effect.c:77
    return _zero(m, range(m.num_components), 1)
