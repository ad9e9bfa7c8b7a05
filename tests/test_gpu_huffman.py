"""GPU Huffman encoder through the C ABI (mjx_huff_encode_host) against a small pure-Python
restatement of the baseline JPEG entropy coding procedure (ITU-T T.81 F.1.2 with the Annex K.3
luminance tables; edge dummy blocks as libjpeg's jctrans.c makes them).  The byte-parity against
libjpeg itself is in tests/test_dropin_shim.py; this one pins the ABI, the error path and odd
geometries with arbitrary coefficient values."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DC_BITS = [0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0]
DC_VALS = list(range(12))
AC_BITS = [0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d]
AC_VALS = [0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08,
           0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28,
           0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
           0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89,
           0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6,
           0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
           0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa]
ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36,
          29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


from libmodjpeg_amd.entropy import HuffComponent, HuffScan, HuffTable  # the ctypes mirrors of include/mjx.h (layout pinned by tests/test_abi.py)


def derive(bits, vals):
    t, code, k = HuffTable(), 0, 0
    for length in range(1, 17):
        for _ in range(bits[length - 1]):
            t.code[vals[k]], t.size[vals[k]] = code, length
            code += 1
            k += 1
        code <<= 1
    return t


def py_encode(planes, geom, mcus_x, mcus_y, dc, ac):
    """planes[c]: int16[hib, bpr, 64]; geom[c] = (wib, hib, hs, vs)"""
    bits = []

    def put(code, size):
        bits.extend((code >> (size - 1 - i)) & 1 for i in range(size))

    last = [0] * len(planes)
    for my in range(mcus_y):
        for mx in range(mcus_x):
            for c, (wib, hib, hs, vs) in enumerate(geom):
                prev_dc = None
                for yi in range(vs):
                    for xi in range(hs):
                        bx, by = mx * hs + xi, my * vs + yi
                        if bx < wib and by < hib:
                            blk = planes[c][by, bx].astype(np.int64)
                        else:
                            blk = np.zeros(64, np.int64)
                            blk[0] = prev_dc
                        prev_dc = int(blk[0])
                        t = int(blk[0]) - last[c]
                        last[c] = int(blk[0])
                        t2 = t
                        if t < 0:
                            t, t2 = -t, t2 - 1
                        n = t.bit_length()
                        put(dc.code[n], dc.size[n])
                        if n:
                            put(t2 & ((1 << n) - 1), n)
                        r = 0
                        for k in range(1, 64):
                            v = int(blk[ZIGZAG[k]])
                            if v == 0:
                                r += 1
                                continue
                            while r > 15:
                                put(ac.code[0xF0], ac.size[0xF0])
                                r -= 16
                            v2 = v
                            if v < 0:
                                v, v2 = -v, v2 - 1
                            n = v.bit_length()
                            put(ac.code[(r << 4) + n], ac.size[(r << 4) + n])
                            put(v2 & ((1 << n) - 1), n)
                            r = 0
                        if r > 0:
                            put(ac.code[0], ac.size[0])
    while len(bits) % 8:
        bits.append(1)
    out = bytearray()
    for i in range(0, len(bits), 8):
        b = int("".join(map(str, bits[i:i + 8])), 2)
        out.append(b)
        if b == 0xFF:
            out.append(0)
    return bytes(out)


def gpu_encode(planes, geom, mcus_x, mcus_y, dc, ac):
    import torch  # noqa: F401
    from libmodjpeg_amd._lib import lib
    sc = HuffScan()
    sc.ncomp, sc.mcus_x, sc.mcus_y, sc.max_coef_bits = len(planes), mcus_x, mcus_y, 10
    keep = []
    for c, (wib, hib, hs, vs) in enumerate(geom):
        p = np.ascontiguousarray(planes[c])
        keep.append(p)
        sc.comp[c] = HuffComponent(p.ctypes.data, p.shape[1], wib, hib, hs, vs, 0, 0)
    sc.dc[0], sc.ac[0] = ctypes.pointer(dc), ctypes.pointer(ac)
    out, n = ctypes.c_void_p(), ctypes.c_size_t()
    rc = lib().mjx_huff_encode_host(ctypes.byref(sc), ctypes.byref(out), ctypes.byref(n))
    if rc != 0:
        return rc, None
    data = ctypes.string_at(out, n.value)
    ctypes.CDLL(None).free(out)
    return 0, data


@pytest.mark.parametrize("shape", [(1, 1), (5, 3), (7, 9)])
def test_interleaved_420_like_scan_with_dummy_blocks(shape):
    wy, hy = shape  # luma size in blocks (odd => dummy blocks on the right / bottom)
    rng = np.random.default_rng(wy * 100 + hy)
    mcus_x, mcus_y = -(-wy // 2), -(-hy // 2)
    geom = [(wy, hy, 2, 2), (mcus_x, mcus_y, 1, 1), (mcus_x, mcus_y, 1, 1)]
    planes = []
    for (wib, hib, hs, vs) in geom:
        bpr = -(-wib // hs) * hs + (2 if hs == 1 else 0)  # a pitch wider than the true width
        p = np.zeros((hib, bpr, 64), np.int16)
        p[:, :, 0] = rng.integers(-1023, 1024, (hib, bpr))
        mask = rng.random((hib, bpr, 63)) < 0.25
        p[:, :, 1:] = np.where(mask, rng.integers(-1023, 1024, (hib, bpr, 63)), 0)
        p[0, 0, 1:] = 0  # an all-zero AC block (EOB only)
        if hib * wib > 2:
            p[hib - 1, wib - 1, 1:] = rng.integers(1, 1024, 63)  # dense block, no EOB
            p[0, min(1, wib - 1), 1:62] = 0  # long zero runs -> ZRL
            p[0, min(1, wib - 1), 63] = -1
        planes.append(p)
    dc, ac = derive(DC_BITS, DC_VALS), derive(AC_BITS, AC_VALS)
    rc, got = gpu_encode(planes, geom, mcus_x, mcus_y, dc, ac)
    assert rc == 0
    assert got == py_encode(planes, geom, mcus_x, mcus_y, dc, ac)


def test_coefficient_outside_the_alphabet_is_an_error():
    geom = [(2, 2, 1, 1)]
    p = np.zeros((2, 2, 64), np.int16)
    p[1, 1, 5] = 1024  # 11 bits: libjpeg raises JERR_BAD_DCT_COEF
    dc, ac = derive(DC_BITS, DC_VALS), derive(AC_BITS, AC_VALS)
    rc, _ = gpu_encode([p], geom, 2, 2, dc, ac)
    assert rc == 4  # MJX_ERR_BAD_ARG
    p[1, 1, 5] = 1023
    rc, got = gpu_encode([p], geom, 2, 2, dc, ac)
    assert rc == 0 and got == py_encode([p], geom, 2, 2, dc, ac)


def test_pending_dc_effects_are_applied_while_the_encoder_reads():
    """mjx_huff_component_t.dc_fx_*: a device-resident image whose luminance / tint calls were only noted
    (host/effects.c) is written with them applied by the encoder, block by block; the bytes must equal
    those of a plane the effects were run over (oracle orc_effect_dc, reference effect.c:207-217) -
    clamps and int16 wraps included - for real blocks, dummy blocks (which repeat the PREVIOUS block's
    effective DC) and pitch padding; the plane handed in stays untouched"""
    from tests import oracle_lib as O
    rng = np.random.default_rng(77)
    wy, hy = 7, 5
    mcus_x, mcus_y = -(-wy // 2), -(-hy // 2)
    geom = [(wy, hy, 2, 2), (mcus_x, mcus_y, 1, 1), (mcus_x, mcus_y, 1, 1)]
    q0 = [3, 17, 255]
    pending = [[40, -2500, 7, 9000], [-30], []]  # luma: four values incl. a clamp and an int16 wrap; Cb: one; Cr: none
    planes, applied = [], []
    for (wib, hib, hs, vs), q, values in zip(geom, q0, pending):
        bpr = -(-wib // hs) * hs + 2
        p = np.zeros((hib, bpr, 64), np.int16)
        p[:, :, 0] = rng.integers(-1023, 1024, (hib, bpr))
        p[:, :, 1:] = np.where(rng.random((hib, bpr, 63)) < 0.2, rng.integers(-300, 300, (hib, bpr, 63)), 0)
        planes.append(p)
        e = p.copy().reshape(-1)
        for v in values:
            O.effect_dc(e, bpr, hib, wib, q, v)
        applied.append(e.reshape(p.shape))
    dc, ac = derive(DC_BITS, DC_VALS), derive(AC_BITS, AC_VALS)

    def encode(ps, with_pending):
        import torch  # noqa: F401
        from libmodjpeg_amd._lib import lib
        sc = HuffScan()
        sc.ncomp, sc.mcus_x, sc.mcus_y, sc.max_coef_bits = 3, mcus_x, mcus_y, 11  # a shifted DC may need the 11-bit category
        keep = []
        for c, ((wib, hib, hs, vs), p) in enumerate(zip(geom, ps)):
            p = np.ascontiguousarray(p)
            keep.append(p)
            sc.comp[c] = HuffComponent(p.ctypes.data, p.shape[1], wib, hib, hs, vs, 0, 0)
            if with_pending and pending[c]:
                sc.comp[c].dc_fx_n = len(pending[c])
                for i, v in enumerate(pending[c]):
                    sc.comp[c].dc_fx_value[i] = v
                sc.comp[c].dc_fx_q0 = q0[c]
        sc.dc[0], sc.ac[0] = ctypes.pointer(dc), ctypes.pointer(ac)
        out, n = ctypes.c_void_p(), ctypes.c_size_t()
        rc = lib().mjx_huff_encode_host(ctypes.byref(sc), ctypes.byref(out), ctypes.byref(n))
        assert rc == 0, rc
        data = ctypes.string_at(out, n.value)
        ctypes.CDLL(None).free(out)
        return data

    before = [p.copy() for p in planes]
    got = encode(planes, True)
    assert all(np.array_equal(a, b) for a, b in zip(planes, before)), "the encoder changed the plane it was handed"
    assert got == encode(applied, False)
    assert got == py_encode(applied, geom, mcus_x, mcus_y, dc, ac)
    assert got != encode(planes, False)
