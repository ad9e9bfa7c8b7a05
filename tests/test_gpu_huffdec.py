"""GPU Huffman decoder through the C ABI (mjx_huff_decode_host): round trips against the encoder
of the same ABI and against the pure-Python restatement of ITU-T T.81 F.1.2 in
tests/test_gpu_huffman.py (which also writes restart segments here).  The parity with libjpeg's own
jpeg_read_coefficients() on real files is in tests/test_dropin_shim.py."""
import ctypes

import numpy as np
import pytest

from tests.test_gpu_huffman import AC_BITS, AC_VALS, DC_BITS, DC_VALS, ZIGZAG, derive, gpu_encode

pytestmark = pytest.mark.gpu


# the ctypes mirrors of include/mjx.h (layout pinned by tests/test_abi.py)
from libmodjpeg_amd.entropy import HuffDComponent as DComponent, HuffDScan as DScan, HuffSpec  # noqa: E402


def spec(bits, vals):
    s = HuffSpec()
    for i, b in enumerate(bits):
        s.bits[i] = b
    for i, v in enumerate(vals):
        s.vals[i] = v
    return s


EOI = b"\xff\xd9"


def gpu_decode(data, geom, mcus_x, mcus_y, restart=0, pitch_extra=0, tables=None, table_ids=None):
    """data: scan data as in a file (stuffed, RSTn, EOI); geom[c] = (hs, vs); returns rc, planes[c] int16[mcus_y*vs, bpr, 64].
    table_ids = (dc id of every component, ac id of every component) with tables = (list of DC specs, list of AC specs)"""
    import torch  # noqa: F401
    from libmodjpeg_amd._lib import lib
    sc = DScan()
    sc.ncomp, sc.mcus_x, sc.mcus_y, sc.restart_interval = len(geom), mcus_x, mcus_y, restart
    planes = []
    for c, (hs, vs) in enumerate(geom):
        p = np.full((mcus_y * vs, mcus_x * hs + pitch_extra, 64), 0x5555, np.int16)
        planes.append(p)
        sc.comp[c] = DComponent(p.ctypes.data, p.shape[1], hs, vs, table_ids[0][c] if table_ids else 0, table_ids[1][c] if table_ids else 0)
    if table_ids:
        keep = [list(tables[0]), list(tables[1])]
        for i, t in enumerate(keep[0]):
            sc.dc[i] = ctypes.pointer(t)
        for i, t in enumerate(keep[1]):
            sc.ac[i] = ctypes.pointer(t)
    else:
        dc, ac = tables or (spec(DC_BITS, DC_VALS), spec(AC_BITS, AC_VALS))
        sc.dc[0], sc.ac[0] = ctypes.pointer(dc), ctypes.pointer(ac)
    buf = np.frombuffer(data, np.uint8).copy()
    sc.data, sc.nbytes = buf.ctypes.data, len(data)
    rc = lib().mjx_huff_decode_host(ctypes.byref(sc))
    return rc, planes


def random_planes(rng, geom, mcus_x, mcus_y, density, amplitude):
    planes = []
    for (hs, vs) in geom:
        p = np.zeros((mcus_y * vs, mcus_x * hs, 64), np.int16)
        mask = rng.random(p.shape) < density
        p[mask] = rng.integers(-amplitude, amplitude + 1, int(mask.sum()), dtype=np.int16)
        p[..., 0] = rng.integers(-1000, 1001, p.shape[:2], dtype=np.int16)
        planes.append(p)
    return planes


@pytest.mark.parametrize("mcus", [(1, 1), (3, 2), (40, 30), (120, 68)])
@pytest.mark.parametrize("density", [0.02, 0.3, 1.0])
def test_round_trip_with_the_gpu_encoder_420(mcus, density):
    mcus_x, mcus_y = mcus
    rng = np.random.default_rng(mcus_x * 1000 + mcus_y + int(density * 100))
    geom = [(2, 2), (1, 1), (1, 1)]
    planes = random_planes(rng, geom, mcus_x, mcus_y, density, 1023 if density < 1 else 40)
    egeom = [(p.shape[1], p.shape[0], hs, vs) for p, (hs, vs) in zip(planes, geom)]
    rc, data = gpu_encode(planes, egeom, mcus_x, mcus_y, derive(DC_BITS, DC_VALS), derive(AC_BITS, AC_VALS))
    assert rc == 0
    rc, got = gpu_decode(data + EOI + b"trailing bytes are not looked at", geom, mcus_x, mcus_y, pitch_extra=3)
    assert rc == 0
    for c in range(3):
        assert np.array_equal(got[c][:, :planes[c].shape[1]], planes[c]), c
        assert np.all(got[c][:, planes[c].shape[1]:] == 0)  # pitch padding comes back zeroed


def py_encode_restart(planes, geom, mcus_x, mcus_y, dc, ac, restart):
    """like test_gpu_huffman.py_encode for full MCU grids, with RSTn markers every `restart` MCUs"""
    out, bits = bytearray(), []

    def put(code, size):
        bits.extend((code >> (size - 1 - i)) & 1 for i in range(size))

    def flush():
        while len(bits) % 8:
            bits.append(1)
        for i in range(0, len(bits), 8):
            b = int("".join(map(str, bits[i:i + 8])), 2)
            out.append(b)
            if b == 0xFF:
                out.append(0)
        bits.clear()

    dcs = dc if isinstance(dc, (list, tuple)) else [dc] * len(planes)
    acs = ac if isinstance(ac, (list, tuple)) else [ac] * len(planes)
    last, n_rst, mcu = [0] * len(planes), 0, 0
    for my in range(mcus_y):
        for mx in range(mcus_x):
            if restart and mcu and mcu % restart == 0:
                flush()
                out.extend([0xFF, 0xD0 + (n_rst & 7)])
                n_rst += 1
                last = [0] * len(planes)
            mcu += 1
            for c, (hs, vs) in enumerate(geom):
                dc, ac = dcs[c], acs[c]
                for yi in range(vs):
                    for xi in range(hs):
                        blk = planes[c][my * vs + yi, mx * hs + xi].astype(np.int64)
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
    flush()
    return bytes(out)


@pytest.mark.parametrize("restart", [0, 1, 3, 7])
@pytest.mark.parametrize("geom", [[(1, 1)], [(2, 1), (1, 1), (1, 1)], [(1, 1), (1, 1), (1, 1)], [(2, 2), (1, 1), (1, 1), (2, 2)]])
def test_against_the_python_encoder_with_restart_segments(restart, geom):
    mcus_x, mcus_y = 9, 5
    rng = np.random.default_rng(restart * 10 + len(geom))
    planes = random_planes(rng, geom, mcus_x, mcus_y, 0.15, 300)
    data = py_encode_restart(planes, geom, mcus_x, mcus_y, derive(DC_BITS, DC_VALS), derive(AC_BITS, AC_VALS), restart)
    assert data.count(b"\xff\xd0") + data.count(b"\xff\xd1") >= (1 if restart else 0)
    rc, got = gpu_decode(data + EOI, geom, mcus_x, mcus_y, restart=restart)
    assert rc == 0
    for c in range(len(geom)):
        assert np.array_equal(got[c], planes[c]), c


# T.81 Annex K.3.3 tables for chrominance, and two more DC tables and one more AC table that are merely valid
DC_C_BITS, DC_C_VALS = [0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0], list(range(12))
AC_C_BITS = [0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77]
AC_C_VALS = [0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09,
             0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38,
             0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74,
             0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5,
             0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6,
             0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa]
DC_FLAT_BITS, DC_FLAT_VALS = [0, 0, 0, 12] + [0] * 12, list(range(12))          # twelve codes of four bits
DC_PAIR_BITS, DC_PAIR_VALS = [0, 2, 2, 2, 2, 2, 2] + [0] * 9, list(range(11, -1, -1))  # two codes of every length 2..7, the categories the other way round
AC_REV_BITS, AC_REV_VALS = AC_BITS, list(reversed(AC_VALS))                      # the luminance lengths over the symbols in reverse: EOB is a 16-bit code


@pytest.mark.parametrize("ids", [((0, 1, 2), (0, 1, 1)), ((0, 1, 2, 3), (0, 1, 0, 1)), ((0, 1), (0, 0)), ((0, 0, 0), (0, 1, 2)), ((1, 0, 1), (1, 0, 1))],
                         ids=["3dc2ac", "4dc2ac", "2dc1ac", "1dc3ac-lane-rounds", "std-swapped"])
def test_components_with_their_own_tables(ids):
    """every mix of DC and AC tables the scan header allows: two AC and up to four DC tables go through the wave-per-chunk rounds
    (the usual 2 + 2 and 1 + 1 have their own instantiation, the others the generic one), three AC tables through the lane-per-chunk rounds"""
    dc_ids, ac_ids = ids
    geom = [(2, 1)] + [(1, 1)] * (len(dc_ids) - 1)
    mcus_x, mcus_y = 40, 24
    rng = np.random.default_rng(len(dc_ids) * 7 + sum(ac_ids))
    planes = random_planes(rng, geom, mcus_x, mcus_y, 0.12, 500)
    dcs = [(DC_BITS, DC_VALS), (DC_C_BITS, DC_C_VALS), (DC_FLAT_BITS, DC_FLAT_VALS), (DC_PAIR_BITS, DC_PAIR_VALS)]
    acs = [(AC_BITS, AC_VALS), (AC_C_BITS, AC_C_VALS), (AC_REV_BITS, AC_REV_VALS)]
    data = py_encode_restart(planes, geom, mcus_x, mcus_y, [derive(*dcs[i]) for i in dc_ids], [derive(*acs[i]) for i in ac_ids], 0)
    assert len(data) * 8 > 20 * 1024  # a good many chunks
    rc, got = gpu_decode(data + EOI, geom, mcus_x, mcus_y, tables=([spec(*t) for t in dcs], [spec(*t) for t in acs]), table_ids=(dc_ids, ac_ids))
    assert rc == 0
    for c in range(len(geom)):
        assert np.array_equal(got[c], planes[c]), c


def test_dc_wraps_like_a_jcoef_and_long_zero_runs():
    geom, mcus_x, mcus_y = [(1, 1)], 300, 1
    p = np.zeros((1, 300, 64), np.int16)
    p[0, :, 0] = np.where(np.arange(300) % 2 == 0, 1023, -1024)  # differences of +-2047
    p[0, ::3, 63] = -1                                              # 62 zeros then a value: three ZRLs
    p[0, 1::3, 33] = 1023
    data = py_encode_restart([p], geom, mcus_x, mcus_y, derive(DC_BITS, DC_VALS), derive(AC_BITS, AC_VALS), 0)
    rc, got = gpu_decode(data + EOI, geom, mcus_x, mcus_y)
    assert rc == 0 and np.array_equal(got[0], p)


def test_corrupt_or_mismatched_streams_are_refused_not_decoded():
    geom, mcus_x, mcus_y = [(2, 2), (1, 1), (1, 1)], 20, 10
    rng = np.random.default_rng(5)
    planes = random_planes(rng, geom, mcus_x, mcus_y, 0.2, 200)
    data = py_encode_restart(planes, geom, mcus_x, mcus_y, derive(DC_BITS, DC_VALS), derive(AC_BITS, AC_VALS), 0)
    ok = data + EOI
    assert gpu_decode(ok, geom, mcus_x, mcus_y)[0] == 0
    assert gpu_decode(data, geom, mcus_x, mcus_y)[0] == 4                                  # no EOI
    assert gpu_decode(data[:len(data) // 2] + EOI, geom, mcus_x, mcus_y)[0] == 4           # truncated
    assert gpu_decode(ok, geom, mcus_x, mcus_y + 1)[0] == 4                                # more MCUs than the data holds
    assert gpu_decode(ok, geom, mcus_x, mcus_y - 1)[0] == 4                                # fewer
    assert gpu_decode(ok, geom, mcus_x, mcus_y, restart=50)[0] == 4                        # restart segments promised, none there
    assert gpu_decode(data[:100] + b"\xff\xd0" + data[100:] + EOI, geom, mcus_x, mcus_y)[0] == 4  # a restart marker nobody announced
    assert gpu_decode(data[:100] + b"\xff\xff" + data[100:] + EOI, geom, mcus_x, mcus_y)[0] == 4  # fill bytes: libjpeg's business
    with_rst = py_encode_restart(planes, geom, mcus_x, mcus_y, derive(DC_BITS, DC_VALS), derive(AC_BITS, AC_VALS), 50)
    assert gpu_decode(with_rst + EOI, geom, mcus_x, mcus_y, restart=50)[0] == 0
    assert gpu_decode(with_rst.replace(b"\xff\xd1", b"\xff\xd2") + EOI, geom, mcus_x, mcus_y, restart=50)[0] == 4  # out of sequence
    bad = spec([0, 5] + [0] * 14, [0, 1, 2, 3, 4])                                          # five codes of length 2 are not a prefix code
    assert gpu_decode(ok, geom, mcus_x, mcus_y, tables=(bad, spec(AC_BITS, AC_VALS)))[0] == 4
    noise = bytes(b if b != 0xFF else 0xFE for b in rng.integers(0, 256, len(data), dtype=np.uint8))
    assert gpu_decode(noise + EOI, geom, mcus_x, mcus_y)[0] == 4


def test_decode_and_encode_on_planes_that_stay_on_the_device():
    """MJX_HUFF_PLANES_ON_DEVICE on both entropy coders: scan data -> device planes -> scan data, and
    the planes read back equal the ones the stream was made from (the batch pipeline's data path)"""
    import torch  # noqa: F401
    from libmodjpeg_amd._lib import lib
    from tests.test_gpu_huffman import HuffComponent, HuffScan
    L = lib()
    geom, mcus_x, mcus_y = [(2, 2), (1, 1), (1, 1)], 60, 34
    rng = np.random.default_rng(11)
    planes = random_planes(rng, geom, mcus_x, mcus_y, 0.1, 500)
    data = py_encode_restart(planes, geom, mcus_x, mcus_y, derive(DC_BITS, DC_VALS), derive(AC_BITS, AC_VALS), 0)
    dev = []
    for p in planes:
        d = ctypes.c_void_p()
        assert L.mjx_malloc(ctypes.byref(d), p.nbytes) == 0
        dev.append(d)
    sc = DScan()
    sc.ncomp, sc.mcus_x, sc.mcus_y, sc.flags = 3, mcus_x, mcus_y, 1
    for c, (hs, vs) in enumerate(geom):
        sc.comp[c] = DComponent(dev[c].value, planes[c].shape[1], hs, vs, 0, 0)
    dc, ac = spec(DC_BITS, DC_VALS), spec(AC_BITS, AC_VALS)
    sc.dc[0], sc.ac[0] = ctypes.pointer(dc), ctypes.pointer(ac)
    buf = np.frombuffer(data + EOI, np.uint8).copy()
    sc.data, sc.nbytes = buf.ctypes.data, buf.size
    assert L.mjx_huff_decode_host(ctypes.byref(sc)) == 0
    for c, p in enumerate(planes):
        back = np.empty_like(p)
        assert L.mjx_memcpy_d2h(back.ctypes.data, dev[c], p.nbytes, None) == 0
        assert np.array_equal(back, p), c
    es = HuffScan()
    es.ncomp, es.mcus_x, es.mcus_y, es.max_coef_bits, es.flags = 3, mcus_x, mcus_y, 10, 1
    for c, (hs, vs) in enumerate(geom):
        es.comp[c] = HuffComponent(dev[c].value, planes[c].shape[1], planes[c].shape[1], planes[c].shape[0], hs, vs, 0, 0)
    edc, eac = derive(DC_BITS, DC_VALS), derive(AC_BITS, AC_VALS)
    es.dc[0], es.ac[0] = ctypes.pointer(edc), ctypes.pointer(eac)
    out, n = ctypes.c_void_p(), ctypes.c_size_t()
    assert L.mjx_huff_encode_host(ctypes.byref(es), ctypes.byref(out), ctypes.byref(n)) == 0
    assert ctypes.string_at(out, n.value) == data
    ctypes.CDLL(None).free(out)
    for d in dev:
        L.mjx_free(d)


def test_optimised_tables_round_trip_through_the_python_mirror():
    """entropy.encode_scan(optimize=True) builds tables for the image on the device pass + host table
    generation; the decoder, given those tables, returns the planes; the stream is not longer than
    the one coded with the Annex K tables"""
    from libmodjpeg_amd import entropy as E
    geom, mcus_x, mcus_y = [(2, 2), (1, 1), (1, 1)], 50, 30
    rng = np.random.default_rng(21)
    planes = random_planes(rng, geom, mcus_x, mcus_y, 0.12, 300)
    true_dims = [(p.shape[1], p.shape[0]) for p in planes]
    plain = E.encode_scan(planes, true_dims, geom, mcus_x, mcus_y)
    data, dc_t, ac_t = E.encode_scan(planes, true_dims, geom, mcus_x, mcus_y, optimize=True)
    assert len(data) <= len(plain) and len(dc_t) == 2 and len(ac_t) == 2
    back = [np.full_like(p, 0x5555) for p in planes]
    E.decode_scan(data + E.EOI, back, geom, mcus_x, mcus_y, dc_tables=dc_t, ac_tables=ac_t)
    for c in range(3):
        assert np.array_equal(back[c], planes[c]), c
