"""The entropy decoder's host step (mjx_huff_unstuff, pure host code): scan data as in the file -> bytes
without 0xFF00 stuffing and RSTn markers + restart segment offsets.  The AVX2 form and the portable form
against a Python restatement of the rule (ITU-T T.81 B.1.1.5 / F.1.2.3: a 0xFF in entropy-coded data is
followed by 0x00; RSTn in sequence between segments; EOI ends the scan), on streams that put 0xFF at every
position relative to the 32-byte vectors, back to back, last in the buffer, and on everything the decoder
must refuse (libjpeg then owns the file: host/gpudecode.c)."""
import ctypes

import numpy as np
import pytest


@pytest.fixture(scope="module")
def L():
    import __graft_entry__ as g
    g.build_hip()
    from libmodjpeg_amd._lib import lib
    return lib()


def unstuff_py(data, max_segs):
    out, segs, i, expect = bytearray(), [0], 0, 0
    n = len(data)
    while True:
        if i >= n:
            return None
        if data[i] != 0xFF:
            out.append(data[i])
            i += 1
            continue
        if i + 1 >= n:
            return None
        m = data[i + 1]
        if m == 0x00:
            out.append(0xFF)
        elif 0xD0 <= m <= 0xD7:
            if m != 0xD0 + expect or len(segs) >= max_segs:
                return None
            expect = (expect + 1) & 7
            segs.append(len(out))
        elif m == 0xD9:
            return bytes(out), segs
        else:
            return None
        i += 2


def call(L, data, max_segs, form):
    src = np.frombuffer(bytes(data), np.uint8).copy()
    out = np.zeros(len(data) + 64, np.uint8)
    segs = np.zeros(max_segs + 1, np.uint64)
    n, ns = ctypes.c_size_t(), ctypes.c_int64()
    rc = L.mjx_huff_unstuff(src.ctypes.data, len(data), out.ctypes.data, ctypes.byref(n), segs.ctypes.data, max_segs, ctypes.byref(ns), form)
    if rc != 0:
        return rc, None
    return 0, (bytes(out[:n.value]), [int(v) for v in segs[:ns.value]])


def forms(L):
    return [0, 1] + [f for f in (2, 3) if call(L, b"\x00\xff\xd9", 1, f)[0] != 1]  # 2: AVX2, 3: AVX-512 byte compress, where the CPU has them


def test_random_streams_with_markers_everywhere(L):
    rng = np.random.default_rng(3)
    for trial in range(300):
        n = int(rng.integers(0, 400))
        body = bytearray()
        rst = 0
        for _ in range(n):
            r = rng.random()
            if r < 0.08:
                body += b"\xff\x00"                      # a stuffed 0xFF
            elif r < 0.10 and trial % 3 == 0:
                body += bytes([0xFF, 0xD0 + rst])          # RSTn in sequence
                rst = (rst + 1) & 7
            else:
                body.append(int(rng.integers(0, 255)))     # anything but 0xFF
        if trial % 5 == 0:
            body = b"\xff\x00" * int(rng.integers(1, 40)) + body  # 0xFF at every offset of the first vectors, back to back
        data = bytes(body) + b"\xff\xd9" + bytes(rng.integers(0, 256, int(rng.integers(0, 40)), dtype=np.uint8))  # trailing bytes after EOI are not looked at
        want = unstuff_py(data, 64)
        for form in forms(L):
            rc, got = call(L, data, 64, form)
            if want is None:
                assert rc == 4, (trial, form)
            else:
                assert rc == 0 and got == want, (trial, form, len(data))


@pytest.mark.parametrize("bad", [b"", b"\x12\x34", b"\x12\xff", b"\x12\xff\xd1\xff\xd9", b"\xff\xff\xd9", b"\x01\xff\xc4\xff\xd9", b"\xff\xd0\xff\xd0\xff\xd9",
                                 b"\x00" * 70, b"\x00" * 31 + b"\xff", b"\x00" * 40 + b"\xff\xe0" + b"\x00" * 40 + b"\xff\xd9"])
def test_everything_else_is_refused(L, bad):
    assert unstuff_py(bad, 8) is None
    for form in forms(L):
        assert call(L, bad, 8, form)[0] == 4, (bad, form)


def test_segment_limit_and_long_runs(L):
    data = (b"\x11" * 1000 + b"\xff\xd0") + (b"\x22" * 999 + b"\xff\x00" + b"\xff\xd1") + b"\x33" * 2000 + b"\xff\xd9"
    for form in forms(L):
        rc, got = call(L, data, 3, form)
        assert rc == 0 and got == unstuff_py(data, 3) and got[1] == [0, 1000, 2000]
        assert call(L, data, 2, form)[0] == 4  # more restart segments than the frame header promises


def test_long_streams_with_0xff_at_every_offset_of_the_vector_width(L):
    """0xFF 0x00 pairs and markers placed at every offset modulo 64 (the widest form's block), runs of them back to back,
    a 0xFF in the last byte of a block, stuffed zeros in the first: every form gives what the portable one gives"""
    rng = np.random.default_rng(11)
    for trial in range(40):
        n = int(rng.integers(200, 5000))
        raw = rng.integers(0, 255, n, dtype=np.uint8)  # no 0xFF of its own
        body = bytearray()
        rst = 0
        where = set(int(v) for v in rng.integers(0, n, n // int(rng.integers(3, 60))))
        where |= {o for o in range(trial % 64, n, 64)} | {o for o in range((trial + 63) % 64, n, 64 * 3)}
        for i in range(n):
            if i in where:
                kind = int(rng.integers(0, 20))
                if kind == 0:
                    body += bytes([0xFF, 0xD0 + rst])
                    rst = (rst + 1) & 7
                else:
                    body += b"\xff\x00" * (1 if kind < 15 else int(rng.integers(2, 40)))
            body.append(int(raw[i]))
        data = bytes(body) + b"\xff\xd9" + bytes(rng.integers(0, 256, int(rng.integers(0, 100)), dtype=np.uint8))
        want = call(L, data, 4096, 1)
        assert want[0] == 0 and want[1] == unstuff_py(data, 4096)
        for form in forms(L):
            assert call(L, data, 4096, form) == want, (trial, form, n)
