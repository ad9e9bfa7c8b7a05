"""The tables the entropy decoder's wave-per-chunk rounds look symbols up in (mjx_huff_wave_tables, host code only) against a
restatement in Python of what T.81's canonical codes say about every 16-bit prefix: field layout, EOB, long codes through the
second level, prefixes that are not a code (reference call site of the decoder: image.c:79 -> libjpeg jdhuff.c)."""
import ctypes

import numpy as np
import pytest

from tests.test_gpu_huffdec import (AC_C_BITS, AC_C_VALS, AC_REV_BITS, AC_REV_VALS, DC_C_BITS, DC_C_VALS, DC_FLAT_BITS, DC_FLAT_VALS, DC_PAIR_BITS, DC_PAIR_VALS, DComponent, DScan,
                                    spec)
from tests.test_gpu_huffman import AC_BITS, AC_VALS, DC_BITS, DC_VALS


def canonical(bits, vals):
    """[(code, length, symbol)] of a DHT (T.81 Annex C)"""
    out, code, p = [], 0, 0
    for length in range(1, 17):
        for _ in range(bits[length - 1]):
            out.append((code, length, vals[p]))
            code += 1
            p += 1
        code <<= 1
    return out


def symbol_at(codes, prefix16):
    """the code that the 16 bits start with, or None"""
    for code, length, sym in codes:
        if prefix16 >> (16 - length) == code:
            return length, sym
    return None


def expected_field(codes, prefix16, is_dc, shift):
    hit = symbol_at(codes, prefix16)
    if hit is None:
        return 0 if is_dc else (63 << 16) << shift
    length, sym = hit
    if is_dc:
        return (length + min(sym, 15)) << shift
    s, r = sym & 15, sym >> 4
    advance = r + 1 if s else (16 if r == 15 else 63)  # EOB (and any other run with size 0) ends the block: 63 in this table
    return ((length + s) | (advance << 16)) << shift


DCS = [(DC_BITS, DC_VALS), (DC_C_BITS, DC_C_VALS), (DC_FLAT_BITS, DC_FLAT_VALS), (DC_PAIR_BITS, DC_PAIR_VALS)]
ACS = [(AC_BITS, AC_VALS), (AC_C_BITS, AC_C_VALS), (AC_REV_BITS, AC_REV_VALS)]


def wave_tables(geom, dc_ids, ac_ids):
    from libmodjpeg_amd._lib import lib
    sc = DScan()
    sc.ncomp, sc.mcus_x, sc.mcus_y = len(geom), 1, 1
    keep = [[spec(*t) for t in DCS], [spec(*t) for t in ACS]]
    for i, t in enumerate(keep[0]):
        sc.dc[i] = ctypes.pointer(t)
    for i, t in enumerate(keep[1]):
        sc.ac[i] = ctypes.pointer(t)
    for c, (hs, vs) in enumerate(geom):
        sc.comp[c] = DComponent(0, hs, hs, vs, dc_ids[c], ac_ids[c])
    out = np.zeros(6 * 1024 + 64 * 64, np.uint32)
    info = (ctypes.c_int32 * 4)()
    rc = lib().mjx_huff_wave_tables(ctypes.byref(sc), out.ctypes.data, out.size, info)
    return rc, out, list(info)


@pytest.mark.parametrize("geom,dc_ids,ac_ids", [([(2, 2), (1, 1), (1, 1)], (0, 1, 1), (0, 1, 1)), ([(1, 1)], (0,), (0,)), ([(2, 1), (1, 1), (1, 1), (1, 1)], (0, 1, 2, 3), (1, 0, 1, 0)),
                                                ([(1, 1), (1, 1)], (3, 3), (2, 0))])
def test_every_prefix_of_every_table(geom, dc_ids, ac_ids):
    rc, out, (nac, ndc, nsub, sel) = wave_tables(geom, dc_ids, ac_ids)
    assert rc == 0
    # staged order: the order the components first name the tables; field index = order among the tables of its kind
    dc_order, ac_order = list(dict.fromkeys(dc_ids)), list(dict.fromkeys(ac_ids))
    assert (nac, ndc) == (len(ac_order), len(dc_order))
    slots = [c for c, (hs, vs) in enumerate(geom) for _ in range(hs * vs)]
    for b, c in enumerate(slots):
        assert (sel >> (3 * b)) & 7 == dc_order.index(dc_ids[c]) | (ac_order.index(ac_ids[c]) << 2), (b, c)
    first, second = out[:6 * 1024], out[6 * 1024:]
    links = 0
    for kind, order, tables, base in (("ac", ac_order, ACS, 0), ("dc", dc_order, DCS, 2)):
        for field, tid in enumerate(order):
            codes = canonical(*tables[tid])
            level1 = first[(base + field) * 1024:(base + field + 1) * 1024]
            for i in range(1024):
                e = int(level1[i])
                if e & 0x80000000:  # codes longer than ten bits share this prefix
                    links += 1
                    sub = second[(e & 0xFFFF):(e & 0xFFFF) + 64]
                    assert (e & 0xFFFF) % 64 == 0 and (e & 0xFFFF) // 64 < nsub
                    for j in range(64):
                        assert int(sub[j]) == expected_field(codes, (i << 6) | j, kind == "dc", 8 * field), (kind, tid, i, j)
                else:
                    # a code of at most ten bits, or no code at all, whatever the six bits that follow
                    assert e == expected_field(codes, i << 6, kind == "dc", 8 * field) == expected_field(codes, (i << 6) | 63, kind == "dc", 8 * field), (kind, tid, i)
    assert links > 0 or all(sum(ACS[t][0][10:]) == 0 for t in ac_order)
    # the unused first-level tables stay zero
    for t in range(nac, 2):
        assert not first[t * 1024:(t + 1) * 1024].any()
    for t in range(ndc, 4):
        assert not first[(2 + t) * 1024:(3 + t) * 1024].any()


def test_more_tables_than_the_walk_has_fields_for_are_refused():
    rc, _, _ = wave_tables([(1, 1), (1, 1), (1, 1)], (0, 0, 0), (0, 1, 2))
    assert rc != 0
    from libmodjpeg_amd._lib import lib
    assert lib().mjx_huff_wave_tables(None, None, 0, None) != 0
