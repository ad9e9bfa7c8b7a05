"""ctypes mirror of the baseline Huffman entropy coders of include/mjx.h (mjx_huff_encode_host /
mjx_huff_decode_host): what libjpeg's jpeg_write_coefficients / jpeg_read_coefficients do for the
scan data of a sequential JPEG (reference call sites image.c:194 and :79), on coefficient planes
that live on the host (numpy) or stay on the device (torch tensors, MJX_HUFF_PLANES_ON_DEVICE).

No Python implementation of either coder exists here: the calls go to libmjx_hip.so or fail.
"""
import ctypes

import numpy as np

from ._lib import caller_stream, check, lib

PLANES_ON_DEVICE = 1  # MJX_HUFF_PLANES_ON_DEVICE
OPTIMIZE = 2          # MJX_HUFF_OPTIMIZE
EOI = b"\xff\xd9"

# ITU-T T.81 Annex K.3 (the tables libjpeg installs by default): (bits[16], values)
STD_DC_LUMA = ([0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0], list(range(12)))
STD_DC_CHROMA = ([0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0], list(range(12)))
STD_AC_LUMA = ([0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d], [
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08,
    0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28,
    0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89,
    0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6,
    0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa])
STD_AC_CHROMA = ([0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77], [
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91,
    0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26,
    0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
    0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87,
    0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4,
    0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa])


class HuffTable(ctypes.Structure):
    """struct mjx_huff_table: derived (code, size) per symbol, for the encoder"""
    _fields_ = [("code", ctypes.c_uint32 * 256), ("size", ctypes.c_uint8 * 256)]


class HuffComponent(ctypes.Structure):
    """struct mjx_huff_component"""
    _fields_ = [("plane", ctypes.c_void_p), ("blocks_per_row", ctypes.c_int32), ("width_in_blocks", ctypes.c_int32), ("height_in_blocks", ctypes.c_int32),
                ("h_samp", ctypes.c_int32), ("v_samp", ctypes.c_int32), ("dc_tbl", ctypes.c_int32), ("ac_tbl", ctypes.c_int32),
                ("dc_fx_n", ctypes.c_int32), ("dc_fx_value", ctypes.c_int32 * 4), ("dc_fx_q0", ctypes.c_uint16), ("dc_fx_reserved", ctypes.c_uint16)]


class HuffSpec(ctypes.Structure):
    """struct mjx_huff_spec: a DHT table as the file defines it (decoder input, optimised-table output)"""
    _fields_ = [("bits", ctypes.c_uint8 * 16), ("vals", ctypes.c_uint8 * 256)]


class HuffScan(ctypes.Structure):
    """struct mjx_huff_scan"""
    _fields_ = [("ncomp", ctypes.c_int32), ("mcus_x", ctypes.c_int32), ("mcus_y", ctypes.c_int32), ("comp", HuffComponent * 4),
                ("dc", ctypes.POINTER(HuffTable) * 4), ("ac", ctypes.POINTER(HuffTable) * 4), ("max_coef_bits", ctypes.c_int32), ("flags", ctypes.c_int32),
                ("optimal_dc", ctypes.POINTER(HuffSpec) * 4), ("optimal_ac", ctypes.POINTER(HuffSpec) * 4), ("stream", ctypes.c_void_p)]


class HuffDComponent(ctypes.Structure):
    """struct mjx_huff_dcomponent"""
    _fields_ = [("plane", ctypes.c_void_p), ("blocks_per_row", ctypes.c_int32), ("h_samp", ctypes.c_int32), ("v_samp", ctypes.c_int32),
                ("dc_tbl", ctypes.c_int32), ("ac_tbl", ctypes.c_int32)]


class HuffDScan(ctypes.Structure):
    """struct mjx_huff_dscan"""
    _fields_ = [("ncomp", ctypes.c_int32), ("mcus_x", ctypes.c_int32), ("mcus_y", ctypes.c_int32), ("comp", HuffDComponent * 4),
                ("dc", ctypes.POINTER(HuffSpec) * 4), ("ac", ctypes.POINTER(HuffSpec) * 4), ("data", ctypes.c_void_p), ("nbytes", ctypes.c_size_t),
                ("restart_interval", ctypes.c_int32), ("flags", ctypes.c_int32), ("stream", ctypes.c_void_p)]


def spec(table):
    bits, vals = table
    s = HuffSpec()
    for i, b in enumerate(bits):
        s.bits[i] = b
    for i, v in enumerate(vals):
        s.vals[i] = v
    return s


def derived(table):
    """(code, size) of every symbol: JPEG Annex C canonical code assignment"""
    bits, vals = table
    t, code, k = HuffTable(), 0, 0
    for length in range(1, 17):
        for _ in range(bits[length - 1]):
            t.code[vals[k]], t.size[vals[k]] = code, length
            code += 1
            k += 1
        code <<= 1
    return t


def _ptr(plane):
    """device pointer of a torch tensor / host pointer of a numpy array, and whether it is on the device"""
    if isinstance(plane, np.ndarray):
        assert plane.dtype == np.int16 and plane.flags["C_CONTIGUOUS"]
        return plane.ctypes.data, False
    assert plane.is_contiguous() and plane.element_size() == 2
    return plane.data_ptr(), plane.is_cuda


def encode_scan(planes, true_dims, samp, mcus_x, mcus_y, dc_tables=(STD_DC_LUMA, STD_DC_CHROMA), ac_tables=(STD_AC_LUMA, STD_AC_CHROMA), optimize=False):
    """scan data (stuffed, without EOI) of one interleaved scan.  planes[c]: int16 [rows, blocks_per_row, 64];
    true_dims[c] = (width_in_blocks, height_in_blocks); samp[c] = (h, v); component 0 uses table 0, the others table 1.
    optimize=True: tables built for this image as libjpeg's optimize_coding does; returns (data, dc_tables, ac_tables)"""
    sc = HuffScan()
    sc.ncomp, sc.mcus_x, sc.mcus_y, sc.max_coef_bits = len(planes), mcus_x, mcus_y, 10
    on_dev = []
    for c, p in enumerate(planes):
        ptr, dev = _ptr(p)
        on_dev.append(dev)
        t = 0 if c == 0 else min(1, len(dc_tables) - 1)
        sc.comp[c] = HuffComponent(ptr, p.shape[1], true_dims[c][0], true_dims[c][1], samp[c][0], samp[c][1], t, t)
    assert all(on_dev) or not any(on_dev), "planes must all be on the host or all on the device"
    sc.flags = (PLANES_ON_DEVICE if on_dev[0] else 0) | (OPTIMIZE if optimize else 0)
    if on_dev[0]:
        sc.stream = caller_stream()  # the encoder reads the planes after whatever torch has queued on them (mjx.h, ordering contract)
    keep = [derived(t) for t in dc_tables], [derived(t) for t in ac_tables]
    for i, t in enumerate(keep[0]):
        sc.dc[i] = ctypes.pointer(t)
    for i, t in enumerate(keep[1]):
        sc.ac[i] = ctypes.pointer(t)
    built = [HuffSpec() for _ in range(4)], [HuffSpec() for _ in range(4)]
    for i in range(4):
        sc.optimal_dc[i], sc.optimal_ac[i] = ctypes.pointer(built[0][i]), ctypes.pointer(built[1][i])
    out, n = ctypes.c_void_p(), ctypes.c_size_t()
    check(lib().mjx_huff_encode_host(ctypes.byref(sc), ctypes.byref(out), ctypes.byref(n)), "mjx_huff_encode_host")
    data = ctypes.string_at(out, n.value)
    ctypes.CDLL(None).free(out)
    if not optimize:
        return data
    ntab = 1 if len(planes) == 1 else 2

    def as_table(sp):
        bits = list(sp.bits)
        return bits, list(sp.vals)[:sum(bits)]
    return data, [as_table(built[0][i]) for i in range(ntab)], [as_table(built[1][i]) for i in range(ntab)]


def decode_scan(data, planes, samp, mcus_x, mcus_y, restart_interval=0, dc_tables=(STD_DC_LUMA, STD_DC_CHROMA), ac_tables=(STD_AC_LUMA, STD_AC_CHROMA)):
    """fills planes[c] (int16 [mcus_y*v, blocks_per_row, 64], host numpy or device torch) from scan data
    as it stands in a file (stuffing, RSTn markers, terminated by EOI)"""
    sc = HuffDScan()
    sc.ncomp, sc.mcus_x, sc.mcus_y, sc.restart_interval = len(planes), mcus_x, mcus_y, restart_interval
    on_dev = []
    for c, p in enumerate(planes):
        ptr, dev = _ptr(p)
        on_dev.append(dev)
        assert p.shape[0] == mcus_y * samp[c][1]
        t = 0 if c == 0 else min(1, len(dc_tables) - 1)
        sc.comp[c] = HuffDComponent(ptr, p.shape[1], samp[c][0], samp[c][1], t, t)
    assert all(on_dev) or not any(on_dev)
    sc.flags = PLANES_ON_DEVICE if on_dev[0] else 0
    if on_dev[0]:
        sc.stream = caller_stream()  # the decoder zeroes and writes the planes after whatever torch has queued on them
    keep = [spec(t) for t in dc_tables], [spec(t) for t in ac_tables]
    for i, t in enumerate(keep[0]):
        sc.dc[i] = ctypes.pointer(t)
    for i, t in enumerate(keep[1]):
        sc.ac[i] = ctypes.pointer(t)
    buf = np.frombuffer(data, np.uint8)
    sc.data, sc.nbytes = buf.ctypes.data, buf.size
    check(lib().mjx_huff_decode_host(ctypes.byref(sc)), "mjx_huff_decode_host")
