"""ctypes binding of the C ABI in include/mjx.h (libmjx_hip.so, hand-written gfx950 kernels).

There is no Python or CPU implementation of the hot path in this package: if the shared library is
missing, or no HIP device is usable, every call fails loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MJX_HIP_LIB") or os.path.join(_HERE, "libmjx_hip.so")  # override: kernel A/B experiments

MJX_OK = 0
MJX_ERR_NO_DEVICE = 1
MJX_ERR_MEMORY = 2
MJX_ERR_NULL_DATA = 3
MJX_ERR_BAD_ARG = 4
MJX_ERR_LAUNCH = 5
MJX_PLAN_GROUP_BY_DENSITY = 1
MJX_PLAN_ALPHA_IS_FINAL = 2
MJX_PLAN_TILES_I16 = 4
MJX_PLAN_SERIAL = 8         # the two kernels of a step always one after the other
MJX_PLAN_SIDE_BY_SIDE = 16  # ... always side by side (the default decides by the size of the launch)
MJX_PLAN_TILES_ARE_FINAL = 64  # float tiles keep their contents too: opaque blocks are written from the tile alone
MJX_PLAN_NO_LOW_COLUMN_BUILD = 32  # smooth runs take the general build of the blend kernel like the others
MJX_VERSION = 105  # include/mjx.h; compared with mjx_version() of the library that is loaded
MJX_JOB_TILES_I16 = 1
MJX_RECT_TILES_ON_DEVICE = 1
MJX_RECT_ROWS_ON_DEVICE = 2
MJX_RECT_TILES_I16 = 4
MJX_PREP_TILES_I16 = 1


class MjxError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        super().__init__(f"{where}: mjx error {code} ({error_string(code)})")


class RectJob(ctypes.Structure):
    """struct mjx_rect_job (include/mjx.h)"""
    _fields_ = [("coef", ctypes.c_void_p), ("tile", ctypes.c_void_p), ("alpha", ctypes.c_void_p),
                ("blocks_per_row", ctypes.c_int32), ("plane_rows", ctypes.c_int32),
                ("row0", ctypes.c_int32), ("col0", ctypes.c_int32), ("rows", ctypes.c_int32), ("cols", ctypes.c_int32),
                ("qtable", ctypes.c_int32), ("flags", ctypes.c_int32)]


class HostRect(ctypes.Structure):
    """struct mjx_host_rect (include/mjx.h)"""
    _fields_ = [("rows_ptr", ctypes.c_void_p), ("tile", ctypes.c_void_p), ("alpha", ctypes.c_void_p), ("q", ctypes.c_void_p),
                ("blocks_per_row", ctypes.c_int32), ("col0", ctypes.c_int32), ("rows", ctypes.c_int32), ("cols", ctypes.c_int32),
                ("flags", ctypes.c_int32), ("reserved", ctypes.c_int32), ("schedule", ctypes.c_void_p), ("stream", ctypes.c_void_p)]


class PlanStats(ctypes.Structure):
    """struct mjx_plan_stats (include/mjx.h)"""
    _fields_ = [("blocks", ctypes.c_int64), ("octets", ctypes.c_int64), ("gather_octets", ctypes.c_int64), ("zero_alpha_octets", ctypes.c_int64),
                ("runs", ctypes.c_int64), ("block_terms", ctypes.c_int64), ("union_terms", ctypes.c_int64), ("dc_only_octets", ctypes.c_int64), ("union_terms_consecutive", ctypes.c_int64),
                ("resident_waves", ctypes.c_int32), ("run_octets_max", ctypes.c_int32), ("run_octets_min", ctypes.c_int32), ("runs_beside", ctypes.c_int32),
                ("low_column_octets", ctypes.c_int64), ("runs_low_columns", ctypes.c_int64), ("low_row_octets", ctypes.c_int64),
                ("build_us", ctypes.c_int32 * 5), ("reserved", ctypes.c_int32), ("opaque_octets", ctypes.c_int64)]

    def as_dict(self):
        d = {k: int(getattr(self, k)) for k, _ in self._fields_ if k not in ("reserved", "build_us")}
        d["build_ms"] = dict(zip(("scan", "download", "host_schedule", "upload", "total"), (u / 1e3 for u in self.build_us)))
        real = d["octets"] - d["zero_alpha_octets"]  # (DC-only octets count with a union of one term)
        d["mean_union_terms_per_octet"] = d["union_terms"] / real if real and d["union_terms"] else None
        d["mean_terms_per_block"] = d["block_terms"] / d["blocks"] if d["blocks"] and d["block_terms"] else None
        d["mean_union_terms_per_octet_consecutive"] = d["union_terms_consecutive"] / ((d["blocks"] + 7) // 8) if d["union_terms_consecutive"] else None
        return d


class PlanOptions(ctypes.Structure):
    """struct mjx_plan_options (include/mjx.h): zero = the library's choice"""
    _fields_ = [("flags", ctypes.c_int32), ("run_octets", ctypes.c_int32), ("stream_run_octets", ctypes.c_int32), ("tail_min_octets", ctypes.c_int32),
                ("beside_share", ctypes.c_float), ("reserved", ctypes.c_int32 * 11)]


class DcJob(ctypes.Structure):
    """struct mjx_dc_job (include/mjx.h)"""
    _fields_ = [("coef", ctypes.c_void_p), ("blocks_per_row", ctypes.c_int32), ("rows", ctypes.c_int32), ("cols", ctypes.c_int32),
                ("value", ctypes.c_int32), ("q0", ctypes.c_uint16), ("reserved", ctypes.c_uint16)]


class TilePrep(ctypes.Structure):
    """struct mjx_tileprep (include/mjx.h)"""
    _fields_ = [("image", ctypes.c_void_p), ("alpha", ctypes.c_void_p), ("width", ctypes.c_int32), ("height", ctypes.c_int32),
                ("crop_x", ctypes.c_int32), ("crop_y", ctypes.c_int32), ("crop_w", ctypes.c_int32), ("crop_h", ctypes.c_int32),
                ("off_x", ctypes.c_int32), ("off_y", ctypes.c_int32), ("source", ctypes.c_int32), ("target", ctypes.c_int32), ("ncomp", ctypes.c_int32),
                ("h_samp", ctypes.c_int32 * 4), ("v_samp", ctypes.c_int32 * 4), ("tile", ctypes.c_void_p * 4), ("alpha_tile", ctypes.c_void_p * 4),
                ("flags", ctypes.c_int32), ("reserved", ctypes.c_int32), ("stream", ctypes.c_void_p)]


MJX_STREAM_IDLE = ctypes.c_void_p(-1).value  # "nothing is pending on the device memory handed over" (ordering contract of mjx.h)


def caller_stream():
    """the stream device tensors of the calling thread were last written on (torch's current stream; 0 = the legacy
    default stream): what the synchronous entry points that take device memory are ordered after"""
    import torch
    return torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else 0


MJX_PREP_SOURCE_RGB, MJX_PREP_SOURCE_YCC, MJX_PREP_TARGET_RGB, MJX_PREP_TARGET_YCC = 1, 5, 2, 3

# every symbol include/mjx.h declares: (name, restype, argtypes)
_vp, _i, _sz, _u64, _u32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_uint32
SYMBOLS = [
    ("mjx_version", _i, []),
    ("mjx_device_count", _i, []),
    ("mjx_set_device", _i, [_i]),
    ("mjx_get_device", _i, [ctypes.POINTER(_i)]),
    ("mjx_error_string", ctypes.c_char_p, [_i]),
    ("mjx_malloc", _i, [ctypes.POINTER(_vp), _sz]),
    ("mjx_free", _i, [_vp]),
    ("mjx_memcpy_h2d", _i, [_vp, _vp, _sz, _vp]),
    ("mjx_memcpy_d2h", _i, [_vp, _vp, _sz, _vp]),
    ("mjx_stream_synchronize", _i, [_vp]),
    ("mjx_compose_plan_create", _i, [ctypes.POINTER(_vp), ctypes.POINTER(RectJob), _i, _vp, _i]),
    ("mjx_compose_plan_create_ex", _i, [ctypes.POINTER(_vp), ctypes.POINTER(RectJob), _i, _vp, _i, _i]),
    ("mjx_compose_plan_create_opts", _i, [ctypes.POINTER(_vp), ctypes.POINTER(RectJob), _i, _vp, _i, ctypes.POINTER(PlanOptions)]),
    ("mjx_compose_plan_gather_octets", ctypes.c_int64, [_vp]),
    ("mjx_compose_plan_stats", _i, [_vp, ctypes.POINTER(PlanStats)]),
    ("mjx_tile_schedule_create", _i, [ctypes.POINTER(_vp), _vp, _i, _i, _i, _vp]),
    ("mjx_tile_schedule_destroy", None, [_vp]),
    ("mjx_tile_schedule_gather_octets", ctypes.c_int64, [_vp]),
    ("mjx_tile_schedule_stats", _i, [_vp, ctypes.POINTER(PlanStats)]),
    ("mjx_compose_plan_destroy", None, [_vp]),
    ("mjx_compose_plan_blocks", ctypes.c_int64, [_vp]),
    ("mjx_compose_run", _i, [_vp, _vp]),
    ("mjx_compose_host", _i, [ctypes.POINTER(HostRect), _i]),
    ("mjx_release_thread_cache", _i, []),
    ("mjx_malloc_host", _i, [ctypes.POINTER(_vp), _sz]),
    ("mjx_free_host", _i, [_vp]),
    ("mjx_compose_rect_host", _i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    ("mjx_effect_dc_host", _i, [_vp, _i, _i, _i, ctypes.c_uint16, _i]),
    ("mjx_effect_zero_host", _i, [_vp, _i, _i, _i, _i]),
    ("mjx_tiles_prepare", _i, [ctypes.POINTER(TilePrep)]),
    ("mjx_convolve_host", _i, [_vp, _vp, ctypes.c_float, _i, _i, _i]),
    ("mjx_convolve", _i, [_vp, _vp, ctypes.c_float, _i, _i, _i, _vp]),
    ("mjx_effect_dc", _i, [ctypes.POINTER(DcJob), _i, _vp]),
    ("mjx_dc_plan_create", _i, [ctypes.POINTER(_vp), ctypes.POINTER(DcJob), _i]),
    ("mjx_dc_plan_destroy", None, [_vp]),
    ("mjx_dc_plan_run", _i, [_vp, _vp]),
    ("mjx_effect_zero", _i, [_vp, _i, _i, _i, _i, _vp]),
    ("mjx_effect_dc_device", _i, [_vp, _i, _i, _i, ctypes.c_uint16, _i, _vp]),
    ("mjx_effect_zero_device", _i, [_vp, _i, _i, _i, _i, _vp]),
    ("mjx_huff_encode_host", _i, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_sz)]),
    ("mjx_huff_decode_host", _i, [_vp]),
    ("mjx_huff_unstuff", _i, [_vp, _sz, _vp, ctypes.POINTER(_sz), _vp, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64), _i]),
    ("mjx_huff_wave_tables", _i, [_vp, _vp, _sz, _vp]),
    ("mjx_huff_optimal_table", _i, [_vp, _vp]),
    ("mjx_synth_coef_batch", _i, [_vp, _sz, _u64, _u32, _u32, _u64, _sz, _vp, _vp]),
    ("mjx_synth_tile_batch", _i, [_vp, _sz, _u64, _u32, _u32, _u64, _sz, _vp]),
    ("mjx_synth_alpha_batch", _i, [_vp, _sz, _u64, _u32, _u32, _u64, _sz, _i, _vp]),
    ("mjx_synth_tiles_i16_batch", _i, [_vp, _i, _sz, _u64, _u32, _u32, _u64, _sz, _i, _vp]),
    ("mjx_synth_coef", _i, [_vp, _u64, _u32, _u64, _sz, _vp, _vp]),
    ("mjx_synth_tile", _i, [_vp, _u64, _u32, _u64, _sz, _vp]),
    ("mjx_synth_alpha", _i, [_vp, _u64, _u32, _u64, _sz, _i, _vp]),
]

_lib = None


def lib():
    """The loaded libmjx_hip.so; raises if it has not been built (python -c 'import
    __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with __graft_entry__.build(); "
                               "libmodjpeg_amd has no CPU fallback for the hot path")
        L = ctypes.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            f = getattr(L, name)  # AttributeError if the ABI lost a symbol
            f.restype = res
            f.argtypes = args
        if L.mjx_version() != MJX_VERSION:  # structs grow between versions: a stale library would read garbage (MJX_HIP_LIB overrides)
            raise RuntimeError(f"{LIB_PATH} implements mjx.h version {L.mjx_version()}, this package was written for {MJX_VERSION}: rebuild it")
        _lib = L
    return _lib


def error_string(code):
    try:
        return lib().mjx_error_string(code).decode()
    except Exception:
        return "?"


def check(code, where):
    if code != MJX_OK:
        raise MjxError(code, where)
