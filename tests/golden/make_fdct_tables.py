#!/usr/bin/env python3
"""Generator of libmodjpeg_amd/csrc/mjx_fdct_tables.h: the forward DCTs of the linked libjpeg as exact
integer linear forms.

Tile preparation (reference dropon.c:325-576 via image.c:257-347) is arithmetic of an un-vendored
dependency: IJG libjpeg 9d (/opt/conda/lib/libjpeg.so.9 in this image).  At quality 100 a prepared
tile is   round_half_away(FDCT(samples) / 8)   where FDCT is libjpeg's integer "islow" transform:
jpeg_fdct_islow on 8x8 samples for a component at full resolution and, because IJG >= 7 folds 2:1
chroma downsampling into the transform ("DCT scaling", do_fancy_downsampling), jpeg_fdct_16x16 on
16x16 samples for a component subsampled 2x2, jpeg_fdct_16x8 / jpeg_fdct_8x16 for one subsampled
horizontally / vertically only (4:2:2 / 4:4:0).  All are two passes (rows, then columns) in which
every output is an exact 32-bit integer linear form of the pass's inputs followed by ONE descale
(arithmetic right shift with rounding constant) — the published structure of jfdctint.c (Loeffler,
Ligtenberg, Moschytz 1989, scaled fixed-point, CONST_BITS 13, PASS1_BITS 2).  A factorisation's
intermediate sums are exact, so each pass equals  out[k] = (sum_i M[k][i]*in[i] + B[k]) >> S[k]
with integer M, B, S.  This script MEASURES those integers on the library itself (its transform
functions are exported symbols): least squares over random inputs, rounded, then the two-pass
model is verified bit for bit on fresh random blocks before anything is written.  The header is a
fixture derived from the dependency, like a golden vector; tests/test_fdct_tables.py repeats the
verification against whatever libjpeg the host has and the GPU tile-preparation tests compare
whole tiles with the reference's mj_compile_dropon.

Usage: python tests/golden/make_fdct_tables.py [libjpeg.so] > libmodjpeg_amd/csrc/mjx_fdct_tables.h
"""
import ctypes
import sys

import numpy as np

# (exported symbol, samples per block row = columns, rows)
TRANSFORMS = (("jpeg_fdct_islow", 8, 8), ("jpeg_fdct_16x16", 16, 16), ("jpeg_fdct_16x8", 16, 8), ("jpeg_fdct_8x16", 8, 16))


def make_caller(lib, name, cols, rows):
    fn = getattr(lib, name)
    buf = np.zeros((rows, cols), np.uint8)
    rowptr = (ctypes.POINTER(ctypes.c_ubyte) * rows)(*[ctypes.cast(buf[i].ctypes.data, ctypes.POINTER(ctypes.c_ubyte)) for i in range(rows)])
    out = (ctypes.c_int * 64)()
    o = np.frombuffer(out, dtype=np.int32)

    def call(samples):
        buf[:] = samples
        fn(out, rowptr, ctypes.c_uint(0))
        return o.reshape(8, 8).astype(np.int64).copy()
    return call


def fit_rows(inputs, outputs, gain):
    """outputs[:, k] = gain * ((M[k] . inputs + B[k]) >> S[k]); returns integer M (8 x n), B, S"""
    k_samples, n = inputs.shape
    a = np.concatenate([inputs.astype(np.float64), np.ones((k_samples, 1))], axis=1)
    sol, *_ = np.linalg.lstsq(a, outputs.astype(np.float64) / gain, rcond=None)
    # standard error of a fitted coefficient: the floor's rounding noise (uniform, 1/sqrt(12)) over
    # the regressor's spread; an estimate is accepted as the integer M / 2^S only within 6 sigma
    sigma = (1.0 / np.sqrt(12.0)) / (inputs.std() * np.sqrt(k_samples))
    m_all, b_all, s_all = [], [], []
    for k in range(8):
        coef, const = sol[:n, k], sol[n, k]
        for s in range(0, 21):
            m = coef * (1 << s)
            tol = 6 * sigma * (1 << s)
            if tol > 0.25:
                raise SystemExit(f"output {k}: not enough samples to tell the integers apart at shift {s}")
            if np.abs(m - np.round(m)).max() < tol + 1e-9 and np.abs(np.round(m)).max() >= 1:
                break
        else:
            raise SystemExit(f"no integer scale for output {k}")
        m = np.round(m).astype(np.int64)
        # constant: exact forms have an integer constant; descaled ones round to nearest (B = 2^(S-1))
        b = int(round(const * (1 << s))) if s == 0 else (1 << (s - 1))
        m_all.append(m); b_all.append(b); s_all.append(s)
    return np.array(m_all), np.array(b_all, np.int64), np.array(s_all, np.int64)


def measure(lib, name, cols, rows, samples=160000, seed=0):
    call = make_caller(lib, name, cols, rows)
    rng = np.random.default_rng(seed)
    # pass 2's DC output of equal rows is an exact multiple of pass 1's output of that row; pass 1's
    # DC is (sum of the row - cols * 128) << PASS1_BITS (jfdctint.c, PASS1_BITS = 2)
    gain = call(np.full((rows, cols), 200))[0, 0] // (4 * (cols * 200 - cols * 128))
    x = rng.integers(0, 256, (samples, cols))
    d = np.empty((samples, 8), np.int64)
    for i in range(samples):
        d[i] = call(np.broadcast_to(x[i], (rows, cols)))[0]
    m1, b1, s1 = fit_rows(x, d, gain)
    # columns: every row constant -> pass 1 leaves only its DC, an exact known multiple of the sample
    a = rng.integers(0, 256, (samples, rows))
    dc = (m1[0].sum() * a + b1[0]) >> s1[0]
    o = np.empty((samples, 8), np.int64)
    for i in range(samples):
        o[i] = call(np.repeat(a[i][:, None], cols, axis=1))[:, 0]
    m2, b2, s2 = fit_rows(dc, o, 1)
    return (m1, b1, s1), (m2, b2, s2)


def model(t1, t2, block):
    """the two-pass integer model on a rows x cols block of samples -> 8 x 8"""
    (m1, b1, s1), (m2, b2, s2) = t1, t2
    d1 = (block.astype(np.int64) @ m1.T + b1) >> s1           # (rows, 8)
    return ((m2 @ d1) + b2[:, None]) >> s2[:, None]            # (8, 8): [k][l]


def verify(lib, name, cols, rows, t1, t2, blocks=20000, seed=99):
    call = make_caller(lib, name, cols, rows)
    rng = np.random.default_rng(seed)
    for i in range(blocks):
        kind = i % 4
        if kind == 0:
            blk = rng.integers(0, 256, (rows, cols))
        elif kind == 1:   # smooth
            blk = np.clip(rng.integers(0, 256) + np.add.outer(np.arange(rows) * rng.integers(-9, 10), np.arange(cols) * rng.integers(-9, 10))
                          + rng.integers(-3, 4, (rows, cols)), 0, 255)
        elif kind == 2:   # extremes
            blk = rng.choice([0, 255], (rows, cols))
        else:             # flat with a few outliers
            blk = np.full((rows, cols), rng.integers(0, 256)); blk[rng.integers(0, rows), rng.integers(0, cols)] = rng.integers(0, 256)
        if not np.array_equal(model(t1, t2, blk), call(blk)):
            return False
    return True


def emit(tables):
    out = ["/* mjx_fdct_tables.h - GENERATED by tests/golden/make_fdct_tables.py, do not edit.",
           " *",
           " * The forward DCTs of IJG libjpeg 9d (jpeg_fdct_islow, _16x16, _16x8, _8x16) as exact integer linear",
           " * forms, measured on the library and verified bit for bit against it on random blocks:",
           " *     pass 1 (along a row of C samples)   d[l] = (sum_u M1[l][u] * sample[u] + B1[l]) >> S1[l]",
           " *     pass 2 (down a column of R rows)    c[k] = (sum_r M2[k][r] * d_r      + B2[k]) >> S2[k]",
           " * (arithmetic shifts).  kFdct<C>x<R>: C x R samples per block; 8 = that direction at full",
           " * resolution, 16 = subsampled 2:1, which libjpeg >= 7 folds into the transform.  */",
           "#ifndef MJX_FDCT_TABLES_H_", "#define MJX_FDCT_TABLES_H_", ""]
    for (cols, rows), (t1, t2) in tables.items():
        for p, n, (m, b, s) in (("1", cols, t1), ("2", rows, t2)):
            out.append(f"static const int kFdct{cols}x{rows}M{p}[8][{n}] = {{")
            for k in range(8):
                out.append("    {" + ", ".join(str(int(v)) for v in m[k]) + "},")
            out.append("};")
            out.append(f"static const int kFdct{cols}x{rows}B{p}[8] = {{" + ", ".join(str(int(v)) for v in b) + "};")
            out.append(f"static const int kFdct{cols}x{rows}S{p}[8] = {{" + ", ".join(str(int(v)) for v in s) + "};")
        out.append("")
    out += ["#endif", ""]
    return "\n".join(out)


def main():
    lib = ctypes.CDLL(sys.argv[1] if len(sys.argv) > 1 else "/opt/conda/lib/libjpeg.so.9")
    tables = {}
    for name, cols, rows in TRANSFORMS:
        t1, t2 = measure(lib, name, cols, rows)
        if not verify(lib, name, cols, rows, t1, t2):
            raise SystemExit(f"{name}: the fitted two-pass model differs from the library")
        print(f"{name}: model verified on 20000 blocks", file=sys.stderr)
        tables[(cols, rows)] = (t1, t2)
    sys.stdout.write(emit(tables))


if __name__ == "__main__":
    main()
