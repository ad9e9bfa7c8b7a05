#!/usr/bin/env python3
"""Generate the committed golden fixtures from the UNMODIFIED reference.

Runs only in the build container (needs /root/reference and oracle/_ref built by
`make -C oracle`).  oracle/_ref/ref_dump links the reference compiled from the sources where they
lie and dumps raw arrays; this script packs them into compressed .npz files next to itself:

    convolve_kat.npz   mj_convolve vectors            (reference convolve.c:29)
    block_kat.npz      mj_compose_with_mask vectors   (reference compose.c:237) on fabricated blocks
    compose_kat.npz    mj_compose end-to-end (image.jpg + dropon.png / dropon.jpg,mask.jpg):
                       coefficient planes before, reference-prepared tiles, rectangles after
    effect_kat.npz     mj_effect_* before/after planes (reference effect.c)

Fixtures are data only (inputs and expected outputs); no reference source is stored.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_IMAGES = "/root/reference/src/contrib/images"
DT = {"f4": np.float32, "i2": np.int16, "u2": np.uint16, "i4": np.int32, "u1": np.uint8}


def load_raw(d):
    out = {}
    for line in open(os.path.join(d, "manifest.txt")):
        name, dt, count = line.split()
        a = np.fromfile(os.path.join(d, name + ".bin"), dtype=DT[dt])
        assert a.size == int(count), name
        out[name] = a
    return out


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    with tempfile.TemporaryDirectory() as d:
        subprocess.check_call([os.path.join(ROOT, "oracle/_ref/ref_dump"), REF_IMAGES, d])
        raw = load_raw(d)

    np.savez_compressed(
        os.path.join(HERE, "convolve_kat.npz"),
        x=raw["convolve_x"].reshape(-1, 64), yin=raw["convolve_yin"].reshape(-1, 64),
        w=raw["convolve_w"], kl=raw["convolve_kl"].reshape(-1, 2), yout=raw["convolve_yout"].reshape(-1, 64))

    np.savez_compressed(
        os.path.join(HERE, "block_kat.npz"),
        coef_in=raw["block_coef_in"].reshape(-1, 64), coef_out=raw["block_coef_out"].reshape(-1, 64),
        tile=raw["block_tile"].reshape(-1, 64), alpha=raw["block_alpha"].reshape(-1, 64),
        q=raw["block_q"].reshape(-1, 3, 64), geom=raw["block_geom"].reshape(-1, 3, 8))

    # compose cases: the three cases share one input image -> store `before` once, and for each
    # case only the composed rectangle (+ an FNV hash of the full plane to pin "rest untouched").
    comp = {}
    for c in range(3):
        comp[f"before_c{c}"] = raw[f"tl_before_c{c}"]
        comp[f"q_c{c}"] = raw[f"tl_q_c{c}"]
        for case in ("br", "crop"):
            assert np.array_equal(raw[f"{case}_before_c{c}"], raw[f"tl_before_c{c}"])
    comp["meta"] = raw["tl_meta"]
    for case in ("tl", "br", "crop"):
        comp[f"{case}_call"] = raw[f"{case}_call"]
        comp[f"{case}_dropon_meta"] = raw[f"{case}_dropon_meta"]
        comp[f"{case}_dropon_image"] = raw[f"{case}_dropon_image"]
        comp[f"{case}_dropon_alpha"] = raw[f"{case}_dropon_alpha"]
        for c in range(3):
            wib, hib = raw["tl_meta"][4 + 4 * c], raw["tl_meta"][5 + 4 * c]
            row0, col0, rows, cols, bx, by = raw[f"{case}_rect_c{c}"]
            before = raw[f"{case}_before_c{c}"].reshape(hib, wib, 64)
            after = raw[f"{case}_after_c{c}"].reshape(hib, wib, 64)
            mask = np.ones((hib, wib), bool)
            mask[row0:row0 + rows, col0:col0 + cols] = False
            assert np.array_equal(before[mask], after[mask]), "reference touched blocks outside the rectangle?"
            comp[f"{case}_rect_c{c}"] = raw[f"{case}_rect_c{c}"]
            comp[f"{case}_tile_c{c}"] = raw[f"{case}_tile_c{c}"].reshape(-1, 64)
            comp[f"{case}_alpha_c{c}"] = raw[f"{case}_alpha_c{c}"].reshape(-1, 64)
            comp[f"{case}_after_rect_c{c}"] = after[row0:row0 + rows, col0:col0 + cols].copy()
    for c in range(3):
        # README golden: the committed image_dropon.jpg holds exactly the tl result
        assert np.array_equal(raw[f"tl_committed_c{c}"], raw[f"tl_after_c{c}"]), "image_dropon.jpg != mj_compose output"
    np.savez_compressed(os.path.join(HERE, "compose_kat.npz"), **comp)

    fx = {"meta": raw["fx_meta"], "w_meta": raw["fxw_meta"]}
    for c in range(3):
        fx[f"q_c{c}"] = raw[f"fx_q_c{c}"]
        fx[f"w_q_c{c}"] = raw[f"fxw_q_c{c}"]
        for tag in ("before", "lum100_tint50_0", "clamp", "zero", "pixelate", "grayscale"):
            fx[f"{tag}_c{c}"] = raw[f"fx_{tag}_c{c}"]
        fx[f"w_before_c{c}"] = raw[f"fxw_before_c{c}"]
        fx[f"w_after_c{c}"] = raw[f"fxw_after_c{c}"]
    np.savez_compressed(os.path.join(HERE, "effect_kat.npz"), **fx)

    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    sys.exit(main())
