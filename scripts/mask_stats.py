#!/usr/bin/env python3
"""What the zero patterns of real masks look like and what a schedule makes of them (GPU box).

For each mask the overlay's tiles are prepared on the device (mjx_tiles_prepare, 4:2:0 YCbCr target, as the
drop-in does on a tile-cache miss) and scheduled (mjx_tile_schedule_create).  Printed per mask: blocks,
non-zero weights per block, octets by kind, and the mean number of terms an octet executes (the size of the
UNION of its 8 blocks' patterns) with consecutive octets and with the schedule's."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def masks():
    from libmodjpeg_amd.batch import disc_overlay
    out = {"hard-edged disc 256x256 (regime A5)": disc_overlay(256, 256)}
    yy, xx = np.mgrid[0:256, 0:256]
    r = np.hypot(xx - 128, yy - 128) / 128
    soft = np.stack([255 - xx, ((xx // 8 + yy // 8) % 2) * 190 + 40, yy, np.clip((1.0 - r) / 0.4, 0, 1) * 255], -1).astype(np.uint8)
    out["soft radial mask 256x256 (scripts/cli_batch_rate.sh's logo)"] = soft
    try:
        from PIL import Image
        logo = np.array(Image.open(os.path.join(ROOT, "tests", "golden", "images", "dropon.png")).convert("RGBA"))
        out[f"text logo {logo.shape[1]}x{logo.shape[0]} (the reference's dropon.png)"] = logo
        big = np.array(Image.fromarray(logo).resize((logo.shape[1] * 4, logo.shape[0] * 4), Image.BICUBIC))
        out[f"the same logo scaled x4 ({big.shape[1]}x{big.shape[0]})"] = big
    except Exception as e:  # noqa: BLE001
        print("no PIL / logo:", e)
    opaque = np.full((256, 256, 4), 255, np.uint8)
    opaque[..., :3] = soft[..., :3]
    opaque[:40, :, 3] = 0
    opaque[:, :24, 3] = 0
    out["opaque rectangle with transparent margins 256x256"] = opaque
    return out


def main():
    import torch
    from libmodjpeg_amd._lib import MJX_JOB_TILES_I16, MJX_PREP_SOURCE_RGB, MJX_PREP_TARGET_YCC, MJX_PREP_TILES_I16, PlanStats, TilePrep, check, lib
    L = lib()
    for name, rgba in masks().items():
        h, w = rgba.shape[:2]
        image = np.ascontiguousarray(rgba[..., :3])
        alpha = np.ascontiguousarray(np.repeat(rgba[..., 3:4], 3, axis=2))
        p = TilePrep(image.ctypes.data, alpha.ctypes.data, w, h, 0, 0, w, h, 0, 0, MJX_PREP_SOURCE_RGB, MJX_PREP_TARGET_YCC, 3)
        p.flags = MJX_PREP_TILES_I16
        W, H = -(-w // 16) * 16, -(-h // 16) * 16
        keep, dims = [], []
        for c, (hs, vs) in enumerate([(2, 2), (1, 1), (1, 1)]):
            p.h_samp[c], p.v_samp[c] = hs, vs
            cols, rows = W // (16 // hs), H // (16 // vs)
            t = torch.empty((2, rows * cols, 64), dtype=torch.int16, device="cuda")
            p.tile[c], p.alpha_tile[c] = t[0].data_ptr(), t[1].data_ptr()
            keep.append(t)
            dims.append((rows, cols))
        check(L.mjx_tiles_prepare(ctypes.byref(p)), "mjx_tiles_prepare")
        tot = PlanStats()
        acc = dict(blocks=0, octets=0, gather=0, zero=0, dc=0, terms=0, union=0, union_cons=0)
        for c, (rows, cols) in enumerate(dims):
            sc = ctypes.c_void_p()
            check(L.mjx_tile_schedule_create(ctypes.byref(sc), ctypes.c_void_p(keep[c][1].data_ptr()), rows, cols, MJX_JOB_TILES_I16, None), "mjx_tile_schedule_create")
            st = PlanStats()
            check(L.mjx_tile_schedule_stats(sc, ctypes.byref(st)), "stats")
            L.mjx_tile_schedule_destroy(sc)
            acc["blocks"] += st.blocks; acc["octets"] += st.octets; acc["gather"] += st.gather_octets; acc["zero"] += st.zero_alpha_octets
            acc["dc"] += st.dc_only_octets; acc["terms"] += st.block_terms; acc["union"] += st.union_terms; acc["union_cons"] += st.union_terms_consecutive
        general = acc["octets"] - acc["zero"] - acc["dc"]
        print(f"{name}: {acc['blocks']} blocks, {acc['terms'] / acc['blocks']:.2f} weights per block")
        print(f"    consecutive octets: {acc['union_cons'] / (-(-acc['blocks'] // 8)):.2f} terms per octet")
        print(f"    scheduled: {acc['octets']} octets = {acc['zero']} zero-alpha + {acc['dc']} DC-only (1 term) + {general} general; "
              f"{acc['union'] / max(acc['octets'] - acc['zero'], 1):.2f} terms per octet that has any, {(acc['union'] - acc['dc']) / max(general, 1):.2f} per general octet; gathered: {acc['gather']}")


if __name__ == "__main__":
    main()
