"""Batches of independent images for the hot path: device-resident JCOEF planes, per-image
prepared tiles, and the launch plans of include/mjx.h.

An image batch is what `modjpeg-nginx`-style callers produce: many decoded JPEGs of one geometry,
each composed with its own compiled dropon (the reference compiles tiles per mj_compose call,
reference compose.c:155-157).  Images are independent, so a batch shards across GPUs by image
index with no exchange step: rank r of G owns a contiguous range of image
indices and nothing ever crosses xGMI.

Synthetic contents follow include/mjx_synth.h and are generated on the device.
"""
import ctypes

import numpy as np
import torch

from ._lib import (MJX_JOB_TILES_I16, MJX_PLAN_ALPHA_IS_FINAL, MJX_PLAN_GROUP_BY_DENSITY, PlanStats, MJX_PREP_SOURCE_RGB, MJX_PREP_TARGET_YCC, MJX_PREP_TILES_I16, DcJob, RectJob, TilePrep,
                   check, lib)

_BASE_LUMA = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
                       14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
                       49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99])
_BASE_CHROMA = np.array([17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99,
                         47, 66, 99, 99, 99, 99, 99, 99] + [99] * 32)


def quant_table(chroma, quality=85):
    """JPEG Annex K table in natural order with libjpeg's quality scaling (baseline clamp)."""
    scale = 5000 // quality if quality < 50 else 200 - 2 * quality
    base = _BASE_CHROMA if chroma else _BASE_LUMA
    return np.clip((base * scale + 50) // 100, 1, 255).astype(np.uint16)


def shard_range(n_items, rank, world):
    """contiguous shard [lo, hi) of rank `rank` out of `world` (image i -> rank floor(i*world/n))"""
    lo = -(-n_items * rank // world)
    hi = -(-n_items * (rank + 1) // world)
    return lo, hi


class Geometry420:
    """block geometry of a 4:2:0 YCbCr JPEG as libjpeg lays its coefficient arrays out
    (interleaved scan: every component array is padded to whole MCUs)"""

    def __init__(self, width, height):
        self.width, self.height = width, height
        self.mcus_x = -(-width // 16)
        self.mcus_y = -(-height // 16)
        self.samp = [(2, 2), (1, 1), (1, 1)]
        # true size (jpeg_component_info.width_in_blocks/height_in_blocks) and padded array size
        # jdmaster/jdinput: ceil(image_dim * samp / (max_samp * 8))
        self.true_dims = [(-(-width // 8), -(-height // 8))] + [(-(-width // 16), -(-height // 16))] * 2
        self.array_dims = [(self.mcus_x * h, self.mcus_y * v) for h, v in self.samp]  # (blocks per row, rows)

    def blocks_per_image(self):
        return sum(w * h for w, h in self.array_dims)


REGIME_DISC = 5  # A5: tiles of a real overlay (hard-edged disc mask), prepared on the device from pixels


def disc_overlay(w, h):
    """RGBA pixels of a w x h overlay with a real mask: colour ramps with a checker, alpha = a hard-edged
    disc (one antialiased pixel of rim) — most blocks end up fully inside (1 non-zero alpha-DCT weight)
    or fully outside (0), the blocks the rim crosses need most of the 64 (7.1 on average)"""
    y, x = np.mgrid[0:h, 0:w]
    r = np.hypot(x - (w - 1) / 2.0, y - (h - 1) / 2.0)
    a = np.clip(0.46 * min(w, h) - r + 0.5, 0.0, 1.0)
    rgba = np.empty((h, w, 4), np.uint8)
    rgba[..., 0] = 255 - x * 255 // max(w, 1)
    rgba[..., 1] = np.where((x // 8 + y // 8) % 2 == 1, 230, 40)
    rgba[..., 2] = y * 255 // max(h, 1)
    rgba[..., 3] = np.floor(a * 255.0 + 0.5)
    return rgba


class ComposeBatch:
    """`n` images (global indices image0 .. image0+n-1) of one geometry, one dropon rectangle per
    image at MCU (mcu_x, mcu_y) of dropon_w x dropon_h pixels (multiples of 16), per-image tiles."""

    def __init__(self, n, width, height, dropon_w, dropon_h, mcu_x, mcu_y, regime, image0=0, quality=85, device=None, tiles_i16=False, shared_dropon=False):
        """tiles_i16: the prepared tiles are kept as the int16 coefficients they are made from
        (MJX_JOB_TILES_I16: 512 instead of 768 bytes per block, the kernel converts) instead of floats.
        shared_dropon: ONE set of tiles (those of image `image0`) for every image of the batch - the reference's own use, one overlay
        composed onto many images (secondary variant: 256 B of JCOEF per block cross HBM, the tiles stay in cache)"""
        assert dropon_w % 16 == 0 and dropon_h % 16 == 0
        self.n, self.image0, self.regime, self.tiles_i16 = n, image0, regime, bool(tiles_i16)
        self.shared_dropon = bool(shared_dropon)
        nt = 1 if self.shared_dropon else n  # tile sets
        self.geo = Geometry420(width, height)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.quant = [quant_table(False, quality), quant_table(True, quality), quant_table(True, quality)]
        self.rect = []  # per component (row0, col0, rows, cols)
        for (h, v) in self.geo.samp:
            self.rect.append((mcu_y * v, mcu_x * h, dropon_h // 16 * v, dropon_w // 16 * h))
        for c, (row0, col0, rows, cols) in enumerate(self.rect):
            bpr, prow = self.geo.array_dims[c]
            assert row0 + rows <= prow and col0 + cols <= bpr, "dropon rectangle outside the image"
        self.coef = [torch.empty((n, prow, bpr, 64), dtype=torch.int16, device=self.device) for bpr, prow in self.geo.array_dims]
        tdt = torch.int16 if self.tiles_i16 else torch.float32
        self.tile = [torch.empty((nt, rows * cols, 64), dtype=tdt, device=self.device) for (_, _, rows, cols) in self.rect]
        self.alpha = [torch.empty((nt, rows * cols, 64), dtype=tdt, device=self.device) for (_, _, rows, cols) in self.rect]
        self.plan = ctypes.c_void_p()
        self.blocks_per_image = sum(r[2] * r[3] for r in self.rect)

    # ---- contents -------------------------------------------------------------------------
    def fill_synthetic(self):
        """one generator launch per array (blockIdx.y = image): 9 launches for the whole batch"""
        L = lib()
        vp = ctypes.c_void_p
        s = vp(torch.cuda.current_stream().cuda_stream)
        for c in range(3):
            bpr, prow = self.geo.array_dims[c]
            pblocks = bpr * prow
            tblocks = self.rect[c][2] * self.rect[c][3]
            qp = self.quant[c].ctypes.data_as(vp)
            nt = self.tile[c].shape[0]  # tile sets: one per image, or one for all (shared_dropon)
            check(L.mjx_synth_coef_batch(vp(self.coef[c].data_ptr()), pblocks, self.image0, self.n, c, 0, pblocks, qp, s), "mjx_synth_coef_batch")
            if self.regime != REGIME_DISC and self.tiles_i16:
                check(L.mjx_synth_tiles_i16_batch(vp(self.tile[c].data_ptr()), 0, tblocks, self.image0, nt, c, 0, tblocks, 0, s), "mjx_synth_tiles_i16_batch")
                check(L.mjx_synth_tiles_i16_batch(vp(self.alpha[c].data_ptr()), 1, tblocks, self.image0, nt, c, 0, tblocks, self.regime, s), "mjx_synth_tiles_i16_batch")
            elif self.regime != REGIME_DISC:
                check(L.mjx_synth_tile_batch(vp(self.tile[c].data_ptr()), tblocks, self.image0, nt, c, 0, tblocks, s), "mjx_synth_tile_batch")
                check(L.mjx_synth_alpha_batch(vp(self.alpha[c].data_ptr()), tblocks, self.image0, nt, c, 0, tblocks, self.regime, s), "mjx_synth_alpha_batch")
        if self.regime == REGIME_DISC:
            self._fill_disc_tiles()
        torch.cuda.current_stream().synchronize()

    def _fill_disc_tiles(self):
        """every image gets (its own copy of) the tiles of one real overlay, prepared from its pixels on
        the device by mjx_tiles_prepare exactly as the drop-in does on a tile-cache miss"""
        dh, dw = self.rect[0][2] * 8, self.rect[0][3] * 8
        rgba = disc_overlay(dw, dh)
        image = np.ascontiguousarray(rgba[..., :3])
        alpha = np.ascontiguousarray(np.repeat(rgba[..., 3:4], 3, axis=2))
        p = TilePrep(image.ctypes.data, alpha.ctypes.data, dw, dh, 0, 0, dw, dh, 0, 0, MJX_PREP_SOURCE_RGB, MJX_PREP_TARGET_YCC, 3)
        p.flags = MJX_PREP_TILES_I16 if self.tiles_i16 else 0
        one = []
        for c, (h, v) in enumerate(self.geo.samp):
            p.h_samp[c], p.v_samp[c] = h, v
            t = torch.empty((2, self.rect[c][2] * self.rect[c][3], 64), dtype=self.tile[c].dtype, device=self.device)
            p.tile[c], p.alpha_tile[c] = t[0].data_ptr(), t[1].data_ptr()
            one.append(t)
        check(lib().mjx_tiles_prepare(ctypes.byref(p)), "mjx_tiles_prepare")
        for c in range(3):
            self.tile[c][:] = one[c][0]
            self.alpha[c][:] = one[c][1]

    # ---- launch plan ------------------------------------------------------------------------
    def build_plan(self, group_by_density=True, alpha_is_final=True, side_by_side=None, run_octets=0, stream_run_octets=0, tail_min_octets=0, beside_share=0.0,
                   low_column_build=True, tiles_are_final=None):
        """group_by_density: octets are formed from blocks that need the same terms where a rectangle
        mixes patterns (mjx_compose_plan_create_ex); a schedule only, same results.  alpha_is_final (with
        it): the alpha tiles will not change any more, so blocks without any weight leave their tile and
        alpha blocks behind (the reference skips every term of theirs, convolve.c:32).  side_by_side: None = the
        library decides by the size of the launch whether the blend kernel and the streams of the blocks that need
        no blend run side by side; True / False force it (MJX_PLAN_SIDE_BY_SIDE / MJX_PLAN_SERIAL).  run_octets, stream_run_octets,
        tail_min_octets, beside_share: mjx_plan_options_t (0 = the library's choice); low_column_build=False: MJX_PLAN_NO_LOW_COLUMN_BUILD.
        tiles_are_final (None: as alpha_is_final): MJX_PLAN_TILES_ARE_FINAL, the image tiles keep their contents too, so opaque blocks
        (single weight 0.25, whole-number tile) are written from the tile alone"""
        from ._lib import MJX_PLAN_NO_LOW_COLUMN_BUILD, MJX_PLAN_SERIAL, MJX_PLAN_SIDE_BY_SIDE, MJX_PLAN_TILES_ARE_FINAL, PlanOptions
        tiles_are_final = alpha_is_final if tiles_are_final is None else (tiles_are_final and alpha_is_final)
        how = 0 if side_by_side is None else (MJX_PLAN_SIDE_BY_SIDE if side_by_side else MJX_PLAN_SERIAL)
        jobs = (RectJob * (3 * self.n))()
        j = 0
        for i in range(self.n):
            for c in range(3):
                bpr, prow = self.geo.array_dims[c]
                row0, col0, rows, cols = self.rect[c]
                ti = 0 if self.shared_dropon else i
                jobs[j] = RectJob(self.coef[c][i].data_ptr(), self.tile[c][ti].data_ptr(), self.alpha[c][ti].data_ptr(),
                                  bpr, prow, row0, col0, rows, cols, c, MJX_JOB_TILES_I16 if self.tiles_i16 else 0)
                j += 1
        qt = np.ascontiguousarray(np.stack(self.quant))
        self.destroy_plan()
        opt = PlanOptions()
        opt.flags = ((MJX_PLAN_GROUP_BY_DENSITY | (MJX_PLAN_ALPHA_IS_FINAL if alpha_is_final else 0) | (MJX_PLAN_TILES_ARE_FINAL if tiles_are_final else 0)) if group_by_density else 0) | how | \
            (0 if low_column_build else MJX_PLAN_NO_LOW_COLUMN_BUILD)
        opt.run_octets, opt.stream_run_octets, opt.tail_min_octets, opt.beside_share = run_octets, stream_run_octets, tail_min_octets, beside_share
        check(lib().mjx_compose_plan_create_opts(ctypes.byref(self.plan), jobs, 3 * self.n, qt.ctypes.data_as(ctypes.c_void_p), 3, ctypes.byref(opt)),
              "mjx_compose_plan_create_opts")
        assert lib().mjx_compose_plan_blocks(self.plan) == self.n * self.blocks_per_image
        self.gather_octets = lib().mjx_compose_plan_gather_octets(self.plan)
        return self

    def plan_stats(self):
        st = PlanStats()
        check(lib().mjx_compose_plan_stats(self.plan, ctypes.byref(st)), "mjx_compose_plan_stats")
        return st.as_dict()

    def run(self):
        """one pass of mj_compose_with_mask over every image of the batch (asynchronous)"""
        check(lib().mjx_compose_run(self.plan, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "mjx_compose_run")

    def destroy_plan(self):
        if self.plan:
            lib().mjx_compose_plan_destroy(self.plan)
            self.plan = ctypes.c_void_p()

    def __del__(self):
        try:
            self.destroy_plan()
        except Exception:
            pass

    # ---- inspection -------------------------------------------------------------------------
    def rect_numpy(self, i, c):
        row0, col0, rows, cols = self.rect[c]
        return self.coef[c][i, row0:row0 + rows, col0:col0 + cols].cpu().numpy()

    def total_blocks(self):
        return self.n * self.blocks_per_image

    def tiles_as_float(self, i, c):
        """(tile, alpha) of image i, component c as the float arrays the reference's mj_compileddropon_t
        would hold (for the oracle): the arrays themselves, or converted from the int16 form exactly as the
        reference converts its JCOEFs (dropon.c:476-485, :548-566)"""
        i = 0 if self.shared_dropon else i
        t, a = self.tile[c][i].cpu().numpy().reshape(-1), self.alpha[c][i].cpu().numpy().reshape(-1)
        if not self.tiles_i16:
            return np.ascontiguousarray(t), np.ascontiguousarray(a)
        c0, c1 = 0.3535534, 0.5
        k = np.array([(c0 if j // 8 == 0 else c1) * (c0 if j % 8 == 0 else c1) / 1020.0 for j in range(64)], np.float64)
        af = (a.astype(np.float32).astype(np.float64).reshape(-1, 64) * k).astype(np.float32).reshape(-1)
        return np.ascontiguousarray(t.astype(np.float32)), np.ascontiguousarray(af)

    def rect_clone(self):
        """device copy of every dropon rectangle (compose is in place: the pristine blocks)"""
        out = []
        for c, (row0, col0, rows, cols) in enumerate(self.rect):
            out.append(self.coef[c][:, row0:row0 + rows, col0:col0 + cols].clone())
        return out

    def rect_restore(self, saved):
        for c, (row0, col0, rows, cols) in enumerate(self.rect):
            self.coef[c][:, row0:row0 + rows, col0:col0 + cols].copy_(saved[c])


class EffectBatch:
    """`n` 4:2:0 images for the DC effects: mj_effect_luminance(value) followed by
    mj_effect_tint(cb, cr) over every image (reference effect.c:185, :116).  `fused=True`
    applies the three channel updates in one launch (one pass over the DCs); `fused=False` issues
    the reference's two calls as two launches."""

    def __init__(self, n, width, height, image0=0, quality=85, device=None):
        self.n, self.image0 = n, image0
        self.geo = Geometry420(width, height)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.quant = [quant_table(False, quality), quant_table(True, quality), quant_table(True, quality)]
        self.coef = [torch.empty((n, prow, bpr, 64), dtype=torch.int16, device=self.device) for bpr, prow in self.geo.array_dims]
        self.plans = []

    def fill_synthetic(self):
        L = lib()
        vp = ctypes.c_void_p
        s = vp(torch.cuda.current_stream().cuda_stream)
        for c in range(3):
            bpr, prow = self.geo.array_dims[c]
            pblocks = bpr * prow
            qp = self.quant[c].ctypes.data_as(vp)
            check(L.mjx_synth_coef_batch(vp(self.coef[c].data_ptr()), pblocks, self.image0, self.n, c, 0, pblocks, qp, s), "mjx_synth_coef_batch")
        torch.cuda.current_stream().synchronize()

    def dc_blocks_per_image(self):
        return sum(w * h for w, h in self.geo.true_dims)

    def build_plans(self, lum, cb, cr, fused=True):
        """the effects walk the TRUE width_in_blocks x height_in_blocks of each component
        (effect.c:201-204), not the MCU-padded array"""
        self.destroy_plans()
        groups = [[(0, lum), (1, cb), (2, cr)]] if fused else [[(0, lum)], [(1, cb), (2, cr)]]
        for grp in groups:
            jobs = []
            for c, value in grp:
                if c > 0 and value == 0:
                    continue  # effect.c:134/:158
                bpr, prow = self.geo.array_dims[c]
                tw, th = self.geo.true_dims[c]
                base, step = self.coef[c].data_ptr(), bpr * prow * 128
                q0 = int(self.quant[c][0])
                for i in range(self.n):
                    jobs.append(DcJob(base + i * step, bpr, th, tw, value, q0, 0))
            # grid.y limit: 65535 jobs per launch
            for lo in range(0, len(jobs), 65535):
                part = jobs[lo:lo + 65535]
                arr = (DcJob * len(part))(*part)
                plan = ctypes.c_void_p()
                check(lib().mjx_dc_plan_create(ctypes.byref(plan), arr, len(part)), "mjx_dc_plan_create")
                self.plans.append(plan)
        return self

    def run(self):
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for p in self.plans:
            check(lib().mjx_dc_plan_run(p, s), "mjx_dc_plan_run")

    def destroy_plans(self):
        for p in self.plans:
            lib().mjx_dc_plan_destroy(p)
        self.plans = []

    def __del__(self):
        try:
            self.destroy_plans()
        except Exception:
            pass
