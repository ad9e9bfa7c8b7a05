// mjx_tileprep.hip — overlay pixels -> prepared DCT-domain tiles on the device (SURVEY.md §8 row f-1,
// second half).
//
// Replaces, for the cases listed below, what the reference does on every mj_compose call through
// libjpeg (reference src/dropon.c:325-576 and src/image.c:257-347): pad the cropped overlay to whole
// MCUs, compress image and alpha at quality 100 in the target's colour space and sampling, read the
// coefficients back, convert to float tiles (alpha: DC + 1024, scaling c(i)c(j)/1020).  At quality
// 100 every quantiser is 1, so a prepared coefficient is
//          round_half_away( FDCT(samples) / 8 )
// with FDCT the integer "islow" transform of the linked libjpeg (IJG 9d): jpeg_fdct_islow on 8x8
// samples for a component at full resolution, jpeg_fdct_16x16 / _16x8 / _8x16 on 16x16 / 16x8 /
// 8x16 samples for a component subsampled 2:1 in both / the horizontal / the vertical direction
// (IJG >= 7 folds the downsampling into the transform).  They are reproduced here as
// the exact integer linear forms of mjx_fdct_tables.h, measured on the library and verified bit
// for bit (tests/golden/make_fdct_tables.py, tests/test_fdct_tables.py); the RGB -> YCbCr
// conversion is libjpeg's 16-bit fixed-point table arithmetic (jccolor.c: FIX(x) = x * 65536 + 0.5,
// ONE_HALF rounding for Y, CBCR_OFFSET + ONE_HALF - 1 for the chroma).  The Huffman round trip
// the reference makes is lossless and is simply not made.  Parity: the tiles equal the ones the
// reference's mj_compile_dropon prepared (tests/test_dropin_shim.py, golden compose_kat.npz) and
// the host path's on random overlays, as float bit patterns.
//
// Supported: three components, target YCbCr (overlay RGB: converted; overlay YCbCr: as is) or RGB
// (overlay RGB), every component at full resolution or subsampled 2:1 in either direction (4:4:4,
// 4:2:2, 4:4:0, 4:2:0); one component, target grayscale (overlay RGB: libjpeg's Y; overlay YCbCr: its
// first component); and every other whole-number ratio a component can have (3 and 4 in either direction: 4:1:1, 4:1:0,
// 1x4 luma ...).  libjpeg 9 folds ONE halving per direction into the transform (jcmaster.c: DCT_h/v_scaled_size 16 when
// the ratio is even) and makes what is left of the ratio - 2 of a ratio 4, the whole of a ratio 3 - with jcsample.c's
// downsamplers on the colour-converted samples: h2v1 (bias 0, 1, 0, 1 ...), h2v2 (bias 1, 2, 1, 2 ...), and its general box
// filter int_downsample for every other pair ((sum + n / 2) / n), all reproduced in prep_block.  Everything
// else (fractional ratios, CMYK ...) returns MJX_ERR_BAD_ARG and the caller prepares the tiles with
// libjpeg as before.
//
// One wavefront per output block: the C x R samples go to LDS (colour conversion on the way),
// pass 1 along the rows (R x 8 outputs), pass 2 down the columns (8 x 8), quantisation, float.
#include "mjx_internal.hpp"

#include "mjx_fdct_tables.h"

#include <string.h>
#include <time.h>

namespace mjx {

struct PrepComp {
    float  *tile, *alpha;   // DEVICE out: rows * cols blocks of 64 floats
    int32_t rows, cols;     // blocks
    int32_t nx, ny;         // samples per block: 8 or 16 in each direction
    int32_t first;          // index of this component's first block in the launch
    int32_t pre_x, pre_y;   // the samples are first averaged over pre_x x pre_y pixels (what the transform does not fold: see prep_block)
};

struct PrepJob {
    const unsigned char *image, *alpha; // DEVICE: the crop window, 3 bytes per pixel, crop_w per row
    int32_t  crop_w, crop_h, off_x, off_y;
    int32_t  convert;                   // 1: RGB -> YCbCr
    int32_t  i16;                       // 1: the tiles are written as int16 coefficients (MJX_PREP_TILES_I16)
    int32_t  nblocks;                   // all components
    PrepComp comp[3];
};

// libjpeg jccolor.c rgb_ycc_convert, SCALEBITS 16
__device__ __forceinline__ int ycc_of_rgb(int c, int r, int g, int b) {
    constexpr int kHalf = 1 << 15, kOffset = 128 << 16;
    if(c == 0) return (19595 * r + 38470 * g + 7471 * b + kHalf) >> 16;
    if(c == 1) return (-11059 * r - 21709 * g + 32768 * b + kOffset + kHalf - 1) >> 16;
    return (32768 * r - 27439 * g - 5329 * b + kOffset + kHalf - 1) >> 16;
}

// alpha tile scaling, the reference's own double expressions (dropon.c:548-566)
__device__ __forceinline__ double alpha_scale(int k, int l) {
    const double c0 = 0.3535534, c1 = 0.5;
    return (k == 0 ? c0 : c1) * (l == 0 ? c0 : c1) / 1020.0;
}

// the tables of the C x R transform
template <int C, int R> struct Fdct;
#define MJX_FDCT(C, R)                                                                             \
    template <> struct Fdct<C, R> {                                                                \
        static __device__ __forceinline__ const int *M1() { return &kFdct##C##x##R##M1[0][0]; }    \
        static __device__ __forceinline__ const int *M2() { return &kFdct##C##x##R##M2[0][0]; }    \
        static __device__ __forceinline__ const int *B1() { return kFdct##C##x##R##B1; }           \
        static __device__ __forceinline__ const int *S1() { return kFdct##C##x##R##S1; }           \
        static __device__ __forceinline__ const int *B2() { return kFdct##C##x##R##B2; }           \
        static __device__ __forceinline__ const int *S2() { return kFdct##C##x##R##S2; }           \
    }
MJX_FDCT(8, 8);
MJX_FDCT(16, 16);
MJX_FDCT(16, 8);
MJX_FDCT(8, 16);

template <int C, int R> __device__ __forceinline__ void prep_block(const PrepJob &job, int c, int kind, int brow, int bcol, float *out, int lane, int (*s)[16], int (*d1)[8], int pre_x, int pre_y) {
    using T = Fdct<C, R>;
    const int *M1 = T::M1(), *M2 = T::M2(), *B1 = T::B1(), *S1 = T::S1(), *B2 = T::B2(), *S2 = T::S2();
    const unsigned char *pix = kind == 0 ? job.image : job.alpha;
    // samples: canvas pixel (x, y) is overlay window pixel (x - off_x, y - off_y), zero outside it
    auto sample = [&](int cx, int cy) { // canvas pixel (cx, cy) of this component, colour-converted
        const int wx = cx - job.off_x, wy = cy - job.off_y;
        int       r = 0, g = 0, b = 0;
        if(wx >= 0 && wx < job.crop_w && wy >= 0 && wy < job.crop_h) {
            const unsigned char *p = pix + ((size_t)wy * job.crop_w + wx) * 3;
            r = p[0]; g = p[1]; b = p[2];
        }
        // the alpha canvas is never converted (reference dropon.c:389-402 hands it over in the target's own space)
        return (kind == 0 && job.convert) ? ycc_of_rgb(c, r, g, b) : (c == 0 ? r : (c == 1 ? g : b));
    };
    for(int i = lane; i < C * R; i += 64) {
        const int sy = i / C, sx = i - sy * C;
        const int ox = bcol * C + sx, oy = brow * R + sy; // the sample's place in the component as the transform sees it
        if(pre_x == 1 && pre_y == 1) {
            s[sy][sx] = sample(ox, oy);
        }
        else {
            // what IJG 9 does not fold into the transform it makes with jcsample.c on the colour-converted samples:
            //   2 x 1  h2v1_downsample: (a + b + bias) >> 1, bias 0, 1, 0, 1 ... along the output row          (4:1:1, 4:1:0)
            //   2 x 2  h2v2_downsample: (a + b + c + d + bias) >> 2, bias 1, 2, 1, 2 ...
            //   else   int_downsample:  (sum + n / 2) / n over the pre_x x pre_y box                           (ratios 3; 1 x 2 of a vertical ratio 4)
            int sum = 0;
            for(int j = 0; j < pre_y; ++j)
                for(int i = 0; i < pre_x; ++i) sum += sample(ox * pre_x + i, oy * pre_y + j);
            if(pre_x == 2 && pre_y == 1) s[sy][sx] = (sum + (ox & 1)) >> 1;
            else if(pre_x == 2 && pre_y == 2) s[sy][sx] = (sum + 1 + (ox & 1)) >> 2;
            else s[sy][sx] = (sum + (pre_x * pre_y) / 2) / (pre_x * pre_y);
        }
    }
    __syncthreads();
    for(int i = lane; i < R * 8; i += 64) { // pass 1: row r, output l
        const int r = i >> 3, l = i & 7;
        int       acc = B1[l];
#pragma unroll
        for(int u = 0; u < C; ++u) acc += M1[l * C + u] * s[r][u];
        d1[r][l] = acc >> S1[l];
    }
    __syncthreads();
    {
        const int k = lane >> 3, l = lane & 7; // pass 2: output (k, l)
        int       acc = B2[k];
#pragma unroll
        for(int r = 0; r < R; ++r) acc += M2[k * R + r] * d1[r][l];
        acc >>= S2[k];
        // quantiser 1, divisor 8 (jcdctmgr.c forward_DCT): round half away from zero
        int q = acc < 0 ? -((-acc + 4) >> 3) : ((acc + 4) >> 3);
        if(kind == 1 && (k | l) == 0) q = (int)(short)(unsigned short)((unsigned)q + 1024u); // dropon.c:542 on a JCOEF
        if(job.i16) { // the coefficient itself: the compose kernel converts (MJX_JOB_TILES_I16)
            reinterpret_cast<int16_t *>(out)[k * 8 + l] = (int16_t)q;
        }
        else {
            out[k * 8 + l] = kind == 0 ? (float)(int)(short)q                                   // dropon.c:476-485
                                       : (float)((double)(float)(int)(short)q * alpha_scale(k, l)); // dropon.c:548-566
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(64) void tileprep_kernel(PrepJob job) {
    __shared__ int s[16][16];
    __shared__ int d1[16][8];
    const int lane = threadIdx.x;
    int       id = blockIdx.x;
    const int kind = id >= job.nblocks ? 1 : 0; // 0 image, 1 alpha
    id -= kind * job.nblocks;
    int c = 0;
    if(id >= job.comp[1].first) c = 1;
    if(id >= job.comp[2].first) c = 2;
    const PrepComp pc = job.comp[c];
    const int      blk = id - pc.first, brow = blk / pc.cols, bcol = blk - brow * pc.cols;
    float         *out = (kind == 0 ? pc.tile : pc.alpha) + (size_t)blk * (job.i16 ? 32 : 64); // an int16 block is 32 floats long
    if(pc.nx == 8 && pc.ny == 8) prep_block<8, 8>(job, c, kind, brow, bcol, out, lane, s, d1, pc.pre_x, pc.pre_y);
    else if(pc.nx == 16 && pc.ny == 16) prep_block<16, 16>(job, c, kind, brow, bcol, out, lane, s, d1, pc.pre_x, pc.pre_y);
    else if(pc.nx == 16) prep_block<16, 8>(job, c, kind, brow, bcol, out, lane, s, d1, pc.pre_x, pc.pre_y);
    else prep_block<8, 16>(job, c, kind, brow, bcol, out, lane, s, d1, pc.pre_x, pc.pre_y);
}

} // namespace mjx

using namespace mjx;

extern "C" int mjx_tiles_prepare(const mjx_tileprep_t *p) {
    if(p == nullptr) return MJX_ERR_NULL_DATA;
    if(p->image == nullptr || p->alpha == nullptr) return MJX_ERR_NULL_DATA;
    if((p->flags & ~(MJX_PREP_TILES_I16 | MJX_PREP_WINDOW_PACKED)) != 0) return MJX_ERR_BAD_ARG;
    const bool packed = (p->flags & MJX_PREP_WINDOW_PACKED) != 0;
    if(packed && (p->crop_x != 0 || p->crop_y != 0 || p->width != p->crop_w || p->height != p->crop_h || p->alpha != p->image + (size_t)p->crop_w * 3 * (size_t)p->crop_h))
        return MJX_ERR_BAD_ARG;
    if((p->ncomp != 3 && p->ncomp != 1) || p->crop_w <= 0 || p->crop_h <= 0 || p->off_x < 0 || p->off_y < 0 || p->crop_x < 0 || p->crop_y < 0) return MJX_ERR_BAD_ARG;
    if((p->ncomp == 1) != (p->target == MJX_PREP_TARGET_GRAY)) return MJX_ERR_BAD_ARG;
    const int ncomp = p->ncomp;
    if(p->crop_x + (int64_t)p->crop_w > p->width || p->crop_y + (int64_t)p->crop_h > p->height) return MJX_ERR_BAD_ARG;
    int convert;
    // a grayscale target takes libjpeg's Y of an RGB overlay (jccolor.c rgb_gray_convert: the Y row of the same
    // tables) or the first component of a YCbCr one (grayscale_convert); its alpha canvas is handed over as
    // YCbCr (reference dropon.c:409-413), i.e. its first component as it stands
    if((p->target == MJX_PREP_TARGET_YCC || p->target == MJX_PREP_TARGET_GRAY) && p->source == MJX_PREP_SOURCE_RGB) convert = 1;
    else if(p->target == MJX_PREP_TARGET_GRAY && p->source == MJX_PREP_SOURCE_YCC) convert = 0;
    else if((p->target == MJX_PREP_TARGET_YCC && p->source == MJX_PREP_SOURCE_YCC) || (p->target == MJX_PREP_TARGET_RGB && p->source == MJX_PREP_SOURCE_RGB)) convert = 0;
    else return MJX_ERR_BAD_ARG;
    int hmax = 1, vmax = 1;
    for(int c = 0; c < ncomp; ++c) {
        if(p->h_samp[c] < 1 || p->v_samp[c] < 1) return MJX_ERR_BAD_ARG;
        hmax = p->h_samp[c] > hmax ? p->h_samp[c] : hmax;
        vmax = p->v_samp[c] > vmax ? p->v_samp[c] : vmax;
    }
    // whole MCUs, including the offset of the overlay inside its first MCU (dropon.c:340-350)
    const int64_t mcu_w = 8 * hmax, mcu_h = 8 * vmax;
    const int64_t W = (p->crop_w + (int64_t)p->off_x + mcu_w - 1) / mcu_w * mcu_w, H = (p->crop_h + (int64_t)p->off_y + mcu_h - 1) / mcu_h * mcu_h;
    PrepJob job;
    memset(&job, 0, sizeof(job));
    int64_t total = 0;
    for(int c = 0; c < ncomp; ++c) {
        if(hmax % p->h_samp[c] != 0 || vmax % p->v_samp[c] != 0) return MJX_ERR_BAD_ARG;
        const int fx = hmax / p->h_samp[c], fy = vmax / p->v_samp[c];
        // 8x8, 16x16, 16x8, 8x16 transforms (an even ratio folds one halving into the transform: jcmaster.c, DCT_h/v_scaled_size;
        // never more than a factor 2 between the two sizes, which ratios up to 4 cannot exceed); the rest of the ratio is a box filter
        if(fx < 1 || fx > 4 || fy < 1 || fy > 4) return MJX_ERR_BAD_ARG;
        const int fold_x = fx % 2 == 0 ? 2 : 1, fold_y = fy % 2 == 0 ? 2 : 1;
        if(p->tile[c] == nullptr || p->alpha_tile[c] == nullptr) return MJX_ERR_NULL_DATA;
        job.comp[c].tile = p->tile[c];
        job.comp[c].alpha = p->alpha_tile[c];
        job.comp[c].nx = 8 * fold_x;
        job.comp[c].ny = 8 * fold_y;
        job.comp[c].pre_x = fx / fold_x;
        job.comp[c].pre_y = fy / fold_y;
        job.comp[c].cols = (int32_t)(W / (8 * fx));
        job.comp[c].rows = (int32_t)(H / (8 * fy));
        job.comp[c].first = (int32_t)total;
        total += (int64_t)job.comp[c].rows * job.comp[c].cols;
    }
    if(total * 2 > 0x7FFFFFFFll) return MJX_ERR_BAD_ARG;
    for(int c = ncomp; c < 3; ++c) job.comp[c].first = 0x7FFFFFFF; // no block belongs to a component the target does not have
    if(!ensure_device()) return MJX_ERR_NO_DEVICE;

    // MJX_TILEPREP_DEBUG=1: where the time goes (stderr)
    static const bool dbg = tuning_env("MJX_TILEPREP_DEBUG") != nullptr;
    auto              now = [] {
        struct timespec t;
        clock_gettime(CLOCK_MONOTONIC, &t);
        return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
    };
    const double t0 = dbg ? now() : 0;
    // the crop window of image and alpha, packed: straight out of the caller's pinned copy, or through the leased arena's
    const size_t row = (size_t)p->crop_w * 3, win = row * (size_t)p->crop_h, win_al = packed ? win : ((win + 255) & ~(size_t)255);
    ArenaLease   arena;
    int          err = arena->prepare(win_al + win, packed ? 256 : win_al + win);
    if(err != MJX_OK) return err;
    const unsigned char *src = p->image;
    if(!packed) {
        for(int y = 0; y < p->crop_h; ++y) {
            const size_t at = ((size_t)(y + p->crop_y) * (size_t)p->width + (size_t)p->crop_x) * 3;
            memcpy(arena->pin + (size_t)y * row, p->image + at, row);
            memcpy(arena->pin + win_al + (size_t)y * row, p->alpha + at, row);
        }
        src = reinterpret_cast<const unsigned char *>(arena->pin);
    }
    const double t1 = dbg ? now() : 0;
    hipStream_t  s = arena->stream;
    if((err = arena->order_after(p->stream)) != MJX_OK) return err; // the tile buffers are overwritten: after whoever still reads them (mjx.h)
    MJX_HIP(hipMemcpyAsync(arena->dev, src, win_al + win, hipMemcpyHostToDevice, s), MJX_ERR_LAUNCH);
    if(dbg) (void)hipStreamSynchronize(s);
    const double t2 = dbg ? now() : 0;
    job.image = reinterpret_cast<const unsigned char *>(arena->dev);
    job.alpha = reinterpret_cast<const unsigned char *>(arena->dev) + win_al;
    job.crop_w = p->crop_w;
    job.crop_h = p->crop_h;
    job.off_x = p->off_x;
    job.off_y = p->off_y;
    job.convert = convert;
    job.i16 = (p->flags & MJX_PREP_TILES_I16) ? 1 : 0;
    job.nblocks = (int32_t)total;
    hipLaunchKernelGGL(tileprep_kernel, dim3((unsigned)(2 * total)), dim3(64), 0, s, job);
    err = check_hip(hipGetLastError(), "tileprep_kernel launch", MJX_ERR_LAUNCH);
    const int serr = check_hip(hipStreamSynchronize(s), "hipStreamSynchronize", MJX_ERR_LAUNCH);
    if(dbg) fprintf(stderr, "mjx_tiles_prepare: arena + window into pinned memory %.3f ms, upload %.3f ms, kernel %.3f ms (%lld blocks x 2)\n", t1 - t0, t2 - t1, now() - t2, (long long)total);
    return err != MJX_OK ? err : serr;
}
