/*
 * mjx.h — C ABI of the MI355X-native libmodjpeg hot path (libmjx_hip.so).
 *
 * This is the drop-in boundary below the libmodjpeg.h API: plain C, plain pointers and sizes, no
 * libjpeg types, no torch types.  Everything the reference computes per 8x8 block inside
 *
 *      mj_compose_with_mask()   reference src/compose.c:237-342
 *      mj_convolve()            reference src/convolve.c:29-1099
 *      mj_effect_grayscale()    reference src/effect.c:28-68
 *      mj_effect_pixelate()     reference src/effect.c:70-114
 *      mj_effect_tint()         reference src/effect.c:116-183
 *      mj_effect_luminance()    reference src/effect.c:185-222
 *
 * is done by hand-written gfx950 HIP kernels behind these entry points, bit-exact against the
 * reference CPU path (JCOEF output identical on identical inputs).  The libjpeg-facing
 * libmodjpeg.h shim (libmodjpeg_amd/host/, include/libmodjpeg.h) and the Python mirror
 * (libmodjpeg_amd/api.py) are thin callers of this header.
 *
 * There is NO CPU fallback: without a usable HIP device every compute entry point returns
 * MJX_ERR_NO_DEVICE (and says so on stderr once).
 *
 * Data layout (the reference's own):
 *   JCOEF plane   int16_t[rows_total][blocks_per_row][64], one 8x8 block = 64 coefficients in
 *                 natural (row-major, de-zigzagged) order, index v*8+u; this is libjpeg's
 *                 JBLOCKROW layout (a component's virtual array row = blocks_per_row*128 bytes).
 *   quant table   uint16_t[64], natural order (jpeg_component_info.quant_table->quantval).
 *   tiles         float[rows*cols][64]: prepared dropon blocks (mj_compileddropon_t.image[c]
 *                 .blocks[cols*l+k][i], reference libmodjpeg.h:86-97) made contiguous, tile row l,
 *                 tile column k; alpha likewise (already scaled as reference dropon.c:542-566).
 *   All device pointers must be 16-byte aligned.
 *
 * Threading: a plan may be run from one thread at a time; distinct plans are independent.  Each
 * host thread works on the device selected by its last mjx_set_device() (hipSetDevice semantics).
 */
#ifndef MJX_H_
#define MJX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MJX_VERSION 105

/* error codes (0 = success).  The libmodjpeg.h shim maps them onto MJ_ERR_* (INTEGRATION.md). */
#define MJX_OK             0
#define MJX_ERR_NO_DEVICE  1 /* no HIP device / HIP runtime failure */
#define MJX_ERR_MEMORY     2 /* device or host allocation failed */
#define MJX_ERR_NULL_DATA  3 /* NULL pointer where data is required (MJ_ERR_NULL_DATA) */
#define MJX_ERR_BAD_ARG    4 /* misaligned pointer, negative size, q == 0, rect outside plane */
#define MJX_ERR_LAUNCH     5 /* kernel launch or synchronisation failed */

typedef void *mjx_stream_t; /* a hipStream_t, or NULL for the default stream */

/* ---- ordering contract for DEVICE memory handed to the synchronous entry points ------------------
 * Two kinds of entry point take device memory:
 *  (1) the asynchronous ones that take a `mjx_stream_t stream` argument and only ENQUEUE on it - mjx_compose_run,
 *      mjx_convolve, mjx_effect_zero, mjx_dc_plan_run, the mjx_synth_* generators: stream order is the caller's;
 *  (2) the synchronous, host-style ones that replace a blocking call of the reference (image.c:94 jpeg_read_coefficients,
 *      image.c:194 jpeg_write_coefficients, compose.c:175 mj_compose_with_mask, effect.c) and may be handed memory
 *      that is already on the device: mjx_huff_decode_host / mjx_huff_encode_host with MJX_HUFF_PLANES_ON_DEVICE,
 *      mjx_compose_host with MJX_RECT_ROWS_ON_DEVICE / MJX_RECT_TILES_ON_DEVICE, mjx_tiles_prepare,
 *      mjx_tile_schedule_create, mjx_effect_dc_device, mjx_effect_zero_device.  These work on a PRIVATE non-blocking
 *      stream from the library's pool, which is ordered against nothing by itself.  The contract:
 *        at entry  the library's first device operation is ordered (by an event) after everything that had been
 *                  queued, when the call was made, on the caller's stream: the `stream` member of the call's
 *                  struct, or its `after` argument.  NULL is the legacy default stream (which in turn waits for
 *                  every blocking stream of the device); a struct that was zero-initialised therefore waits for
 *                  the default stream.  (A stream of the caller's own with nothing queued costs a query, no event; for NULL the event is always recorded, since only that makes the default stream wait for the blocking streams.)  MJX_STREAM_IDLE says that nothing is pending on the memory handed over
 *                  (the caller has synchronised, or the library itself produced it in an earlier call of this
 *                  kind): no event, no wait.
 *        at exit   the call returns only after its own device work has completed, so whatever the caller
 *                  queues afterwards, on any stream, sees the results.
 *      Work the caller queues on OTHER streams before the call is the caller's to order (make `stream` wait for it).
 *  mjx_compose_plan_create_ex with MJX_PLAN_GROUP_BY_DENSITY reads the alpha tiles on the legacy default stream
 *  and is synchronous: tiles written on a non-blocking stream must be complete before the call. */
#define MJX_STREAM_IDLE ((mjx_stream_t)(intptr_t)-1)

/* ---- device plumbing (so that C callers need no HIP headers) -------------------------- */
/* MJX_VERSION of the header the library was built from.  Structs grow and arguments are added between versions (104: the `stream`
 * members; 105: mjx_plan_stats_t), so a caller built against another header must not go on: the Python mirror and the
 * libmodjpeg.h shim compare this with their own MJX_VERSION when they load the library. */
int         mjx_version(void);
int         mjx_device_count(void);                 /* number of HIP devices, 0 if none */
int         mjx_set_device(int ordinal);            /* hipSetDevice for the calling thread */
int         mjx_get_device(int *ordinal);           /* the calling thread's current device */
const char *mjx_error_string(int err);
int         mjx_malloc(void **dev, size_t bytes);   /* hipMalloc, 256-byte aligned */
int         mjx_free(void *dev);
int         mjx_memcpy_h2d(void *dev, const void *host, size_t bytes, mjx_stream_t s);
int         mjx_memcpy_d2h(void *host, const void *dev, size_t bytes, mjx_stream_t s);
int         mjx_stream_synchronize(mjx_stream_t s);

/* ---- masked compose: mj_compose_with_mask (reference compose.c:237) -------------------- */

/* One rectangle of one component plane of one image: what one iteration of the component loop
 * at reference compose.c:256-338 visits.  Tile block (l,k) is blended into plane block
 * (row0+l, col0+k); row0/col0 are the reference's height_offset/width_offset
 * (block_y*v_samp, block_x*h_samp, compose.c:264-265). */
typedef struct mjx_rect_job {
    int16_t     *coef;           /* DEVICE: component plane, updated in place */
    const float *tile;           /* DEVICE: rows*cols prepared image blocks */
    const float *alpha;          /* DEVICE: rows*cols prepared alpha blocks */
    int32_t      blocks_per_row; /* plane row pitch in blocks (width_in_blocks of the array) */
    int32_t      plane_rows;     /* plane height in blocks (bounds check only) */
    int32_t      row0, col0;     /* top-left block of the rectangle inside the plane */
    int32_t      rows, cols;     /* rectangle size in blocks == tile height/width in blocks */
    int32_t      qtable;         /* index into the plan's quant tables */
    int32_t      flags;          /* MJX_JOB_*, read by mjx_compose_plan_create_ex ONLY: the field occupies what was tail
                                    padding before version 101, which a caller that fills the struct member by member
                                    leaves indeterminate, so mjx_compose_plan_create ignores it.  Zero-initialise the
                                    struct (memset / = {0}) before handing it to the _ex form. */
} mjx_rect_job_t;
/* (mjx_compose_plan_create_ex only) tile and alpha point at int16_t[rows*cols][64] instead of float: the prepared JCOEFs themselves.  A
 * prepared tile is nothing but an int16 coefficient turned into a float (reference dropon.c:476-485;
 * alpha: `coef[0] += 1024` then (float)coef * K, :542-566), so the kernel can do that conversion
 * itself, bit for bit, and stream 512 instead of 768 bytes per block.  tile[i] holds coef[i], alpha[i]
 * holds coef[i] with the DC offset already added (as a short).  mjx_tiles_prepare writes this form
 * with MJX_PREP_TILES_I16.  All jobs of a plan must agree on it. */
#define MJX_JOB_TILES_I16 1

typedef struct mjx_compose_plan mjx_compose_plan_t;

/* Build a device-resident launch plan for `njobs` rectangles (any mix of images/components).
 * `jobs` and `qtables` (nq tables of 64 uint16, natural order) are HOST arrays, copied.
 * Rectangles must not overlap each other inside one plan. */
int mjx_compose_plan_create(mjx_compose_plan_t **plan, const mjx_rect_job_t *jobs, int njobs,
                            const uint16_t *qtables, int nq);
/* The same with options.
 * MJX_PLAN_GROUP_BY_DENSITY: the alpha tiles (which must hold their final contents) are scanned once
 * on the device for the zero pattern of every block (which of its 64 weights are not zero: the test of
 * reference convolve.c:32, made per block there), and inside every rectangle where it pays the blocks
 * that share a wavefront are chosen by what they need instead of by position: a wavefront executes a
 * term when ANY of its 8 blocks has a weight for it, so what it pays for is the UNION of their
 * patterns; blocks are sorted by density, rows and columns used and pattern, and cut into groups of
 * 8, so that a dense block no longer makes seven sparse neighbours wait for its 64 terms and a
 * block with a horizontal edge (one column of weights) does not share with one that has a vertical
 * edge (one row).  Purely a schedule: the results are identical, whatever alpha holds later.
 * MJX_PLAN_ALPHA_IS_FINAL (with the above): the caller promises that the alpha tiles keep the
 * contents they have now for as long as the plan is used.  Blocks whose 64 weights are all zero
 * then travel without their tile and alpha blocks: the reference skips every term of such a block
 * (Y stays 0), only the de-/re-quantisation of compose.c:277-286 / :327-336 is left, and that still runs.
 * Blocks whose ONLY weight is (0,0) - every block an opaque or evenly translucent part of an overlay
 * covers - travel without their alpha block (the schedule reads the weight once and keeps it in the run table),
 * and their one term runs without the two transpositions the general product needs.  Both kinds are streamed
 * by a lean kernel of their own (8 waves per SIMD, short runs) next to the blend kernel.  Runs of blocks none of whose weights
 * lies in a column l >= 4 (smooth masks) may go to a build of the blend kernel that keeps four z_l and the first 16 bytes of
 * every alpha row only (four workgroups per CU), runs none of whose weights lies in a row k >= 4 leave the second 128-byte line of
 * their alpha blocks unfetched, and a block without weights that comes out as it went in is not written back: also under this promise only.
 * MJX_PLAN_TILES_I16: every job's tiles are int16 coefficients (the same as MJX_JOB_TILES_I16 on every job). */
#define MJX_PLAN_GROUP_BY_DENSITY 1
#define MJX_PLAN_ALPHA_IS_FINAL   2
#define MJX_PLAN_TILES_I16        4
/* How the two kernels of a step are launched when a plan has both kinds of work (blocks that are blended: bound by the vector
 * ALU; blocks that need no blend: bound by the memory system).  By default a plan whose launches fill the device runs them SIDE
 * BY SIDE - a lean build of the blend kernel, a long-lived stream kernel in the registers it leaves on every CU, forked from
 * and joined to the caller's stream by events inside mjx_compose_run (legal in a stream capture) - and a small one runs them
 * one after the other.  MJX_PLAN_SERIAL: always one after the other; MJX_PLAN_SIDE_BY_SIDE: side by side whatever the size
 * (tests; plans with a plane or a tile array of 4 GiB or more always run them one after the other).  Purely a schedule: the
 * results are identical. */
#define MJX_PLAN_SERIAL           8
#define MJX_PLAN_SIDE_BY_SIDE     16
/* MJX_PLAN_NO_LOW_COLUMN_BUILD: blended runs whose weights all lie in columns 0..3 go to the general build of the blend kernel
 * like the others (three workgroups per CU instead of four; measurements, and the parity tests' way of sending smooth masks
 * through the general build).  Purely a schedule. */
#define MJX_PLAN_NO_LOW_COLUMN_BUILD 32
/* MJX_PLAN_TILES_ARE_FINAL (with MJX_PLAN_GROUP_BY_DENSITY | MJX_PLAN_ALPHA_IS_FINAL, float tiles): the caller promises the same
 * for the image tiles.  The scan then also reads them, and a block whose only weight is (0,0) = 0.25 exactly (what an opaque pixel
 * of an overlay compiles to: tiles.c / compose.c:139-157, 255/255 * 1/4... the reference's own normalisation) and whose 64 tile
 * values are whole numbers below 2^23 is an OPAQUE block: the reference's arithmetic for it,
 *     X = tile - (float)(coef * q);  Y = (X * 2) * 0.5 = X;  coef' = (coef * q + (int)Y) / q        (compose.c:277-336, convolve.c)
 * is exact in every step (|coef * q| < 2^23 as well), so coef' = (int)tile / q whatever the image held: the block is written from
 * its tile alone and the image's block is not read (388 bytes per block instead of 520).  With int16 tiles (MJX_PLAN_TILES_I16 /
 * MJX_JOB_TILES_I16) the tile values are whole numbers by type and MJX_PLAN_ALPHA_IS_FINAL is enough - schedules of
 * mjx_tile_schedule_create over int16 tiles included (260 bytes per block instead of 392).  Blocks with that weight whose tile
 * holds anything else (a fraction, 2^23 or more, NaN) stay with the single-weight blocks: the results are the reference's either
 * way. */
#define MJX_PLAN_TILES_ARE_FINAL 64
int mjx_compose_plan_create_ex(mjx_compose_plan_t **plan, const mjx_rect_job_t *jobs, int njobs,
                               const uint16_t *qtables, int nq, int flags);
/* The same with every choice of the schedule that is the caller's to make.  Zero-initialise the struct (memset / = {0}): a zero
 * field leaves the choice to the library, and the reserved fields must be zero.  None of them changes a result
 * (tests/test_gpu_fullsize.py::test_full_batch_schedule_independence_and_oracle_samples composes the 4096-image batch with runs
 * of 16, 1 and 5 octets and compares every block). */
typedef struct mjx_plan_options {
    int32_t flags;             /* MJX_PLAN_* */
    int32_t run_octets;        /* octets (8 blocks) one wavefront of the blend kernel works through in the body of a launch, 1..16.
                                  0: by the size of the launch (2..16: long runs amortise a wave's start, short ones fill the device
                                  when the launch is small).  A caller with many tiny rectangles may know better. */
    int32_t stream_run_octets; /* the same for the runs of blocks that need no blend (MJX_PLAN_ALPHA_IS_FINAL), 1..24; 0: 4 */
    int32_t tail_min_octets;   /* towards the end of a launch runs shrink with the work that is left, not below this; 0: 4; -1: never */
    float   beside_share;      /* side-by-side launches: this share (0 < s <= 1) of the stream runs goes to the kernel beside the blend
                                  kernel; 0: the library's estimate */
    int32_t reserved[11];
} mjx_plan_options_t;
int mjx_compose_plan_create_opts(mjx_compose_plan_t **plan, const mjx_rect_job_t *jobs, int njobs,
                                 const uint16_t *qtables, int nq, const mjx_plan_options_t *opts);
/* octets (8 blocks) of the plan that were formed by pattern rather than by position */
int64_t mjx_compose_plan_gather_octets(const mjx_compose_plan_t *plan);
/* what a plan looks like: `block_terms` = non-zero weights summed over the blocks, `union_terms` = terms
 * summed over the octets (size of the union of the 8 patterns), both 0 for plans made without a scan;
 * mean union size per octet = union_terms / (octets - zero_alpha_octets) */
typedef struct mjx_plan_stats {
    int64_t blocks, octets, gather_octets, zero_alpha_octets, runs;
    int64_t block_terms, union_terms;
    int64_t dc_only_octets; /* octets of blocks whose only weight is (0,0): their alpha blocks do not travel */
    int64_t union_terms_consecutive; /* what union_terms would be with every octet 8 consecutive blocks (over all of them) */
    int32_t resident_waves; /* wavefronts of the compose kernel the device holds at once: what runs are cut for */
    int32_t run_octets_max, run_octets_min;
    int32_t runs_beside; /* runs of blocks that need no blend that are streamed BESIDE the blend kernel (0: the kernels run one after the other) */
    int64_t low_column_octets; /* blended octets none of whose weights lies in a column l >= 4 (smooth masks) ... */
    int64_t runs_low_columns;  /* ... and the runs of them that go to the low-column build of the blend kernel (four workgroups per CU) */
    int64_t low_row_octets;    /* blended octets none of whose weights lies in a row k >= 4: with float tiles the second 128-byte line of
                                  their alpha blocks is not fetched (656 - 672 instead of 768 bytes per block) */
    /* (105) what making the plan cost, host wall clock in microseconds: the reference pays its tile compilation and its zero tests
     * inside every call (compose.c:155-157, convolve.c:32); a plan pays this once.  [0] the scan of the alpha tiles on the device
     * (MJX_PLAN_GROUP_BY_DENSITY; 0 without), [1] its results coming down, [2] the host's schedule (patterns -> octets -> runs),
     * [3] tables going up (runs, gather table, quantisation rows), [4] the whole call. */
    int32_t build_us[5];
    int32_t reserved;
    int64_t opaque_octets;  /* octets of opaque blocks (MJX_PLAN_TILES_ARE_FINAL; int16 tiles): written from the tile alone, not
                               counted in dc_only_octets */
} mjx_plan_stats_t;
int mjx_compose_plan_stats(const mjx_compose_plan_t *plan, mjx_plan_stats_t *out);
void mjx_compose_plan_destroy(mjx_compose_plan_t *plan);
/* total 8x8 blocks the plan composes per run */
int64_t mjx_compose_plan_blocks(const mjx_compose_plan_t *plan);
/* Enqueue one pass over all rectangles on `stream` (asynchronous; no host sync, no allocation:
 * safe inside hipGraph capture).  Host threads may run one plan at the same time (a side-by-side plan serialises its fork and
 * join internally; its side stream then runs the passes' stream kernels one after the other); the passes touch the same
 * planes, so ordering them against each other stays the callers' business.  Two restrictions: a plan runs on the device it was made
 * on (the calling thread's current device must be that one: MJX_ERR_BAD_ARG otherwise), and while a stream capture that contains
 * a side-by-side plan's launches is open (its side stream is part of the capture until the join), no other thread may run the same
 * plan - capture it, end the capture, then share it. */
int mjx_compose_run(const mjx_compose_plan_t *plan, mjx_stream_t stream);

/* Per-image callers (the libmodjpeg.h shim) hold everything in HOST memory.  One call = one
 * staging allocation, asynchronous H2D of every rectangle's blocks and tiles, ONE launch over all
 * rectangles, D2H of the blocks.  `rows_ptr` points at the first plane row the rectangle touches
 * (rows * blocks_per_row blocks, updated in place); a rectangle much narrower than the plane travels
 * alone (gathered into, and scattered back from, pinned memory), a wide one as whole rows. */
typedef struct mjx_host_rect {
    int16_t        *rows_ptr;       /* HOST, in/out */
    const float    *tile;           /* HOST, rows*cols blocks */
    const float    *alpha;          /* HOST, rows*cols blocks */
    const uint16_t *q;              /* HOST[64] */
    int32_t         blocks_per_row; /* plane pitch in blocks */
    int32_t         col0;           /* first block column of the rectangle */
    int32_t         rows, cols;
    int32_t         flags;          /* MJX_RECT_* */
    int32_t         reserved;
    const struct mjx_tile_schedule *schedule; /* NULL, or how the blocks of THESE tiles are best shared out (below) */
    mjx_stream_t    stream;         /* the caller's stream of the ordering contract above, for the DEVICE memory of this
                                       rectangle (MJX_RECT_ROWS_ON_DEVICE / MJX_RECT_TILES_ON_DEVICE); ignored without them */
} mjx_host_rect_t;
/* tile and alpha are DEVICE pointers (tiles kept resident across calls, e.g. one dropon composed
 * into many images): only the plane rows cross PCIe */
#define MJX_RECT_TILES_ON_DEVICE 1
/* rows_ptr is a DEVICE pointer into a plane that lives on the device (an image decoded there and
 * kept there): the rows are blended in place and nothing but the descriptors crosses PCIe */
#define MJX_RECT_ROWS_ON_DEVICE 2
/* tile and alpha hold int16 coefficients (MJX_JOB_TILES_I16; what mjx_tiles_prepare writes with
 * MJX_PREP_TILES_I16): all rectangles of one call must agree */
#define MJX_RECT_TILES_I16 4
int mjx_compose_host(const mjx_host_rect_t *rects, int n);

/* A schedule for ONE set of prepared tiles (rows x cols blocks, DEVICE): the zero patterns of the alpha
 * blocks are read once, and the blocks are grouped as MJX_PLAN_GROUP_BY_DENSITY | MJX_PLAN_ALPHA_IS_FINAL
 * describe.  It depends on the tiles only, not on the image they are composed into, so whoever keeps
 * tiles resident (the drop-in's tile cache: one overlay, many images, reference compose.c:155-177) keeps
 * the schedule with them and hands both to mjx_compose_host.  The alpha tiles must not change while
 * the schedule is in use.  flags: MJX_JOB_TILES_I16 when the tiles are int16. */
typedef struct mjx_tile_schedule mjx_tile_schedule_t;
int     mjx_tile_schedule_create(mjx_tile_schedule_t **sched, const void *alpha /* DEVICE */, int rows, int cols, int flags,
                                 mjx_stream_t after /* ordering contract: the stream alpha was written on */);
void    mjx_tile_schedule_destroy(mjx_tile_schedule_t *sched);
/* octets formed by pattern (0: the tiles are uniform enough for consecutive octets) */
int64_t mjx_tile_schedule_gather_octets(const mjx_tile_schedule_t *sched);
int     mjx_tile_schedule_stats(const mjx_tile_schedule_t *sched, mjx_plan_stats_t *out);
/* The host-buffer calls borrow a staging arena (grow-only device buffer, pinned descriptor buffer,
 * stream) from a small process-wide pool, so that a steady stream of per-image calls from any
 * number of threads allocates nothing.  mjx_release_thread_cache() frees the idle arenas (and the spare host buffers the
 * plan builder keeps between plans: at most 160 MB). */
int mjx_release_thread_cache(void);
/* pinned (page-locked) host memory: plane rows gathered into it upload at full PCIe rate */
int mjx_malloc_host(void **host, size_t bytes);
int mjx_free_host(void *host);

/* single rectangle given the whole plane (blocks_per_row * plane_rows blocks) */
int mjx_compose_rect_host(int16_t *plane, int blocks_per_row, int plane_rows, int row0, int col0, int rows, int cols,
                          const uint16_t *q, const float *tile, const float *alpha);

/* ---- tile preparation on the device -----------------------------------
 * What mj_compile_dropon does through libjpeg (reference dropon.c:325-576, image.c:257-347): the
 * crop window of an overlay (3 bytes per pixel image, 3 bytes per pixel replicated alpha), placed at
 * (off_x, off_y) inside a canvas of whole MCUs, becomes one contiguous array of prepared float
 * tiles per component for image and alpha, identical to the ones the linked libjpeg (IJG 9d) would
 * have produced.  Supported: 3 components; target YCbCr from an RGB or YCbCr overlay, target RGB
 * from an RGB overlay; every component at full resolution or subsampled by a whole number up to 4 in either
 * direction (4:4:4, 4:2:2, 4:4:0, 4:2:0; 4:1:1, 4:1:0, 1x4 or 3x1 luma ...: libjpeg's box filters - h2v1, h2v2, int_downsample -
 * in front of the transform that folds one halving of an even ratio); 1 component, target grayscale from an RGB or YCbCr
 * overlay.  Anything else (fractional ratios, which libjpeg does not compress either; four components) returns
 * MJX_ERR_BAD_ARG before touching the device, and the caller prepares the tiles with libjpeg.
 * Block counts: canvas = crop + offset rounded up to whole MCUs; component c has
 * canvas_w / (8 * hmax / h_samp[c]) columns and canvas_h / (8 * vmax / v_samp[c]) rows. */
#define MJX_PREP_SOURCE_RGB 1
#define MJX_PREP_SOURCE_YCC 5 /* = MJ_COLORSPACE_RGB / MJ_COLORSPACE_YCC of libmodjpeg.h */
#define MJX_PREP_TARGET_GRAY 1 /* one component: libjpeg's Y of an RGB overlay, the first component of a YCbCr one */
#define MJX_PREP_TARGET_RGB 2
#define MJX_PREP_TARGET_YCC 3 /* = JCS_GRAYSCALE / JCS_RGB / JCS_YCbCr of jpeglib.h */
typedef struct mjx_tileprep {
    const unsigned char *image;      /* HOST: width x height pixels, 3 bytes each */
    const unsigned char *alpha;      /* HOST: the same geometry, alpha replicated into 3 bytes */
    int32_t              width, height;
    int32_t              crop_x, crop_y, crop_w, crop_h; /* visible window of the overlay */
    int32_t              off_x, off_y;                   /* its offset inside the first MCU */
    int32_t              source, target;                 /* MJX_PREP_SOURCE_*, MJX_PREP_TARGET_* */
    int32_t              ncomp;                          /* 3, or 1 for MJX_PREP_TARGET_GRAY */
    int32_t              h_samp[4], v_samp[4];           /* the target image's sampling factors */
    float               *tile[4];                        /* DEVICE out, rows*cols*64 floats each */
    float               *alpha_tile[4];
    int32_t              flags;                          /* MJX_PREP_* */
    int32_t              reserved;
    mjx_stream_t         stream;                         /* ordering contract: the stream on which tile[] / alpha_tile[] were last
                                                            used (they are overwritten) */
} mjx_tileprep_t;
/* write the tiles as int16 coefficients (the form MJX_JOB_TILES_I16 consumes): tile[] / alpha_tile[]
 * then point at int16_t[rows*cols][64] */
#define MJX_PREP_TILES_I16 1
/* `image` points at the visible window already packed (crop_w * 3 bytes per row, crop_h rows), immediately
 * followed by the packed window of alpha (alpha == image + crop_w * 3 * crop_h), in PINNED host memory
 * (mjx_malloc_host): it is uploaded as it lies, without a staging copy.  width == crop_w, height == crop_h,
 * crop_x == crop_y == 0.  (The drop-in's tile cache keeps such a copy of the window anyway, to compare
 * later callers' pixels with.) */
#define MJX_PREP_WINDOW_PACKED 2
int mjx_tiles_prepare(const mjx_tileprep_t *p);

/* ---- one (k,l) term: mj_convolve (reference convolve.c:29), test seam ------------------- */
/* y[b][64] += V_k(H_l(x[b][64])) * w for b < nblocks; x, y DEVICE float arrays. */
int mjx_convolve(const float *x, float *y, float w, int k, int l, int nblocks, mjx_stream_t stream);

/* ---- per-block effects (reference effect.c) -------------------------------------------- */

/* One component plane (true width_in_blocks x height_in_blocks inside a pitch of
 * blocks_per_row): DC' = clamp((short)((short)(DC*q0) + value), -2047, 2047) / q0 —
 * mj_effect_luminance on component 0, mj_effect_tint on components 1 and 2
 * (reference effect.c:207-217, :143-153, :167-177). */
typedef struct mjx_dc_job {
    int16_t *coef; /* DEVICE */
    int32_t  blocks_per_row, rows, cols;
    int32_t  value;
    uint16_t q0;
    uint16_t reserved;
} mjx_dc_job_t;

/* all jobs in one launch (fused luminance + tint over a batch = one pass over the DCs) */
int mjx_effect_dc(const mjx_dc_job_t *jobs /* HOST */, int njobs, mjx_stream_t stream);

typedef struct mjx_dc_plan mjx_dc_plan_t;
int  mjx_dc_plan_create(mjx_dc_plan_t **plan, const mjx_dc_job_t *jobs, int njobs);
void mjx_dc_plan_destroy(mjx_dc_plan_t *plan);
int  mjx_dc_plan_run(const mjx_dc_plan_t *plan, mjx_stream_t stream);

/* zero whole blocks (mj_effect_grayscale on components >= 1, effect.c:44-65; keep_dc = 0) or
 * coefficients 1..63 (mj_effect_pixelate, effect.c:82-111; keep_dc = 1) of rows x cols blocks */
int mjx_effect_zero(int16_t *coef /* DEVICE */, int blocks_per_row, int rows, int cols, int keep_dc, mjx_stream_t stream);

/* HOST-buffer forms of the effects and of the mj_convolve seam for per-image callers (H2D,
 * kernel, D2H; synchronous) */
int mjx_effect_dc_host(int16_t *plane, int blocks_per_row, int rows, int cols, uint16_t q0, int value);
int mjx_effect_zero_host(int16_t *plane, int blocks_per_row, int rows, int cols, int keep_dc);
int mjx_convolve_host(const float *x, float *y, float w, int k, int l, int nblocks);
/* the same effects on a plane that is resident on the DEVICE (synchronous, own stream; `after`: the ordering contract
 * at the top of this header - the stream the plane was last touched on, NULL, or MJX_STREAM_IDLE) */
int mjx_effect_dc_device(int16_t *plane, int blocks_per_row, int rows, int cols, uint16_t q0, int value, mjx_stream_t after);
int mjx_effect_zero_device(int16_t *plane, int blocks_per_row, int rows, int cols, int keep_dc, mjx_stream_t after);

/* ---- baseline Huffman entropy encoder (write side) ------------------
 * Replaces, for fixed tables, what the reference gets from libjpeg at image.c:194-196
 * (jpeg_write_coefficients + jpeg_finish_compress).  Encodes the entropy-coded segment of ONE sequential (baseline) JPEG scan - all components
 * interleaved, or the single component of a grayscale image - exactly as libjpeg's
 * jpeg_write_coefficients() does with fixed Huffman tables (jctrans.c compress_output +
 * jchuff.c encode_mcu_huff): MCU order, dummy blocks at the right/bottom edge (AC zero, DC of
 * the previous block), DC prediction, run-length/size symbols with ZRL and EOB, 0xFF00 byte
 * stuffing, final padding with 1-bits.  The caller supplies the coefficient planes (HOST), the
 * geometry and the derived code tables; markers (headers, EOI) stay with the caller. */
typedef struct mjx_huff_table {
    uint32_t code[256]; /* Huffman code of symbol s, right-aligned */
    uint8_t  size[256]; /* its length in bits, 0 = symbol not in the table */
} mjx_huff_table_t;

typedef struct mjx_huff_component {
    const int16_t *plane;           /* HOST (DEVICE with MJX_HUFF_PLANES_ON_DEVICE): int16[rows][blocks_per_row][64], natural order */
    int32_t        blocks_per_row;  /* array pitch in blocks (>= width_in_blocks) */
    int32_t        width_in_blocks; /* true size: blocks beyond it inside an MCU are dummies */
    int32_t        height_in_blocks;
    int32_t        h_samp, v_samp;  /* blocks per MCU in x and y (1,1 for a single-component scan) */
    int32_t        dc_tbl, ac_tbl;  /* indices into dc[] / ac[] */
    /* DC effects still pending on this plane (an image that stays on the device between read and write:
     * mj_effect_luminance / mj_effect_tint there only note their value): the encoder applies them, in
     * order, to the DC of every real block as it reads it - DC' = fx(value[n-1], .. fx(value[0], DC)),
     * fx = the arithmetic of mjx_effect_dc with quantiser dc_fx_q0 (reference effect.c:207-217) - so the
     * separate pass over the plane (one 128-byte line read and written per 2-byte DC) never runs.  The
     * plane itself is not changed.  dc_fx_n = 0: none. */
    int32_t        dc_fx_n;         /* 0 .. MJX_HUFF_MAX_DC_FX */
    int32_t        dc_fx_value[4];
    uint16_t       dc_fx_q0, dc_fx_reserved; /* dc_fx_reserved must be 0: zero-initialise the struct (memset / = {0}) */
} mjx_huff_component_t;
#define MJX_HUFF_MAX_DC_FX 4

typedef struct mjx_huff_scan {
    int32_t                 ncomp;          /* 1..4 */
    int32_t                 mcus_x, mcus_y; /* MCUs per row / MCU rows */
    mjx_huff_component_t    comp[4];
    const mjx_huff_table_t *dc[4];
    const mjx_huff_table_t *ac[4];
    int32_t                 max_coef_bits;  /* 10 for 8-bit JPEG: larger categories are an error as in libjpeg */
    int32_t                 flags;          /* MJX_HUFF_* */
    struct mjx_huff_spec   *optimal_dc[4];  /* MJX_HUFF_OPTIMIZE: out, the tables that were built and used ... */
    struct mjx_huff_spec   *optimal_ac[4];  /* ... for every table index a component refers to (caller's storage) */
    mjx_stream_t            stream;         /* MJX_HUFF_PLANES_ON_DEVICE: the caller's stream of the ordering contract at the top of
                                               this header (the planes are READ after everything queued on it; NULL = legacy default
                                               stream, MJX_STREAM_IDLE = nothing pending).  The call returns after its device work. */
} mjx_huff_scan_t;
/* the component planes are DEVICE pointers (an image decoded on the device and kept there): no
 * coefficient crosses PCIe, only the compressed bytes come back */
#define MJX_HUFF_PLANES_ON_DEVICE 1
/* libjpeg's optimize_coding (the reference's MJ_OPTION_OPTIMIZE, image.c:170): dc[]/ac[] are ignored,
 * the symbol statistics of this image are gathered on the device, the optimal tables are built as
 * libjpeg 9d builds them (mjx_huff_optimal_table) and returned through optimal_dc[]/optimal_ac[] for
 * the caller's DHT segments, and the scan is encoded with them */
#define MJX_HUFF_OPTIMIZE 2

/* *out is malloc()ed (caller free()s); returns MJX_ERR_BAD_ARG when a coefficient does not fit
 * the JPEG symbol alphabet (libjpeg: JERR_BAD_DCT_COEF) or a needed symbol has no code. */
int mjx_huff_encode_host(const mjx_huff_scan_t *scan, unsigned char **out, size_t *outlen);

/* Optimal Huffman table for a symbol histogram, as libjpeg 9d's jpeg_gen_optimal_table builds it for
 * optimize_coding (jchuff.c; ITU-T T.81 K.2 with code lengths limited to 16 bits, the all-ones code
 * reserved, symbols of equal length ordered by falling frequency).  Pure host code, no device.
 * freq[s] = occurrences of symbol s (0..255); returns MJX_ERR_BAD_ARG when no symbol occurs. */
typedef struct mjx_huff_spec mjx_huff_spec_t;
int mjx_huff_optimal_table(const uint32_t freq[256], mjx_huff_spec_t *table);

/* ---- baseline Huffman entropy decoder (row f-4, read side) --------------------------------
 * Replaces, for sequential Huffman files, what the reference gets from libjpeg at image.c:79
 * (jpeg_read_coefficients).  Decodes the entropy-coded segment of one sequential interleaved (or single-component) scan into
 * coefficient planes, as libjpeg's jpeg_read_coefficients() does (jdhuff.c decode_mcu): every block
 * slot of every MCU is stored, including the padding blocks of partial MCUs, into arrays of
 * mcus_x*h_samp by mcus_y*v_samp blocks.  The bitstream is sequential; it is cut into fixed-size
 * chunks that are decoded speculatively and re-synchronised iteratively until every chunk starts
 * exactly where its predecessor ended (self-synchronisation of Huffman codes), then decoded once
 * more for real.  `data` points at the first byte after the SOS header, as in the file: 0xFF00
 * stuffing and RSTn markers included; the scan must end with EOI inside `nbytes`. */
struct mjx_huff_spec {
    uint8_t bits[16];  /* number of codes of length 1..16 (DHT) */
    uint8_t vals[256]; /* symbols in code order */
};

typedef struct mjx_huff_dcomponent {
    int16_t *plane;          /* HOST (DEVICE with the flag), out: int16[mcus_y*v_samp][blocks_per_row][64], written whole (pitch padding = 0) */
    int32_t  blocks_per_row; /* >= mcus_x*h_samp */
    int32_t  h_samp, v_samp;
    int32_t  dc_tbl, ac_tbl;
} mjx_huff_dcomponent_t;

typedef struct mjx_huff_dscan {
    int32_t                ncomp, mcus_x, mcus_y;
    mjx_huff_dcomponent_t  comp[4];
    const mjx_huff_spec_t *dc[4];
    const mjx_huff_spec_t *ac[4];
    const uint8_t         *data;             /* HOST: scan data as in the file */
    size_t                 nbytes;           /* bytes available at `data` (may extend past EOI) */
    int32_t                restart_interval; /* MCUs per restart segment (DRI), 0 = none */
    int32_t                flags;            /* MJX_HUFF_PLANES_ON_DEVICE: the planes are DEVICE pointers and stay there */
    mjx_stream_t           stream;           /* MJX_HUFF_PLANES_ON_DEVICE: the caller's stream of the ordering contract at the top of this
                                                header: the planes are zeroed and WRITTEN after everything queued on it (a fill, an
                                                earlier reader); NULL = legacy default stream, MJX_STREAM_IDLE = nothing pending.  The
                                                call returns after the planes are complete. */
} mjx_huff_dscan_t;

/* MJX_ERR_BAD_ARG when the data is not a clean baseline scan of this geometry: markers other than
 * in-sequence RSTn before EOI, a wrong number of restart segments or blocks, codes or categories
 * outside the tables.  The caller then uses its ordinary decoder, which owns those semantics. */
int mjx_huff_decode_host(const mjx_huff_dscan_t *scan);

/* The decoder's first step on its own (pure host code, no device; a test seam): scan data as it stands in
 * the file -> the bytes without 0xFF00 stuffing and RSTn markers, and the offset (in `out`) at which every
 * restart segment starts (segment_offsets[0] = 0).  `out` needs nbytes + 64 bytes.  form 0: what the
 * decoder uses on this host (AVX-512 byte compress or AVX2 when the CPU has them), 1: the portable form, 2: the AVX2
 * form, 3: the AVX-512 form (MJX_ERR_NO_DEVICE when the CPU lacks it).  MJX_ERR_BAD_ARG: not a plain run of [segment RSTn]* segment EOI. */
int mjx_huff_unstuff(const uint8_t *data, size_t nbytes, uint8_t *out, size_t *outlen, uint64_t *segment_offsets, int64_t max_segments, int64_t *nsegments,
                     int form);

/* Another host-only test seam: the tables the decoder's wave-per-chunk rounds look symbols up in, as the host builds them
 * from the scan's Huffman tables (scan->data and the planes are not looked at).  out: 6 first-level tables of 1024 words —
 * 0, 1: the AC tables in the order the components first name them, 2..5: the DC tables — then info[2] second-level tables of
 * 64 words.  A first-level word is either the symbol's field, shifted into place, or 0x80000000 | offset of a second-level
 * table (in words from the end of the first level) for codes longer than 10 bits.  Field of AC table a: total length
 * (code + value bits) << 8a | zigzag advance << (16 + 8a), EOB's advance being 63; not a code: advance 63, length 0.
 * Field of DC table d: total length << 8d; not a code: 0.  info[0], info[1]: AC and DC tables in use; info[3]: 3 bits per
 * block slot of the MCU, DC field index | AC field index << 2.  Returns MJX_ERR_BAD_ARG for tables that are not prefix codes,
 * for more than two AC tables (such scans take the lane-per-chunk rounds) or when out_words is too small. */
int mjx_huff_wave_tables(const mjx_huff_dscan_t *scan, uint32_t *out, size_t out_words, int32_t info[4]);

/* ---- synthetic workload generators (bench / test support; definition: mjx_synth.h) ------
 * The *_batch forms fill one array of a whole batch with ONE launch: image i of the batch (global
 * index image0 + i) starts image_stride_blocks * 64 elements after image i-1.  regime: MJX_ALPHA_*
 * of mjx_synth.h (1 uniform, 2 smooth, 3 dense, 4 mixed).  mjx_synth_coef* return after the
 * generator has finished; the float generators are asynchronous on `stream`. */
int mjx_synth_coef_batch(int16_t *dst /* DEVICE */, size_t image_stride_blocks, uint64_t image0, uint32_t nimages, uint32_t comp, uint64_t first_block,
                         size_t nblocks, const uint16_t *q /* HOST[64] */, mjx_stream_t stream);
int mjx_synth_tile_batch(float *dst, size_t image_stride_blocks, uint64_t image0, uint32_t nimages, uint32_t comp, uint64_t first_block, size_t nblocks,
                         mjx_stream_t stream);
int mjx_synth_alpha_batch(float *dst, size_t image_stride_blocks, uint64_t image0, uint32_t nimages, uint32_t comp, uint64_t first_block, size_t nblocks,
                          int regime, mjx_stream_t stream);
/* the same tiles as int16 coefficients (MJX_JOB_TILES_I16): kind 0 = image tile, 1 = alpha tile of `regime` */
int mjx_synth_tiles_i16_batch(int16_t *dst, int kind, size_t image_stride_blocks, uint64_t image0, uint32_t nimages, uint32_t comp, uint64_t first_block,
                              size_t nblocks, int regime, mjx_stream_t stream);
/* one image */
int mjx_synth_coef(int16_t *dst /* DEVICE */, uint64_t image, uint32_t comp, uint64_t first_block, size_t nblocks,
                   const uint16_t *q /* HOST[64] */, mjx_stream_t stream);
int mjx_synth_tile(float *dst, uint64_t image, uint32_t comp, uint64_t first_block, size_t nblocks, mjx_stream_t stream);
int mjx_synth_alpha(float *dst, uint64_t image, uint32_t comp, uint64_t first_block, size_t nblocks, int regime, mjx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MJX_H_ */
