/*
 * gsr.h - C ABI of libgsr.so, the MI355X (gfx950) Gaussian-splat rasterizer.
 *
 * The reference has no FFI: its render path is the Python method
 * GaussianRenderer.render (GaussianRenderer.py:52-113) built from ATen ops.  Each entry point
 * below replaces the group of reference statements cited in its comment; the Python
 * autograd.Function in 3d-gaussian-splatting_amd/rasterizer.py binds them with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - all buffers (inputs, outputs, workspaces) are allocated by the caller; the library
 *     allocates nothing and keeps no state between calls;
 *   - every kernel is enqueued on `stream` (a hipStream_t passed as void*); no call blocks;
 *   - return value: 0 on success, a negative GSR_ERR_* code otherwise (gsr_last_error() gives
 *     the text for the calling thread);
 *   - "N-indexed" = one row per Gaussian of the model (size, not capacity);
 *   - images are [width][height][3] fp32, x (width) is the slow axis (GaussianRenderer.py:84).
 */
#ifndef GSR_H
#define GSR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_OK 0
#define GSR_ERR_ARG (-1)
#define GSR_ERR_SINGULAR (-2) /* reported through gsr_counters.error, see below */
#define GSR_ERR_LAUNCH (-3)

/* One view.  The two matrices are built on the host (Utils/Camera.py:25-28). */
typedef struct gsr_view {
    int32_t n;           /* Gaussians */
    int32_t sh_degree;   /* 0..3 (GaussianModel.curr_sh_degree) */
    int32_t width;       /* cam.width  (slow axis) */
    int32_t height;      /* cam.height (fast axis) */
    int32_t tile_length; /* edge of a reference tile, GaussianRenderer.py:52 */
    float w2m[16];       /* cam.world2model_matrix, row-major */
    float proj[16];      /* cam.projection_matrix, row-major */
    float cam_pos[3];    /* cam.transform.position */
} gsr_view;

/* Device-side counters written by gsr_preprocess_fwd / gsr_bin_offsets (8 x u32; 8-byte aligned). */
typedef struct gsr_counters {
    uint32_t num_visible; /* M = mask.sum() */
    uint32_t num_pairs;   /* I = sum over tiles of list length */
    uint32_t error;       /* bit 0: a Gaussian that lands in a tile has a singular 2-D covariance
                             (the reference raises in torch.linalg.inv, Utils/Probability.py:15);
                             bit 1 (gsr_bin_offsets): the number of (tile, Gaussian) pairs does not fit
                             32 bits - num_pairs and the slots are then meaningless and the caller must
                             stop (the reference would build lists of that total length,
                             GaussianRenderer.py:89-97) */
    uint32_t reserved0;
    uint32_t pairs_wide[2]; /* written by gsr_preprocess_fwd[_views] (8-byte aligned, low word first): the 64-bit sum
                             of the view's tile counts = the number of pairs gsr_bin_offsets will report, known
                             before the depth order exists.  A caller that sizes its per-pair buffers on the host
                             reads it back while the sort runs; a sum of 2^32 or more is the error of bit 1 above */
    uint32_t reserved[2];
} gsr_counters;

/* Per-Gaussian splat record produced by stage 1 and consumed by the compositing kernels. */
/* x, y, A', B', D', opacity, r, g, b, a, (b+c)/2, d   (48 bytes).  [a b; c d] = inverse 2-D
 * covariance; A' = k a, B' = k (b+c), D' = k d with k = -log2(e)/2, so G = exp2(A'dx^2+B'dxdy+D'dy^2) */
#define GSR_REC_FLOATS 12
/* The same record in per-tile list order, as the compositing kernels stream it: 16 dwords (64 bytes,
 * one cache line, fetched with one scalar load).  Values used as the scalar operand of packed
 * instructions sit at even dwords (an aligned register pair needs no copy):
 *   0 y   1 x   2 D'   3 A'   4 opacity   5 B'   6 r   7 a   8 g   9 slot (u32)   10 b
 *   11 flags (u32)   12 (b+c)/2   13 0   14 d   15 0
 * When the tile is a single 16x16 block (tile_length <= 16) the two hint bits ride in dword 9 above the slot
 * (slot | flags << 30; slots stay below 2^26) and dwords 11, 13, 15 hold b/100, r/100, g/100: the colour times the
 * lower clamp of alpha (0.01, GaussianRenderer.py:104-105), with which the compositing kernels blend the halves on
 * which alpha sits at that clamp - 0.01 T colour as T times these - without forming the weight.
 * slot  = the pair's emission slot (row of pairgrad its gradient record goes to);
 * flags = hints for the compositing waves: a tile of edge L is composited by S*S waves, S = ceil(L/16),
 *         one per 16x16 block (block c = bx * S + by); bits 2c / 2c+1 are set when alpha is 0.01 on
 *         every pixel of the top / bottom 8 rows of block c (S <= 4; for larger tiles all 32 bits
 *         carry the answer for the whole tile). */
#define GSR_STREAM_DWORDS 16
/* Per-(tile,Gaussian) gradient record produced by the compositing backward. */
/* With h = -2 dL/d(exponent) per pixel, d = centre - pixel and v = inverse covariance * d:
 * sum h v_x, sum h v_y, sum h v_x v_x, sum h v_x v_y, sum h v_y v_y, dL/d(opacity), dL/d(r), dL/d(g),
 * then FOUR partial sums whose total is dL/d(b) */
#define GSR_PAIRGRAD_FLOATS 12

const char *gsr_last_error(void);
/* Version of the library in the low byte, above it the row widths the caller allocates by:
 * GSR_VERSION | GSR_REC_FLOATS << 8 | GSR_STREAM_DWORDS << 16 | GSR_PAIRGRAD_FLOATS << 24.  A caller compiled or written
 * against other widths must refuse the library (a library that writes rows wider than the caller allocated them
 * writes out of bounds): compare with GSR_VERSION_WORD. */
#define GSR_VERSION 2
#define GSR_VERSION_WORD (GSR_VERSION | GSR_REC_FLOATS << 8 | GSR_STREAM_DWORDS << 16 | GSR_PAIRGRAD_FLOATS << 24)
int gsr_version(void);

/* ---- stage 1: per-Gaussian ------------------------------------------------------------
 * Replaces: GaussianModel.py:42-65,99-108 (activations, cov3d), Utils/Transform.py:41-100,
 * Utils/Camera.py:89-117 + 74-87 (project, cull, viewport, cov2d), GaussianRenderer.py:70-77
 * (opacity, SH colour, radius), Utils/Screen.py:30-36 (integer bounding boxes).
 * Outputs (N-indexed): mask u8, depth_key u32 (bits of the NDC depth, 0xFFFFFFFF when culled),
 * pos2d[N,3], cov2d[N,4], radius[N], bbox[N,4] i32 (xmin,ymin,xmax,ymax), ntiles[N] u32,
 * rec[N,12].  counters->num_visible, ->pairs_wide and ->error are accumulated (zero them first).
 * screen_coords may be NULL (= zeros, GaussianRenderer.py:64). */
int gsr_preprocess_fwd(const gsr_view *view_host, const float *coords, const float *scale_raw,
                       const float *rot_raw, const float *opac_raw, const float *sh,
                       const float *screen_coords, uint8_t *mask, uint32_t *depth_key,
                       float *pos2d, float *cov2d, float *radius, int32_t *bbox, uint32_t *ntiles,
                       float *rec, gsr_counters *counters, void *stream);

/* gsr_preprocess_fwd for num_views (1..8) views of the SAME Gaussians in one pass over the parameters - the
 * views of one optimisation step (Trainer.py:54-57 renders one; a view minibatch renders several against the
 * same model).  The parameters are read once and the view-independent statements (GaussianModel.py:42-65,
 * 99-108) are evaluated once; per view the results equal gsr_preprocess_fwd's bit for bit.  views_host: the
 * views (same n and sh degree).  Every output array holds the views one after the other: view k's rows
 * start k * n rows in (mask[num_views*N], pos2d[num_views*N,3], ...); counters[num_views].  screen_coords
 * are taken as zeros.  pos2d and cov2d may each be NULL: no later call reads them, and a caller that does not
 * either saves their 28 bytes per Gaussian and view. */
int gsr_preprocess_fwd_views(const gsr_view *views_host, uint32_t num_views, const float *coords,
                             const float *scale_raw, const float *rot_raw, const float *opac_raw, const float *sh,
                             uint8_t *mask, uint32_t *depth_key, float *pos2d, float *cov2d, float *radius,
                             int32_t *bbox, uint32_t *ntiles, float *rec, gsr_counters *counters, void *stream);

/* Backward of stage 1.  Replaces the autograd graph of the statements listed above
 * (Trainer.py:57).  pairgrad (written by gsr_composite_bwd for the same view) holds the gradient records of
 * the (tile, Gaussian) pairs at their emission slots: Gaussian g owns the contiguous slots
 * slot_base[g] .. slot_base[g]+ntiles[g], gsr_pairgrad_parts(tile_length) rows each.
 * A Gaussian's records are summed in a fixed order, so the result is reproducible run to run.
 * Writes (or, with accumulate != 0, adds to) the dense gradients of the raw parameters
 * g_coords[N,3] g_scale[N,3] g_rot[N,4] g_opac[N] g_sh[N,3,16], and always overwrites
 * g_screen[N,3] (the gradient of this view's screen_coords leaf).  g_sh beyond the active degree,
 * g_screen[:,2] and the rows of culled Gaussians are 0.
 * workspace: gsr_preprocess_bwd_workspace_bytes(n) (the per-Gaussian sums). */
size_t gsr_preprocess_bwd_workspace_bytes(uint32_t n);
int gsr_preprocess_bwd(const gsr_view *view_host, const float *coords, const float *scale_raw,
                       const float *rot_raw, const float *opac_raw, const float *sh,
                       const uint8_t *mask, const uint32_t *slot_base, const uint32_t *ntiles,
                       const float *pairgrad, float *g_coords, float *g_scale, float *g_rot,
                       float *g_opac, float *g_sh, float *g_screen, int accumulate, void *workspace,
                       size_t workspace_bytes, void *stream);

/* The two halves of gsr_preprocess_bwd, for view minibatches (the reference renders one view per
 * optimiser step, Trainer.py:52-57; a minibatch sums the gradients of several renders of the same
 * parameters).  gsr_pair_reduce sums one view's gradient records per Gaussian into gsum[N,12]
 * (the nine sums of a record; floats 9..11 are 0); parts = gsr_pairgrad_parts(tile_length) of the
 * view whose compositing backward wrote pairgrad.  gsr_preprocess_bwd_views then chains the sums of
 * num_views views (host arrays of per-view device pointers) to the raw parameters in one pass: the
 * dense gradients are written (accumulate == 0) or added to (accumulate != 0) ONCE for all views,
 * where per-view calls of gsr_preprocess_bwd re-read and re-write them for every view.
 * g_screen_host[v][N,3] receives view v's screen_coords gradient.  All views must share n and
 * sh_degree.  At most GSR_MAX_VIEWS views go into one kernel launch; more are processed in groups. */
#define GSR_MAX_VIEWS 8
int gsr_pair_reduce(uint32_t n, uint32_t parts, const uint8_t *mask, const uint32_t *slot_base,
                    const uint32_t *ntiles, const float *pairgrad, float *gsum, void *stream);
int gsr_preprocess_bwd_views(const gsr_view *views_host, uint32_t num_views, const float *coords,
                             const float *scale_raw, const float *rot_raw, const float *opac_raw,
                             const float *sh, const uint8_t *const *mask_host, const float *const *gsum_host,
                             float *g_coords, float *g_scale, float *g_rot, float *g_opac, float *g_sh,
                             float *const *g_screen_host, float *const *g_color_host, int accumulate,
                             void *stream);

/* The sh gradient of a view minibatch is a sum over the views of outer products: per Gaussian and
 * colour channel, dL/d(colour of the view) times the view's SH basis (where the colour passes its
 * clamp, GaussianRenderer.py:74).  Ranks that hold the same parameters therefore need not exchange
 * the 48 sh gradient floats per Gaussian: gsr_preprocess_bwd_views with g_color_host != NULL also
 * writes g_color_host[v][N,3], the colour gradient of view v as it reaches the clamp (zero rows where
 * the Gaussian is culled), and with g_sh == NULL leaves the sh gradient out (the colour gradient of view v is also
 * columns 6..8 of gsum_host[v], so a caller that wants it before this pass has run copies it from there and passes
 * g_sh == NULL and g_color_host == NULL); after the ranks have
 * gathered the colour gradients of ALL views of the minibatch, gsr_sh_grad_views rebuilds
 * g_sh[N,3,16] (written, or with accumulate != 0 added to) with the arithmetic and the view order
 * gsr_preprocess_bwd_views itself uses, so the result does not depend on the number of ranks.
 * views_host: every view of the minibatch in minibatch order (camera positions and sh_degree are
 * read); coords and sh: the parameters. */
int gsr_sh_grad_views(const gsr_view *views_host, uint32_t num_views, const float *coords, const float *sh,
                      const float *const *g_color_host, float *g_sh, int accumulate, void *stream);

/* ---- stage 2: binning and sorting -----------------------------------------------------
 * Replaces Utils/Screen.py:38-62 (tile geometry + membership masks) and the per-tile argsort
 * with its gathers (GaussianRenderer.py:85-95).  The (tile, depth) radix sort is least-
 * significant-digit first: the depth digits are sorted once per Gaussian (before a Gaussian is
 * duplicated into its tiles), the tile digits once per pair. */

/* bytes of scratch gsr_sort_pairs needs for n items */
size_t gsr_sort_workspace_bytes(uint32_t n);
/* Stable LSD radix sort of (key,value) u32 pairs on key bits [begin_bit, end_bit).
 * keys_in/vals_in are clobbered; the result is in keys_out/vals_out. */
int gsr_sort_pairs(uint32_t *keys_in, uint32_t *vals_in, uint32_t *keys_out, uint32_t *vals_out,
                   uint32_t n, int begin_bit, int end_bit, void *workspace, size_t workspace_bytes,
                   void *stream);
/* The same with 64-bit values (same workspace size). */
int gsr_sort_pairs64(uint32_t *keys_in, uint64_t *vals_in, uint32_t *keys_out, uint64_t *vals_out,
                     uint32_t n, int begin_bit, int end_bit, void *workspace, size_t workspace_bytes,
                     void *stream);
/* bytes of scratch gsr_exclusive_scan needs */
size_t gsr_scan_workspace_bytes(uint32_t n);
/* out[i] = sum(in[0..i)); *total (device, may be NULL) = sum(in[0..n)).  in == out allowed. */
int gsr_exclusive_scan(const uint32_t *in, uint32_t *out, uint32_t n, uint32_t *total,
                       void *workspace, size_t workspace_bytes, void *stream);

/* gsr_exclusive_scan for callers that cannot allocate device memory or hand over a workspace (code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the
 * reference.  Same results as gsr_exclusive_scan, bit for bit; the block sums live in a 32 KiB array
 * inside the library's code object (one copy per GPU, held until the process ends), so
 * n <= GSR_SCAN_STATIC_MAX_N - larger inputs are refused with GSR_ERR_ARG.  The call allocates and frees
 * nothing; the first call on a device looks up the array's device address (no allocation either).
 * All calls on one GPU share that array: calls that overlap on one GPU (two streams, two host threads)
 * race - order them, or use gsr_exclusive_scan with separate workspaces. */
#define GSR_SCAN_STATIC_MAX_N (1u << 24)
int gsr_exclusive_scan_static(const uint32_t *in, uint32_t *out, uint32_t n, uint32_t *total, void *stream);

/* Depth order of the Gaussians and the slot ranges of their pairs.
 * order[N]: Gaussian indices by (depth_key, index) ascending - visible ones first;
 * slot_base[N] (N-indexed): first emission slot of each Gaussian; slots follow depth order, so
 * slot_base[order[s]] is the exclusive scan of ntiles[order[s]]; counters->num_pairs = total.
 * slot_seq[N+1] (may be NULL): the same slot bases in depth order, slot_seq[s] = slot_base[order[s]],
 * with the total at slot_seq[N] - consumers that walk the depth order read it instead of gathering.
 * depth_key is clobbered.  workspace: gsr_bin_offsets_workspace_bytes(n). */
size_t gsr_bin_offsets_workspace_bytes(uint32_t n);
int gsr_bin_offsets(uint32_t *depth_key, const uint32_t *ntiles, uint32_t n, uint32_t *order,
                    uint32_t *slot_base, uint32_t *slot_seq, gsr_counters *counters, void *workspace,
                    size_t workspace_bytes, void *stream);

/* gsr_bin_offsets for num_views views of the same n Gaussians with one set of launches (the arrays of
 * gsr_preprocess_fwd_views: view k's depth_key / ntiles / order / slot_base start k * n elements in, its
 * slot_seq k * (n + 1), its counters at counters[k]).  Per view the results equal gsr_bin_offsets' bit for
 * bit.  Replaces, like gsr_bin_offsets, the depth argsort of GaussianRenderer.py:89-97 - for all views of a
 * step at once.  workspace: gsr_bin_offsets_views_workspace_bytes(n, num_views). */
size_t gsr_bin_offsets_views_workspace_bytes(uint32_t n, uint32_t num_views);
int gsr_bin_offsets_views(uint32_t *depth_key, const uint32_t *ntiles, uint32_t n, uint32_t num_views, uint32_t *order,
                          uint32_t *slot_base, uint32_t *slot_seq, gsr_counters *counters, void *workspace,
                          size_t workspace_bytes, void *stream);

/* gsr_bin_offsets_views accepts slot_base == NULL when slot_seq is given: the N-indexed slot bases are then left to
 * this call, slot_base[order[s]] = slot_seq[s] for num_views views laid out as above.  gsr_bin_lists reads slot_seq;
 * the N-indexed array is only read by gsr_pair_reduce, in the backward - a caller with several streams writes it
 * beside the first view's lists instead of in front of them (4-byte writes at random, 0.07 ms for 8 x 1 M). */
int gsr_slot_bases(const uint32_t *order, const uint32_t *slot_seq, uint32_t n, uint32_t num_views, uint32_t *slot_base,
                   void *stream);

/* Per-tile lists.  order, slot_base and slot_seq (may be NULL) come from gsr_bin_offsets; num_pairs
 * must be the value read back from counters->num_pairs.
 * Outputs: tile_offsets[T+1] u32 (tile id = yi + n_block_h * xi, Utils/Screen.py:38-39),
 * pair_slot[I] (emission slot of the pair at each sorted position; may be NULL), sorted_gauss[I] (Gaussian
 * index at each sorted position - the concatenated per-tile depth-ordered lists),
 * stream_rec[I,GSR_STREAM_DWORDS] (the stream records in that order, layout above).
 * num_pairs * 64 must stay below 2^32 (the kernels address the stream with 32-bit byte offsets).  workspace: gsr_bin_lists_workspace_bytes(num_pairs). */
size_t gsr_bin_lists_workspace_bytes(uint32_t num_pairs);
int gsr_bin_lists(const gsr_view *view_host, const uint32_t *order, const uint32_t *slot_base,
                  const uint32_t *slot_seq, const int32_t *bbox, const uint32_t *ntiles, const float *rec, uint32_t num_pairs,
                  uint32_t *tile_offsets, uint32_t *pair_slot, uint32_t *sorted_gauss,
                  float *stream_rec, void *workspace, size_t workspace_bytes, void *stream);

/* Order in which the compositing launches take the tiles: tile_order[T] = the tile ids by
 * descending list length (in 1024 length classes; the order inside a class is unspecified).  Purely a
 * scheduling aid - the grid of a compositing launch is one to two rounds of resident waves and the
 * lists of a view differ in length by an order of magnitude, so longest-first shortens the drain;
 * results do not depend on it. */
int gsr_tile_order(const gsr_view *view_host, const uint32_t *tile_offsets, uint32_t *tile_order, void *stream);

/* gsr_bin_lists + gsr_tile_order for num_views (1..8) views of the same Gaussians, image size and tile edge with one
 * set of launches - the per-tile lists of all views of one optimisation step (GaussianRenderer.py:85-97 once per
 * view).  Per view the results equal the one-view calls bit for bit.  The per-Gaussian inputs are the arrays of
 * gsr_preprocess_fwd_views / gsr_bin_offsets_views (view k from k * n, slot_seq from k * (n + 1)); num_pairs: HOST array
 * of the views' pair counts (read back from counters[k].num_pairs).  The per-pair outputs hold the views one after
 * the other, view k from element pair_base[k] (gsr_bin_lists_views_layout fills pair_base, a multiple of 4096 each,
 * and returns the total element count to allocate: sorted_gauss[total], stream_rec[total, GSR_STREAM_DWORDS]);
 * tile_offsets[num_views, T+1]; tile_order[num_views, T] (may be NULL).
 * gsr_bin_lists_views_supported: 1 when the views qualify - same sizes, and tile id, Gaussian index and emission slot
 * fit one 64-bit word together - else 0 (use gsr_bin_lists per view). */
int gsr_bin_lists_views_supported(const gsr_view *views_host, uint32_t num_views, const uint32_t *num_pairs);
uint64_t gsr_bin_lists_views_layout(uint32_t num_views, const uint32_t *num_pairs, uint32_t *pair_base);
size_t gsr_bin_lists_views_workspace_bytes(uint32_t num_views, const uint32_t *num_pairs);
int gsr_bin_lists_views(const gsr_view *views_host, uint32_t num_views, const uint32_t *order, const uint32_t *slot_base,
                        const uint32_t *slot_seq, const int32_t *bbox, const uint32_t *ntiles, const float *rec,
                        const uint32_t *num_pairs, uint32_t *tile_offsets, uint32_t *tile_order, uint32_t *sorted_gauss,
                        float *stream_rec, void *workspace, size_t workspace_bytes, void *stream);

/* ---- stage 3: compositing -------------------------------------------------------------
 * Replaces Utils/Probability.py:4-23 and GaussianRenderer.py:100-112 for every tile:
 * G = exp(-1/2 d^T Sigma^-1 d) at integer pixel coordinates, alpha = clamp(o*G, 0.01, 0.99),
 * front-to-back blend over the WHOLE list (no early exit, no background).
 * img[W,H,3] and final_T[W,H] are fully overwritten.  tile_order: from gsr_tile_order, or NULL for
 * tile-id order. */
int gsr_composite_fwd(const gsr_view *view_host, const uint32_t *tile_offsets, const uint32_t *tile_order,
                      const float *stream_rec, float *img, float *final_T, void *stream);

/* Backward of the blend (autograd of the statements above).  dimg[W,H,3] is dL/dimg.
 * pairgrad[I * parts, 12], parts = gsr_pairgrad_parts(tile_length), is fully overwritten: the pair at sorted
 * position p with emission slot s (its stream record's slot field, = pair_slot[p]) owns the rows
 * parts * s .. parts * s + parts - 1, whose sum is its gradient record.  parts is 1 when a tile is one 16x16
 * block (tile_length <= 16) and 1, 3 or 4 for tiles of 4, 9 or 16 blocks (the reference's default tile edge
 * of 64, GaussianRenderer.py:49, has 16): the blocks of such a tile are composited by `parts` workgroups, each
 * of which adds its blocks' sums in a fixed order and stores its own row, so the gradients are reproducible
 * run to run.  Only tiles of more than 16 blocks (tile_length > 64) use float atomics (parts = 1). */
uint32_t gsr_pairgrad_parts(int tile_length);
int gsr_composite_bwd(const gsr_view *view_host, const uint32_t *tile_offsets, const uint32_t *tile_order,
                      const float *stream_rec, const float *final_T, const float *dimg,
                      uint32_t num_pairs, float *pairgrad, void *stream);

/* ---- tiles of 2x2, 3x3 or 4x4 blocks: one compact list per 16x16 block -----------------------
 * A second way to composite the same lists (GaussianRenderer.py:100-112 over the whole list of the tile, alpha
 * clamped to [0.01, 0.99]) for tile edges of 32, 48 and 64 - 64 is the reference's default (GaussianRenderer.py:49).
 * An entry that does not reach a block of its tile has alpha = 0.01 on all of the block's pixels; a run of n such
 * entries multiplies the block's transmittance by 0.99^n and adds transmittance x (sum_i 0.99^i 0.01 c_i) to its
 * colour - two constants that do not depend on the pixel.  gsr_block_lists rewrites the stream of a view, per block,
 * as the entries the block evaluates plus ONE record per run (runs end at multiples of 32 list positions), in the
 * record layout of one-block tiles, and the kernels of one-block tiles composite these streams.  Results agree with
 * gsr_composite_fwd / gsr_composite_bwd to rounding (the product 0.99^n is formed in another order); the gradients
 * are reproducible run to run.
 *   gsr_block_lists_supported   1 for tile_length 32 / 48 / 64 with num_pairs x blocks-per-tile < 2^28, else 0
 *   gsr_block_combo_bytes       bytes per word of combo: 4, or 8 from 2^26 - 1 pair-block combinations on
 *   gsr_block_count             T16 = ceil(width / 16) * ceil(height / 16); block id = Y16 + ceil(height / 16) * X16
 *   gsr_block_lists             from tile_offsets and stream_rec of gsr_bin_lists: block_offsets[T16 + 1],
 *                               block_stream[R, 16] (R <= num_pairs x S^2 records, S = tile_length / 16: allocate that;
 *                               every record carries its index in its block's stream), combo[num_pairs x S^2] words
 *                               of gsr_block_combo_bytes (per list position and block: all ones outside the image,
 *                               r << 6 | 32 evaluated by record r, r << 6 | index inside run r).  gsr_tile_order with
 *                               tile_length = 16 orders the blocks.  The streams may exceed 4 GiB.
 *   gsr_composite_blocks_fwd    img, final_T as gsr_composite_fwd
 *   gsr_composite_blocks_bwd    block_grad[R, 12] (allocate num_pairs x S^2 rows): row r belongs to stream record r -
 *                               the gradient record of an evaluated entry, or, for a run, the six sums over the block
 *                               of (transmittance in front of the run) x cotangent (red, green, four parts of blue)
 *   gsr_block_rows_reduce       pairgrad[num_pairs, 12] at the pairs' emission slots, ONE row per pair (sum over the
 *                               blocks in a fixed order, plus 0.01 x 0.99^i x run sum for the blocks on which the entry
 *                               was inside a run): what gsr_pair_reduce takes with parts = 1 */
int gsr_block_lists_supported(const gsr_view *view_host, uint32_t num_pairs);
uint32_t gsr_block_combo_bytes(const gsr_view *view_host, uint32_t num_pairs);
uint32_t gsr_block_count(const gsr_view *view_host);
size_t gsr_block_lists_workspace_bytes(const gsr_view *view_host, uint32_t num_pairs);
int gsr_block_lists(const gsr_view *view_host, const uint32_t *tile_offsets, const float *stream_rec, uint32_t num_pairs,
                    uint32_t *block_offsets, float *block_stream, void *combo, void *workspace, size_t workspace_bytes,
                    void *stream);
int gsr_composite_blocks_fwd(const gsr_view *view_host, uint32_t num_pairs, const uint32_t *block_offsets,
                             const uint32_t *block_order, const float *block_stream, float *img, float *final_T, void *stream);
int gsr_composite_blocks_bwd(const gsr_view *view_host, uint32_t num_pairs, const uint32_t *block_offsets,
                             const uint32_t *block_order, const float *block_stream, const float *final_T, const float *dimg,
                             float *block_grad, void *stream);
int gsr_block_rows_reduce(const gsr_view *view_host, uint32_t num_pairs, const float *stream_rec, const void *combo,
                          const float *block_grad, float *pairgrad, void *stream);

/* ---- training loss of one view ----------------------------------------------------------
 * Replaces Trainer.compute_loss (Trainer.py:35-38) with its two terms l1_loss (Utils/Loss.py:4-5)
 * and ssim_loss (Utils/Loss.py:59-97) and their autograd, for pred and gt of shape [W,H,3] as the
 * renderer and the data loader (Utils/DataLoader.py:81-89) hand them over:
 *   loss = (1 - loss_lambda) * mean|pred - gt| + loss_lambda * (1 - ssim).
 * As in the reference, the SSIM window code takes dimension -3 of a [W,H,3] tensor as the channel
 * count (Utils/Loss.py:70): each image column x is one [H,3] plane, and the 11x11 Gaussian window
 * (sigma 1.5, zero padding 5) runs along y and across the three colours.
 * out[3] (device) = {loss, l1, ssim}.  dpred[W,H,3] (device, may be NULL) = grad_scale * dloss/dpred,
 * fully overwritten; loss_lambda = 0 gives the gradient of l1_loss, loss_lambda = 1 with
 * grad_scale = -1 that of ssim_loss.  workspace: gsr_loss_workspace_bytes(width, height). */
size_t gsr_loss_workspace_bytes(int width, int height);
int gsr_loss_fwd_bwd(const float *pred, const float *gt, int width, int height, float loss_lambda,
                     float grad_scale, float *out, float *dpred, void *workspace,
                     size_t workspace_bytes, void *stream);

/* ---- optimiser step ------------------------------------------------------------------------
 * Replaces optimizer.step() (Trainer.py:80) of the torch.optim.Adam built at Trainer.py:24
 * (lr, betas; eps 1e-8, no weight decay, no amsgrad) for up to GSR_ADAM_MAX_TENSORS tensors in one
 * launch.  The *_host arguments are HOST arrays of num_tensors entries: device pointers of the
 * parameter, its gradient and its two moment buffers (exp_avg, exp_avg_sq: zeros before the first
 * step), the element counts, and the 1-based count of THIS step per tensor (a tensor that had no
 * gradient in some iteration is left out of that call and keeps its count, as PyTorch does).
 * Per element: m = lerp(m, g, 1-beta1); v = v*beta2 + (1-beta2)*g*g;
 * p -= lr/(1-beta1^t) * m / (sqrt(v)/sqrt(1-beta2^t) + eps). */
#define GSR_ADAM_MAX_TENSORS 8
int gsr_adam_step(uint32_t num_tensors, float *const *param_host, const float *const *grad_host,
                  float *const *exp_avg_host, float *const *exp_avg_sq_host, const uint64_t *numel_host,
                  const uint32_t *step_host, double lr, double beta1, double beta2, double eps, void *stream);

/* ---- densification bookkeeping --------------------------------------------------------------
 * Rows are the reference's parameter rows of a model with `size` live rows inside tensors of
 * `capacity` rows: coords[.,3] scale_raw[.,3] rot_raw[.,4] opac_raw[.,1] sh[.,3,16], plus the
 * statistics coords_grads[.,1] f32 and denom[.,1] i32 (GaussianModel.py:19-26,93-94). */

/* GaussianModel.update_densify_stats (GaussianModel.py:230-232): for the n live rows,
 * coords_grads[i] += |g_screen[i, :2]| and denom[i] += 1 where mask[i]. */
int gsr_densify_stats(uint32_t n, const uint8_t *mask, const float *g_screen, float *coords_grads,
                      int32_t *denom, void *stream);

/* gsr_densify_stats for the num_views views of one step with one launch per GSR_MAX_VIEWS views (Trainer.py:61 calls
 * update_densify_stats once per rendered view; a step of several views calls it once per view).  mask_host /
 * g_screen_host: HOST arrays of num_views device pointers.  A row's contributions are added in the order of the
 * views, so the result equals num_views calls of gsr_densify_stats bit for bit. */
int gsr_densify_stats_views(uint32_t n, uint32_t num_views, const uint8_t *const *mask_host,
                            const float *const *g_screen_host, float *coords_grads, int32_t *denom, void *stream);

/* Selection of GaussianModel.densify (GaussianModel.py:236-244): with g = coords_grads / denom
 * (nan -> 0) and s = the largest RAW scale of the row, row i < size is cloned when
 * |g| >= grad_threshold and s <= scale_threshold, split when |g| >= grad_threshold and
 * s > scale_threshold (the reference passes 0.0002 and 0.01 * scene_radius).
 * Outputs: cls[size] u8 (0 none, 1 clone, 2 split), clone_rank[size] / split_rank[size] = number of
 * cloned / split rows below i, counts[2] (device) = {num_clone, num_split}. */
size_t gsr_densify_select_workspace_bytes(uint32_t size);
int gsr_densify_select(uint32_t size, const float *coords_grads, const int32_t *denom, const float *scale_raw,
                       float grad_threshold, float scale_threshold, uint8_t *cls, uint32_t *clone_rank,
                       uint32_t *split_rank, uint32_t *counts, void *workspace, size_t workspace_bytes,
                       void *stream);

/* GaussianModel.clone (GaussianModel.py:189-193) followed by GaussianModel.split
 * (GaussianModel.py:209-224, split_factor 2) for the selection above; the tensors must hold at
 * least size + num_clone + num_split rows (the caller grows them as GaussianModel.py:184-187,
 * 204-207 do).  Cloned row i is copied to row size + clone_rank[i].  Split row i, with k =
 * split_rank[i], std = exp(scale_raw[i]) and R = the rotation matrix of its normalised quaternion,
 * gets coords = R (noise[k] * std) and the new row size + num_clone + k gets
 * coords = R (noise[num_split + k] * std); both get scale_raw = log(std / 1.6); the new row copies
 * rotation, opacity and sh.  noise[2 * num_split, 3]: the standard-normal draws behind the
 * reference's torch.normal(0, std) (GaussianModel.py:210), supplied by the caller.  As in the
 * reference the old centre is not added to the new centres. */
int gsr_densify_apply(uint32_t size, uint32_t num_clone, uint32_t num_split, const uint8_t *cls,
                      const uint32_t *clone_rank, const uint32_t *split_rank, const float *noise,
                      float *coords, float *scale_raw, float *rot_raw, float *opac_raw, float *sh,
                      void *stream);

/* GaussianModel.remove(mask, inplace=True) (GaussianModel.py:148-164), literally: with r = the
 * number of set mask bytes, the last r KEPT rows (in order) are written to rows
 * [size - 2r, size - r) - opacity, rotation and scale in their activated form (sigmoid, normalised,
 * exp), coords, sh and the statistics as they are - and the caller then lowers the size by r.
 * gsr_remove_select: keep_rank[i] = number of kept rows below i, *num_kept (device) = size - r.
 * gsr_remove_apply: needs the num_removed read back from that; fails when 2 r > size (the
 * reference raises there). */
size_t gsr_remove_select_workspace_bytes(uint32_t size);
int gsr_remove_select(uint32_t size, const uint8_t *mask, uint32_t *keep_rank, uint32_t *num_kept,
                      void *workspace, size_t workspace_bytes, void *stream);
size_t gsr_remove_apply_workspace_bytes(uint32_t num_removed);
int gsr_remove_apply(uint32_t size, uint32_t num_removed, const uint8_t *mask, const uint32_t *keep_rank,
                     float *coords, float *scale_raw, float *rot_raw, float *opac_raw, float *sh,
                     float *coords_grads, int32_t *denom, void *workspace, size_t workspace_bytes,
                     void *stream);

/* ---- initialisation from a point cloud ------------------------------------------------------
 * Replaces distCUDA2 of the simple_knn extension as GaussianModel.from_pcd calls it
 * (GaussianModel.py:82): dist2[i] = mean of the squared distances from points[i] to its three nearest
 * other points (a point's own index is skipped; coincident points count with distance zero).
 * points[N,3], dist2[N]; N >= 4.  Exhaustive search, O(N^2). */
int gsr_knn_dist2(uint32_t n, const float *points, float *dist2, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GSR_H */
