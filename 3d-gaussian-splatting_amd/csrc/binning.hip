// Stage 2: depth order of the Gaussians, per-pair slots, per-tile lists.
//
// The reference builds, for every tile, a boolean membership mask over all visible Gaussians and
// argsorts the members by NDC depth (Utils/Screen.py:58-62, GaussianRenderer.py:85-95).  Here the
// same lists come out of one least-significant-digit radix sort of the composite key
// (tile id, depth bits): the four depth digits are sorted once per Gaussian - they are equal for
// all tiles a Gaussian touches - and the tile digits once per (tile, Gaussian) pair.  Both sorts
// are stable and pairs are emitted in depth order, so ties in depth keep ascending Gaussian index.
#include <stdlib.h>

#include "gsr_common.hpp"

namespace {

// The lists of several views of the same Gaussians built by one set of launches (blockIdx.y = view): view y has
// n[y] pairs from element base[y] (a multiple of 4096) of the per-pair arrays; the per-Gaussian inputs of view y
// start y * N elements (rows) into their arrays, its tile offsets y * (T + 1), its tile order y * T.  count == 0: one view.
struct ListSegs {
    uint32_t count;
    uint32_t n[GSR_MAX_SEGS], base[GSR_MAX_SEGS];
};
__device__ inline uint32_t seg_pick(const uint32_t (&a)[GSR_MAX_SEGS], uint32_t y)
{
    uint32_t r = a[0];
#pragma unroll
    for (uint32_t k = 1; k < GSR_MAX_SEGS; ++k) r = y == k ? a[k] : r;
    return r;
}

// A (tile, Gaussian, slot) triple in one 64-bit word: slot in the low `slot` bits, the Gaussian above it, the
// tile id above that.  The tile sort then moves 8 bytes per pair and pass instead of a 4-byte key and an
// 8-byte value.  Used whenever the three fields fit 64 bits (gsr_bin_lists decides).
struct PairBits {
    int slot, gauss;   // widths; the tile id starts at bit slot + gauss
};
// x / d for x below 2^24 (exact in fp32) from the reciprocal of d: the rounded product is off by one at most, which
// the remainder shows.  A 32-bit integer division is some twenty vector instructions, this is seven.
__device__ inline uint32_t div_below_2p24(uint32_t x, uint32_t d, float inv_d)
{
    if (x >= (1u << 24)) return x / d;
    uint32_t q = (uint32_t)((float)x * inv_d);
    const int32_t r = (int32_t)(x - q * d);
    q = r < 0 ? q - 1u : (r >= (int32_t)d ? q + 1u : q);
    return q;
}

__device__ inline uint64_t pack_pair(PairBits b, uint32_t tile, uint32_t gauss, uint32_t slot)
{
    return ((uint64_t)tile << (b.slot + b.gauss)) | ((uint64_t)gauss << b.slot) | (uint64_t)slot;
}

// Emits (tile id, owner | slot) for every slot.  Slot e of Gaussian g: e = slot_base[g] + (xi - x0) * ny
// + (yi - y0).  A wave takes 64 Gaussians that are consecutive in DEPTH order: their slots form one
// contiguous range, which the wave writes 64 slots at a time - lane j owns slot start + 64 k + j and
// finds the Gaussian it belongs to with a binary search over the 64 slot bases (held one per lane).
// All global writes are fully coalesced; one thread per Gaussian writing its own slots is not.
// slot_seq (may be null): the slot bases in depth order with the total at [n] - read in place of the
// two per-Gaussian gathers of slot_base and ntiles.
__global__ __launch_bounds__(256) void emit_pairs_kernel(ViewK v, const uint32_t *__restrict__ order,
                                                         const uint32_t *__restrict__ slot_seq,
                                                         const uint32_t *__restrict__ slot_base,
                                                         const int32_t *__restrict__ bbox,
                                                         const uint32_t *__restrict__ ntiles,
                                                         uint32_t *__restrict__ tile_key,
                                                         uint64_t *__restrict__ owner_slot, uint32_t num_pairs,
                                                         PairBits bits, ListSegs segs)
{
    if (segs.count) {
        const size_t at = (size_t)blockIdx.y * (size_t)v.n;
        order += at; slot_base += at; bbox += 4 * at; ntiles += at;
        if (slot_seq) slot_seq += at + blockIdx.y;          // (n + 1 entries per view)
        num_pairs = seg_pick(segs.n, blockIdx.y);
        owner_slot += seg_pick(segs.base, blockIdx.y);
    }
    const int s = blockIdx.x * blockDim.x + threadIdx.x;   // position in depth order
    uint32_t g = 0, sb = 0xFFFFFFFFu, nt = 0;
    int x0 = 0, y0 = 0, ny = 1;
    if (s < v.n) {
        g = order[s];
        if (slot_seq) {
            sb = slot_seq[s];
            nt = slot_seq[s + 1] - sb;
        } else {
            nt = ntiles[g];
            sb = slot_base[g];
        }
        if (nt) {
            const int4 bb = reinterpret_cast<const int4 *>(bbox)[g];
            int x1, y1;
            gsr_tile_span(bb.x, bb.z, v.L, x0, x1);
            gsr_tile_span(bb.y, bb.w, v.L, y0, y1);
            ny = y1 - y0 + 1;
        }
    }
    // the wave's slot range; lanes past the end of the model hold sb = 0xFFFFFFFF and nt = 0
    const uint32_t start = __shfl(sb, 0, 64);
    uint32_t end = sb == 0xFFFFFFFFu ? 0u : sb + nt;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) end = max(end, (uint32_t)__shfl_xor(end, d, 64));
    end = min(end, num_pairs);   // guards a stale num_pairs; no effect on consistent input
    const int lane = threadIdx.x & 63;
    // (the trip count is wave-uniform: the shuffles below must see every lane)
    for (uint32_t base = start; base < end; base += 64) {
        const uint32_t pos = base + lane;
        // owner = last lane whose first slot is <= pos (slot bases do not decrease along the wave;
        // a Gaussian without tiles shares its base with its successor and is never the last one)
        int lo = 0, hi = 63;
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            const int mid = (lo + hi + 1) >> 1;
            const uint32_t b = __shfl(sb, mid, 64);
            if (b <= pos) lo = mid; else hi = mid - 1;
        }
        const uint32_t og = __shfl(g, lo, 64), osb = __shfl(sb, lo, 64);
        const int ox0 = __shfl(x0, lo, 64), oy0 = __shfl(y0, lo, 64), ony = __shfl(ny, lo, 64);
        if (pos < end) {
            const uint32_t local = pos - osb;
            const uint32_t cx = div_below_2p24(local, (uint32_t)ony, __builtin_amdgcn_rcpf((float)ony)), cy = local - cx * (uint32_t)ony;
            const uint32_t tile = (uint32_t)(oy0 + (int)cy + v.nbh * (ox0 + (int)cx));
            if (tile_key) {
                tile_key[pos] = tile;
                owner_slot[pos] = ((uint64_t)og << 32) | (uint64_t)pos;   // carried through the tile sort
            } else {
                owner_slot[pos] = pack_pair(bits, tile, og, pos);          // one word per pair, sorted by its tile bits
            }
        }
    }
}

// Hint test of one block: bit 0 set when alpha = clamp(opacity*G, 0.01, 0.99) is pinned at 0.01 on EVERY pixel of
// the rectangle [x0,x1] x [ya,ym] (the top rows of the block), bit 1 the same for [x0,x1] x [ym+1,yb] (inclusive
// integer pixels; `top` / `bottom` false: that half has no pixels inside the tile and counts as clamped), with a
// margin that covers the fp32 evaluation of the exponent in the compositing kernels.  The quadratic form is convex
// (checked), so its minimum over a rectangle is 0 when the centre lies inside; otherwise it sits on an edge that
// FACES the centre - along a ray from the centre the form grows, so a point of an edge that looks away from the
// centre is beaten by the point where its ray entered the rectangle - and there is at most one such vertical and
// one such horizontal edge.  Conservative: a half the test misses is simply evaluated in full.  Both halves share
// the columns, hence everything that belongs to the vertical edge but the clamp of its minimiser.  Branch-free:
// the lanes of a wave hold unrelated pairs.
struct HalfRows {
    float lo, hi;   // d = centre - pixel over the rows of the half: lo = gy - last row, hi = gy - first row
};
__device__ inline uint32_t block_halves_clamped_low(float gx, float gy, float A, float B, float D, float o, float x0, float x1,
                                                    float ya, float ym, float yb, bool top, bool bottom)
{
    // A, B, D are the exp2-domain coefficients (negative-definite for a valid Gaussian)
    const float a = -A, bh = -0.5f * B, d = -D;          // f(dx,dy) = a dx^2 + 2 bh dx dy + d dy^2 = -log2(G)
    const bool convex = a > 0.0f && d > 0.0f && a * d > 1.000001f * bh * bh;
    const float dx0 = gx - x1, dx1 = gx - x0;            // d = centre - pixel
    const bool x_in = dx0 <= 0.0f && dx1 >= 0.0f;
    const float ex = dx0 > 0.0f ? dx0 : dx1;             // the vertical edge that faces the centre (if !x_in)
    // (hardware reciprocals, one ulp: the minimiser along an edge only has to be close - the form is evaluated AT
    // it, and the slack below covers the rest)
    const float bh2 = bh + bh;
    const float t_free = -bh * ex * __builtin_amdgcn_rcpf(d);     // minimiser along dx = ex
    const float over_a = -bh * __builtin_amdgcn_rcpf(a);
    const float aex2 = a * ex * ex, bex2 = bh2 * ex;
    const float mx = fmaxf(fabsf(dx0), fabsf(dx1));
    const float amx2 = a * mx * mx, bmx2 = fabsf(bh2) * mx;
    // raw < 0.01  <=>  log2(o) - f < log2(0.01); slack: rounding of f here and in the kernels
    const float need = __log2f(o) + 1.0e-4f + 6.6438561898f;       // low  <=>  need + 4e-6 mag < fmin
    const HalfRows rows[2] = {{gy - ym, gy - ya}, {gy - yb, gy - (ym + 1.0f)}};
    uint32_t flags = 0u;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float dy0 = rows[h].lo, dy1 = rows[h].hi;
        const bool y_in = dy0 <= 0.0f && dy1 >= 0.0f;
        const float t = fminf(fmaxf(t_free, dy0), dy1);
        const float fx = fmaf(t, fmaf(d, t, bex2), aex2);                 // a ex^2 + 2 bh ex t + d t^2
        const float ey = dy0 > 0.0f ? dy0 : dy1;                           // the horizontal edge that faces the centre
        const float u = fminf(fmaxf(ey * over_a, dx0), dx1);
        const float fy = fmaf(u, fmaf(a, u, bh2 * ey), d * ey * ey);      // a u^2 + 2 bh u ey + d ey^2
        const float fmin = x_in ? (y_in ? 0.0f : fy) : (y_in ? fx : fminf(fx, fy));
        const float my = fmaxf(fabsf(dy0), fabsf(dy1));
        const float mag = fmaf(my, fmaf(d, my, bmx2), amx2);              // bound on the summed terms
        const bool low = fmaf(4.0e-6f, mag, need) < fmin;
        flags |= (low ? 1u : 0u) << h;
    }
    // opacity underflowed to 0: raw = 0 everywhere; a negative or NaN opacity, a form that is not convex: no hint
    flags = o > 0.0f ? (convex ? flags : 0u) : (o == 0.0f ? 3u : 0u);
    return flags | (top ? 0u : 1u) | (bottom ? 0u : 2u);
}

// the whole rectangle [x0,x1] x [y0,y1] at once (tiles of more than 16 blocks carry one answer for all blocks)
__device__ inline bool tile_all_clamped_low(float gx, float gy, float A, float B, float D, float o, float x0,
                                            float x1, float y0, float y1)
{
    return (block_halves_clamped_low(gx, gy, A, B, D, o, x0, x1, y0, y1, y1, true, false) & 1u) != 0u;
}

// One thread per sorted position p: write the owner's splat record in stream layout (with the
// pair's emission slot and the hint flags: the halves of the tile's block on which alpha is 0.01
// everywhere, where the compositing kernels can skip their exponentials), note the owner, mark tile
// boundaries.
// (ONE_BLOCK: the tile is a single 16x16 block, the headline case, without the loops over a tile's blocks)
template <bool ONE_BLOCK>
__global__ __launch_bounds__(256) void finalize_lists_kernel(
    ViewK v, const uint32_t *__restrict__ sorted_tile, const uint64_t *__restrict__ sorted_owner_slot,
    const float *__restrict__ rec, uint32_t num_pairs, uint32_t T, uint32_t *__restrict__ tile_offsets,
    uint32_t *__restrict__ pair_slot, uint32_t *__restrict__ sorted_gauss, float *__restrict__ stream_rec, PairBits bits,
    ListSegs segs)
{
    if (segs.count) {
        num_pairs = seg_pick(segs.n, blockIdx.y);
        const uint32_t base = seg_pick(segs.base, blockIdx.y);
        if (blockIdx.x * blockDim.x >= num_pairs) return;
        sorted_owner_slot += base;
        if (pair_slot) pair_slot += base;
        sorted_gauss += base;
        stream_rec += (size_t)base * GSR_STREAM_DWORDS;
        rec += (size_t)blockIdx.y * (size_t)v.n * GSR_REC_FLOATS;
        tile_offsets += (size_t)blockIdx.y * (T + 1u);
    }
    // (sorted_tile == null: sorted_owner_slot holds packed words, see pack_pair)
    // The 64-byte records of a wave's 64 pairs are one contiguous 4 KiB piece of the stream: they go
    // through LDS so that every store instruction writes 1 KiB of consecutive addresses.
    __shared__ float4 tile[256 * 4];
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    float4 *wave_tile = tile + (threadIdx.x - lane) * 4;
    const uint32_t wave_first = p - lane;
    if (wave_first >= num_pairs) return;   // whole wave out of range
    float4 out[4] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f),
                     make_float4(0.f, 0.f, 0.f, 0.f)};
    if (p < num_pairs) {
    const uint64_t word = sorted_owner_slot[p];
    const bool packed = sorted_tile == nullptr;
    const uint32_t g = packed ? (uint32_t)(word >> bits.slot) & ((1u << bits.gauss) - 1u) : (uint32_t)(word >> 32);
    const uint32_t slot = packed ? (uint32_t)word & (uint32_t)((1ull << bits.slot) - 1ull) : (uint32_t)word;
    if (pair_slot) pair_slot[p] = slot;
    sorted_gauss[p] = g;
    const float4 *src = reinterpret_cast<const float4 *>(rec + (size_t)g * GSR_REC_FLOATS);
    const float4 a = src[0], b = src[1], c = src[2];   // x y A' B' | D' o r g | b ca cbh cd
    const uint32_t cur = packed ? (uint32_t)(word >> (bits.slot + bits.gauss)) : sorted_tile[p];
    uint32_t flags;
    {
        const int xi = (int)div_below_2p24(cur, (uint32_t)v.nbh, __builtin_amdgcn_rcpf((float)v.nbh)), yi = (int)(cur - (uint32_t)xi * v.nbh);
        const float x0 = (float)(xi * v.L), y0 = (float)(yi * v.L);
        const float x1 = (float)(min((xi + 1) * v.L, v.W) - 1), y1 = (float)(min((yi + 1) * v.L, v.H) - 1);
        // bits 2c and 2c+1: the top and bottom 8 rows of the tile's c-th 16x16 block (c = bx * S + by,
        // the wave that composites it); a half that lies outside the tile counts as clamped
        const int S = ONE_BLOCK ? 1 : (v.L + 15) >> 4;
        flags = 0u;
        if (ONE_BLOCK) {
            flags = block_halves_clamped_low(a.x, a.y, a.z, a.w, b.x, b.y, x0, x1, y0, fminf(y0 + 7.0f, y1), y1, true, y0 + 7.0f < y1);
        } else if (S <= 4) {
            for (int bx = 0; bx < S; ++bx) {
                const float bx0 = x0 + 16.0f * (float)bx, bx1 = fminf(bx0 + 15.0f, x1);
                for (int by = 0; by < S; ++by) {
                    const float by0 = y0 + 16.0f * (float)by, bym = fminf(by0 + 7.0f, y1), by1 = fminf(by0 + 15.0f, y1);
                    const bool some = bx0 <= x1 && by0 <= y1;
                    flags |= block_halves_clamped_low(a.x, a.y, a.z, a.w, b.x, b.y, bx0, bx1, by0, bym, by1, some, some && bym < by1)
                             << (2 * (bx * S + by));
                }
            }
        } else {
            // more than 16 blocks per tile: one test for the whole tile, the same answer for every block
            flags = tile_all_clamped_low(a.x, a.y, a.z, a.w, b.x, b.y, x0, x1, y0, y1) ? 0xFFFFFFFFu : 0u;
        }
    }
    out[0] = make_float4(a.y, a.x, b.x, a.z);
    out[1] = make_float4(b.y, a.w, b.z, c.y);
    if (ONE_BLOCK) {
        // stream layout of a one-block tile: y x D' A' | o B' r ca | g slot+flags b b/100 | cbh r/100 cd g/100
        // (the two hint bits ride above the slot, which stays below 2^26; the colour times alpha's lower clamp serves
        // the halves on which alpha sits at that clamp: the blend adds 0.01 T colour there, GaussianRenderer.py:104-112)
        constexpr float kAlphaFloor = 0.01f;
        out[2] = make_float4(b.w, __uint_as_float(slot | (flags << 30)), c.x, kAlphaFloor * c.x);
        out[3] = make_float4(c.z, kAlphaFloor * b.z, c.w, kAlphaFloor * b.w);
    } else {
        // stream layout of larger tiles: y x D' A' | o B' r ca | g slot b flags | cbh 0 cd 0
        out[2] = make_float4(b.w, __uint_as_float(slot), c.x, __uint_as_float(flags));
        out[3] = make_float4(c.z, 0.0f, c.w, 0.0f);
    }
    const uint32_t first = p == 0 ? 0u : (packed ? (uint32_t)(sorted_owner_slot[p - 1] >> (bits.slot + bits.gauss)) : sorted_tile[p - 1]) + 1u;
    for (uint32_t t = first; t <= cur; ++t) tile_offsets[t] = p;  // tiles (prev, cur] start at p
    if (p + 1 == num_pairs)
        for (uint32_t t = cur + 1; t <= T; ++t) tile_offsets[t] = num_pairs;
    }
    // record of lane l = float4 4 l .. 4 l + 3 of the wave's piece
#pragma unroll
    for (int q = 0; q < 4; ++q) wave_tile[lane * 4 + q] = out[q];
    __builtin_amdgcn_wave_barrier();
    float4 *dst = reinterpret_cast<float4 *>(stream_rec + (size_t)wave_first * GSR_STREAM_DWORDS);
    const uint32_t wave_f4 = min(64u, num_pairs - wave_first) * 4u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t f = q * 64 + lane;
        if (f < wave_f4) { typedef float v4f __attribute__((ext_vector_type(4))); const float4 t4 = wave_tile[f]; __builtin_nontemporal_store(v4f{t4.x, t4.y, t4.z, t4.w}, reinterpret_cast<v4f *>(&dst[f])); }
    }
}

// finalize_lists_kernel<true> for tiles of one block, <false> for larger tiles
#define GSR_LAUNCH_FINALIZE(one_block, ...)                                        \
    do {                                                                           \
        if (one_block) hipLaunchKernelGGL(finalize_lists_kernel<true>, __VA_ARGS__);   \
        else hipLaunchKernelGGL(finalize_lists_kernel<false>, __VA_ARGS__);            \
    } while (0)

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// GSR_WIDE_PAIRS=1 in the environment keeps the (32-bit key, 64-bit value) form of the pairs even when the packed
// word would fit - the form that workloads with very many tiles, Gaussians and pairs fall back to, kept testable.
inline bool gsr_force_wide_pairs()
{
    const char *e = getenv("GSR_WIDE_PAIRS");   // read at every call: a test switches it inside one process
    return e && e[0] == '1';
}

inline int bits_for(uint32_t count)
{
    int b = 0;
    while (b < 32 && (1ull << b) < (unsigned long long)count) ++b;
    return b;
}

// ---- order in which the compositing grid takes the tiles ------------------------------------------
// One wave composites one 16x16 block over its tile's whole list, lists differ in length by a factor
// of twenty within a view, and the grid is only one to two rounds of resident waves: in tile-id
// order the waves still running at the end are whichever long lists came late.  Longest lists first
// keeps the drain short.  A counting sort over 1024 length classes (exact order inside a class does
// not matter) by one workgroup: T is a few thousand to a few hundred thousand.
constexpr int kOrderThreads = 1024;
constexpr int kOrderBuckets = 1024;

__global__ __launch_bounds__(kOrderThreads) void tile_order_kernel(uint32_t T, const uint32_t *__restrict__ tile_offsets,
                                                                   uint32_t *__restrict__ tile_order)
{
    __shared__ uint32_t s_count[kOrderBuckets];
    __shared__ uint32_t s_max[kOrderThreads / GSR_WAVE];
    tile_offsets += (size_t)blockIdx.y * (T + 1u);   // (blockIdx.y: one of several views, one after the other)
    tile_order += (size_t)blockIdx.y * T;
    const uint32_t t = threadIdx.x;
    uint32_t mx = 0;
    for (uint32_t i = t; i < T; i += kOrderThreads) mx = max(mx, tile_offsets[i + 1] - tile_offsets[i]);
    for (int m = 32; m >= 1; m >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, m));
    if ((t & 63) == 0) s_max[t >> 6] = mx;
    s_count[t] = 0;
    __syncthreads();
    mx = 0;
    for (int w = 0; w < kOrderThreads / GSR_WAVE; ++w) mx = max(mx, s_max[w]);
    // class 0 holds the longest lists
    const float to_class = (float)(kOrderBuckets - 1) / (float)max(mx, 1u);
    auto class_of = [&](uint32_t len) {
        const int c = (kOrderBuckets - 1) - (int)((float)len * to_class);
        return (uint32_t)min(max(c, 0), kOrderBuckets - 1);
    };
    for (uint32_t i = t; i < T; i += kOrderThreads) atomicAdd(&s_count[class_of(tile_offsets[i + 1] - tile_offsets[i])], 1u);
    __syncthreads();
    // exclusive scan of the 1024 class counts: one per thread
    const uint32_t mine = s_count[t];
    uint32_t incl = mine;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
        if ((int)(t & 63) >= d) incl += up;
    }
    __syncthreads();
    if ((t & 63) == 63) s_max[t >> 6] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < (t >> 6); ++w) base += s_max[w];
    s_count[t] = base + incl - mine;
    __syncthreads();
    for (uint32_t i = t; i < T; i += kOrderThreads) {
        const uint32_t pos = atomicAdd(&s_count[class_of(tile_offsets[i + 1] - tile_offsets[i])], 1u);
        tile_order[pos] = i;
    }
}

}  // namespace

extern "C" int gsr_tile_order(const gsr_view *view_host, const uint32_t *tile_offsets, uint32_t *tile_order, void *stream)
{
    ViewK v;
    if (int e = gsr_make_view(view_host, &v)) return e;
    if (!tile_offsets || !tile_order) { gsr_set_error("gsr_tile_order: null pointer"); return GSR_ERR_ARG; }
    hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(kOrderThreads), 0, (hipStream_t)stream, (uint32_t)v.T, tile_offsets,
                       tile_order);
    return gsr_check_launch("tile_order");
}

extern "C" size_t gsr_bin_offsets_workspace_bytes(uint32_t n)
{
    const size_t arr = align256((size_t)(n ? n : 1) * sizeof(uint32_t));
    return 3 * arr + gsr_sort_workspace_bytes(n) + gsr_scan_workspace_bytes(n);
}

extern "C" int gsr_bin_offsets(uint32_t *depth_key, const uint32_t *ntiles, uint32_t n, uint32_t *order,
                               uint32_t *slot_base, uint32_t *slot_seq, gsr_counters *counters, void *workspace,
                               size_t workspace_bytes, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!workspace || workspace_bytes < gsr_bin_offsets_workspace_bytes(n)) {
        gsr_set_error("gsr_bin_offsets: workspace too small");
        return GSR_ERR_ARG;
    }
    if (!counters) { gsr_set_error("gsr_bin_offsets: null counters"); return GSR_ERR_ARG; }
    if (n == 0) {
        (void)hipMemsetAsync(&counters->num_pairs, 0, sizeof(uint32_t), st);
        if (slot_seq) (void)hipMemsetAsync(slot_seq, 0, sizeof(uint32_t), st);
        return GSR_OK;
    }
    if (!depth_key || !ntiles || !order || !slot_base) { gsr_set_error("gsr_bin_offsets: null pointer"); return GSR_ERR_ARG; }
    const size_t arr = align256((size_t)n * sizeof(uint32_t));
    char *ws = (char *)workspace;
    uint32_t *ids_b = (uint32_t *)ws;             // second value buffer of the sort
    uint32_t *keys_b = (uint32_t *)(ws + arr);    // second key buffer of the sort
    void *sort_ws = ws + 3 * arr;
    const size_t sort_bytes = gsr_sort_workspace_bytes(n);
    void *scan_ws = (char *)sort_ws + sort_bytes;
    const size_t scan_bytes = gsr_scan_workspace_bytes(n);
    // an argsort: the first pass takes a key's position as its value.  32 key bits = an even number of passes: the
    // sorted indices end in `order`
    bool in_first;
    if (int e = gsr_sort_pingpong32(depth_key, order, keys_b, ids_b, n, 0, 32, sort_ws, sort_bytes, stream, &in_first, true))
        return e;
    if (!in_first) { gsr_set_error("gsr_bin_offsets: unexpected number of sort passes"); return GSR_ERR_LAUNCH; }
    // slot_base[order[s]] = sum over s' < s of ntiles[order[s']]
    // (keys_b, the sort's second key buffer, is free again: the scan keeps the gathered tile counts there)
    if (int e = gsr_exclusive_scan_along(ntiles, order, slot_base, n, &counters->num_pairs, scan_ws, scan_bytes, stream,
                                         slot_seq, &counters->error, keys_b))
        return e;
    return gsr_check_launch("bin_offsets");
}

extern "C" size_t gsr_bin_offsets_views_workspace_bytes(uint32_t n, uint32_t num_views)
{
    const size_t arr = align256((size_t)(n ? n : 1) * (num_views ? num_views : 1) * sizeof(uint32_t));
    return 2 * arr + (size_t)(num_views ? num_views : 1) * (gsr_sort_workspace_bytes(n) + gsr_scan_workspace_bytes(n));
}

extern "C" int gsr_bin_offsets_views(uint32_t *depth_key, const uint32_t *ntiles, uint32_t n, uint32_t num_views,
                                     uint32_t *order, uint32_t *slot_base, uint32_t *slot_seq, gsr_counters *counters,
                                     void *workspace, size_t workspace_bytes, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (num_views == 0) return GSR_OK;
    if (!workspace || workspace_bytes < gsr_bin_offsets_views_workspace_bytes(n, num_views)) {
        gsr_set_error("gsr_bin_offsets_views: workspace too small");
        return GSR_ERR_ARG;
    }
    if (!counters) { gsr_set_error("gsr_bin_offsets_views: null counters"); return GSR_ERR_ARG; }
    if ((unsigned long long)n * num_views >= (1ull << 32)) { gsr_set_error("gsr_bin_offsets_views: too many keys"); return GSR_ERR_ARG; }
    if (n == 0) {
        for (uint32_t k = 0; k < num_views; ++k) {
            (void)hipMemsetAsync(&counters[k].num_pairs, 0, sizeof(uint32_t), st);
            if (slot_seq) (void)hipMemsetAsync(slot_seq + k, 0, sizeof(uint32_t), st);
        }
        return GSR_OK;
    }
    if (!depth_key || !ntiles || !order || (!slot_base && !slot_seq)) { gsr_set_error("gsr_bin_offsets_views: null pointer"); return GSR_ERR_ARG; }
    const size_t arr = align256((size_t)n * num_views * sizeof(uint32_t));
    char *ws = (char *)workspace;
    uint32_t *ids_b = (uint32_t *)ws;             // second value buffer of the sort
    uint32_t *keys_b = (uint32_t *)(ws + arr);    // second key buffer of the sort
    void *sort_ws = ws + 2 * arr;
    const size_t sort_bytes = (size_t)num_views * gsr_sort_workspace_bytes(n);
    void *scan_ws = (char *)sort_ws + sort_bytes;
    const size_t scan_bytes = (size_t)num_views * gsr_scan_workspace_bytes(n);
    bool in_first;
    if (int e = gsr_sort_pingpong32_batch(depth_key, order, keys_b, ids_b, n, num_views, 0, 32, sort_ws, sort_bytes, stream,
                                          &in_first, true))
        return e;
    if (!in_first) { gsr_set_error("gsr_bin_offsets_views: unexpected number of sort passes"); return GSR_ERR_LAUNCH; }
    if (int e = gsr_exclusive_scan_along_batch(ntiles, order, slot_base, n, num_views, &counters->num_pairs, &counters->error,
                                               (uint32_t)(sizeof(gsr_counters) / sizeof(uint32_t)), scan_ws, scan_bytes, stream,
                                               slot_seq, keys_b))
        return e;
    return gsr_check_launch("bin_offsets_views");
}

namespace {
// slot_base[order[s]] = slot_seq[s] for the views of a step (blockIdx.y = view; n + 1 entries of slot_seq per view)
__global__ __launch_bounds__(256) void slot_bases_kernel(const uint32_t *__restrict__ order, const uint32_t *__restrict__ slot_seq,
                                                         uint32_t n, uint32_t *__restrict__ slot_base)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const size_t at = (size_t)blockIdx.y * n;
    slot_base[at + order[at + s]] = slot_seq[at + blockIdx.y + s];
}
}  // namespace

extern "C" int gsr_slot_bases(const uint32_t *order, const uint32_t *slot_seq, uint32_t n, uint32_t num_views,
                              uint32_t *slot_base, void *stream)
{
    if (n == 0 || num_views == 0) return GSR_OK;
    if (!order || !slot_seq || !slot_base) { gsr_set_error("gsr_slot_bases: null pointer"); return GSR_ERR_ARG; }
    hipLaunchKernelGGL(slot_bases_kernel, dim3(gsr_div_up(n, 256), num_views), dim3(256), 0, (hipStream_t)stream, order,
                       slot_seq, n, slot_base);
    return gsr_check_launch("slot_bases");
}

extern "C" size_t gsr_bin_lists_workspace_bytes(uint32_t num_pairs)
{
    const size_t arr = align256((size_t)(num_pairs ? num_pairs : 1) * sizeof(uint32_t));
    return 6 * arr + gsr_sort_workspace_bytes(num_pairs);   // 2 key arrays + 2 arrays of 64-bit payloads
}

extern "C" int gsr_bin_lists(const gsr_view *view_host, const uint32_t *order, const uint32_t *slot_base,
                             const uint32_t *slot_seq, const int32_t *bbox, const uint32_t *ntiles, const float *rec, uint32_t num_pairs,
                             uint32_t *tile_offsets, uint32_t *pair_slot, uint32_t *sorted_gauss,
                             float *stream_rec, void *workspace, size_t workspace_bytes, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    ViewK v;
    if (int e = gsr_make_view(view_host, &v)) return e;
    if (!tile_offsets) { gsr_set_error("gsr_bin_lists: null tile_offsets"); return GSR_ERR_ARG; }
    if (num_pairs == 0) {
        (void)hipMemsetAsync(tile_offsets, 0, (size_t)(v.T + 1) * sizeof(uint32_t), st);
        return GSR_OK;
    }
    if ((unsigned long long)num_pairs * GSR_STREAM_DWORDS * 4ull >= (1ull << 32)) {
        gsr_set_error("gsr_bin_lists: %u pairs: the record stream would reach 4 GiB (32-bit stream offsets)", num_pairs);
        return GSR_ERR_ARG;
    }
    if (!workspace || workspace_bytes < gsr_bin_lists_workspace_bytes(num_pairs)) {
        gsr_set_error("gsr_bin_lists: workspace too small");
        return GSR_ERR_ARG;
    }
    if (!order || !slot_base || !bbox || !ntiles || !rec || !sorted_gauss || !stream_rec) {
        gsr_set_error("gsr_bin_lists: null pointer");
        return GSR_ERR_ARG;
    }
    const size_t arr = align256((size_t)num_pairs * sizeof(uint32_t));
    char *ws = (char *)workspace;
    uint32_t *tile_key = (uint32_t *)ws;
    uint32_t *sorted_tile = (uint32_t *)(ws + arr);
    uint64_t *owner_slot = (uint64_t *)(ws + 2 * arr);
    uint64_t *sorted_owner_slot = (uint64_t *)(ws + 4 * arr);
    void *sort_ws = ws + 6 * arr;
    const size_t sort_bytes = workspace_bytes - 6 * arr;
    // One word per pair whenever tile, Gaussian and slot fit 64 bits together (1 M Gaussians at 1080p: 13 + 20 + 23;
    // 6 M at 2160p: 15 + 23 + 25), else a 32-bit tile key beside a 64-bit (Gaussian, slot) value.
    const PairBits bits = {bits_for(num_pairs), bits_for((uint32_t)v.n)};
    const int tile_bits = bits_for((uint32_t)v.T);
    const bool packed = bits.slot + bits.gauss + tile_bits <= 64 && bits.gauss < 32 && bits.slot < 32 && !gsr_force_wide_pairs();
    bool in_first;
    if (packed) {
        hipLaunchKernelGGL(emit_pairs_kernel, dim3(gsr_div_up(v.n, 256)), dim3(256), 0, st, v, order, slot_seq, slot_base,
                           bbox, ntiles, (uint32_t *)nullptr, owner_slot, num_pairs, bits, ListSegs{});
        if (int e = gsr_sort_words64(owner_slot, sorted_owner_slot, num_pairs, bits.slot + bits.gauss,
                                     bits.slot + bits.gauss + tile_bits, sort_ws, sort_bytes, stream, &in_first))
            return e;
        GSR_LAUNCH_FINALIZE(v.L <= 16, dim3(gsr_div_up(num_pairs, 256)), dim3(256), 0, st, v,
                           (const uint32_t *)nullptr, in_first ? owner_slot : sorted_owner_slot, rec, num_pairs, (uint32_t)v.T,
                           tile_offsets, pair_slot, sorted_gauss, stream_rec, bits, ListSegs{});
        return gsr_check_launch("bin_lists");
    }
    hipLaunchKernelGGL(emit_pairs_kernel, dim3(gsr_div_up(v.n, 256)), dim3(256), 0, st, v, order, slot_seq, slot_base,
                       bbox, ntiles, tile_key, owner_slot, num_pairs, bits, ListSegs{});
    if (int e = gsr_sort_pingpong64(tile_key, owner_slot, sorted_tile, sorted_owner_slot, num_pairs, 0, tile_bits, sort_ws,
                                    sort_bytes, stream, &in_first))
        return e;
    GSR_LAUNCH_FINALIZE(v.L <= 16, dim3(gsr_div_up(num_pairs, 256)), dim3(256), 0, st, v,
                       in_first ? tile_key : sorted_tile, in_first ? owner_slot : sorted_owner_slot, rec, num_pairs,
                       (uint32_t)v.T, tile_offsets, pair_slot, sorted_gauss, stream_rec, bits, ListSegs{});
    return gsr_check_launch("bin_lists");
}

// ---- the lists of several views at once ------------------------------------------------------------------------
static bool lists_views_fit(const ViewK &v, uint32_t most_pairs, PairBits *bits, int *tile_bits)
{
    bits->slot = bits_for(most_pairs);
    bits->gauss = bits_for((uint32_t)v.n);
    *tile_bits = bits_for((uint32_t)v.T);
    return bits->slot + bits->gauss + *tile_bits <= 64 && bits->gauss < 32 && bits->slot < 32 && !gsr_force_wide_pairs();
}

static int lists_views_check(const gsr_view *views_host, uint32_t num_views, const uint32_t *num_pairs, ViewK *v, uint32_t *most)
{
    if (!views_host || !num_pairs || num_views == 0 || num_views > GSR_MAX_SEGS) {
        gsr_set_error("gsr_bin_lists_views: %u views (1..%d expected)", num_views, GSR_MAX_SEGS);
        return GSR_ERR_ARG;
    }
    if (int e = gsr_make_view(&views_host[0], v)) return e;
    *most = 0;
    for (uint32_t k = 0; k < num_views; ++k) {
        ViewK o;
        if (int e = gsr_make_view(&views_host[k], &o)) return e;
        if (o.n != v->n || o.W != v->W || o.H != v->H || o.L != v->L) {
            gsr_set_error("gsr_bin_lists_views: view %u differs from view 0 in the number of Gaussians, the image size or the tile edge", k);
            return GSR_ERR_ARG;
        }
        if ((unsigned long long)num_pairs[k] * GSR_STREAM_DWORDS * 4ull >= (1ull << 32)) {
            gsr_set_error("gsr_bin_lists_views: %u pairs in view %u: the record stream would reach 4 GiB", num_pairs[k], k);
            return GSR_ERR_ARG;
        }
        *most = num_pairs[k] > *most ? num_pairs[k] : *most;
    }
    return GSR_OK;
}

extern "C" int gsr_bin_lists_views_supported(const gsr_view *views_host, uint32_t num_views, const uint32_t *num_pairs)
{
    ViewK v;
    uint32_t most;
    if (lists_views_check(views_host, num_views, num_pairs, &v, &most)) return 0;
    PairBits bits;
    int tile_bits;
    return lists_views_fit(v, most, &bits, &tile_bits) ? 1 : 0;
}

// Element offset of every view's pairs in the concatenated per-pair arrays (multiples of 4096) and their total length.
extern "C" uint64_t gsr_bin_lists_views_layout(uint32_t num_views, const uint32_t *num_pairs, uint32_t *pair_base)
{
    uint64_t at = 0;
    for (uint32_t k = 0; k < num_views; ++k) {
        if (pair_base) pair_base[k] = (uint32_t)at;
        at += (uint64_t)gsr_div_up(num_pairs[k], 4096u) * 4096u;
    }
    return at ? at : 4096u;
}

extern "C" size_t gsr_bin_lists_views_workspace_bytes(uint32_t num_views, const uint32_t *num_pairs)
{
    const uint64_t total = gsr_bin_lists_views_layout(num_views, num_pairs, nullptr);
    return 2 * align256((size_t)total * sizeof(uint64_t)) + gsr_sort_words64_segs_bytes(num_pairs, num_views);
}

extern "C" int gsr_bin_lists_views(const gsr_view *views_host, uint32_t num_views, const uint32_t *order, const uint32_t *slot_base,
                                   const uint32_t *slot_seq, const int32_t *bbox, const uint32_t *ntiles, const float *rec,
                                   const uint32_t *num_pairs, uint32_t *tile_offsets, uint32_t *tile_order, uint32_t *sorted_gauss,
                                   float *stream_rec, void *workspace, size_t workspace_bytes, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    ViewK v;
    uint32_t most;
    if (int e = lists_views_check(views_host, num_views, num_pairs, &v, &most)) return e;
    PairBits bits;
    int tile_bits;
    if (!lists_views_fit(v, most, &bits, &tile_bits)) {
        gsr_set_error("gsr_bin_lists_views: tile, Gaussian and slot do not fit one 64-bit word (see gsr_bin_lists_views_supported)");
        return GSR_ERR_ARG;
    }
    if (!order || !slot_base || !bbox || !ntiles || !rec || !tile_offsets || !sorted_gauss || !stream_rec || !workspace ||
        workspace_bytes < gsr_bin_lists_views_workspace_bytes(num_views, num_pairs)) {
        gsr_set_error("gsr_bin_lists_views: null pointer or workspace too small");
        return GSR_ERR_ARG;
    }
    ListSegs segs = {};
    segs.count = num_views;
    uint32_t base[GSR_MAX_SEGS];
    const uint64_t total = gsr_bin_lists_views_layout(num_views, num_pairs, base);
    if (total >= (1ull << 32)) { gsr_set_error("gsr_bin_lists_views: too many pairs in all"); return GSR_ERR_ARG; }
    for (uint32_t k = 0; k < num_views; ++k) { segs.n[k] = num_pairs[k]; segs.base[k] = base[k]; }
    // views without pairs have all-zero tile offsets; the others are written in full by finalize_lists
    (void)hipMemsetAsync(tile_offsets, 0, (size_t)num_views * (size_t)(v.T + 1) * sizeof(uint32_t), st);
    if (most == 0) return GSR_OK;
    const size_t arr = align256((size_t)total * sizeof(uint64_t));
    uint64_t *words = (uint64_t *)workspace, *words_b = (uint64_t *)((char *)workspace + arr);
    void *sort_ws = (char *)workspace + 2 * arr;
    hipLaunchKernelGGL(emit_pairs_kernel, dim3(gsr_div_up(v.n, 256), num_views), dim3(256), 0, st, v, order, slot_seq, slot_base,
                       bbox, ntiles, (uint32_t *)nullptr, words, 0u, bits, segs);
    bool in_first;
    if (int e = gsr_sort_words64_segs(words, words_b, segs.n, segs.base, num_views, bits.slot + bits.gauss,
                                      bits.slot + bits.gauss + tile_bits, sort_ws, workspace_bytes - 2 * arr, stream, &in_first))
        return e;
    GSR_LAUNCH_FINALIZE(v.L <= 16, dim3(gsr_div_up(most, 256), num_views), dim3(256), 0, st, v, (const uint32_t *)nullptr,
                       in_first ? words : words_b, rec, 0u, (uint32_t)v.T, tile_offsets, (uint32_t *)nullptr, sorted_gauss, stream_rec,
                       bits, segs);
    if (tile_order)
        hipLaunchKernelGGL(tile_order_kernel, dim3(1, num_views), dim3(kOrderThreads), 0, st, (uint32_t)v.T, tile_offsets, tile_order);
    return gsr_check_launch("bin_lists_views");
}

