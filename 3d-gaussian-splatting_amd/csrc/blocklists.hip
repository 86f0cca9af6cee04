// Per-block lists for tiles of 2x2, 3x3 or 4x4 blocks (tile edges 32, 48, 64 - the reference's default is 64,
// GaussianRenderer.py:49).
//
// The reference blends the WHOLE list of a tile into every pixel of the tile (GaussianRenderer.py:100-112), and alpha is
// clamped from below at 0.01, so a Gaussian that does not reach a 16x16 block of its tile still contributes there:
// alpha = 0.01 on every pixel of the block, the same for all of them.  At a tile edge of 64 that is 89 % of the entries a
// block walks, in runs of nine.  A run of n such entries is one affine map of the block's state,
//     T <- T * 0.99^n,      colour <- colour + T * sum_i 0.99^i * 0.01 * c_i,
// whose two constants do not depend on the pixel.  This file turns the list of a tile into one COMPACT stream per block:
// the entries the block has to evaluate keep their 64-byte record (in the layout of a one-block tile), every run of
// entries that are clamped on the whole block becomes ONE record that carries the two constants.  The compositing
// kernels of one-block tiles then walk these streams (composite.hip, `RUNS`), a fifth of the records.
//
// Runs end at multiples of 32 list positions of their tile (a wave handles 64 consecutive entries of a tile on its own,
// and the index of an entry inside its run fits five bits), so the kernels below need no carries between waves.
//
// Outputs, for a view of T16 = ceil(W/16) * ceil(H/16) blocks (block id = Y16 + ceil(H/16) * X16, the tile id of the
// same view at a tile edge of 16):
//   block_offsets[T16 + 1]  range of each block's records in block_stream
//   block_stream[R, 16]     the records; every record carries its index inside its block's stream - the compositing
//                           backward writes row r (the record's position in block_stream) for it, so a wave's stores
//                           follow its walk through the stream
//   combo[I * S^2]          per list position p and block c (32-bit words, 64-bit ones from 2^26 - 1 combinations on):
//                           all ones = the block lies outside the image, else r << 6 | 32 = the block evaluates the entry
//                           (record r), or r << 6 | index inside run r
#include "gsr_common.hpp"

#include <stdlib.h>

namespace {

constexpr uint32_t kChunk = 64;
constexpr uint32_t kEvaluatedBit = 32u, kRunIndexMask = 31u;
template <class W> __device__ constexpr W outside() { return ~(W)0; }   // combination word of a block outside the image
constexpr unsigned long long kRunBreaks = (1ull << 32) | 1ull;   // lanes at which a run always starts anew
constexpr float kAlphaFloor = 0.01f;

struct BlockGrid {
    int S, S2, nbw16, nbh16, T16;
};
__host__ __device__ inline BlockGrid block_grid(const ViewK &v)
{
    BlockGrid g;
    g.S = v.L / 16;
    g.S2 = g.S * g.S;
    g.nbw16 = (v.W + 15) / 16;
    g.nbh16 = (v.H + 15) / 16;
    g.T16 = g.nbw16 * g.nbh16;
    return g;
}

// exclusive scan of f(0..n-1) into out[0..n] by one workgroup of 1024 threads
template <class F>
__device__ inline void workgroup_scan(F f, uint32_t n, uint32_t *__restrict__ out, uint32_t *s)
{
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t begin = min(tid * per, n), end = min(begin + per, n);
    uint32_t sum = 0;
    for (uint32_t i = begin; i < end; ++i) sum += f(i);
    s[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {
        const uint32_t t = tid >= d ? s[tid - d] : 0u;
        __syncthreads();
        s[tid] += t;
        __syncthreads();
    }
    uint32_t run = s[tid] - sum;
    for (uint32_t i = begin; i < end; ++i) { out[i] = run; run += f(i); }
    if (tid == 1023u) out[n] = s[1023];
    __syncthreads();
}

// chunks of 64 list positions: per tile (chunk_base_tile[T + 1]) and per block, in block-id order
// (chunk_base_block[T16 + 1]; a block has the chunks of its tile)
__global__ __launch_bounds__(1024) void chunk_bases_kernel(ViewK v, const uint32_t *__restrict__ tile_offsets,
                                                           uint32_t *__restrict__ chunk_base_tile,
                                                           uint32_t *__restrict__ chunk_base_block)
{
    __shared__ uint32_t s[1024];
    const BlockGrid g = block_grid(v);
    auto chunks_of_tile = [&](uint32_t t) { return (tile_offsets[t + 1] - tile_offsets[t] + kChunk - 1u) / kChunk; };
    workgroup_scan(chunks_of_tile, (uint32_t)v.T, chunk_base_tile, s);
    auto chunks_of_block = [&](uint32_t b) {
        const int X16 = (int)b / g.nbh16, Y16 = (int)b - X16 * g.nbh16;
        return chunks_of_tile((uint32_t)((Y16 / g.S) + v.nbh * (X16 / g.S)));
    };
    workgroup_scan(chunks_of_block, (uint32_t)g.T16, chunk_base_block, s);
}

// the wave's chunk: tile t, chunk k of it
__device__ inline bool wave_chunk(const ViewK &v, const uint32_t *__restrict__ chunk_base_tile, uint32_t &t, uint32_t &k)
{
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    if (w >= chunk_base_tile[v.T]) return false;
    uint32_t lo = 0, hi = (uint32_t)v.T;       // first index with chunk_base_tile[idx] > w lies in (lo, hi]
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (chunk_base_tile[mid] <= w) lo = mid; else hi = mid;
    }
    t = lo;
    k = w - chunk_base_tile[lo];
    return true;
}

struct BlockOfTile {
    bool inside;
    uint32_t id;   // block id in the image's 16-pixel grid
};
__device__ inline BlockOfTile block_of_tile(const ViewK &v, const BlockGrid &g, uint32_t t, int c)
{
    const int xi = (int)t / v.nbh, yi = (int)t - xi * v.nbh;
    const int bx = c / g.S, by = c - bx * g.S;
    const int X16 = xi * g.S + bx, Y16 = yi * g.S + by;
    BlockOfTile b;
    b.inside = X16 < g.nbw16 && Y16 < g.nbh16;
    b.id = (uint32_t)(Y16 + g.nbh16 * X16);
    return b;
}

// lanes that leave a record in the block's stream: every evaluated entry, and the last entry of every run
// `joined`: the clamped lanes that continue the run of the lane below them
__device__ inline unsigned long long emitters(bool valid, bool clamped, int lane, unsigned long long &cm, unsigned long long &joined)
{
    cm = __ballot(clamped);
    joined = cm & (cm << 1) & ~kRunBreaks;
    const bool next_joined = lane < 63 && ((joined >> (lane + 1)) & 1ull);
    return __ballot(valid && (!clamped || !next_joined));
}

__global__ __launch_bounds__(256) void block_count_kernel(ViewK v, const uint32_t *__restrict__ tile_offsets,
                                                          const float *__restrict__ stream_rec,
                                                          const uint32_t *__restrict__ chunk_base_tile,
                                                          const uint32_t *__restrict__ chunk_base_block,
                                                          uint32_t *__restrict__ counts)
{
    uint32_t t, k;
    if (!wave_chunk(v, chunk_base_tile, t, k)) return;
    const BlockGrid g = block_grid(v);
    const int lane = threadIdx.x & 63;
    const uint32_t p = tile_offsets[t] + k * kChunk + (uint32_t)lane;
    const bool valid = p < tile_offsets[t + 1];
    const uint32_t flags = valid ? __float_as_uint(stream_rec[(size_t)p * GSR_STREAM_DWORDS + 11]) : 0u;
    // lane c keeps the count of block c and stores it after the loop (one dependent load per wave, not one per block)
    const BlockOfTile own = block_of_tile(v, g, t, min(lane, g.S2 - 1));
    const bool owner = lane < g.S2 && own.inside;
    const uint32_t own_at = owner ? chunk_base_block[own.id] + k : 0u;
    uint32_t own_count = 0;
    for (int c = 0; c < g.S2; ++c) {
        const bool clamped = valid && ((flags >> (2 * c)) & 3u) == 3u;
        unsigned long long cm, joined;
        const unsigned long long em = emitters(valid, clamped, lane, cm, joined);
        if (lane == c) own_count = (uint32_t)__popcll(em);
    }
    if (owner) counts[own_at] = own_count;
}

// 0.99^n for n = 0 .. 64, each the exact power of the float 0.99 rounded ONCE (formed in double at compile time).  A
// product of rounded float factors would have the same error for every run of the same length, and over the few hundred
// runs of a list that adds up (2e-5 of the transmittance at a tile edge of 64) instead of averaging out as the roundings
// of the entry-by-entry product do.
struct Pow99 {
    float v[kChunk + 1];
    constexpr Pow99() : v()
    {
        double p = 1.0;
        for (uint32_t i = 0; i <= kChunk; ++i) { v[i] = (float)p; p *= (double)(1.0f - kAlphaFloor); }
    }
};
__constant__ Pow99 kPow99 = Pow99();

template <int S, class W>
__global__ __launch_bounds__(256) void block_write_kernel(ViewK v, const uint32_t *__restrict__ tile_offsets,
                                                          const float *__restrict__ stream_rec,
                                                          const uint32_t *__restrict__ chunk_base_tile,
                                                          const uint32_t *__restrict__ chunk_base_block,
                                                          const uint32_t *__restrict__ first_record,   // scanned counts
                                                          float *__restrict__ block_stream, W *__restrict__ combo)
{
    constexpr int S2 = S * S;
    __shared__ float pow99[kChunk + 1];
    // the records a wave leaves in one block's stream are consecutive there: they are put together in LDS and leave as
    // whole 16-byte pieces of consecutive addresses (a lane storing its own record costs the address unit a cache line
    // per lane and instruction: 0.19 -> 0.16 ms for the pass at 1 M Gaussians, 1080p, tile edge 64)
    __shared__ float4 staged[4 * kChunk * 4];
    if (threadIdx.x <= kChunk) pow99[threadIdx.x] = kPow99.v[threadIdx.x];
    __syncthreads();
    float4 *wave_staged = staged + (threadIdx.x >> 6) * (kChunk * 4);
    uint32_t t, k;
    if (!wave_chunk(v, chunk_base_tile, t, k)) return;
    const BlockGrid g = block_grid(v);
    const int lane = threadIdx.x & 63;
    const uint32_t p = tile_offsets[t] + k * kChunk + (uint32_t)lane;
    const bool valid = p < tile_offsets[t + 1];
    float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, q2 = q0, q3 = q0;
    if (valid) {
        const float4 *src = reinterpret_cast<const float4 *>(stream_rec + (size_t)p * GSR_STREAM_DWORDS);
        q0 = src[0]; q1 = src[1]; q2 = src[2]; q3 = src[3];   // y x D' A' | o B' r ca | g slot b flags | cbh 0 cd 0
    }
    const uint32_t flags = __float_as_uint(q2.w);
    const float red = q1.z, green = q2.x, blue = q2.z;
    const unsigned long long below = (1ull << lane) - 1ull;
    // lane c fetches where block c's records of this chunk start (one dependent load per wave, not one per block)
    const BlockOfTile own = block_of_tile(v, g, t, min(lane, S2 - 1));
    const bool owner = lane < S2 && own.inside;
    const uint32_t own_first = owner ? first_record[chunk_base_block[own.id] + k] : 0u;
    const uint32_t own_block_first = owner ? first_record[chunk_base_block[own.id]] : 0u;   // the block's first record
    W cmb[S2];   // the entry's word of every block: stored together after the loop (S2 consecutive words per entry)
#pragma unroll
    for (int c = 0; c < S2; ++c) {
        const BlockOfTile b = block_of_tile(v, g, t, c);
        const uint32_t first = (uint32_t)__shfl((int)own_first, c, 64);
        const uint32_t block_first = (uint32_t)__shfl((int)own_block_first, c, 64);
        cmb[c] = outside<W>();
        if (!b.inside) continue;
        const uint32_t f = (flags >> (2 * c)) & 3u;
        const bool clamped = valid && f == 3u;
        unsigned long long cm, joined;
        const unsigned long long em = emitters(valid, clamped, lane, cm, joined);
        const uint32_t rid = first + (uint32_t)__popcll(em & below);
        // index of the lane inside its run: distance to the nearest lane at or below it that starts a run
        const unsigned long long heads = cm & ~joined;
        const unsigned long long mine = heads & (below | (1ull << lane));
        const int idx = clamped ? lane - (63 - __clzll((long long)mine)) : 0;
        // the run's colour sum_i 0.99^i 0.01 c_i: every entry weighted by the power of its index, then an inclusive
        // sum inside the run
        const float wi = pow99[idx] * kAlphaFloor;
        float sr = wi * red, sg = wi * green, sb = wi * blue;
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) {
            const float r1 = __shfl_up(sr, d, 64), g1 = __shfl_up(sg, d, 64), b1 = __shfl_up(sb, d, 64);
            if (idx >= d) { sr += r1; sg += g1; sb += b1; }
        }
        const bool last = clamped && (lane == 63 || !((joined >> (lane + 1)) & 1ull));
        if (valid) cmb[c] = ((W)rid << 6) | (W)(clamped ? (uint32_t)idx : kEvaluatedBit);
        const uint32_t local = rid - block_first;   // the record's index in its block's stream
        if (valid && (!clamped || last)) {
            float4 *at = wave_staged + (rid - first) * 4u;
            if (!clamped) {
                // the entry's own record in the layout of a one-block tile: ... | g index+flags b b/100 | cbh r/100 cd g/100
                at[0] = q0; at[1] = q1;
                at[2] = make_float4(green, __uint_as_float(local | (f << 30)), blue, kAlphaFloor * blue);
                at[3] = make_float4(q3.x, kAlphaFloor * red, q3.z, kAlphaFloor * green);
            } else {
                // a run: dword 4 = 0.99^n, dword 5 = its reciprocal, dword 9 = the record's own index with both hint bits
                // set, dwords 11, 13, 15 = the run's colour (blue, red, green)
                const float P = pow99[idx + 1];
                at[0] = make_float4(0.f, 0.f, 0.f, 0.f);
                at[1] = make_float4(P, 1.0f / P, 0.f, 0.f);
                at[2] = make_float4(0.f, __uint_as_float(local | (3u << 30)), 0.f, sb);
                at[3] = make_float4(0.f, sr, 0.f, sg);
            }
        }
        __builtin_amdgcn_wave_barrier();
        {
            const uint32_t pieces = (uint32_t)__popcll(em) * 4u;
            float4 *dst = reinterpret_cast<float4 *>(block_stream + (size_t)first * GSR_STREAM_DWORDS);
            for (uint32_t q = (uint32_t)lane; q < pieces; q += 64u) dst[q] = wave_staged[q];
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (valid) {
        W *dst = combo + (size_t)p * (size_t)S2;
        constexpr int kPerPiece = 16 / (int)sizeof(W);
        if (S2 % kPerPiece == 0) {
#pragma unroll
            for (int c = 0; c < S2; c += kPerPiece) {
                uint4 piece;
                if (sizeof(W) == 4) piece = make_uint4((uint32_t)cmb[c], (uint32_t)cmb[c + 1], (uint32_t)cmb[c + 2], (uint32_t)cmb[c + 3]);
                else piece = make_uint4((uint32_t)cmb[c], (uint32_t)((unsigned long long)cmb[c] >> 32), (uint32_t)cmb[c + 1], (uint32_t)((unsigned long long)cmb[c + 1] >> 32));
                reinterpret_cast<uint4 *>(dst)[c / kPerPiece] = piece;
            }
        } else {
#pragma unroll
            for (int c = 0; c < S2; ++c) dst[c] = cmb[c];
        }
    }
}

__global__ __launch_bounds__(256) void block_offsets_kernel(uint32_t T16, const uint32_t *__restrict__ chunk_base_block,
                                                            const uint32_t *__restrict__ first_record,
                                                            const uint32_t *__restrict__ total, uint32_t *__restrict__ block_offsets)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > T16) return;
    const uint32_t i = chunk_base_block[b];
    block_offsets[b] = i < chunk_base_block[T16] ? first_record[i] : *total;
}

// One gradient record per pair from what the blocks of its tile left in block_grad (one row per record of the block
// streams, written by the compositing backward): the rows of the records with which blocks evaluated the entry, and for
// the blocks on which the entry was part of a run the colour gradient 0.01 * 0.99^i * U, U = the run's sum over the
// block's pixels of (transmittance in front of the run) x (cotangent), the first six floats of the run's row (red,
// green, four partial sums of blue).
// A workgroup takes 64 consecutive list positions; a wave walks them for one block at a time (wave w: blocks w, w + 4,
// ...), so that neighbouring lanes - neighbouring entries of a list, which mostly sit in the same run - read the same
// run sums.  The combination words arrive through LDS (loaded as they lie, read by block), the four waves' partial
// sums meet in LDS and are added in wave order: fixed orders throughout, the result is reproducible.
template <int S2, class W>
__global__ __launch_bounds__(256) void block_rows_reduce_kernel(uint32_t num_pairs, const float *__restrict__ stream_rec,
                                                                const W *__restrict__ combo,
                                                                const float *__restrict__ block_grad, float *__restrict__ pairgrad)
{
    __shared__ W words[S2 * 65];               // [block][entry], rows padded to 65
    __shared__ float part[4 * 9 * 64];         // [wave][sum][entry]
    const uint32_t p0 = blockIdx.x * 64u;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const uint32_t count = min(64u, num_pairs - p0);
    for (uint32_t i = threadIdx.x; i < count * (uint32_t)S2; i += 256u) {
        const uint32_t e = i / (uint32_t)S2, c = i - e * (uint32_t)S2;
        words[c * 65u + e] = combo[(size_t)p0 * S2 + i];
    }
    __syncthreads();
    float val[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) val[j] = 0.0f;
    if ((uint32_t)l < count) {
        // (all loads of the wave's blocks are issued before the first sum)
        constexpr int kTurns = (S2 + 3) / 4;
        W m[kTurns];
        float4 a[kTurns], b[kTurns], d[kTurns];
#pragma unroll
        for (int q = 0; q < kTurns; ++q) {
            const int c = w + 4 * q;
            m[q] = c < S2 ? words[c * 65 + l] : outside<W>();
            a[q] = b[q] = d[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m[q] == outside<W>()) continue;
            const float4 *src = reinterpret_cast<const float4 *>(block_grad + (size_t)(m[q] >> 6) * GSR_PAIRGRAD_FLOATS);
            a[q] = src[0]; b[q] = src[1];                                 // (a run: r g b0 b1 | b2 b3 - -)
            if (m[q] & kEvaluatedBit) d[q] = src[2];
        }
#pragma unroll
        for (int q = 0; q < kTurns; ++q) {
            if (m[q] == outside<W>()) continue;
            if (m[q] & kEvaluatedBit) {
                val[0] += a[q].x; val[1] += a[q].y; val[2] += a[q].z; val[3] += a[q].w;
                val[4] += b[q].x; val[5] += b[q].y; val[6] += b[q].z; val[7] += b[q].w;
                val[8] += (d[q].x + d[q].y) + (d[q].z + d[q].w);
            } else {
                const float wgt = kAlphaFloor * __builtin_amdgcn_exp2f((float)((uint32_t)m[q] & kRunIndexMask) * -0.014499569695115089f);   // 0.01 * 0.99^i
                val[6] = fmaf(wgt, a[q].x, val[6]);
                val[7] = fmaf(wgt, a[q].y, val[7]);
                val[8] = fmaf(wgt, (a[q].z + a[q].w) + (b[q].x + b[q].y), val[8]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) part[(w * 9 + j) * 64 + l] = val[j];
    __syncthreads();
    // 64 entries x 3 float4 of output: thread t writes float4 (t % 3) of entry t / 3 (the first 192 threads)
    const uint32_t e = threadIdx.x / 3u, q = threadIdx.x - e * 3u;
    if (threadIdx.x < 192u && e < count) {
        auto total = [&](int j) { return ((part[(0 * 9 + j) * 64 + e] + part[(1 * 9 + j) * 64 + e]) + part[(2 * 9 + j) * 64 + e]) + part[(3 * 9 + j) * 64 + e]; };
        const uint32_t slot = __float_as_uint(stream_rec[(size_t)(p0 + e) * GSR_STREAM_DWORDS + 9]);
        float4 *dst = reinterpret_cast<float4 *>(pairgrad + (size_t)slot * GSR_PAIRGRAD_FLOATS);
        dst[q] = q == 0 ? make_float4(total(0), total(1), total(2), total(3))
                        : (q == 1 ? make_float4(total(4), total(5), total(6), total(7)) : make_float4(total(8), 0.f, 0.f, 0.f));
    }
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Layout {
    uint32_t n_counts;
    size_t off_tile, off_block, off_counts, off_total, off_scan, scan_bytes, bytes;
};
inline Layout layout_of(const ViewK &v, uint32_t num_pairs)
{
    const BlockGrid g = block_grid(v);
    Layout l;
    // every block of a tile has the tile's chunks: at most S^2 (pairs / 64 + tiles) counts
    l.n_counts = (uint32_t)g.S2 * (num_pairs / kChunk + (uint32_t)v.T) + 1u;
    l.off_tile = 0;
    l.off_block = l.off_tile + align256(((size_t)v.T + 1) * 4);
    l.off_counts = l.off_block + align256(((size_t)g.T16 + 1) * 4);
    l.off_total = l.off_counts + align256((size_t)l.n_counts * 4);
    l.off_scan = l.off_total + 256;
    l.scan_bytes = gsr_scan_workspace_bytes(l.n_counts);
    l.bytes = l.off_scan + align256(l.scan_bytes);
    return l;
}

}  // namespace

// pair-block combinations of a view (the upper bound of its block records)
static unsigned long long combinations(const gsr_view *view_host, uint32_t num_pairs)
{
    const unsigned long long S = (unsigned long long)(view_host->tile_length / 16);
    return (unsigned long long)num_pairs * S * S;
}

// 64-bit combination words: from 2^26 - 1 combinations on (a record's index rides above six bits), or when
// GSR_BLOCK_WIDE=1 asks for them (read at every call: a test switches it inside one process)
static bool wide_words(const gsr_view *view_host, uint32_t num_pairs)
{
    const char *e = getenv("GSR_BLOCK_WIDE");
    return (e && e[0] == '1') || combinations(view_host, num_pairs) >= (1ull << 26) - 1ull;
}

extern "C" int gsr_block_lists_supported(const gsr_view *view_host, uint32_t num_pairs)
{
    if (!view_host) return 0;
    const int L = view_host->tile_length;
    if (!(L == 32 || L == 48 || L == 64)) return 0;
    // records are counted in 32 bits and the buffers are sized by the number of combinations: 2^28 of them are 31 GB
    return combinations(view_host, num_pairs) < (1ull << 28) ? 1 : 0;
}

extern "C" uint32_t gsr_block_combo_bytes(const gsr_view *view_host, uint32_t num_pairs)
{
    return view_host && wide_words(view_host, num_pairs) ? 8u : 4u;
}

extern "C" uint32_t gsr_block_count(const gsr_view *view_host)
{
    return view_host ? (uint32_t)(((view_host->width + 15) / 16) * ((view_host->height + 15) / 16)) : 0u;
}

extern "C" size_t gsr_block_lists_workspace_bytes(const gsr_view *view_host, uint32_t num_pairs)
{
    ViewK v;
    if (gsr_make_view(view_host, &v)) return 0;
    return layout_of(v, num_pairs).bytes;
}

template <class W>
static void launch_block_write(int S, dim3 grid, dim3 block, hipStream_t s, const ViewK &v, const uint32_t *tile_offsets,
                               const float *stream_rec, const uint32_t *chunk_base_tile, const uint32_t *chunk_base_block,
                               const uint32_t *counts, float *block_stream, W *combo)
{
    if (S == 2)
        hipLaunchKernelGGL((block_write_kernel<2, W>), grid, block, 0, s, v, tile_offsets, stream_rec, chunk_base_tile, chunk_base_block, counts,
                           block_stream, combo);
    else if (S == 3)
        hipLaunchKernelGGL((block_write_kernel<3, W>), grid, block, 0, s, v, tile_offsets, stream_rec, chunk_base_tile, chunk_base_block, counts,
                           block_stream, combo);
    else
        hipLaunchKernelGGL((block_write_kernel<4, W>), grid, block, 0, s, v, tile_offsets, stream_rec, chunk_base_tile, chunk_base_block, counts,
                           block_stream, combo);
}

template <class W>
static void launch_rows_reduce(int S, dim3 grid, dim3 block, hipStream_t s, uint32_t num_pairs, const float *stream_rec, const W *combo,
                               const float *block_grad, float *pairgrad)
{
    if (S == 2)
        hipLaunchKernelGGL((block_rows_reduce_kernel<4, W>), grid, block, 0, s, num_pairs, stream_rec, combo, block_grad, pairgrad);
    else if (S == 3)
        hipLaunchKernelGGL((block_rows_reduce_kernel<9, W>), grid, block, 0, s, num_pairs, stream_rec, combo, block_grad, pairgrad);
    else
        hipLaunchKernelGGL((block_rows_reduce_kernel<16, W>), grid, block, 0, s, num_pairs, stream_rec, combo, block_grad, pairgrad);
}

extern "C" int gsr_block_lists(const gsr_view *view_host, const uint32_t *tile_offsets, const float *stream_rec, uint32_t num_pairs,
                               uint32_t *block_offsets, float *block_stream, void *combo, void *workspace,
                               size_t workspace_bytes, void *stream)
{
    ViewK v;
    if (int e = gsr_make_view(view_host, &v)) return e;
    if (!gsr_block_lists_supported(view_host, num_pairs)) {
        gsr_set_error("gsr_block_lists: tile edge %d with %u pairs is not supported (edges 32, 48, 64; pairs x blocks per tile < 2^28)",
                      view_host->tile_length, num_pairs);
        return GSR_ERR_ARG;
    }
    if (!tile_offsets || !stream_rec || !block_offsets || !block_stream || !combo || !workspace) {
        gsr_set_error("gsr_block_lists: null pointer");
        return GSR_ERR_ARG;
    }
    const Layout l = layout_of(v, num_pairs);
    if (workspace_bytes < l.bytes) { gsr_set_error("gsr_block_lists: workspace too small"); return GSR_ERR_ARG; }
    const BlockGrid g = block_grid(v);
    char *ws = static_cast<char *>(workspace);
    uint32_t *chunk_base_tile = reinterpret_cast<uint32_t *>(ws + l.off_tile);
    uint32_t *chunk_base_block = reinterpret_cast<uint32_t *>(ws + l.off_block);
    uint32_t *counts = reinterpret_cast<uint32_t *>(ws + l.off_counts);
    uint32_t *total = reinterpret_cast<uint32_t *>(ws + l.off_total);
    hipStream_t s = (hipStream_t)stream;
    (void)hipMemsetAsync(counts, 0, (size_t)l.n_counts * 4, s);
    hipLaunchKernelGGL(chunk_bases_kernel, dim3(1), dim3(1024), 0, s, v, tile_offsets, chunk_base_tile, chunk_base_block);
    const unsigned waves = num_pairs / kChunk + (unsigned)v.T;   // at least the chunks of all tiles
    hipLaunchKernelGGL(block_count_kernel, dim3(gsr_div_up(waves, 4)), dim3(256), 0, s, v, tile_offsets, stream_rec, chunk_base_tile,
                       chunk_base_block, counts);
    if (int e = gsr_exclusive_scan(counts, counts, l.n_counts, total, ws + l.off_scan, l.scan_bytes, stream)) return e;
    const dim3 grid(gsr_div_up(waves, 4)), block(256);
    if (wide_words(view_host, num_pairs))
        launch_block_write(g.S, grid, block, s, v, tile_offsets, stream_rec, chunk_base_tile, chunk_base_block, counts, block_stream,
                           static_cast<unsigned long long *>(combo));
    else
        launch_block_write(g.S, grid, block, s, v, tile_offsets, stream_rec, chunk_base_tile, chunk_base_block, counts, block_stream,
                           static_cast<uint32_t *>(combo));
    hipLaunchKernelGGL(block_offsets_kernel, dim3(gsr_div_up((unsigned)g.T16 + 1u, 256)), dim3(256), 0, s, (uint32_t)g.T16,
                       chunk_base_block, counts, total, block_offsets);
    return gsr_check_launch("block_lists");
}

extern "C" int gsr_block_rows_reduce(const gsr_view *view_host, uint32_t num_pairs, const float *stream_rec, const void *combo,
                                     const float *block_grad, float *pairgrad, void *stream)
{
    ViewK v;
    if (int e = gsr_make_view(view_host, &v)) return e;
    if (num_pairs == 0) return GSR_OK;
    if (!gsr_block_lists_supported(view_host, num_pairs) || !stream_rec || !combo || !block_grad || !pairgrad) {
        gsr_set_error("gsr_block_rows_reduce: bad arguments");
        return GSR_ERR_ARG;
    }
    const BlockGrid g = block_grid(v);
    const dim3 grid(gsr_div_up(num_pairs, 64)), block(256);
    if (wide_words(view_host, num_pairs))
        launch_rows_reduce(g.S, grid, block, (hipStream_t)stream, num_pairs, stream_rec, static_cast<const unsigned long long *>(combo), block_grad, pairgrad);
    else
        launch_rows_reduce(g.S, grid, block, (hipStream_t)stream, num_pairs, stream_rec, static_cast<const uint32_t *>(combo), block_grad, pairgrad);
    return gsr_check_launch("block_rows_reduce");
}
