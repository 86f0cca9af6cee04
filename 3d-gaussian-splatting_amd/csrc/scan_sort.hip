// Device-wide exclusive scan and stable least-significant-digit radix sort of (u32 key, u32 value)
// pairs.  Both are bandwidth-bound passes over HBM; wave64 primitives (ballot, shuffles) do the
// in-block work.  Nothing here allocates: scratch comes from the caller's workspace.
#include <mutex>

#include "gsr_common.hpp"

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;  // 2048 elements per block

constexpr int RADIX_BITS = 8;
constexpr int RADIX_SIZE = 1 << RADIX_BITS;
constexpr int SORT_THREADS = 256;
constexpr int SORT_WAVES = SORT_THREADS / GSR_WAVE;
constexpr int SORT_ITEMS = 16;
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;  // 4096 keys per block

__device__ inline uint32_t wave_inclusive_scan(uint32_t x)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    return x;
}

// Exclusive scan across the block of one value per thread; returns the exclusive prefix and the
// block total.  lds must hold blockDim.x/64 + 1 words.
__device__ inline uint32_t block_exclusive_scan(uint32_t x, uint32_t *lds, uint32_t &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint32_t inc = wave_inclusive_scan(x);
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int w = 0; w < nw; ++w) { const uint32_t t = lds[w]; lds[w] = run; run += t; }
        lds[nw] = run;
    }
    __syncthreads();
    const uint32_t base = lds[wave];
    total = lds[nw];
    __syncthreads();
    return base + inc - x;
}

// Several independent sequences of the same length can be processed by one launch: sequence y = blockIdx.y
// starts y * n elements into in / idx / out (y * (n + 1) into out_seq, y * nb into block_sums) and owns the
// y-th `Counters` record (total, overflow flag).
struct SegCounters {
    uint32_t *total;           // may be null
    uint32_t *overflow_flag;   // may be null
    uint32_t stride;           // words between the records of two sequences
};

// Which sequence and which of its blocks a workgroup of a (blocks, sequences) grid works on.  Consecutive workgroups go
// to consecutive XCDs (8, each with an L2 of its own), so the sequences are interleaved over the workgroup numbers:
// with eight sequences - the views of a step - every XCD gathers from and scatters into the arrays of ONE view, which
// its L2 holds, instead of all of them.
struct SeqBlock {
    uint32_t seq, block;
};
__device__ inline SeqBlock seq_block()
{
    if (gridDim.y == 1) return {0u, blockIdx.x};
    const uint32_t linear = blockIdx.y * gridDim.x + blockIdx.x;
    return {linear % gridDim.y, linear / gridDim.y};
}

// With idx != null element i of the sequence is in[idx[i]] and its result goes to out[idx[i]]
// (a scan along a permutation; in and out may then not alias).
// gathered (may be null, only with idx): the elements in sequence order, gathered[i] = in[idx[i]], written here so
// that scan_apply_kernel reads them in order instead of gathering a second time.
__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_kernel(const uint32_t *__restrict__ in,
                                                                   const uint32_t *__restrict__ idx, uint32_t n,
                                                                   uint32_t *__restrict__ block_sums, SegCounters cnt,
                                                                   uint32_t *__restrict__ gathered)
{
    __shared__ uint32_t lds[SCAN_THREADS / 64 + 1];
    __shared__ unsigned long long wide;
    const SeqBlock wg = seq_block();
    in += (size_t)wg.seq * n;
    if (idx) idx += (size_t)wg.seq * n;
    if (gathered) gathered += (size_t)wg.seq * n;
    block_sums += (size_t)wg.seq * gridDim.x;
    uint32_t *overflow_flag = cnt.overflow_flag ? cnt.overflow_flag + (size_t)wg.seq * cnt.stride : nullptr;
    // (only the block's total matters here: thread t takes the elements t, t + 256, ... of the block, so that the
    // index reads and the gathered writes of a wave are 256 consecutive bytes per instruction)
    const uint32_t base = wg.block * SCAN_TILE + threadIdx.x;
    uint32_t s = 0;
    unsigned long long s64 = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const uint32_t i = base + k * SCAN_THREADS;
        if (i < n) {
            const uint32_t x = in[idx ? idx[i] : i];
            if (gathered) gathered[i] = x;
            s += x;
            s64 += x;
        }
    }
    if (overflow_flag) {
        // the block's sum in 64 bits: a sum that does not fit 32 bits raises bit 1 of the flag word
        if (threadIdx.x == 0) wide = 0ull;
        __syncthreads();
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s64 += __shfl_xor(s64, d, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(&wide, s64);
        __syncthreads();
        if (threadIdx.x == 0 && (wide >> 32) != 0ull) atomicOr(overflow_flag, 2u);
    }
    uint32_t total;
    block_exclusive_scan(s, lds, total);
    if (threadIdx.x == 0) block_sums[wg.block] = total;
}

// One block: exclusive scan of block_sums[0..nb) in place, total to *total_out (may be null).
__global__ __launch_bounds__(1024) void scan_sums_kernel(uint32_t *__restrict__ block_sums, uint32_t nb, SegCounters cnt)
{
    __shared__ uint32_t lds[1024 / 64 + 1];
    __shared__ unsigned long long wide;
    block_sums += (size_t)blockIdx.y * nb;
    uint32_t *total_out = cnt.total ? cnt.total + (size_t)blockIdx.y * cnt.stride : nullptr;
    uint32_t *overflow_flag = cnt.overflow_flag ? cnt.overflow_flag + (size_t)blockIdx.y * cnt.stride : nullptr;
    if (threadIdx.x == 0) wide = 0ull;
    uint32_t carry = 0;
    unsigned long long mine = 0;
    for (uint32_t start = 0; start < nb; start += 1024) {
        const uint32_t i = start + threadIdx.x;
        const uint32_t x = i < nb ? block_sums[i] : 0u;
        mine += x;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan(x, lds, total);
        if (i < nb) block_sums[i] = carry + ex;
        carry += total;
    }
    if (overflow_flag) {
        // the grand total in 64 bits (bit 1 of the flag word: it does not fit 32 bits)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) atomicAdd(&wide, mine);
        __syncthreads();
        if (threadIdx.x == 0 && (wide >> 32) != 0ull) atomicOr(overflow_flag, 2u);
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry;
}

// out_seq (may be null): the same prefix sums in SEQUENCE order, out_seq[i] for i < n and the total at
// out_seq[n]
// gathered (may be null): the elements in sequence order, as scan_reduce_kernel left them - read in place of in[idx[i]]
__global__ __launch_bounds__(SCAN_THREADS) void scan_apply_kernel(const uint32_t *in, const uint32_t *__restrict__ idx,
                                                                  uint32_t *out, uint32_t n,
                                                                  const uint32_t *__restrict__ block_sums,
                                                                  uint32_t *__restrict__ out_seq,
                                                                  const uint32_t *__restrict__ gathered)
{
    __shared__ uint32_t lds[SCAN_THREADS / 64 + 1];
    const SeqBlock wg = seq_block();
    if (gathered) gathered += (size_t)wg.seq * n;
    in += (size_t)wg.seq * n;
    out += (size_t)wg.seq * n;
    if (idx) idx += (size_t)wg.seq * n;
    if (out_seq) out_seq += (size_t)wg.seq * (n + 1);
    block_sums += (size_t)wg.seq * gridDim.x;
    // Wave w takes the elements [w * 512, (w + 1) * 512) of the block's 2048, 64 consecutive ones per load: every
    // instruction of a wave reads (and, for out_seq, writes) 256 consecutive bytes.  The prefix of an element is the
    // sum of the rows before it in the wave, the lanes before it in its row, the waves before it in the block.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t base = wg.block * SCAN_TILE + wave * (SCAN_ITEMS * 64) + lane;
    uint32_t v[SCAN_ITEMS], at[SCAN_ITEMS], before[SCAN_ITEMS];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const uint32_t i = base + k * 64;
        // (out == null - only with gathered and out_seq: the results are wanted in sequence order only, the permutation
        // is not read)
        at[k] = i < n && (out || !gathered) ? (idx ? idx[i] : i) : 0u;
        v[k] = i < n ? (gathered ? gathered[i] : in[at[k]]) : 0u;
    }
    uint32_t rows = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const uint32_t inc = wave_inclusive_scan(v[k]);
        before[k] = rows + inc - v[k];
        rows += __shfl(inc, 63, 64);
    }
    if (lane == 0) lds[wave] = rows;
    __syncthreads();
    uint32_t run = block_sums[wg.block];
    for (int w = 0; w < wave; ++w) run += lds[w];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const uint32_t i = base + k * 64;
        if (i < n) {
            const uint32_t r = run + before[k];
            if (out) out[at[k]] = r;
            if (out_seq) {
                out_seq[i] = r;
                if (i == n - 1) out_seq[n] = r + v[k];
            }
        }
    }
}

// ---- radix sort -------------------------------------------------------------------------

// Sequences of DIFFERENT lengths sorted by one set of launches (blockIdx.y = sequence): sequence y has n[y] items
// from element base[y] of the buffers (a multiple of the 4096 items of a sort block), nb[y] sort blocks, and its
// per-block digit counts from word hbase[y] of the histogram buffer.  count == 0: the sequences all have the
// kernel's n (or there is one).
struct SortSegs {
    uint32_t count;
    uint32_t n[GSR_MAX_SEGS], base[GSR_MAX_SEGS], nb[GSR_MAX_SEGS], hbase[GSR_MAX_SEGS];
};
// a[y] without a dynamically indexed copy of the kernel arguments in scratch memory
__device__ inline uint32_t seg_pick(const uint32_t (&a)[GSR_MAX_SEGS], uint32_t y)
{
    uint32_t r = a[0];
#pragma unroll
    for (uint32_t k = 1; k < GSR_MAX_SEGS; ++k) r = y == k ? a[k] : r;
    return r;
}

// hist[digit * nb + block] = number of keys of the block's tile with that digit.
// K = uint32_t: the keys of a (key, value) sort; K = uint64_t: 64-bit words sorted by some of their own bits.
template <typename K>
__global__ __launch_bounds__(SORT_THREADS) void radix_hist_kernel(const K *__restrict__ keys, uint32_t n,
                                                                  int shift, uint32_t mask, uint32_t nb,
                                                                  uint32_t *__restrict__ hist, SortSegs segs)
{
    __shared__ uint32_t h[RADIX_SIZE];
    // (equally long sequences are interleaved over the workgroup numbers, see seq_block)
    const SeqBlock wg = segs.count ? SeqBlock{blockIdx.y, blockIdx.x} : seq_block();
    if (segs.count) {
        n = seg_pick(segs.n, wg.seq);
        nb = seg_pick(segs.nb, wg.seq);
        if (wg.block >= nb) return;
        keys += seg_pick(segs.base, wg.seq);
        hist += seg_pick(segs.hbase, wg.seq);
    } else {
        // (sequence blockIdx.y of a batch of equally long ones: y * n keys and y * 256 * nb counts further on)
        keys += (size_t)wg.seq * n;
        hist += (size_t)wg.seq * RADIX_SIZE * nb;
    }
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = wg.block * SORT_TILE;
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; ++it) {
        const uint32_t i = base + it * SORT_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & mask], 1u);
    }
    __syncthreads();
    hist[threadIdx.x * nb + wg.block] = h[threadIdx.x];
}

// Turns the per-block digit counts into per-block offsets inside each digit: block d of this kernel
// scans row d of hist (the nb counts of digit d, one per sort block) in place and writes the row total
// to digit_total[d].  The scatter kernel adds the digit's own base (a scan of the 256 totals).
__global__ __launch_bounds__(256) void radix_rows_kernel(uint32_t *__restrict__ hist, uint32_t nb,
                                                         uint32_t *__restrict__ digit_total, SortSegs segs)
{
    __shared__ uint32_t lds[256 / 64 + 1];
    if (segs.count) {
        nb = seg_pick(segs.nb, blockIdx.y);
        hist += seg_pick(segs.hbase, blockIdx.y);
    } else {
        hist += (size_t)blockIdx.y * RADIX_SIZE * nb;
    }
    digit_total += (size_t)blockIdx.y * RADIX_SIZE;
    uint32_t *row = hist + (size_t)blockIdx.x * nb;
    uint32_t carry = 0;
    for (uint32_t start = 0; start < nb; start += 256) {
        const uint32_t i = start + threadIdx.x;
        const uint32_t x = i < nb ? row[i] : 0u;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan(x, lds, total);
        if (i < nb) row[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) digit_total[blockIdx.x] = carry;
}

// Stable scatter of one tile.  Wave w owns keys [w*1024, (w+1)*1024) of the tile, item `it` of a
// wave is the 64 consecutive keys at it*64, so key order = (wave, item, lane).  Ranks inside a
// wave come from ballots (lanes with an equal digit), per-wave digit counters live in LDS.
// The tile is first put in digit order inside LDS and written out from there: consecutive threads
// then write consecutive addresses of a digit's run instead of 64 scattered words per instruction.
// (STAGED = false writes straight from registers: for digits that are spread evenly over the 256
// values - the depth bits - a digit's run inside a tile is only ~16 items and the detour through LDS
// does not pay.)
template <typename V>
__device__ inline uint32_t word_key(V word, int shift) { return (uint32_t)((unsigned long long)word >> shift); }

// WORDS (V = uint64_t, STAGED): there is no key array - the digit is taken from bits shift.. of the value itself.
// vals_in == null (not with WORDS): the value of an item is its position in its sequence (the first pass of an argsort).
template <typename V, bool STAGED, bool WORDS = false>
__global__ __launch_bounds__(SORT_THREADS) void radix_scatter_kernel(
    const uint32_t *__restrict__ keys_in, const V *__restrict__ vals_in, uint32_t *__restrict__ keys_out,
    V *__restrict__ vals_out, uint32_t n, int shift, uint32_t mask, uint32_t nb,
    const uint32_t *__restrict__ hist_rows, const uint32_t *__restrict__ digit_total, SortSegs segs)
{
    __shared__ uint32_t cnt[SORT_WAVES][RADIX_SIZE];   // per-wave digit counts, then per-wave local offsets
    __shared__ uint32_t lstart[RADIX_SIZE];            // first local position of a digit inside the tile
    __shared__ uint32_t gbase[RADIX_SIZE];             // first global position of the tile's run of a digit
    __shared__ uint32_t scan_lds[SORT_THREADS / 64 + 1];
    // (32-bit keys with 32-bit values are staged together, as one 64-bit word per item: one pass through LDS)
    constexpr bool PAIRED = STAGED && !WORDS && sizeof(V) == 4;
    __shared__ V stage[STAGED && !PAIRED ? SORT_TILE : 1];        // keys (as words), then values, in digit order
    __shared__ unsigned long long paired[PAIRED ? SORT_TILE : 1];
    uint32_t *stage_keys = reinterpret_cast<uint32_t *>(stage);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const SeqBlock wg = segs.count ? SeqBlock{blockIdx.y, blockIdx.x} : seq_block();
    if (segs.count) {
        n = seg_pick(segs.n, wg.seq);
        nb = seg_pick(segs.nb, wg.seq);
        if (wg.block >= nb) return;
        const uint32_t base = seg_pick(segs.base, wg.seq);
        if (!WORDS) { keys_in += base; keys_out += base; }
        if (vals_in) vals_in += base;
        vals_out += base;
        hist_rows += seg_pick(segs.hbase, wg.seq);
    } else {
        if (!WORDS) { keys_in += (size_t)wg.seq * n; keys_out += (size_t)wg.seq * n; }
        if (vals_in) vals_in += (size_t)wg.seq * n;
        vals_out += (size_t)wg.seq * n;
        hist_rows += (size_t)wg.seq * RADIX_SIZE * nb;
    }
    const int word_shift = shift;   // WORDS: position of the digit inside the 64-bit word
    if (WORDS) shift = 0;
    const int digit_bits = __popc(mask);
    digit_total += (size_t)wg.seq * RADIX_SIZE;
    // first position of digit d in the output: exclusive scan of the 256 digit totals
    uint32_t digit_base;
    {
        uint32_t all;
        digit_base = block_exclusive_scan(digit_total[threadIdx.x], scan_lds, all);
    }
#pragma unroll
    for (int w = 0; w < SORT_WAVES; ++w) cnt[w][threadIdx.x] = 0;
    __syncthreads();

    const uint32_t tile_base = wg.block * SORT_TILE;
    const uint32_t tile_count = min((uint32_t)SORT_TILE, n - tile_base);
    const uint32_t wbase = tile_base + wave * (SORT_ITEMS * 64);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t key[SORT_ITEMS], rank[SORT_ITEMS];
    V word[WORDS ? SORT_ITEMS : 1];
    volatile uint32_t *mycnt = cnt[wave];
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; ++it) {
        const uint32_t i = wbase + it * 64 + lane;
        const bool valid = i < n;
        // (WORDS: the 32 bits of the word from its digit on; the code below then finds the digit at bit 0)
        if (WORDS) {
            word[it] = valid ? vals_in[i] : (V)0;
            key[it] = valid ? word_key(word[it], word_shift) : 0xFFFFFFFFu;
        } else {
            key[it] = valid ? keys_in[i] : 0xFFFFFFFFu;
        }
        const uint32_t d = (key[it] >> shift) & mask;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < RADIX_BITS; ++b) {
            if (WORDS && b >= digit_bits) break;   // (a digit of fewer than eight bits: the others are zero in every lane)
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        const uint32_t prior = mycnt[d];
        const uint32_t below = (uint32_t)__popcll(peers & lt_mask);
        rank[it] = prior + below;
        __builtin_amdgcn_wave_barrier();
        if (valid && below == 0) mycnt[d] = prior + (uint32_t)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    {
        // thread d: the tile's count of digit d -> local start (scan over digits), per-wave offsets
        const uint32_t d = threadIdx.x;
        uint32_t c[SORT_WAVES], sum = 0;
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) { c[w] = cnt[w][d]; sum += c[w]; }
        const uint32_t gb = digit_base + hist_rows[d * nb + wg.block];
        uint32_t run;
        if (STAGED) {
            uint32_t all;
            run = block_exclusive_scan(sum, scan_lds, all);   // local positions
            lstart[d] = run;
            gbase[d] = gb;
        } else {
            run = gb;                                         // global positions right away
        }
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) { cnt[w][d] = run; run += c[w]; }
    }
    __syncthreads();
    if (!STAGED) {
#pragma unroll
        for (int it = 0; it < SORT_ITEMS; ++it) {
            const uint32_t i = wbase + it * 64 + lane;
            if (i < n) {
                const uint32_t d = (key[it] >> shift) & mask;
                const uint32_t pos = cnt[wave][d] + rank[it];
                keys_out[pos] = key[it];
                vals_out[pos] = vals_in ? vals_in[i] : (V)i;
            }
        }
        return;
    }
    if (WORDS) {
        // the words themselves into digit order (their digit is read back from the staged word)
#pragma unroll
        for (int it = 0; it < SORT_ITEMS; ++it) {
            const uint32_t i = wbase + it * 64 + lane;
            const uint32_t d = key[it] & mask;
            if (i < n) stage[cnt[wave][d] + rank[it]] = word[it];
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < SORT_ITEMS; ++it) {
            const uint32_t l = it * SORT_THREADS + threadIdx.x;
            if (l < tile_count) {
                const V word = stage[l];
                const uint32_t d = word_key(word, word_shift) & mask;
                vals_out[gbase[d] + (l - lstart[d])] = word;
            }
        }
        return;
    }
    if (PAIRED) {
#pragma unroll
        for (int it = 0; it < SORT_ITEMS; ++it) {
            const uint32_t i = wbase + it * 64 + lane;
            const uint32_t d = (key[it] >> shift) & mask;
            if (i < n) paired[cnt[wave][d] + rank[it]] = (unsigned long long)key[it] | ((unsigned long long)(vals_in ? (uint32_t)vals_in[i] : i) << 32);
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < SORT_ITEMS; ++it) {
            const uint32_t l = it * SORT_THREADS + threadIdx.x;
            if (l < tile_count) {
                const unsigned long long kv = paired[l];
                const uint32_t k = (uint32_t)kv;
                const uint32_t d = (k >> shift) & mask;
                const uint32_t at = gbase[d] + (l - lstart[d]);
                keys_out[at] = k;
                vals_out[at] = (V)(uint32_t)(kv >> 32);
            }
        }
        return;
    }
    // keys into digit order
    uint32_t lpos[SORT_ITEMS];
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; ++it) {
        const uint32_t i = wbase + it * 64 + lane;
        const uint32_t d = (key[it] >> shift) & mask;
        lpos[it] = cnt[wave][d] + rank[it];
        if (i < n) stage_keys[lpos[it]] = key[it];
    }
    __syncthreads();
    uint32_t gpos[SORT_ITEMS];
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; ++it) {
        const uint32_t l = it * SORT_THREADS + threadIdx.x;   // local position in digit order
        if (l < tile_count) {
            const uint32_t k = stage_keys[l];
            const uint32_t d = (k >> shift) & mask;
            gpos[it] = gbase[d] + (l - lstart[d]);
            if (!WORDS) keys_out[gpos[it]] = k;
        }
    }
    __syncthreads();
    // values the same way
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; ++it) {
        const uint32_t i = wbase + it * 64 + lane;
        if (i < n) stage[lpos[it]] = vals_in ? vals_in[i] : (V)i;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; ++it) {
        const uint32_t l = it * SORT_THREADS + threadIdx.x;
        if (l < tile_count) vals_out[gpos[it]] = stage[l];
    }
}

}  // namespace

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
static const SortSegs no_segs = {};

extern "C" size_t gsr_scan_workspace_bytes(uint32_t n)
{
    return align256((size_t)gsr_div_up(n ? n : 1, SCAN_TILE) * sizeof(uint32_t));
}

// library-internal: out[idx[i]] = sum over j < i of in[idx[j]]  (idx == null: the plain scan), for `nseq`
// independent sequences of n elements each that lie one after the other in every array (out_seq: n + 1 per
// sequence); total and overflow_flag point at the first sequence's words, `counter_stride` words apart from
// the next one's.  The workspace holds nseq times gsr_scan_workspace_bytes(n).
int gsr_exclusive_scan_along_batch(const uint32_t *in, const uint32_t *idx, uint32_t *out, uint32_t n, uint32_t nseq,
                                   uint32_t *total, uint32_t *overflow_flag, uint32_t counter_stride, void *workspace,
                                   size_t workspace_bytes, void *stream, uint32_t *out_seq, uint32_t *gathered)
{
    hipStream_t st = (hipStream_t)stream;
    if (nseq == 0) return GSR_OK;
    if (workspace_bytes < nseq * gsr_scan_workspace_bytes(n) || !workspace) {
        gsr_set_error("gsr_exclusive_scan: workspace too small (%zu < %zu)", workspace_bytes,
                      nseq * gsr_scan_workspace_bytes(n));
        return GSR_ERR_ARG;
    }
    if (n == 0) {
        for (uint32_t y = 0; y < nseq; ++y) {
            if (total) (void)hipMemsetAsync(total + (size_t)y * counter_stride, 0, sizeof(uint32_t), st);
            if (out_seq) (void)hipMemsetAsync(out_seq + y, 0, sizeof(uint32_t), st);
        }
        return GSR_OK;
    }
    if (!in || (!out && !(idx && gathered && out_seq))) { gsr_set_error("gsr_exclusive_scan: null pointer"); return GSR_ERR_ARG; }
    uint32_t *sums = (uint32_t *)workspace;
    const uint32_t nb = gsr_div_up(n, SCAN_TILE);
    const SegCounters cnt = {total, overflow_flag, counter_stride};
    if (!idx) gathered = nullptr;
    hipLaunchKernelGGL(scan_reduce_kernel, dim3(nb, nseq), dim3(SCAN_THREADS), 0, st, in, idx, n, sums, cnt, gathered);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1, nseq), dim3(1024), 0, st, sums, nb, cnt);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(nb, nseq), dim3(SCAN_THREADS), 0, st, in, idx, out, n, sums, out_seq,
                       (const uint32_t *)gathered);
    return gsr_check_launch("exclusive_scan");
}

int gsr_exclusive_scan_along(const uint32_t *in, const uint32_t *idx, uint32_t *out, uint32_t n, uint32_t *total,
                             void *workspace, size_t workspace_bytes, void *stream, uint32_t *out_seq,
                             uint32_t *overflow_flag, uint32_t *gathered)
{
    return gsr_exclusive_scan_along_batch(in, idx, out, n, 1, total, overflow_flag, 0, workspace, workspace_bytes, stream, out_seq, gathered);
}

extern "C" int gsr_exclusive_scan(const uint32_t *in, uint32_t *out, uint32_t n, uint32_t *total, void *workspace,
                                  size_t workspace_bytes, void *stream)
{
    return gsr_exclusive_scan_along(in, nullptr, out, n, total, workspace, workspace_bytes, stream, nullptr, nullptr, nullptr);
}

// ---- exclusive scan without a workspace argument ---------------------------------------------
// For callers that cannot allocate (code under HIP graph capture, tight serving loops): the block sums,
// which gsr_exclusive_scan keeps in the caller's workspace, live in a file-scope device array.  Every GPU
// the library runs on holds its own copy of the array at its own address for the life of the process;
// the address is looked up once per device, with that device current, and kept.
constexpr uint32_t kStaticScanMaxN = 1u << 24;                        // 16 777 216 elements
constexpr uint32_t kStaticScanSums = kStaticScanMaxN / SCAN_TILE;     // 8192 block sums = 32 KiB
static __device__ __align__(256) uint32_t scratch[kStaticScanSums];

constexpr int kMaxDevices = 64;
static uint32_t *g_scan_scratch[kMaxDevices];      // device address of `scratch` per device, null until looked up
static std::mutex g_scan_scratch_lock;

// The device address of this device's copy of `scratch` (never the array's name: in host code that is a
// host-side stand-in, not a device address).
static int static_scan_scratch(uint32_t **out)
{
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= kMaxDevices) {
        gsr_set_error("gsr_exclusive_scan_static: no current device (%s)", e != hipSuccess ? hipGetErrorString(e) : "index out of range");
        return GSR_ERR_LAUNCH;
    }
    std::lock_guard<std::mutex> hold(g_scan_scratch_lock);
    if (!g_scan_scratch[dev]) {
        void *ptr = nullptr;
        e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) {
            gsr_set_error("gsr_exclusive_scan_static: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
            return GSR_ERR_LAUNCH;
        }
        g_scan_scratch[dev] = (uint32_t *)ptr;
    }
    *out = g_scan_scratch[dev];
    return GSR_OK;
}

extern "C" int gsr_exclusive_scan_static(const uint32_t *in, uint32_t *out, uint32_t n, uint32_t *total, void *stream)
{
    if (n > kStaticScanMaxN) {
        gsr_set_error("gsr_exclusive_scan_static: %u elements exceed the built-in scratch (at most %u); use "
                      "gsr_exclusive_scan with a workspace", n, kStaticScanMaxN);
        return GSR_ERR_ARG;
    }
    uint32_t *sums = nullptr;
    if (int e = static_scan_scratch(&sums)) return e;
    // the kernels overwrite the block sums they read (one per 2048 elements), nothing has to start at zero
    return gsr_exclusive_scan_along(in, nullptr, out, n, total, sums, kStaticScanSums * sizeof(uint32_t), stream, nullptr,
                                    nullptr, nullptr);
}

extern "C" size_t gsr_sort_workspace_bytes(uint32_t n)
{
    const uint32_t nb = gsr_div_up(n ? n : 1, SORT_TILE);
    const size_t hist = align256((size_t)nb * RADIX_SIZE * sizeof(uint32_t));
    return hist + align256(RADIX_SIZE * sizeof(uint32_t));
}

// Sorts between the two buffer pairs; *result_in_first tells where the sorted data ended up (the
// *_a buffers after an even number of passes).  No copies.
// (nseq independent sequences of n pairs each, one after the other in all four buffers; the workspace holds
// nseq times gsr_sort_workspace_bytes(n))
template <typename V>
static int sort_pingpong(uint32_t *keys_a, V *vals_a, uint32_t *keys_b, V *vals_b, uint32_t n, int begin_bit, int end_bit,
                         void *workspace, size_t workspace_bytes, void *stream, bool *result_in_first, uint32_t nseq = 1,
                         bool positions_as_values = false)
{
    hipStream_t st = (hipStream_t)stream;
    *result_in_first = true;
    if (begin_bit < 0 || end_bit > 32 || begin_bit > end_bit) {
        gsr_set_error("gsr_sort_pairs: bad bit range [%d,%d)", begin_bit, end_bit);
        return GSR_ERR_ARG;
    }
    if (workspace_bytes < nseq * gsr_sort_workspace_bytes(n) || !workspace) {
        gsr_set_error("gsr_sort_pairs: workspace too small (%zu < %zu)", workspace_bytes, nseq * gsr_sort_workspace_bytes(n));
        return GSR_ERR_ARG;
    }
    if (n == 0 || nseq == 0) return GSR_OK;
    if (!keys_a || !vals_a || !keys_b || !vals_b) { gsr_set_error("gsr_sort_pairs: null pointer"); return GSR_ERR_ARG; }
    const uint32_t nb = gsr_div_up(n, SORT_TILE);
    uint32_t *hist = (uint32_t *)workspace;
    const size_t hist_bytes = align256((size_t)nseq * nb * RADIX_SIZE * sizeof(uint32_t));
    uint32_t *digit_total = (uint32_t *)((char *)workspace + hist_bytes);   // nseq * 256 words
    uint32_t *ki = keys_a, *ko = keys_b;
    V *vi = vals_a, *vo = vals_b;
    // 32-bit values: one launch over a few million keys is bound by its scattered 4-byte writes and gains from
    // putting a block's keys in digit order in LDS first (8 x 1 M keys: 0.87 -> 0.68 ms for the four passes); a
    // single million is bound by latency and loses (see radix_scatter_kernel)
    const bool staged32 = (unsigned long long)n * nseq >= 3000000ull;
    for (int bit = begin_bit; bit < end_bit; bit += RADIX_BITS) {
        const int nbits = end_bit - bit < RADIX_BITS ? end_bit - bit : RADIX_BITS;
        const uint32_t mask = (1u << nbits) - 1u;
        hipLaunchKernelGGL(radix_hist_kernel<uint32_t>, dim3(nb, nseq), dim3(SORT_THREADS), 0, st, ki, n, bit, mask, nb, hist, no_segs);
        hipLaunchKernelGGL(radix_rows_kernel, dim3(RADIX_SIZE, nseq), dim3(256), 0, st, hist, nb, digit_total, no_segs);
        // (positions_as_values: the values of the first pass are the positions 0 .. n-1, which nobody has to write first)
        const V *values = positions_as_values && bit == begin_bit ? nullptr : vi;
        if (sizeof(V) > 4 || staged32)
            hipLaunchKernelGGL((radix_scatter_kernel<V, true>), dim3(nb, nseq), dim3(SORT_THREADS), 0, st, ki, values, ko,
                               vo, n, bit, mask, nb, hist, digit_total, no_segs);
        else
            hipLaunchKernelGGL((radix_scatter_kernel<V, false>), dim3(nb, nseq), dim3(SORT_THREADS), 0, st, ki, values, ko,
                               vo, n, bit, mask, nb, hist, digit_total, no_segs);
        uint32_t *t = ki; ki = ko; ko = t;
        V *tv = vi; vi = vo; vo = tv;
        *result_in_first = !*result_in_first;
    }
    return gsr_check_launch("sort_pairs");
}

// Bits of the next digit when the bits [bit, end_bit) are split evenly over as few passes of at most RADIX_BITS as
// they need (13 tile bits: 7 + 6 instead of 8 + 5).  The scatter kernel ranks a digit with one ballot per bit.
static int even_digit_bits(int bit, int end_bit)
{
    const int left = end_bit - bit, passes = (left + RADIX_BITS - 1) / RADIX_BITS;
    return (left + passes - 1) / passes;
}

// 64-bit words sorted (stably) by their bits [begin_bit, end_bit): ping-pong between words_a and words_b as above.
int gsr_sort_words64(uint64_t *words_a, uint64_t *words_b, uint32_t n, int begin_bit, int end_bit, void *workspace,
                     size_t workspace_bytes, void *stream, bool *result_in_first)
{
    hipStream_t st = (hipStream_t)stream;
    *result_in_first = true;
    if (begin_bit < 0 || end_bit > 64 || begin_bit > end_bit) {
        gsr_set_error("gsr_sort_words64: bad bit range [%d,%d)", begin_bit, end_bit);
        return GSR_ERR_ARG;
    }
    if (workspace_bytes < gsr_sort_workspace_bytes(n) || !workspace) {
        gsr_set_error("gsr_sort_words64: workspace too small (%zu < %zu)", workspace_bytes, gsr_sort_workspace_bytes(n));
        return GSR_ERR_ARG;
    }
    if (n == 0) return GSR_OK;
    if (!words_a || !words_b) { gsr_set_error("gsr_sort_words64: null pointer"); return GSR_ERR_ARG; }
    const uint32_t nb = gsr_div_up(n, SORT_TILE);
    uint32_t *hist = (uint32_t *)workspace;
    uint32_t *digit_total = (uint32_t *)((char *)workspace + align256((size_t)nb * RADIX_SIZE * sizeof(uint32_t)));
    uint64_t *wi = words_a, *wo = words_b;
    for (int bit = begin_bit, nbits; bit < end_bit; bit += nbits) {
        nbits = even_digit_bits(bit, end_bit);
        const uint32_t mask = (1u << nbits) - 1u;
        hipLaunchKernelGGL(radix_hist_kernel<uint64_t>, dim3(nb), dim3(SORT_THREADS), 0, st, wi, n, bit, mask, nb, hist, no_segs);
        hipLaunchKernelGGL(radix_rows_kernel, dim3(RADIX_SIZE), dim3(256), 0, st, hist, nb, digit_total, no_segs);
        hipLaunchKernelGGL((radix_scatter_kernel<uint64_t, true, true>), dim3(nb), dim3(SORT_THREADS), 0, st,
                           (const uint32_t *)nullptr, wi, (uint32_t *)nullptr, wo, n, bit, mask, nb, hist, digit_total, no_segs);
        uint64_t *t = wi; wi = wo; wo = t;
        *result_in_first = !*result_in_first;
    }
    return gsr_check_launch("sort_words64");
}

// gsr_sort_words64 for `count` sequences of different lengths with one set of launches: sequence y has n[y] words
// from element base[y] (a multiple of 4096) of both buffers.  The workspace holds gsr_sort_words64_segs_bytes().
size_t gsr_sort_words64_segs_bytes(const uint32_t *n, uint32_t count)
{
    size_t blocks = 0;
    for (uint32_t y = 0; y < count; ++y) blocks += gsr_div_up(n[y], SORT_TILE);
    return align256(blocks * RADIX_SIZE * sizeof(uint32_t)) + align256((size_t)count * RADIX_SIZE * sizeof(uint32_t));
}

int gsr_sort_words64_segs(uint64_t *words_a, uint64_t *words_b, const uint32_t *n, const uint32_t *base, uint32_t count,
                          int begin_bit, int end_bit, void *workspace, size_t workspace_bytes, void *stream,
                          bool *result_in_first)
{
    hipStream_t st = (hipStream_t)stream;
    *result_in_first = true;
    if (count == 0 || count > GSR_MAX_SEGS || begin_bit < 0 || end_bit > 64 || begin_bit > end_bit) {
        gsr_set_error("gsr_sort_words64_segs: bad sequence count or bit range");
        return GSR_ERR_ARG;
    }
    if (!workspace || workspace_bytes < gsr_sort_words64_segs_bytes(n, count) || !words_a || !words_b) {
        gsr_set_error("gsr_sort_words64_segs: null pointer or workspace too small");
        return GSR_ERR_ARG;
    }
    SortSegs segs = {};
    segs.count = count;
    uint32_t blocks = 0, most = 0;
    for (uint32_t y = 0; y < count; ++y) {
        if (base[y] % SORT_TILE) { gsr_set_error("gsr_sort_words64_segs: a sequence does not start at a multiple of %d", SORT_TILE); return GSR_ERR_ARG; }
        segs.n[y] = n[y];
        segs.base[y] = base[y];
        segs.nb[y] = gsr_div_up(n[y], SORT_TILE);
        segs.hbase[y] = blocks * RADIX_SIZE;
        blocks += segs.nb[y];
        most = segs.nb[y] > most ? segs.nb[y] : most;
    }
    if (most == 0) return GSR_OK;
    uint32_t *hist = (uint32_t *)workspace;
    uint32_t *digit_total = (uint32_t *)((char *)workspace + align256((size_t)blocks * RADIX_SIZE * sizeof(uint32_t)));
    uint64_t *wi = words_a, *wo = words_b;
    for (int bit = begin_bit, nbits; bit < end_bit; bit += nbits) {
        nbits = even_digit_bits(bit, end_bit);
        const uint32_t mask = (1u << nbits) - 1u;
        hipLaunchKernelGGL(radix_hist_kernel<uint64_t>, dim3(most, count), dim3(SORT_THREADS), 0, st, wi, 0u, bit, mask, 0u, hist, segs);
        hipLaunchKernelGGL(radix_rows_kernel, dim3(RADIX_SIZE, count), dim3(256), 0, st, hist, 0u, digit_total, segs);
        hipLaunchKernelGGL((radix_scatter_kernel<uint64_t, true, true>), dim3(most, count), dim3(SORT_THREADS), 0, st,
                           (const uint32_t *)nullptr, wi, (uint32_t *)nullptr, wo, 0u, bit, mask, 0u, hist, digit_total, segs);
        uint64_t *t = wi; wi = wo; wo = t;
        *result_in_first = !*result_in_first;
    }
    return gsr_check_launch("sort_words64_segs");
}

// library-internal entry points (binning.hip): sort without the final copy
int gsr_sort_pingpong32(uint32_t *keys_a, uint32_t *vals_a, uint32_t *keys_b, uint32_t *vals_b, uint32_t n, int begin_bit,
                        int end_bit, void *workspace, size_t workspace_bytes, void *stream, bool *result_in_first,
                        bool positions_as_values)
{
    return sort_pingpong<uint32_t>(keys_a, vals_a, keys_b, vals_b, n, begin_bit, end_bit, workspace, workspace_bytes,
                                   stream, result_in_first, 1, positions_as_values);
}
int gsr_sort_pingpong32_batch(uint32_t *keys_a, uint32_t *vals_a, uint32_t *keys_b, uint32_t *vals_b, uint32_t n, uint32_t nseq,
                              int begin_bit, int end_bit, void *workspace, size_t workspace_bytes, void *stream,
                              bool *result_in_first, bool positions_as_values)
{
    return sort_pingpong<uint32_t>(keys_a, vals_a, keys_b, vals_b, n, begin_bit, end_bit, workspace, workspace_bytes,
                                   stream, result_in_first, nseq, positions_as_values);
}
int gsr_sort_pingpong64(uint32_t *keys_a, uint64_t *vals_a, uint32_t *keys_b, uint64_t *vals_b, uint32_t n, int begin_bit,
                        int end_bit, void *workspace, size_t workspace_bytes, void *stream, bool *result_in_first)
{
    return sort_pingpong<uint64_t>(keys_a, vals_a, keys_b, vals_b, n, begin_bit, end_bit, workspace, workspace_bytes,
                                   stream, result_in_first);
}

template <typename V>
static int sort_pairs_impl(uint32_t *keys_in, V *vals_in, uint32_t *keys_out, V *vals_out, uint32_t n, int begin_bit,
                           int end_bit, void *workspace, size_t workspace_bytes, void *stream)
{
    bool in_first;
    if (int e = sort_pingpong<V>(keys_in, vals_in, keys_out, vals_out, n, begin_bit, end_bit, workspace, workspace_bytes,
                                 stream, &in_first))
        return e;
    if (n && in_first) {  // even number of passes (or none): the result sits in the *_in buffers
        (void)hipMemcpyAsync(keys_out, keys_in, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream);
        (void)hipMemcpyAsync(vals_out, vals_in, (size_t)n * sizeof(V), hipMemcpyDeviceToDevice, (hipStream_t)stream);
    }
    return gsr_check_launch("sort_pairs");
}

extern "C" int gsr_sort_pairs(uint32_t *keys_in, uint32_t *vals_in, uint32_t *keys_out, uint32_t *vals_out,
                              uint32_t n, int begin_bit, int end_bit, void *workspace, size_t workspace_bytes,
                              void *stream)
{
    return sort_pairs_impl<uint32_t>(keys_in, vals_in, keys_out, vals_out, n, begin_bit, end_bit, workspace,
                                     workspace_bytes, stream);
}

extern "C" int gsr_sort_pairs64(uint32_t *keys_in, uint64_t *vals_in, uint32_t *keys_out, uint64_t *vals_out,
                                uint32_t n, int begin_bit, int end_bit, void *workspace, size_t workspace_bytes,
                                void *stream)
{
    return sort_pairs_impl<uint64_t>(keys_in, vals_in, keys_out, vals_out, n, begin_bit, end_bit, workspace,
                                     workspace_bytes, stream);
}
