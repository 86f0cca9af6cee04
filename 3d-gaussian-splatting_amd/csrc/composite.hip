// Stage 3: per-tile front-to-back blend and its backward.
//
// Work decomposition (CDNA4): one 64-lane wave owns one 16x16 pixel block, 4 pixels per lane
// (lane -> column x = lane/4, rows y0, y0+4, y0+8, y0+12 with y0 = lane%4).  The lane's first two
// pixels lie in the top 8 rows of the block and the other two in the bottom 8 rows; each pair is
// processed with packed fp32 instructions, and a pair is the unit of the clamped-low fast path
// (see the hint flags of a record below).  A reference tile of edge L is covered
// by ceil(L/16)^2 such waves, all streaming the tile's list.  The list is a contiguous stream of
// 64-byte splat records in depth order; the record address is wave-uniform, so a record is fetched
// with ONE scalar load into 16 SGPRs and broadcast to the vector ALU for free - no LDS staging,
// no barriers, waves are fully independent.  The loops are bound by the ISSUE of vector instructions
// (SQ counters: the vector ALU is busy 80-90 % of a launch at 4.3-4.4 cycles per instruction, the scalar
// unit a quarter of it), so what counts is the number of vector instructions per list entry: packed
// pairs, wave-uniform values kept in scalar registers, no register copies at the loop's back edge
// (see the loop structure below), no reduction where a running sum does (bwd_sweep).
//
// Semantics follow the reference exactly (GaussianRenderer.py:100-112, Utils/Probability.py):
// integer pixel coordinates, alpha = clamp(opacity*G, 0.01, 0.99) for EVERY list entry, the
// whole list is blended (no early exit), no background term.
#include "gsr_common.hpp"

namespace {

constexpr int WAVES_PER_BLOCK = 4;
constexpr float kAlphaMin = 0.01f;
constexpr float kAlphaMax = 0.99f;
// The backward recovers the transmittance in front of an entry by dividing the one behind it, which
// works for as long as the transmittance is a normal fp32 number with room below it for the exact
// remainder of the correctly rounded quotient (2^-24 of it): kTinyT.  A wave whose smallest final
// transmittance lies between the smallest normal number and kTinyT scales its transmittances up by
// kRescue and its cotangents down by it (a power of two, exact) and still takes the plain sweep.  Only a wave with a pixel whose
// transmittance went denormal or to zero takes the checked sweep (see composite_bwd_kernel), which
// re-runs the forward recurrence, costs about two plain sweeps, and drops, per pixel, the entries
// behind the point where the transmittance falls below kTinyT: their weights are below 1e-26 of the
// pixel's cotangent.
constexpr float kTinyT = 1e-26f;
constexpr float kMinNormalT = 1.17549435e-38f;   // FLT_MIN
constexpr float kRescue = 1099511627776.0f;      // 2^40: FLT_MIN * 2^40 > kTinyT
// records between the L2 touch and the scalar fetch in the forward and in the backward (the forward alone is 3 %
// faster with 8 than with 16 and slower again below 6; the backward runs the same from 2 to 16; beside the list kernels
// of the next view on a second stream 24 / 48 and 16 / 96 change nothing either)
constexpr uint32_t kTouchAhead = 8, kTouchAheadBwd = 16;

struct WaveGeom {
    int tile, x, y0;     // pixel column and first row of this lane; its rows are y0 + 4 j, j = 0..3
    bool vx;             // column inside the tile and the image
    int ylim;            // rows < ylim are inside
    uint32_t begin, end; // list range (wave-uniform), RELATIVE to the list's first record: begin = 0, end = its length
    uint32_t total;      // records from the list's first one to the end of the stream (capped at 2^26 - 1)
    uint32_t first;      // position of the list's first record in the stream
    uint32_t flag_shift; // position of this block's two hint bits in a record's flags
    uint32_t turn;       // rank of the tile in the order the grid takes the tiles
};

// tile_order (may be null): the tiles in the order in which the waves of the grid take them - longest
// lists first, so that the waves still running when the grid drains are short ones.
// List positions are RELATIVE to the list's first record (w.first): the kernels move their stream pointer there
// (list_start below), so their 32-bit byte offsets only have to span one list and the touch-ahead behind it - a stream
// may be longer than 4 GiB.
__device__ inline bool wave_setup(const ViewK &v, const uint32_t *__restrict__ tile_offsets,
                                  const uint32_t *__restrict__ tile_order, WaveGeom &w)
{
    const int S = (v.L + 15) >> 4;
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned wv = blockIdx.x * (blockDim.x >> 6) + wave_in_block;
    const unsigned per_tile = (unsigned)(S * S);
    const unsigned turn = wv / per_tile;
    if (turn >= (unsigned)v.T) return false;
    const unsigned sub = wv - turn * per_tile;
    const unsigned tile = tile_order ? tile_order[turn] : turn;
    const int sxi = (int)(sub / (unsigned)S), syi = (int)(sub - (unsigned)sxi * S);
    const int xi = (int)(tile / (unsigned)v.nbh), yi = (int)(tile - (unsigned)xi * v.nbh);
    const int lane = threadIdx.x & 63;
    w.tile = (int)tile;
    w.turn = turn;
    w.flag_shift = 2u * (sub & 15u);
    w.x = xi * v.L + sxi * 16 + (lane >> 2);
    w.y0 = yi * v.L + syi * 16 + (lane & 3);
    const int xlim = min((xi + 1) * v.L, v.W);
    w.ylim = min((yi + 1) * v.L, v.H);
    w.vx = w.x < xlim;
    // (wave-uniform by construction; said explicitly, because the kernels use them as scalar operands of asm statements)
    const uint32_t begin = (uint32_t)__builtin_amdgcn_readfirstlane((int)tile_offsets[tile]);
    const uint32_t end = (uint32_t)__builtin_amdgcn_readfirstlane((int)tile_offsets[tile + 1]);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readfirstlane((int)tile_offsets[v.T]);
    w.first = begin;
    w.begin = 0u;
    w.end = end - begin;
    w.total = min(total - begin, (1u << 26) - 1u);
    return true;
}

// The compositing waves are bound by instruction issue.  When memory-bound kernels of another view
// run beside them on a second stream (minibatch.render_views), the SIMD's issue arbiter should serve
// these waves first and give the others the cycles they leave.
__device__ inline void raise_wave_priority(const WaveGeom &w, int T)
{
    // and among themselves: the longer the list, the higher the priority (lists are taken longest first, `turn`
    // is the list's rank).  The waves with the longest lists are the critical path of a launch; served first,
    // they finish early and the SIMD's issue slots go to the shorter ones instead of running out at the end.
    const unsigned quarter = (unsigned)w.turn * 4u / (unsigned)max(T, 1);
    if (quarter == 0u) __builtin_amdgcn_s_setprio(3);
    else if (quarter == 1u) __builtin_amdgcn_s_setprio(2);
    else if (quarter == 2u) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

// One 64-byte stream record (layout: GSR_STREAM_DWORDS in gsr.h) held in 16 SGPRs.  Values that
// serve as the scalar operand of packed instructions sit at even positions: an aligned SGPR pair
// whose low half is broadcast (op_sel_hi = 0) needs no copy.
struct Rec {
    f32x16 d;
};
// (two layouts, see GSR_STREAM_DWORDS in gsr.h: `ONE` = the tile is a single 16x16 block)
#define R_gy(R) ((R).d[0])
#define R_gx(R) ((R).d[1])
#define R_D(R) ((R).d[2])
#define R_A(R) ((R).d[3])
#define R_o(R) ((R).d[4])
#define R_B(R) ((R).d[5])
#define R_r(R) ((R).d[6])
#define R_ca(R) ((R).d[7])
#define R_g(R) ((R).d[8])
#define R_b(R) ((R).d[10])
#define R_cbh(R) ((R).d[12])
#define R_cd(R) ((R).d[14])
// one-block tiles: 0.01 x colour, for the halves on which alpha sits at its lower clamp
#define R_b1(R) ((R).d[11])
#define R_r1(R) ((R).d[13])
#define R_g1(R) ((R).d[15])
// run records of per-block streams (blocklists.hip): 0.99^n and its reciprocal
#define R_P(R) ((R).d[4])
#define R_rcpP(R) ((R).d[5])
constexpr uint32_t kSlotBits = 30;   // one-block tiles: dword 9 = emission slot (below 2^26) | the two hint bits << 30
template <bool ONE>
__device__ inline uint32_t rec_slot(const f32x16 &d)
{
    return ONE ? __float_as_uint(d[9]) & ((1u << kSlotBits) - 1u) : __float_as_uint(d[9]);
}
template <bool ONE>
__device__ inline uint32_t rec_flags(const f32x16 &d)
{
    return ONE ? __float_as_uint(d[9]) >> kSlotBits : __float_as_uint(d[11]);
}

// Scalar load of the record at byte offset `off` of the stream, issued now and waited for later
// (rec_wait), so that the latency of one record's fetch (a few hundred cycles from L2) overlaps the
// arithmetic of the previous one.  hipcc would otherwise sink the load below its last use to reuse
// the registers.  The waits are written by hand because the compiler does not count loads issued
// from inline asm.
__device__ inline void rec_request(const float *__restrict__ stream_rec, uint32_t off, Rec &R)
{
    asm volatile("s_load_dwordx16 %0, %1, %2" : "=&s"(R.d) : "s"(stream_rec), "s"(off) : "memory");
}
__device__ inline void rec_wait(Rec &R) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(R.d)::"memory"); }
__device__ inline void rec_wait(Rec &R, Rec &Q) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(R.d), "+s"(Q.d)::"memory"); }

// A wave-uniform address as a pair of scalar registers.  Where the compiler does not see that a stream offset
// is uniform it computes the address on the vector ALU and this reads it back with v_readfirstlane; where it
// does, this costs nothing.
__device__ inline const char *uniform_address(const char *p)
{
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    return reinterpret_cast<const char *>(((unsigned long long)hi << 32) | lo);
}
// the stream from the first record of the wave's list on, as a pair of scalar registers
__device__ inline const float *list_start(const float *stream_rec, const WaveGeom &w)
{
    return reinterpret_cast<const float *>(uniform_address(reinterpret_cast<const char *>(stream_rec + (size_t)w.first * GSR_STREAM_DWORDS)));
}
// A vector memory instruction must not read a scalar register within five wait states of a VECTOR instruction
// (v_readfirstlane) writing it.  The compiler inserts those wait states for its own instructions but cannot see
// into an asm statement, so every asm below whose memory instruction takes such an address starts with them.
#define kSgprReady "s_nop 4\n\t"

// max(a, b) - b on the scalar unit.  Written as an expression the compiler makes it a saturating subtraction, which only
// the vector ALU has, and brings the result back with v_readfirstlane: four vector instructions per touch of the
// backward, 3 % of its loop.
__device__ inline uint32_t scalar_sub_floor0(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("s_max_u32 %0, %1, %2\n\ts_sub_u32 %0, %0, %2" : "=&s"(r) : "s"(a), "s"(b) : "scc");
    return r;
}

// Pulls the cache lines of records some entries further on into L2 with a vector load whose result
// is never used (`lane_off` is 0 for half of the lanes and 64 for the others: a call touches two
// neighbouring records, and one call per two entries touches every line of the stream; `off` is
// wave-uniform and added to the base on the scalar unit).  The scalar record loads then
// hit L2 instead of waiting for HBM; a wave has only one or two records in flight, so HBM latency would
// otherwise bound the stream rate.  `sink` must stay live until the wave drains its vector loads
// (l2_touch_drain), and the drain must come before any code that follows the loop: the compiler counts the
// register as written when the load is ISSUED, so once the loop is left it may move the value elsewhere and
// reuse the register while a load is still on its way (tools/audit_touch.py checks the generated code).
__device__ inline void l2_touch(const float *stream_rec, uint32_t off, uint32_t lane_off, float &sink)
{
    const char *base = uniform_address(reinterpret_cast<const char *>(stream_rec) + off);
    asm volatile(kSgprReady "global_load_dword %0, %1, %2" : "+v"(sink) : "v"(lane_off), "s"(base) : "memory");
}
__device__ inline void l2_touch_drain(float &sink) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink)::"memory"); }

// Hint flags of a record: the halves of a 16x16 block on which alpha is pinned at 0.01 for every
// pixel (bit 0: top 8 rows, bit 1: bottom 8 rows).  A tile of up to 4x4 blocks carries two bits per
// block, block c at bits 2c and 2c+1 (MULTI: the wave shifts its own pair down; a tile that is one block
// has them at bits 0 and 1).  Wave-uniform: scalar compares and branches.
constexpr uint32_t kLowTop = 1u, kLowBottom = 2u, kLowBoth = 3u;
template <bool MULTI>
__device__ inline uint32_t block_flags(uint32_t record_flags, uint32_t shift)
{
    return MULTI ? (record_flags >> shift) & 3u : record_flags;
}

__device__ inline float clamp_alpha(float raw) { return fminf(fmaxf(raw, kAlphaMin), kAlphaMax); }

// Two pixels per operation.  On gfx950 a v_fma_f32 costs about 4.4 cycles of a SIMD and a
// v_pk_fma_f32 about 5.7 for two lanes' worth (measured, tools/ubench/valu_rate.hip), so the
// per-pixel fused multiply-adds are written on pairs; a lane's 4 pixels are 2 pairs.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ inline f2 splat(float s) { return f2{s, s}; }
// x + y of a pair with ONE plain add: left to itself the compiler writes v_pk_add_f32 with swapped halves, a packed
// instruction (two lanes' worth of issue time) for one useful sum
__device__ inline float hsum(f2 s)
{
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(s.x), "v"(s.y));
    return r;
}
__device__ inline f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ inline f2 exp2_2(f2 q) { return f2{__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)}; }
__device__ inline f2 clamp_alpha2(f2 raw) { return f2{clamp_alpha(raw.x), clamp_alpha(raw.y)}; }

// RUNS: the lists are the per-block streams of blocklists.hip - one-block layout, and a record with both hint bits set
// stands for a run of entries clamped on the whole block: transmittance times 0.99^n (dword 4), colour plus
// transmittance times the run's colour (dwords 11, 13, 15)
template <bool MULTI, bool RUNS = false>
__global__ __launch_bounds__(256) void composite_fwd_kernel(ViewK v, const uint32_t *__restrict__ tile_offsets,
                                                            const uint32_t *__restrict__ tile_order,
                                                            const float *__restrict__ stream_rec,
                                                            float *__restrict__ img, float *__restrict__ final_T)
{
    WaveGeom w;
    if (!wave_setup(v, tile_offsets, tile_order, w)) return;
    stream_rec = list_start(stream_rec, w);
    raise_wave_priority(w, v.T);
    const float pxf = (float)w.x;
    f2 pyf[2], T[2], cr[2], cg[2], cb[2];   // pixel rows (y0, y0+4): top half, (y0+8, y0+12): bottom half
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        pyf[h] = f2{(float)(w.y0 + 8 * h), (float)(w.y0 + 8 * h + 4)};
        T[h] = splat(1.0f);
        cr[h] = cg[h] = cb[h] = splat(0.0f);
    }

    // alpha is 0.01 on this half of the block: no exponent, no exp
    // (the record of a one-block tile carries 0.01 x colour for this path: the weight 0.01 T is never formed)
    auto half_low = [&](int h, const Rec &R, float red, float keep) {
        if (!MULTI) {
            cr[h] = pk_fma(T[h], splat(R_r1(R)), cr[h]);
            cg[h] = pk_fma(T[h], splat(R_g1(R)), cg[h]);
            cb[h] = pk_fma(T[h], splat(R_b1(R)), cb[h]);
        } else {
            const f2 wgt = splat(kAlphaMin) * T[h];
            cr[h] = pk_fma(wgt, splat(red), cr[h]);
            cg[h] = pk_fma(wgt, splat(R_g(R)), cg[h]);
            cb[h] = pk_fma(wgt, splat(R_b(R)), cb[h]);
        }
        T[h] = T[h] * splat(keep);
        asm("" : "+v"(T[h]));   // keeps this multiply here: merged with half_full's it needs its factor in registers
    };
    auto half_full = [&](int h, const Rec &R, float red, float o, float bdx, float adx2) {
        const f2 dy = splat(R_gy(R)) - pyf[h];
        const f2 q = pk_fma(pk_fma(splat(R_D(R)), dy, splat(bdx)), dy, splat(adx2));
        const f2 a = clamp_alpha2(splat(o) * exp2_2(q));
        const f2 wgt = a * T[h];
        cr[h] = pk_fma(wgt, splat(red), cr[h]);
        cg[h] = pk_fma(wgt, splat(R_g(R)), cg[h]);
        cb[h] = pk_fma(wgt, splat(R_b(R)), cb[h]);
        T[h] = T[h] * (splat(1.0f) - a);   // same rounding as the reference's cumprod of (1 - alpha)
    };
    auto blend = [&](const Rec &R) {
        const float red = R_r(R);
        const uint32_t flags = block_flags<MULTI>(rec_flags<!MULTI>(R.d), w.flag_shift);
        constexpr float keep1 = 1.0f - kAlphaMin;
        if (flags == kLowBoth) {
            const float keep = RUNS ? R_P(R) : keep1;
            half_low(0, R, red, keep);
            half_low(1, R, red, keep);
            return;
        }
        const float o = R_o(R);
        const float dx = R_gx(R) - pxf;
        const float adx2 = R_A(R) * dx * dx;
        const float bdx = R_B(R) * dx;
        if (flags & kLowTop) half_low(0, R, red, keep1); else half_full(0, R, red, o, bdx, adx2);
        if (flags & kLowBottom) half_low(1, R, red, keep1); else half_full(1, R, red, o, bdx, adx2);
    };
    // even lanes touch the record named, odd lanes the one before it; the named record is never the stream's
    // first one unless the stream has no other (touch_floor)
    const uint32_t touch_back = w.total > 1u ? 64u : 0u;
    const uint32_t touch_off = (threadIdx.x & 1) ? 0u : touch_back;
    float sink = 0.0f;
    if (w.begin < w.end) {
        // Record buffers kept by hand: while some are blended the others are in flight.  All stream
        // positions are 32-bit byte offsets (64 bytes per record).  Several entries per turn and ONE exit, the
        // last entries after the loop: with an exit between two entries as well, the compiler copied all
        // sixteen accumulator registers twice per turn (a quarter of the loop's vector instructions).
        const uint32_t off_last = (w.end - 1) * 64u;
        const uint32_t off_stream_last = (w.total - 1) * 64u;   // the touches may run into the next lists
        const uint32_t touch_floor = min(64u, off_stream_last);
        uint32_t off = w.begin * 64u;
        const uint32_t n = w.end - w.begin;
        // Four record buffers, two fetches in flight while two records are blended: a wave waits for ALL its scalar
        // loads at once (they may return out of order), so with one fetch per wait a run of clamped entries - ten
        // vector instructions each - leaves eight waves per SIMD short of work for the 500 cycles a fetch from L2
        // takes; two fetches per wait halve the waits.  (The backward has no scalar registers left for this.)
        Rec A, B, C, D;
        rec_request(stream_rec, off, A);
        rec_request(stream_rec, min(off + 64u, off_last), B);
        for (uint32_t i = n >> 2; i != 0; --i) {
            rec_wait(A, B);
            rec_request(stream_rec, min(off + 128u, off_last), C);
            rec_request(stream_rec, min(off + 192u, off_last), D);
            l2_touch(stream_rec, max(min(off + kTouchAhead * 64u, off_stream_last), touch_floor) - touch_back, touch_off, sink);
            blend(A);
            blend(B);
            rec_wait(C, D);
            rec_request(stream_rec, min(off + 256u, off_last), A);   // (the last record again past the end of the list)
            rec_request(stream_rec, min(off + 320u, off_last), B);
            l2_touch(stream_rec, max(min(off + (kTouchAhead + 2u) * 64u, off_stream_last), touch_floor) - touch_back, touch_off, sink);
            blend(C);
            blend(D);
            off += 256u;
        }
        rec_wait(A, B);
        l2_touch_drain(sink);   // before any other code: once the loop is left the compiler may reuse the sink's register
        const uint32_t rest = n & 3u;
        if (rest >= 1u) blend(A);
        if (rest >= 2u) blend(B);
        if (rest == 3u) {
            rec_request(stream_rec, off + 128u, C);
            rec_wait(C);
            blend(C);
        }
    }
    if (!w.vx) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int y = w.y0 + 4 * j;
        if (y < w.ylim) {
            const size_t pix = (size_t)w.x * v.H + y;
            img[pix * 3 + 0] = cr[j >> 1][j & 1]; img[pix * 3 + 1] = cg[j >> 1][j & 1]; img[pix * 3 + 2] = cb[j >> 1][j & 1];
            final_T[pix] = T[j >> 1][j & 1];
        }
    }
}

// ---- wave reductions of the gradient sums ------------------------------------------------------
// Nine sums per entry have to be taken over the 64 lanes.  A plain butterfly costs six
// cross-lane adds per value.  Here values are MERGED while they are reduced: one
// v_permlane32_swap (gfx950) exchanges the upper half of one register with the lower half of
// another, so a single add afterwards reduces two values by one level and leaves them in one
// register (one value per half-wave); v_permlane16_swap does the same one level down (one value
// per 16-lane row), a bank-masked DPP move once more (one value per 8 lanes).  Eight values end up
// in ONE register and share the last three butterfly steps.

template <int CTRL>
__device__ inline float dpp_add(float val)
{
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(val), CTRL, 0xF, 0xF, false);
    return val + __int_as_float(moved);
}
constexpr int DPP_QUAD_1032 = 0xB1, DPP_QUAD_2301 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140, DPP_ROR8 = 0x128;

// lanes 0..31: a[l] + a[l+32];  lanes 32..63: b[l-32] + b[l]
__device__ inline float merge_halves(float a, float b)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// rows (16 lanes) 0..3: a.row0 + a.row1, b.row0 + b.row1, a.row2 + a.row3, b.row2 + b.row3
__device__ inline float merge_rows(float a, float b)
{
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// in every row: lanes 0..7: a[l] + a[l+8];  lanes 8..15: b[l-8] + b[l]
__device__ inline float merge_half_rows(float a, float b)
{
    const float sa = dpp_add<DPP_ROR8>(a), sb = dpp_add<DPP_ROR8>(b);
    // identity move of sb into banks 2 and 3 (lanes 8..15 of each row), the rest keeps sa
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(sa), __float_as_int(sb), 0xE4, 0xF, 0xC, false));
}
// every lane ends with the sum over its group of 8 (resp. its row of 16)
__device__ inline float sum8(float x) { return dpp_add<DPP_HALF_MIRROR>(dpp_add<DPP_QUAD_2301>(dpp_add<DPP_QUAD_1032>(x))); }
__device__ inline float sum16(float x) { return dpp_add<DPP_MIRROR>(sum8(x)); }

// Which lane writes which float of a gradient record, fixed per lane.
// Full records: group g of 8 lanes (g = lane / 8) holds record float g, row r (r = lane / 16) holds
// the row's part of the blue sum, record float 8 + r.
// Colour-only records: after merge_halves(red, green) and merge_rows(.., blue) the rows hold red,
// blue (rows 0-1), green, blue (rows 2-3): floats 6, 8, 7, 9; the other eight floats are zeros.
// The twelve writers of a record store with ONE instruction under a constant exec mask.
constexpr unsigned long long kFullWriters = 0x0103010301030103ull;   // lanes 8 g and lanes 16 r + 1
constexpr unsigned long long kColWriters = (0x7Full | (3ull << 10)) | (1ull << 24) | (1ull << 39) | (1ull << 57);
struct RecordLanes {
    uint32_t full_off, col_off;   // byte offset of the lane's float inside a record
    bool full_main, col_zero;     // the lane holds one of the 8 merged sums / writes a zero
    float colour;                 // 1 on the six lanes that write a colour float of a full record (floats 6..11), else 0
};
__device__ inline RecordLanes record_lanes(int lane)
{
    RecordLanes L;
    L.full_main = (lane & 7) == 0;
    L.full_off = 4u * (L.full_main ? (uint32_t)(lane >> 3) : 8u + (uint32_t)(lane >> 4));
    L.col_zero = lane < 6 || lane == 10 || lane == 11;
    L.col_off = 4u * (uint32_t)(lane & 15);
    L.colour = (((kFullWriters >> lane) & 1ull) && L.full_off >= 24u) ? 1.0f : 0.0f;
    return L;
}

// How the gradient record of a (tile, Gaussian) pair reaches pairgrad.
//   kOneBlock:  the tile is one 16x16 block, its wave stores the record (plain stores);
//   kTileGroup: the tile has 4, 9 or 16 blocks; they are composited by workgroups of 4 (3 for 9 blocks) waves,
//               one wave per block.  Every wave leaves its part of a record in LDS, the workgroup adds the parts
//               in block order, a chunk of entries at a time, and stores the sum in the workgroup's own row of
//               pairgrad: a pair owns gsr_pairgrad_parts() consecutive rows, one per workgroup of its tile, and
//               the per-Gaussian sums add them in row order.  No atomics: the result does not depend on timing.
//               (Small workgroups, several per compute unit: while one waits at its barrier the others run.)
//   kAtomic:    more than 16 blocks per tile: independent waves add into a zeroed pairgrad with float atomics.
//   kBlockRuns: the per-block streams of blocklists.hip: as kOneBlock, with one row per STREAM record (the record carries its
//               own index): twelve floats for an evaluated entry, the run's six colour sums for a run record.
enum StoreMode { kOneBlock = 0, kAtomic = 1, kTileGroup = 2, kBlockRuns = 3 };
constexpr bool one_block_layout(int mode) { return mode == kOneBlock || mode == kBlockRuns; }
// the lanes of kFullWriters that hold a colour float of a full record (floats 6..11: lanes 48, 56 and 1, 17, 33, 49)
constexpr unsigned long long kRunWriters = (1ull << 48) | (1ull << 56) | (1ull << 1) | (1ull << 17) | (1ull << 33) | (1ull << 49);

constexpr uint32_t kGroupChunk = 32;   // entries between two reductions of a tile group (two chunk buffers; 16 to 64 time alike)
constexpr uint32_t kGroupMaxWaves = 4;

__host__ __device__ inline int group_waves(int S) { return S == 3 ? 3 : 4; }   // blocks per workgroup, S = blocks per tile edge

// Where a wave of a tile group leaves its parts (unused by the other modes).
typedef __attribute__((address_space(3))) float lds_float;      // (LDS pointers: 32-bit address arithmetic)
typedef __attribute__((address_space(3))) uint32_t lds_u32;
struct GroupSlot {
    lds_float *parts;    // LDS [2][kGroupChunk][kGroupMaxWaves][GSR_PAIRGRAD_FLOATS]: the chunk being filled and the one before
    lds_u32 *slots;      // LDS [2][kGroupChunk]: emission slot of each entry
    uint32_t block, blocks, entry;   // this wave's block inside the group, blocks of the group, entry of the chunk being filled
    uint32_t buf;        // chunk buffer being filled (0 / 1)
    uint32_t part, nparts;           // the group's row among the rows of a pair, rows per pair
};

// dst = pairgrad + byte offset `rec_off` of the record; the writers are the lanes of `writers`.
template <int MODE>
__device__ inline void put_record(float *pairgrad, uint32_t rec_off, uint32_t lane_off, float val,
                                  unsigned long long writers, const GroupSlot &gs)
{
    if (MODE == kAtomic) {
        const uint32_t voff = lane_off + rec_off;
        // several waves share the record; it was zeroed by the launcher
        if (((writers >> (threadIdx.x & 63)) & 1ull) && val != 0.0f)
            atomicAdd(reinterpret_cast<float *>(reinterpret_cast<char *>(pairgrad) + voff), val);
    } else if (MODE == kTileGroup) {
        const uint32_t e = gs.buf * kGroupChunk + gs.entry;
        // (the entry's part of the index is wave-uniform and the lane's part never changes: a scalar multiply and one add)
        const uint32_t lane_idx = gs.block * GSR_PAIRGRAD_FLOATS + (lane_off >> 2);
        const uint32_t entry_idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)(e * (kGroupMaxWaves * GSR_PAIRGRAD_FLOATS)));
        if ((writers >> (threadIdx.x & 63)) & 1ull)
            gs.parts[entry_idx + lane_idx] = val;
        if (gs.block == 0 && (threadIdx.x & 63) == 0) gs.slots[e] = rec_off / (uint32_t)(GSR_PAIRGRAD_FLOATS * sizeof(float));
    } else {
        // every lane of the wave is active here: set the writers, store, restore.  The record's address is
        // wave-uniform and added on the scalar unit; the lane's offset inside a record never changes.
        const char *record = uniform_address(reinterpret_cast<const char *>(pairgrad) + rec_off);
        asm volatile(kSgprReady "s_mov_b64 exec, %3\n\t"
                     "global_store_dword %0, %1, %2\n\t"
                     "s_mov_b64 exec, -1"
                     :
                     : "v"(lane_off), "v"(val), "s"(record), "s"(writers)
                     : "memory");
    }
}

// Adds up the parts of the `count` entries a tile group has filled in and stores the sums (all threads of
// the workgroup), then turns to the other chunk buffer.  One barrier per chunk: a wave writes into this buffer
// again only after the NEXT chunk's barrier, which every wave passes after its share of these reads.
__device__ inline void group_flush(GroupSlot &gs, uint32_t count, float *pairgrad)
{
    __syncthreads();
    const uint32_t first = gs.buf * kGroupChunk;
    for (uint32_t i = threadIdx.x; i < count * GSR_PAIRGRAD_FLOATS; i += blockDim.x) {
        const uint32_t e = i / GSR_PAIRGRAD_FLOATS, f = i - e * GSR_PAIRGRAD_FLOATS;
        const lds_float *part = gs.parts + (first + e) * kGroupMaxWaves * GSR_PAIRGRAD_FLOATS + f;
        float sum = part[0];
        for (uint32_t b = 1; b < gs.blocks; ++b) sum += part[b * GSR_PAIRGRAD_FLOATS];   // block order: reproducible
        pairgrad[((size_t)gs.slots[first + e] * gs.nparts + gs.part) * GSR_PAIRGRAD_FLOATS + f] = sum;
    }
    gs.buf ^= 1u;
    gs.entry = 0;
}

// a * b with 0 * x = 0 for every x, NaN and infinities included (v_mul_legacy_f32)
__device__ inline float mul_zero_wins(float a, float b)
{
    float r;
    asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Full record, floats: sum h v_x, sum h v_y, sum h v_x^2, sum h v_x v_y, sum h v_y^2, dL/d(opacity),
// dL/d(red), dL/d(green), then four partial sums of dL/d(blue).
// Returns, on every writer lane of a full record, the float that lane stores (before any rescaling).
__device__ inline float reduce_full(const RecordLanes &L, float vx, float vy, float xx, float xy, float yy, float o, float r, float g, float b)
{
    // pairing order chosen so that group g of 8 lanes ends up with record float g
    const float m0 = merge_halves(vx, yy), m1 = merge_halves(xx, r), m2 = merge_halves(vy, o), m3 = merge_halves(xy, g);
    const float n0 = merge_rows(m0, m1);   // rows: vx, xx, yy, r
    const float n1 = merge_rows(m2, m3);   // rows: vy, xy, o,  g
    const float all8 = sum8(merge_half_rows(n0, n1));
    const float blue = sum16(b);
    return L.full_main ? all8 : blue;
}

// Colour-only record (the clamp blocks every other gradient of the entry).
template <int MODE>
__device__ inline void reduce_store_colour(const RecordLanes &L, float *pairgrad, uint32_t rec_off, const GroupSlot &gs, float r,
                                           float g, float b)
{
    const float rows = sum16(merge_rows(merge_halves(r, g), b));   // rows: red, blue(0-1), green, blue(2-3)
    put_record<MODE>(pairgrad, rec_off, L.col_off, L.col_zero ? 0.0f : rows, kColWriters, gs);
}

// Reverse sweep of one list for the 4 pixels of a lane.  CHECK: pixel j only takes part in
// entries with index (k - begin) < nstop[j].
template <bool CHECK, int MODE>
__device__ inline void bwd_sweep(const WaveGeom &w, const float *__restrict__ stream_rec, const float pxf,
                                 const f2 (&pyf)[2], f2 (&T)[2], const f2 (&gr)[2], const f2 (&gg)[2],
                                 const f2 (&gb)[2], const uint32_t (&nstop)[4],
                                 float *__restrict__ pairgrad, GroupSlot gs)
{
    f2 suffix[2] = {splat(0.f), splat(0.f)};
    const f2 py_top = pyf[0];
    const RecordLanes lanes = record_lanes(threadIdx.x & 63);
    // Plain sweep: the colour gradient of an entry that is clamped low on the whole block needs no sums.  With
    // U_c(k) = sum over the block's pixels of T_k g_c (T_k: transmittance in front of entry k, g_c: the pixel's
    // cotangent in channel c), every entry has  U_c(k) = U_c(k+1) + dL/dc_k  (T_k = T_{k+1} + alpha_k T_k, and
    // dL/dc_k = sum alpha_k T_k g_c), and an entry with alpha = 0.01 everywhere has  dL/dc_k = 0.01 U_c(k) =
    // (0.01 / 0.99) U_c(k+1).  U is kept on the six lanes that store the colour floats of a record (in the
    // record's layout: red, green, four partial sums of blue over the rows of lanes) and follows from the sums
    // every other entry reduces anyway.
    const float low_gain = lanes.colour * (kAlphaMin / (1.0f - kAlphaMin));
    float U = 0.0f;
    if (!CHECK) {
        const f2 ur = pk_fma(T[1], gr[1], T[0] * gr[0]), ug = pk_fma(T[1], gg[1], T[0] * gg[0]), ub = pk_fma(T[1], gb[1], T[0] * gb[0]);
        U = reduce_full(lanes, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, hsum(ur), hsum(ug), hsum(ub)) * lanes.colour;
    }
    // Sums of one entry over the lane's pixels (pairs are added up at the end).
    f2 s_vx, s_vy, s_xx, s_xy, s_yy, s_o, s_r, s_g, s_b;

    // One half of the block with alpha = 0.01 on every pixel: only the transmittance recurrence, the
    // colour sums and the suffix remain; the clamp blocks every geometric gradient.
    // (`first`: the half is the first one of its entry to add to the colour sums - it sets them)
    // (`run`: the record is a run of clamped entries - its 0.99^n and reciprocal instead of 0.99, its colour instead of
    // 0.01 x colour, and `wgt` is the transmittance in front of the run)
    auto half_low = [&](int h, const Rec &R, uint32_t k, float red, const bool first, const bool with_colour, const bool run) {
        // T_k = T_{k+1} / 0.99, correctly rounded (see half_full).  Two operations - the product with the rounded
        // reciprocal, then a fused correction by the reciprocal's own rounding error - would do for one step, but their
        // error is one-sided: the quotient of a forward product lies next to a float by construction, and over 6000
        // clamped entries the transmittance drifts by 7e-5 (5e-6 with the exact remainder).
        const float om = run ? R_P(R) : 1.0f - kAlphaMin, rcp = run ? R_rcpP(R) : 1.0f / (1.0f - kAlphaMin);
        const f2 t0 = T[h] * splat(rcp);
        f2 Tk = pk_fma(pk_fma(splat(-om), t0, T[h]), splat(rcp), t0);
        f2 wgt = splat(0.0f), suf;
        if (run) {
            wgt = Tk;
            const f2 cgd1 = pk_fma(splat(R_b1(R)), gb[h], pk_fma(splat(R_g1(R)), gg[h], splat(R_r1(R)) * gr[h]));
            suf = pk_fma(Tk, cgd1, suffix[h]);
        } else if (with_colour || CHECK || MODE != kOneBlock) {
            wgt = splat(kAlphaMin) * Tk;
            const f2 cgd = pk_fma(splat(R_b(R)), gb[h], pk_fma(splat(R_g(R)), gg[h], splat(red) * gr[h]));
            suf = pk_fma(wgt, cgd, suffix[h]);
        } else {
            // (0.01 x colour from the record of a one-block tile: the weight is never formed)
            const f2 cgd1 = pk_fma(splat(R_b1(R)), gb[h], pk_fma(splat(R_g1(R)), gg[h], splat(R_r1(R)) * gr[h]));
            suf = pk_fma(Tk, cgd1, suffix[h]);
        }
        if (CHECK) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const bool on = (k - w.begin) < nstop[2 * h + e];
                Tk[e] = on ? Tk[e] : T[h][e];
                suf[e] = on ? suf[e] : suffix[h][e];
                wgt[e] = on ? wgt[e] : 0.0f;
            }
        }
        T[h] = Tk;
        suffix[h] = suf;
        if (with_colour) {
            s_r = first ? wgt * gr[h] : pk_fma(wgt, gr[h], s_r);
            s_g = first ? wgt * gg[h] : pk_fma(wgt, gg[h], s_g);
            s_b = first ? wgt * gb[h] : pk_fma(wgt, gb[h], s_b);
        }
    };
    // (`first_geo`: the first half of its entry to add to the geometric sums)
    auto half_full = [&](int h, const Rec &R, uint32_t k, float red, float o, float dx, float adx2, float bdx, const bool first,
                         const bool first_geo) {
        const float cadx = R_ca(R) * dx;   // with v = conic * d:  v_x = a dx + bh dy,  v_y = bh dx + d dy
        const float cbdx = R_cbh(R) * dx;
        // bottom-half rows are the top-half rows + 8 (exact in fp32): one register pair less to keep
        const f2 dy = splat(R_gy(R)) - (h == 0 ? py_top : py_top + splat(8.0f));
        const f2 q = pk_fma(pk_fma(splat(R_D(R)), dy, splat(bdx)), dy, splat(adx2));
        const f2 G = exp2_2(q);
        const f2 raw = splat(o) * G;
        const f2 a = clamp_alpha2(raw);
        // T_k = T_{k+1} / (1 - a), correctly rounded: quotient estimate, exact remainder (fma), one
        // correction.  A plain T * rcp(1 - a) is not good enough here: most entries of a long list
        // have the same a (0.01), so the reciprocal's rounding error repeats with the same sign
        // at every step and the transmittance drifts by ~1e-4 over a few thousand entries.
        const f2 om = splat(1.0f) - a;
        const f2 rcp = f2{__builtin_amdgcn_rcpf(om.x), __builtin_amdgcn_rcpf(om.y)};
        const f2 t0 = T[h] * rcp;
        f2 Tk = pk_fma(pk_fma(-om, t0, T[h]), rcp, t0);   // transmittance in front of this entry
        f2 wgt = a * Tk;
        const f2 cgd = pk_fma(splat(R_b(R)), gb[h], pk_fma(splat(R_g(R)), gg[h], splat(red) * gr[h]));
        f2 d_a = pk_fma(Tk, cgd, -(suffix[h] * rcp));
        f2 suf = pk_fma(wgt, cgd, suffix[h]);
        if (CHECK) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const bool on = (k - w.begin) < nstop[2 * h + e];
                Tk[e] = on ? Tk[e] : T[h][e];
                suf[e] = on ? suf[e] : suffix[h][e];
                wgt[e] = on ? wgt[e] : 0.0f;
                d_a[e] = on ? d_a[e] : 0.0f;
            }
        }
        T[h] = Tk;
        suffix[h] = suf;
        s_r = first ? wgt * gr[h] : pk_fma(wgt, gr[h], s_r);
        s_g = first ? wgt * gg[h] : pk_fma(wgt, gg[h], s_g);
        s_b = first ? wgt * gb[h] : pk_fma(wgt, gb[h], s_b);
        // clamp passes gradient on [0.01, 0.99]: exactly when the clamped value equals the input
        const bool p0 = (a.x == raw.x), p1 = (a.y == raw.y);
        const f2 d_raw = f2{p0 ? d_a.x : 0.0f, p1 ? d_a.y : 0.0f};
        s_o = first_geo ? d_raw * G : pk_fma(d_raw, G, s_o);
        const f2 hh = d_raw * a;                       // = -2 dL/d(exponent); a == raw when it matters
        // Sums are taken in the conic-weighted coordinates v: summing h*d and weighting by the
        // conic afterwards would cancel large coherent sums for elongated Gaussians.
        const f2 vx = pk_fma(splat(R_cbh(R)), dy, splat(cadx));
        const f2 vy = pk_fma(splat(R_cd(R)), dy, splat(cbdx));
        const f2 hvx = hh * vx, hvy = hh * vy;
        s_vx = first_geo ? hvx : s_vx + hvx;
        s_vy = first_geo ? hvy : s_vy + hvy;
        s_xx = first_geo ? hvx * vx : pk_fma(hvx, vx, s_xx);
        s_xy = first_geo ? hvx * vy : pk_fma(hvx, vy, s_xy);
        s_yy = first_geo ? hvy * vy : pk_fma(hvy, vy, s_yy);
    };
    auto entry = [&](const Rec &R, uint32_t k) {
        const uint32_t rec_off = rec_slot<one_block_layout(MODE)>(R.d) * (uint32_t)(GSR_PAIRGRAD_FLOATS * sizeof(float));   // a Gaussian's records are contiguous
        const uint32_t flags = block_flags<!one_block_layout(MODE)>(rec_flags<one_block_layout(MODE)>(R.d), w.flag_shift);
        const float red = R_r(R);
        if (MODE == kBlockRuns && flags == kLowBoth) {
            // a run: U in front of it = U behind it / 0.99^n; the entries' colour gradients follow from it elsewhere
            if (CHECK) {
                half_low(0, R, k, red, true, true, true);
                half_low(1, R, k, red, false, true, true);
                const float val = reduce_full(lanes, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, hsum(s_r), hsum(s_g), hsum(s_b));
                put_record<kOneBlock>(pairgrad, rec_off, lanes.full_off - 24u, val, kRunWriters, gs);
            } else {
                half_low(0, R, k, red, true, false, true);
                half_low(1, R, k, red, false, false, true);
                U = U * R_rcpP(R);
                put_record<kOneBlock>(pairgrad, rec_off, lanes.full_off - 24u, U, kRunWriters, gs);
            }
            return;
        }
        if (flags == kLowBoth) {
            if (CHECK) {
                half_low(0, R, k, red, true, true, false);
                half_low(1, R, k, red, false, true, false);
                reduce_store_colour<MODE>(lanes, pairgrad, rec_off, gs, hsum(s_r), hsum(s_g), hsum(s_b));
            } else {
                half_low(0, R, k, red, true, false, false);
                half_low(1, R, k, red, false, false, false);
                // (legacy multiply: 0 * x = 0 whatever x is, so the six geometric floats of the record are zeros even when a
                // NaN of a diverged model has reached U on the lanes that do not hold a colour sum)
                put_record<MODE>(pairgrad, rec_off, lanes.full_off, mul_zero_wins(U, low_gain), kFullWriters, gs);
                U = fmaf(U, low_gain, U);
            }
            return;
        }
        const float o = R_o(R);
        const float dx = R_gx(R) - pxf;
        const float adx2 = R_A(R) * dx * dx;
        const float bdx = R_B(R) * dx;
        if (flags & kLowTop) {
            half_low(0, R, k, red, true, true, false);
            half_full(1, R, k, red, o, dx, adx2, bdx, false, true);   // (not both halves are flagged here)
        } else {
            half_full(0, R, k, red, o, dx, adx2, bdx, true, true);
            if (flags & kLowBottom) half_low(1, R, k, red, false, true, false); else half_full(1, R, k, red, o, dx, adx2, bdx, false, false);
        }
        // (an entry that is not flagged but whose clamp blocks every pixel anyway - one in a thousand -
        // simply stores zero sums)
        const float val = reduce_full(lanes, hsum(s_vx), hsum(s_vy), hsum(s_xx), hsum(s_xy), hsum(s_yy), hsum(s_o), hsum(s_r), hsum(s_g), hsum(s_b));
        if (!CHECK) U = fmaf(val, lanes.colour, U);
        put_record<MODE>(pairgrad, rec_off, lanes.full_off, val, kFullWriters, gs);
    };
    // back to front, record buffers kept by hand (see composite_fwd_kernel); k is only needed by
    // the checked variant
    uint32_t off = (w.end - 1) * 64u;   // the caller guarantees a non-empty list
    uint32_t k = w.end - 1;
    // (odd lanes touch the record after the one named; a stream of a single record has none)
    const uint32_t touch_off = ((threadIdx.x & 1) && w.total > 1u) ? 64u : 0u;
    float sink = 0.0f;
    Rec A, B;
    const uint32_t n = w.end - w.begin;
    rec_request(stream_rec, off, A);
    if (!CHECK && one_block_layout(MODE)) {
        // plain sweep of a one-block tile: four record buffers, two fetches in flight per wait (see composite_fwd_kernel)
        // (the first record again past the front; positions are relative to the list's first record, and the subtraction
        // that stops at zero is written for the scalar unit: as an expression it becomes a vector instruction)
        auto back = [&](uint32_t records) { return scalar_sub_floor0(off, records * 64u); };
        Rec C, D;
        rec_request(stream_rec, back(1u), B);
        for (uint32_t i = n >> 2; i != 0; --i) {
            rec_wait(A, B);
            rec_request(stream_rec, back(2u), C);
            rec_request(stream_rec, back(3u), D);
            l2_touch(stream_rec, scalar_sub_floor0(off, (kTouchAheadBwd + 1) * 64u), touch_off, sink);
            entry(A, k);
            entry(B, k);
            rec_wait(C, D);
            rec_request(stream_rec, back(4u), A);
            rec_request(stream_rec, back(5u), B);
            l2_touch(stream_rec, scalar_sub_floor0(off, (kTouchAheadBwd + 3) * 64u), touch_off, sink);
            entry(C, k);
            entry(D, k);
            off = back(4u);
        }
        rec_wait(A, B);
        l2_touch_drain(sink);
        const uint32_t rest = n & 3u;
        if (rest >= 1u) entry(A, k);
        if (rest >= 2u) entry(B, k);
        if (rest == 3u) {
            rec_request(stream_rec, back(2u), C);
            rec_wait(C);
            entry(C, k);
        }
        return;
    }
    // two entries per turn and a single exit: the registers of the running sums stay put (a loop that can also
    // end between the two entries made the compiler copy all of them twice per turn)
    for (uint32_t i = n >> 1; i != 0; --i) {
        rec_wait(A);
        rec_request(stream_rec, off - 64u, B);
        // the two lines below the one sixteen entries down (never below the start of the stream)
        l2_touch(stream_rec, scalar_sub_floor0(off, (kTouchAheadBwd + 1) * 64u), touch_off, sink);
        entry(A, k);
        if (MODE == kTileGroup && ++gs.entry == kGroupChunk) group_flush(gs, kGroupChunk, pairgrad);
        rec_wait(B);
        rec_request(stream_rec, scalar_sub_floor0(off, 128u), A);   // (the first record again after the last pair)
        entry(B, k - 1u);
        if (MODE == kTileGroup && ++gs.entry == kGroupChunk) group_flush(gs, kGroupChunk, pairgrad);
        off -= 128u;
        if (CHECK) k -= 2u;
    }
    rec_wait(A);
    l2_touch_drain(sink);   // (before the odd entry: see composite_fwd_kernel)
    if (n & 1u) {
        entry(A, k);
        if (MODE == kTileGroup && ++gs.entry == kGroupChunk) group_flush(gs, kGroupChunk, pairgrad);
    }
    // (every wave of a tile group walks the same list: the barriers inside the flushes match up)
    if (MODE == kTileGroup && gs.entry != 0) group_flush(gs, gs.entry, pairgrad);
}

// MODE: how the blocks of a tile combine their sums (StoreMode above).
// A wave none of whose pixels ran below kTinyT takes the plain sweep, the others the checked one.
// Both live in one launch: the checked waves are few and three times as long, and a launch of their
// own ran at the latency of a single wave per SIMD.  (Five waves per SIMD, 96 registers: a tile that is one block
// runs the same with four - its loop is bound by instruction issue - but the tile groups, whose waves also wait
// for each other at the chunk barriers, gain 5 % from the fifth (1.98 -> 1.87 ms per view at a tile edge of 64);
// six waves would spill.)
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void composite_bwd_kernel(ViewK v, const uint32_t *__restrict__ tile_offsets,
                                                            const uint32_t *__restrict__ tile_order,
                                                            const float *__restrict__ stream_rec,
                                                            const float *__restrict__ final_T,
                                                            const float *__restrict__ dimg,
                                                            float *__restrict__ pairgrad)
{
    WaveGeom w;
    if (!wave_setup(v, tile_offsets, tile_order, w)) return;
    if (w.begin >= w.end) return;   // (the same for every wave of a tile group)
    stream_rec = list_start(stream_rec, w);
    // per-block streams: a record names its row by its position in the block's own stream
    if (MODE == kBlockRuns) pairgrad += (size_t)w.first * GSR_PAIRGRAD_FLOATS;
    raise_wave_priority(w, v.T);
    GroupSlot gs = {nullptr, nullptr, 0u, 1u, 0u, 0u, 0u, 1u};
    if (MODE == kTileGroup) {
        __shared__ float group_parts[2 * kGroupChunk * kGroupMaxWaves * GSR_PAIRGRAD_FLOATS];   // 12 KiB
        __shared__ uint32_t group_slots[2 * kGroupChunk];
        const int S = (v.L + 15) >> 4;
        gs.parts = (lds_float *)group_parts;
        gs.slots = (lds_u32 *)group_slots;
        gs.blocks = blockDim.x >> 6;                  // = group_waves(S): the launch made it so
        gs.block = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        gs.nparts = (uint32_t)(S * S) / gs.blocks;
        gs.part = blockIdx.x % gs.nparts;             // workgroups of a tile are consecutive
    }
    const float pxf = (float)w.x;
    f2 pyf[2], T[2], gr[2], gg[2], gb[2];
    bool tiny = false, small = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int y = w.y0 + 4 * j;   // j = 0, 1: top half of the block; 2, 3: bottom half
        const int h = j >> 1, e = j & 1;
        pyf[h][e] = (float)y;
        T[h][e] = 0.0f; gr[h][e] = gg[h][e] = gb[h][e] = 0.0f;   // a pixel outside the tile contributes exact zeros
        if (w.vx && y < w.ylim) {
            const size_t pix = (size_t)w.x * v.H + y;
            T[h][e] = final_T[pix];
            gr[h][e] = dimg[pix * 3 + 0]; gg[h][e] = dimg[pix * 3 + 1]; gb[h][e] = dimg[pix * 3 + 2];
            small |= !(T[h][e] >= kTinyT);
            tiny |= !(T[h][e] >= kMinNormalT);
        }
    }
    uint32_t nstop[4] = {0u, 0u, 0u, 0u};
    if (__ballot(tiny) != 0ull) {
        // Some pixel's transmittance went denormal or underflowed to zero (long or opaque lists), where
        // dividing back is impossible.  Re-run the forward recurrence to find, per
        // pixel, the entry after which T < kTinyT and the T there; later entries are dropped for
        // that pixel: their weights are below 1e-26 of the pixel's cotangent.
        const uint32_t len = w.end - w.begin;
        float Tf[4], Ts[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { Tf[j] = 1.0f; Ts[j] = 0.0f; nstop[j] = len; }
        bool inside[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) inside[j] = w.vx && (w.y0 + 4 * j) < w.ylim;
        // some pixel of the wave (inside the image) has not reached its stop yet
        auto wave_live = [&]() {
            bool live = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) live |= inside[j] && nstop[j] == len;
            return __ballot(live) != 0ull;
        };
        bool all_stopped = false;
        auto forward_step = [&](const Rec &R, uint32_t idx) {
            const uint32_t bf = block_flags<!one_block_layout(MODE)>(rec_flags<one_block_layout(MODE)>(R.d), w.flag_shift);
            const bool low[2] = {(bf & kLowTop) != 0, (bf & kLowBottom) != 0};
            const bool run = MODE == kBlockRuns && bf == kLowBoth;
            const float o = R_o(R);
            const float dx = R_gx(R) - pxf;
            const float adx2 = R_A(R) * dx * dx;
            const float bdx = R_B(R) * dx;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dy = R_gy(R) - pyf[j >> 1][j & 1];
                const float q = fmaf(fmaf(R_D(R), dy, bdx), dy, adx2);
                const float a = low[j >> 1] ? kAlphaMin : clamp_alpha(o * __builtin_amdgcn_exp2f(q));
                Tf[j] = Tf[j] * (run ? R_P(R) : 1.0f - a);
                const bool hit = (Tf[j] < kTinyT) && (nstop[j] == len);
                nstop[j] = hit ? (idx + 1u) : nstop[j];
                Ts[j] = hit ? Tf[j] : Ts[j];
            }
        };
        {
            // front to back with two record buffers and the L2 touch-ahead, as in composite_fwd_kernel: a
            // loop that waits for every record before it asks for the next one runs at the latency of L2
            const uint32_t off_last = (w.end - 1) * 64u, off_stream_last = (w.total - 1) * 64u;
            const uint32_t touch_floor = min(64u, off_stream_last);
            const uint32_t touch_back = w.total > 1u ? 64u : 0u;
            const uint32_t touch_off = (threadIdx.x & 1) ? 0u : touch_back;
            uint32_t off = w.begin * 64u, idx = 0u;
            float sink = 0.0f;
            Rec A, B;
            rec_request(stream_rec, off, A);
            for (uint32_t i = len >> 1; i != 0; --i) {
                rec_wait(A);
                rec_request(stream_rec, off + 64u, B);
                l2_touch(stream_rec, max(min(off + kTouchAhead * 64u, off_stream_last), touch_floor) - touch_back, touch_off, sink);
                forward_step(A, idx);
                rec_wait(B);
                rec_request(stream_rec, min(off + 128u, off_last), A);   // (the last record again after the last pair)
                forward_step(B, idx + 1u);
                off += 128u;
                idx += 2u;
                // (one wave per list: once every pixel has stopped, the rest of the list matters to none of them - see below)
                if (one_block_layout(MODE) && (idx & 14u) == 0u && !wave_live()) { all_stopped = true; break; }
            }
            rec_wait(A);
            l2_touch_drain(sink);
            if ((len & 1u) && !all_stopped) forward_step(A, idx);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (nstop[j] != len) T[j >> 1][j & 1] = inside[j] ? Ts[j] : 0.0f;
        }
        if (one_block_layout(MODE) && !wave_live()) {
            // Every pixel of the block dropped out before the list ended (at a tile edge of 64 with a few million
            // Gaussians: two thirds into a list of 9000 entries): the records behind the last stop get zero rows without
            // being blended, and the sweep starts at that stop.
            uint32_t last = 0u;
#pragma unroll
            for (int j = 0; j < 4; ++j) last = inside[j] ? max(last, nstop[j]) : last;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) last = max(last, (uint32_t)__shfl_xor((int)last, d, 64));
            last = (uint32_t)__builtin_amdgcn_readfirstlane((int)last);
            const int lane = threadIdx.x & 63;
            const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
            if (MODE == kBlockRuns) {
                // (a record's row is its position in the stream: three lanes per row, 21 rows per turn)
                if (lane < 63)
                    for (uint32_t r = last + (uint32_t)(lane / 3); r < len; r += 21u)
                        reinterpret_cast<float4 *>(pairgrad + (size_t)r * GSR_PAIRGRAD_FLOATS)[lane % 3] = zero;
            } else {
                // (a record's row is named by its slot field: a lane per record)
                for (uint32_t r = last + (uint32_t)lane; r < len; r += 64u) {
                    const uint32_t slot = __float_as_uint(stream_rec[(size_t)r * GSR_STREAM_DWORDS + 9]) & ((1u << kSlotBits) - 1u);
                    float4 *dst = reinterpret_cast<float4 *>(pairgrad + (size_t)slot * GSR_PAIRGRAD_FLOATS);
                    dst[0] = zero; dst[1] = zero; dst[2] = zero;
                }
            }
            if (last == 0u) return;
            WaveGeom head = w;
            head.end = last;
            bwd_sweep<true, MODE>(head, stream_rec, pxf, pyf, T, gr, gg, gb, nstop, pairgrad, gs);
            return;
        }
        bwd_sweep<true, MODE>(w, stream_rec, pxf, pyf, T, gr, gg, gb, nstop, pairgrad, gs);
    } else {
        // Between kTinyT and the smallest normal number the plain sweep still works on a transmittance
        // scaled up by a power of two (exact).  Everything the sweep accumulates is a product of one
        // transmittance and one cotangent, so the wave scales its cotangents down by the same factor and
        // its sums come out unscaled.
        if (__ballot(small) != 0ull) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                T[h] = T[h] * splat(kRescue);
                gr[h] = gr[h] * splat(1.0f / kRescue); gg[h] = gg[h] * splat(1.0f / kRescue); gb[h] = gb[h] * splat(1.0f / kRescue);
            }
        }
        bwd_sweep<false, MODE>(w, stream_rec, pxf, pyf, T, gr, gg, gb, nstop, pairgrad, gs);
    }
}

}  // namespace

extern "C" int gsr_composite_fwd(const gsr_view *view_host, const uint32_t *tile_offsets, const uint32_t *tile_order,
                                 const float *stream_rec, float *img, float *final_T, void *stream)
{
    ViewK v;
    if (int e = gsr_make_view(view_host, &v)) return e;
    if (!tile_offsets || !img || !final_T) { gsr_set_error("gsr_composite_fwd: null pointer"); return GSR_ERR_ARG; }
    const int S = (v.L + 15) >> 4;
    const unsigned waves = (unsigned)v.T * S * S;
    if (S > 1)
        hipLaunchKernelGGL(composite_fwd_kernel<true>, dim3(gsr_div_up(waves, WAVES_PER_BLOCK)), dim3(64 * WAVES_PER_BLOCK), 0,
                           (hipStream_t)stream, v, tile_offsets, tile_order, stream_rec, img, final_T);
    else
        hipLaunchKernelGGL(composite_fwd_kernel<false>, dim3(gsr_div_up(waves, WAVES_PER_BLOCK)), dim3(64 * WAVES_PER_BLOCK), 0,
                           (hipStream_t)stream, v, tile_offsets, tile_order, stream_rec, img, final_T);
    return gsr_check_launch("composite_fwd");
}

extern "C" uint32_t gsr_pairgrad_parts(int tile_length)
{
    const int S = (tile_length + 15) >> 4;
    return (S > 1 && S <= 4) ? (uint32_t)(S * S / group_waves(S)) : 1u;
}

extern "C" int gsr_composite_bwd(const gsr_view *view_host, const uint32_t *tile_offsets, const uint32_t *tile_order,
                                 const float *stream_rec, const float *final_T, const float *dimg, uint32_t num_pairs,
                                 float *pairgrad, void *stream)
{
    ViewK v;
    if (int e = gsr_make_view(view_host, &v)) return e;
    if (num_pairs == 0) return GSR_OK;
    if ((unsigned long long)num_pairs * 64ull >= (1ull << 32)) {
        gsr_set_error("gsr_composite_bwd: %u pairs: the record stream would reach 4 GiB", num_pairs);
        return GSR_ERR_ARG;
    }
    if (!tile_offsets || !stream_rec || !final_T || !dimg || !pairgrad) {
        gsr_set_error("gsr_composite_bwd: null pointer");
        return GSR_ERR_ARG;
    }
    const int S = (v.L + 15) >> 4;
    const unsigned waves = (unsigned)v.T * S * S;
    const dim3 grid(gsr_div_up(waves, WAVES_PER_BLOCK)), block(64 * WAVES_PER_BLOCK);
    if (S > 4) {
        (void)hipMemsetAsync(pairgrad, 0, (size_t)num_pairs * GSR_PAIRGRAD_FLOATS * sizeof(float), (hipStream_t)stream);
        hipLaunchKernelGGL((composite_bwd_kernel<kAtomic>), grid, block, 0, (hipStream_t)stream, v, tile_offsets, tile_order,
                           stream_rec, final_T, dimg, pairgrad);
    } else if (S > 1) {
        // S * S / group_waves(S) workgroups per tile, one wave per block of the tile
        const int G = group_waves(S);
        hipLaunchKernelGGL((composite_bwd_kernel<kTileGroup>), dim3((unsigned)v.T * (unsigned)(S * S / G)), dim3(64 * G), 0,
                           (hipStream_t)stream, v, tile_offsets, tile_order, stream_rec, final_T, dimg, pairgrad);
    } else {
        hipLaunchKernelGGL((composite_bwd_kernel<kOneBlock>), grid, block, 0, (hipStream_t)stream, v, tile_offsets, tile_order,
                           stream_rec, final_T, dimg, pairgrad);
    }
    return gsr_check_launch("composite_bwd");
}

// The compositing kernels of one-block tiles over the per-block streams of gsr_block_lists: `view_host` is the view
// at ITS tile edge (32, 48 or 64), the kernels run on its grid of 16-pixel blocks.
static int blocks_view(const gsr_view *view_host, uint32_t num_pairs, ViewK *v, const char *who)
{
    if (!view_host || !gsr_block_lists_supported(view_host, num_pairs)) {
        gsr_set_error("%s: the view has no per-block lists (tile edges 32, 48, 64; pairs x blocks per tile < 2^28)", who);
        return GSR_ERR_ARG;
    }
    gsr_view v16 = *view_host;
    v16.tile_length = 16;
    return gsr_make_view(&v16, v);
}

extern "C" int gsr_composite_blocks_fwd(const gsr_view *view_host, uint32_t num_pairs, const uint32_t *block_offsets,
                                        const uint32_t *block_order, const float *block_stream, float *img, float *final_T,
                                        void *stream)
{
    ViewK v;
    if (int e = blocks_view(view_host, num_pairs, &v, "gsr_composite_blocks_fwd")) return e;
    if (!block_offsets || !block_stream || !img || !final_T) { gsr_set_error("gsr_composite_blocks_fwd: null pointer"); return GSR_ERR_ARG; }
    hipLaunchKernelGGL((composite_fwd_kernel<false, true>), dim3(gsr_div_up((unsigned)v.T, WAVES_PER_BLOCK)), dim3(64 * WAVES_PER_BLOCK), 0,
                       (hipStream_t)stream, v, block_offsets, block_order, block_stream, img, final_T);
    return gsr_check_launch("composite_blocks_fwd");
}

extern "C" int gsr_composite_blocks_bwd(const gsr_view *view_host, uint32_t num_pairs, const uint32_t *block_offsets,
                                        const uint32_t *block_order, const float *block_stream, const float *final_T,
                                        const float *dimg, float *block_grad, void *stream)
{
    ViewK v;
    if (int e = blocks_view(view_host, num_pairs, &v, "gsr_composite_blocks_bwd")) return e;
    if (num_pairs == 0) return GSR_OK;
    if (!block_offsets || !block_stream || !final_T || !dimg || !block_grad) {
        gsr_set_error("gsr_composite_blocks_bwd: null pointer");
        return GSR_ERR_ARG;
    }
    hipLaunchKernelGGL((composite_bwd_kernel<kBlockRuns>), dim3(gsr_div_up((unsigned)v.T, WAVES_PER_BLOCK)), dim3(64 * WAVES_PER_BLOCK), 0,
                       (hipStream_t)stream, v, block_offsets, block_order, block_stream, final_T, dimg, block_grad);
    return gsr_check_launch("composite_blocks_bwd");
}
