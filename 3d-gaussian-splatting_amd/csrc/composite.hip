// Stage 3: per-tile front-to-back blend and its backward.
//
// Work decomposition (CDNA4): one 64-lane wave owns one 16x16 pixel block, 4 pixels per lane
// (lane -> column x = lane/4, rows y0, y0+4, y0+8, y0+12 with y0 = lane%4).  The lane's first two
// pixels lie in the top 8 rows of the block and the other two in the bottom 8 rows; each pair is
// processed with packed fp32 instructions, and a pair is the unit of the clamped-low fast path
// (see rec_low_top / rec_low_bottom).  A reference tile of edge L is covered
// by ceil(L/16)^2 such waves, all streaming the tile's list.  The list is a contiguous stream of
// 48-byte splat records in depth order; the record address is wave-uniform, so records are
// fetched with scalar loads into SGPRs and broadcast to the vector ALU for free - no LDS
// staging, no barriers, waves are fully independent.  The loop is FP32-VALU / transcendental
// bound (one v_exp_f32 per pixel-Gaussian pair), not HBM bound.
//
// Semantics follow the reference exactly (GaussianRenderer.py:100-112, Utils/Probability.py):
// integer pixel coordinates, alpha = clamp(opacity*G, 0.01, 0.99) for EVERY list entry, the
// whole list is blended (no early exit), no background term.
#include "gsr_common.hpp"

namespace {

constexpr int WAVES_PER_BLOCK = 4;
constexpr float kAlphaMin = 0.01f;
constexpr float kAlphaMax = 0.99f;
// Below this transmittance a pixel's remaining entries cannot change any fp32 result; the
// backward stops dividing there (see composite_bwd_kernel).
constexpr float kTinyT = 1e-20f;
constexpr uint32_t kTouchAhead = 16;   // records (768 bytes) between the L2 touch and the scalar fetch

struct WaveGeom {
    int tile, x, y0;     // pixel column and first row of this lane; its rows are y0 + 4 j, j = 0..3
    bool vx;             // column inside the tile and the image
    int ylim;            // rows < ylim are inside
    uint32_t begin, end; // list range (wave-uniform)
};

__device__ inline bool wave_setup(const ViewK &v, const uint32_t *__restrict__ tile_offsets, WaveGeom &w)
{
    const int S = (v.L + 15) >> 4;
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned wv = blockIdx.x * WAVES_PER_BLOCK + wave_in_block;
    const unsigned per_tile = (unsigned)(S * S);
    const unsigned tile = wv / per_tile;
    if (tile >= (unsigned)v.T) return false;
    const unsigned sub = wv - tile * per_tile;
    const int sxi = (int)(sub / (unsigned)S), syi = (int)(sub - (unsigned)sxi * S);
    const int xi = (int)(tile / (unsigned)v.nbh), yi = (int)(tile - (unsigned)xi * v.nbh);
    const int lane = threadIdx.x & 63;
    w.tile = (int)tile;
    w.x = xi * v.L + sxi * 16 + (lane >> 2);
    w.y0 = yi * v.L + syi * 16 + (lane & 3);
    const int xlim = min((xi + 1) * v.L, v.W);
    w.ylim = min((yi + 1) * v.L, v.H);
    w.vx = w.x < xlim;
    w.begin = tile_offsets[tile];
    w.end = tile_offsets[tile + 1];
    return true;
}

typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// One 48-byte splat record held in 12 SGPRs.
struct Rec {
    f32x8 a;   // gx, gy, A, B, D, o, r, g   centre, pre-scaled conic (exp2 domain), opacity, red, green
    f32x4 b;   // b, ca, cbh, cd             blue, conic a, (b+c)/2, d as they are (backward only)
};
#define R_gx(R) ((R).a[0])
#define R_gy(R) ((R).a[1])
#define R_A(R) ((R).a[2])
#define R_B(R) ((R).a[3])
#define R_D(R) ((R).a[4])
#define R_o(R) ((R).a[5])
#define R_r(R) ((R).a[6])
#define R_g(R) ((R).a[7])
#define R_b(R) ((R).b[0])
#define R_ca(R) ((R).b[1])
#define R_cbh(R) ((R).b[2])
#define R_cd(R) ((R).b[3])

// Scalar loads of record k, issued now and waited for later (rec_wait), so that the latency of
// one record's fetch (a few hundred cycles from L2) overlaps the arithmetic of the previous one.
// hipcc would otherwise sink the load below its last use to reuse the registers.  The waits are
// written by hand because the compiler does not count loads issued from inline asm.
__device__ inline void rec_request(const float *__restrict__ stream_rec, uint32_t k, Rec &R)
{
    const float *p = stream_rec + (size_t)k * GSR_REC_FLOATS;
    asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dwordx4 %1, %2, 0x20" : "=&s"(R.a), "=&s"(R.b) : "s"(p) : "memory");
}
__device__ inline void rec_wait(Rec &R) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(R.a), "+s"(R.b)::"memory"); }


// The same plus the record's emission slot (where its gradient record goes), for the backward.
__device__ inline void rec_request(const float *__restrict__ stream_rec, const uint32_t *__restrict__ pair_slot,
                                   uint32_t k, Rec &R, uint32_t &slot)
{
    const float *p = stream_rec + (size_t)k * GSR_REC_FLOATS;
    const uint32_t *q = pair_slot + k;
    asm volatile("s_load_dwordx8 %0, %3, 0x0\n\ts_load_dwordx4 %1, %3, 0x20\n\ts_load_dword %2, %4, 0x0"
                 : "=&s"(R.a), "=&s"(R.b), "=&s"(slot)
                 : "s"(p), "s"(q)
                 : "memory");
}
__device__ inline void rec_wait(Rec &R, uint32_t &slot)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(R.a), "+s"(R.b), "+s"(slot)::"memory");
}

// Pulls the cache lines of a record some entries further on into L2 with a vector load whose
// result is never used: lanes alternate between the record's first and last dword (byte offsets 0
// and 44).  Issued once per two entries (96 bytes of stream), the touched addresses are at most 52
// bytes apart, so every 64-byte line of the stream is touched and no address leaves a record.
// The scalar record loads then hit L2 instead of waiting for HBM; a wave has only one record in
// flight, so HBM latency would otherwise bound the stream rate.  `sink` must stay live until the
// wave drains its vector loads (l2_touch_drain).
__device__ inline void l2_touch(const float *p, uint32_t lane_off, float &sink)
{
    asm volatile("global_load_dword %0, %1, %2" : "+v"(sink) : "v"(lane_off), "s"(p) : "memory");
}
__device__ inline void l2_touch_drain(float &sink) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink)::"memory"); }

// The list builder marks, per record, the halves of the 16x16 block on which alpha is pinned at
// 0.01 for every pixel: sign bit of the opacity = top 8 rows, sign bit of the red colour = bottom 8
// rows (both values are >= 0 by construction; read them through rec_opacity / rec_red).  For tiles larger than one block both bits describe the whole
// tile.  Wave-uniform: scalar compares and branches.
__device__ inline bool rec_low_top(const Rec &R) { return __float_as_int(R_o(R)) < 0; }
__device__ inline bool rec_low_bottom(const Rec &R) { return __float_as_int(R_r(R)) < 0; }
__device__ inline float strip_sign(float x) { return __int_as_float(__float_as_int(x) & 0x7fffffff); }
__device__ inline float rec_opacity(const Rec &R) { return strip_sign(R_o(R)); }
__device__ inline float rec_red(const Rec &R) { return strip_sign(R_r(R)); }

__device__ inline float clamp_alpha(float raw) { return fminf(fmaxf(raw, kAlphaMin), kAlphaMax); }

// Two pixels per operation.  On gfx950 a v_fma_f32 costs about 4.4 cycles of a SIMD and a
// v_pk_fma_f32 about 5.6 for two lanes' worth (measured, tools/ubench/valu_rate.hip), so the
// per-pixel fused multiply-adds are written on pairs; a lane's 4 pixels are 2 pairs.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ inline f2 splat(float s) { return f2{s, s}; }
__device__ inline f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ inline f2 exp2_2(f2 q) { return f2{__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)}; }
__device__ inline f2 clamp_alpha2(f2 raw) { return f2{clamp_alpha(raw.x), clamp_alpha(raw.y)}; }

__global__ __launch_bounds__(256) void composite_fwd_kernel(ViewK v, const uint32_t *__restrict__ tile_offsets,
                                                            const float *__restrict__ stream_rec,
                                                            float *__restrict__ img, float *__restrict__ final_T)
{
    WaveGeom w;
    if (!wave_setup(v, tile_offsets, w)) return;
    const float pxf = (float)w.x;
    f2 pyf[2], T[2], cr[2], cg[2], cb[2];   // pixel rows (y0, y0+4): top half, (y0+8, y0+12): bottom half
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        pyf[h] = f2{(float)(w.y0 + 8 * h), (float)(w.y0 + 8 * h + 4)};
        T[h] = splat(1.0f);
        cr[h] = cg[h] = cb[h] = splat(0.0f);
    }

    // alpha is 0.01 on this half of the block: no exponent, no exp
    auto half_low = [&](int h, const Rec &R, float red) {
        const f2 wgt = splat(kAlphaMin) * T[h];
        cr[h] = pk_fma(wgt, splat(red), cr[h]);
        cg[h] = pk_fma(wgt, splat(R_g(R)), cg[h]);
        cb[h] = pk_fma(wgt, splat(R_b(R)), cb[h]);
        T[h] = T[h] * splat(1.0f - kAlphaMin);
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
        const bool low0 = rec_low_top(R), low1 = rec_low_bottom(R);
        const float red = rec_red(R);
        if (low0 && low1) {
            half_low(0, R, red);
            half_low(1, R, red);
            return;
        }
        const float o = rec_opacity(R);
        const float dx = R_gx(R) - pxf;
        const float adx2 = R_A(R) * dx * dx;
        const float bdx = R_B(R) * dx;
        // (the forward pass is bound by the rate of the scalar record fetches, not by arithmetic:
        // treating a single clamped-low half separately only adds branches here)
        half_full(0, R, red, o, bdx, adx2);
        half_full(1, R, red, o, bdx, adx2);
    };
    const uint32_t touch_off = (threadIdx.x & 1) * 44u;
    float sink = 0.0f;
    if (w.begin < w.end) {
        // two record buffers, alternating: while one is blended the other is in flight
        const uint32_t last = w.end - 1;
        uint32_t k = w.begin;
        Rec A, B;
        rec_request(stream_rec, k, A);
        for (;;) {
            rec_wait(A);
            rec_request(stream_rec, min(k + 1, last), B);
            l2_touch(stream_rec + (size_t)min(k + kTouchAhead, last) * GSR_REC_FLOATS, touch_off, sink);
            blend(A);
            if (++k >= w.end) break;
            rec_wait(B);
            rec_request(stream_rec, min(k + 1, last), A);
            blend(B);
            if (++k >= w.end) break;
        }
        rec_wait(A);   // nothing may be in flight when the wave ends
        rec_wait(B);
        l2_touch_drain(sink);
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
struct RecordLanes {
    bool full_writer, full_main;   // lane writes a float of a full record / it is one of the 8 merged sums
    uint32_t full_slot;
    bool col_writer, col_zero;
    uint32_t col_slot;
};
__device__ inline RecordLanes record_lanes(int lane)
{
    RecordLanes L;
    L.full_main = (lane & 7) == 0;
    L.full_writer = L.full_main || (lane & 15) == 1;
    L.full_slot = L.full_main ? (uint32_t)(lane >> 3) : 8u + (uint32_t)(lane >> 4);
    L.col_zero = lane < 6 || lane == 10 || lane == 11;
    L.col_writer = L.col_zero || lane == 6 || lane == 24 || lane == 39 || lane == 57;
    L.col_slot = (uint32_t)(lane & 15);
    return L;
}

template <bool ATOMIC>
__device__ inline void put_record_float(float *dst, float val)
{
    if (ATOMIC) {
        if (val != 0.0f) atomicAdd(dst, val);   // several waves share the record; it was zeroed by the launcher
    } else {
        *dst = val;
    }
}

// Full record, floats: sum h v_x, sum h v_y, sum h v_x^2, sum h v_x v_y, sum h v_y^2, dL/d(opacity),
// dL/d(red), dL/d(green), then four partial sums of dL/d(blue).
template <bool ATOMIC>
__device__ inline void reduce_store_full(const RecordLanes &L, float *dst, float vx, float vy, float xx, float xy,
                                         float yy, float o, float r, float g, float b)
{
    // pairing order chosen so that group g of 8 lanes ends up with record float g
    const float m0 = merge_halves(vx, yy), m1 = merge_halves(xx, r), m2 = merge_halves(vy, o), m3 = merge_halves(xy, g);
    const float n0 = merge_rows(m0, m1);   // rows: vx, xx, yy, r
    const float n1 = merge_rows(m2, m3);   // rows: vy, xy, o,  g
    const float all8 = sum8(merge_half_rows(n0, n1));
    const float blue = sum16(b);
    if (L.full_writer) put_record_float<ATOMIC>(dst + L.full_slot, L.full_main ? all8 : blue);
}

// Colour-only record (the clamp blocks every other gradient of the entry).
template <bool ATOMIC>
__device__ inline void reduce_store_colour(const RecordLanes &L, float *dst, float r, float g, float b)
{
    const float rows = sum16(merge_rows(merge_halves(r, g), b));   // rows: red, blue(0-1), green, blue(2-3)
    if (L.col_writer) put_record_float<ATOMIC>(dst + L.col_slot, L.col_zero ? 0.0f : rows);
}

// Reverse sweep of one list for the 4 pixels of a lane.  CHECK: pixel j only takes part in
// entries with index (k - begin) < nstop[j].
template <bool CHECK, bool ATOMIC>
__device__ inline void bwd_sweep(const WaveGeom &w, const float *__restrict__ stream_rec, const float pxf,
                                 const f2 (&pyf)[2], f2 (&T)[2], const f2 (&gr)[2], const f2 (&gg)[2],
                                 const f2 (&gb)[2], const uint32_t (&nstop)[4],
                                 const uint32_t *__restrict__ pair_slot, float *__restrict__ pairgrad)
{
    f2 suffix[2] = {splat(0.f), splat(0.f)};
    const f2 py_top = pyf[0];
    const RecordLanes lanes = record_lanes(threadIdx.x & 63);
    // Sums of one entry over the lane's pixels (pairs are added up at the end).
    f2 s_vx, s_vy, s_xx, s_xy, s_yy, s_o, s_r, s_g, s_b;
    bool any_pass;

    // One half of the block with alpha = 0.01 on every pixel: only the transmittance recurrence, the
    // colour sums and the suffix remain; the clamp blocks every geometric gradient.
    auto half_low = [&](int h, const Rec &R, uint32_t k, float red) {
        const float om = 1.0f - kAlphaMin, rcp = 1.0f / (1.0f - kAlphaMin);
        const f2 t0 = T[h] * splat(rcp);
        f2 Tk = pk_fma(pk_fma(splat(-om), t0, T[h]), splat(rcp), t0);
        f2 wgt = splat(kAlphaMin) * Tk;
        const f2 cgd = pk_fma(splat(R_b(R)), gb[h], pk_fma(splat(R_g(R)), gg[h], splat(red) * gr[h]));
        f2 suf = pk_fma(wgt, cgd, suffix[h]);
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
        s_r = pk_fma(wgt, gr[h], s_r);
        s_g = pk_fma(wgt, gg[h], s_g);
        s_b = pk_fma(wgt, gb[h], s_b);
    };
    auto half_full = [&](int h, const Rec &R, uint32_t k, float red, float o, float dx, float adx2, float bdx) {
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
        s_r = pk_fma(wgt, gr[h], s_r);
        s_g = pk_fma(wgt, gg[h], s_g);
        s_b = pk_fma(wgt, gb[h], s_b);
        // clamp passes gradient on [0.01, 0.99]: exactly when the clamped value equals the input
        const bool p0 = (a.x == raw.x), p1 = (a.y == raw.y);
        any_pass |= p0 | p1;
        const f2 d_raw = f2{p0 ? d_a.x : 0.0f, p1 ? d_a.y : 0.0f};
        s_o = pk_fma(d_raw, G, s_o);
        const f2 hh = d_raw * a;                       // = -2 dL/d(exponent); a == raw when it matters
        // Sums are taken in the conic-weighted coordinates v: summing h*d and weighting by the
        // conic afterwards would cancel large coherent sums for elongated Gaussians.
        const f2 vx = pk_fma(splat(R_cbh(R)), dy, splat(cadx));
        const f2 vy = pk_fma(splat(R_cd(R)), dy, splat(cbdx));
        const f2 hvx = hh * vx, hvy = hh * vy;
        s_vx += hvx;
        s_vy += hvy;
        s_xx = pk_fma(hvx, vx, s_xx);
        s_xy = pk_fma(hvx, vy, s_xy);
        s_yy = pk_fma(hvy, vy, s_yy);
    };
    auto entry = [&](const Rec &R, uint32_t k, uint32_t slot) {
        float *dst = pairgrad + (size_t)slot * GSR_PAIRGRAD_FLOATS;   // a Gaussian's records are contiguous
        const bool low0 = rec_low_top(R), low1 = rec_low_bottom(R);
        const f2 z = splat(0.f);
        s_r = s_g = s_b = z;
        const float red = rec_red(R);
        if (low0 && low1) {
            half_low(0, R, k, red);
            half_low(1, R, k, red);
            reduce_store_colour<ATOMIC>(lanes, dst, s_r.x + s_r.y, s_g.x + s_g.y, s_b.x + s_b.y);
            return;
        }
        s_vx = s_vy = s_xx = s_xy = s_yy = s_o = z;
        any_pass = false;
        const float o = rec_opacity(R);
        const float dx = R_gx(R) - pxf;
        const float adx2 = R_A(R) * dx * dx;
        const float bdx = R_B(R) * dx;
        if (low0) {
            half_low(0, R, k, red);
            half_full(1, R, k, red, o, dx, adx2, bdx);
        } else if (low1) {
            half_full(0, R, k, red, o, dx, adx2, bdx);
            half_low(1, R, k, red);
        } else {
            half_full(0, R, k, red, o, dx, adx2, bdx);
            half_full(1, R, k, red, o, dx, adx2, bdx);
        }
        const bool any_geom = __ballot(any_pass) != 0ull;
        if (any_geom)
            reduce_store_full<ATOMIC>(lanes, dst, s_vx.x + s_vx.y, s_vy.x + s_vy.y, s_xx.x + s_xx.y, s_xy.x + s_xy.y,
                                      s_yy.x + s_yy.y, s_o.x + s_o.y, s_r.x + s_r.y, s_g.x + s_g.y, s_b.x + s_b.y);
        else
            reduce_store_colour<ATOMIC>(lanes, dst, s_r.x + s_r.y, s_g.x + s_g.y, s_b.x + s_b.y);
    };
    // back to front, two record buffers alternating (see composite_fwd_kernel)
    uint32_t k = w.end - 1;   // the caller guarantees a non-empty list
    Rec A, B;
    uint32_t sa, sb;
    const uint32_t touch_off = (threadIdx.x & 1) * 44u;
    float sink = 0.0f;
    rec_request(stream_rec, pair_slot, k, A, sa);
    for (;;) {
        rec_wait(A, sa);
        rec_request(stream_rec, pair_slot, k > w.begin ? k - 1 : k, B, sb);
        l2_touch(stream_rec + (size_t)(k - w.begin > kTouchAhead ? k - kTouchAhead : w.begin) * GSR_REC_FLOATS,
                 touch_off, sink);
        entry(A, k, sa);
        if (k == w.begin) break;
        --k;
        rec_wait(B, sb);
        rec_request(stream_rec, pair_slot, k > w.begin ? k - 1 : k, A, sa);
        entry(B, k, sb);
        if (k == w.begin) break;
        --k;
    }
    rec_wait(A, sa);
    rec_wait(B, sb);
    l2_touch_drain(sink);
}

// ATOMIC = several waves share a tile (L > 16): their sums meet in pairgrad through float atomics
// (pairgrad zeroed by the launcher).  Otherwise each record is written once with plain stores.
// The kernel is launched twice.  TINY = false handles the waves none of whose pixels ran below
// kTinyT and returns at once in the others; TINY = true does the opposite.  The plain sweep then
// does not pay (in registers, hence in waves per SIMD) for the rarely needed checked one.
template <bool TINY, bool ATOMIC>
__global__ __launch_bounds__(256) void composite_bwd_kernel(ViewK v, const uint32_t *__restrict__ tile_offsets,
                                                            const float *__restrict__ stream_rec,
                                                            const float *__restrict__ final_T,
                                                            const float *__restrict__ dimg,
                                                            const uint32_t *__restrict__ pair_slot,
                                                            float *__restrict__ pairgrad)
{
    WaveGeom w;
    if (!wave_setup(v, tile_offsets, w)) return;
    if (w.begin >= w.end) return;
    const float pxf = (float)w.x;
    f2 pyf[2], T[2], gr[2], gg[2], gb[2];
    bool tiny = false;
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
            tiny |= !(T[h][e] >= kTinyT);
        }
    }
    uint32_t nstop[4] = {0u, 0u, 0u, 0u};
    if ((__ballot(tiny) != 0ull) != TINY) return;
    if (TINY) {
        // Some pixel's transmittance ran below kTinyT (long or opaque lists; it may have underflowed
        // to zero, where dividing back is impossible).  Re-run the forward recurrence to find, per
        // pixel, the entry after which T < kTinyT and the T there; later entries are dropped for
        // that pixel: their weights are below 1e-20 of the pixel's cotangent.
        const uint32_t len = w.end - w.begin;
        float Tf[4], Ts[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { Tf[j] = 1.0f; Ts[j] = 0.0f; nstop[j] = len; }
        for (uint32_t k = w.begin; k < w.end; ++k) {
            Rec R;
            rec_request(stream_rec, k, R);
            rec_wait(R);
            const bool low[2] = {rec_low_top(R), rec_low_bottom(R)};
            const float o = rec_opacity(R);
            const float dx = R_gx(R) - pxf;
            const float adx2 = R_A(R) * dx * dx;
            const float bdx = R_B(R) * dx;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dy = R_gy(R) - pyf[j >> 1][j & 1];
                const float q = fmaf(fmaf(R_D(R), dy, bdx), dy, adx2);
                const float a = low[j >> 1] ? kAlphaMin : clamp_alpha(o * __builtin_amdgcn_exp2f(q));
                Tf[j] = Tf[j] * (1.0f - a);
                const bool hit = (Tf[j] < kTinyT) && (nstop[j] == len);
                nstop[j] = hit ? (k - w.begin + 1u) : nstop[j];
                Ts[j] = hit ? Tf[j] : Ts[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool inside = w.vx && (w.y0 + 4 * j) < w.ylim;
            if (nstop[j] != len) T[j >> 1][j & 1] = inside ? Ts[j] : 0.0f;
        }
        bwd_sweep<true, ATOMIC>(w, stream_rec, pxf, pyf, T, gr, gg, gb, nstop, pair_slot, pairgrad);
    } else {
        bwd_sweep<false, ATOMIC>(w, stream_rec, pxf, pyf, T, gr, gg, gb, nstop, pair_slot, pairgrad);
    }
}

}  // namespace

extern "C" int gsr_composite_fwd(const gsr_view *view_host, const uint32_t *tile_offsets, const float *stream_rec,
                                 float *img, float *final_T, void *stream)
{
    ViewK v;
    if (int e = gsr_make_view(view_host, &v)) return e;
    if (!tile_offsets || !img || !final_T) { gsr_set_error("gsr_composite_fwd: null pointer"); return GSR_ERR_ARG; }
    const int S = (v.L + 15) >> 4;
    const unsigned waves = (unsigned)v.T * S * S;
    hipLaunchKernelGGL(composite_fwd_kernel, dim3(gsr_div_up(waves, WAVES_PER_BLOCK)), dim3(64 * WAVES_PER_BLOCK), 0,
                       (hipStream_t)stream, v, tile_offsets, stream_rec, img, final_T);
    return gsr_check_launch("composite_fwd");
}

extern "C" int gsr_composite_bwd(const gsr_view *view_host, const uint32_t *tile_offsets, const float *stream_rec,
                                 const uint32_t *pair_slot, const float *final_T, const float *dimg,
                                 uint32_t num_pairs, float *pairgrad, void *stream)
{
    ViewK v;
    if (int e = gsr_make_view(view_host, &v)) return e;
    if (num_pairs == 0) return GSR_OK;
    if (!tile_offsets || !stream_rec || !pair_slot || !final_T || !dimg || !pairgrad) {
        gsr_set_error("gsr_composite_bwd: null pointer");
        return GSR_ERR_ARG;
    }
    const int S = (v.L + 15) >> 4;
    const unsigned waves = (unsigned)v.T * S * S;
    const dim3 grid(gsr_div_up(waves, WAVES_PER_BLOCK)), block(64 * WAVES_PER_BLOCK);
    if (S > 1) {
        (void)hipMemsetAsync(pairgrad, 0, (size_t)num_pairs * GSR_PAIRGRAD_FLOATS * sizeof(float), (hipStream_t)stream);
        hipLaunchKernelGGL((composite_bwd_kernel<false, true>), grid, block, 0, (hipStream_t)stream, v, tile_offsets,
                           stream_rec, final_T, dimg, pair_slot, pairgrad);
        hipLaunchKernelGGL((composite_bwd_kernel<true, true>), grid, block, 0, (hipStream_t)stream, v, tile_offsets,
                           stream_rec, final_T, dimg, pair_slot, pairgrad);
    } else {
        hipLaunchKernelGGL((composite_bwd_kernel<false, false>), grid, block, 0, (hipStream_t)stream, v, tile_offsets,
                           stream_rec, final_T, dimg, pair_slot, pairgrad);
        hipLaunchKernelGGL((composite_bwd_kernel<true, false>), grid, block, 0, (hipStream_t)stream, v, tile_offsets,
                           stream_rec, final_T, dimg, pair_slot, pairgrad);
    }
    return gsr_check_launch("composite_bwd");
}
