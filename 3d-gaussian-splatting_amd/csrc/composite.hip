// Stage 3: per-tile front-to-back blend and its backward.
//
// Work decomposition (CDNA4): one 64-lane wave owns one 16x16 pixel block, 4 pixels per lane
// (lane -> column x = lane/4, rows y0..y0+3 with y0 = 4*(lane%4); y is the fast image axis, so
// a lane's pixels are 12 contiguous floats of the image).  A reference tile of edge L is covered
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

struct WaveGeom {
    int tile, x, y0;     // pixel column and first row of this lane
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
    w.y0 = yi * v.L + syi * 16 + (lane & 3) * 4;
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

// The list builder sets the sign bit of the opacity when alpha is pinned at 0.01 on every pixel of
// the tile (wave-uniform: a scalar compare and branch).
__device__ inline bool rec_all_clamped_low(const Rec &R) { return __float_as_int(R_o(R)) < 0; }

__device__ inline float clamp_alpha(float raw) { return fminf(fmaxf(raw, kAlphaMin), kAlphaMax); }

__global__ __launch_bounds__(256) void composite_fwd_kernel(ViewK v, const uint32_t *__restrict__ tile_offsets,
                                                            const float *__restrict__ stream_rec,
                                                            float *__restrict__ img, float *__restrict__ final_T)
{
    WaveGeom w;
    if (!wave_setup(v, tile_offsets, w)) return;
    const float pxf = (float)w.x;
    float pyf[4], T[4], cr[4], cg[4], cb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { pyf[j] = (float)(w.y0 + j); T[j] = 1.0f; cr[j] = cg[j] = cb[j] = 0.0f; }

    auto blend = [&](const Rec &R) {
        if (rec_all_clamped_low(R)) {
            // alpha is 0.01 on the whole tile: no exponent, no exp
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float wgt = kAlphaMin * T[j];
                cr[j] = fmaf(wgt, R_r(R), cr[j]);
                cg[j] = fmaf(wgt, R_g(R), cg[j]);
                cb[j] = fmaf(wgt, R_b(R), cb[j]);
                T[j] = T[j] * (1.0f - kAlphaMin);
            }
            return;
        }
        const float dx = R_gx(R) - pxf;
        const float adx2 = R_A(R) * dx * dx;
        const float bdx = R_B(R) * dx;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float dy = R_gy(R) - pyf[j];
            const float q = fmaf(fmaf(R_D(R), dy, bdx), dy, adx2);
            const float G = __builtin_amdgcn_exp2f(q);
            const float a = clamp_alpha(R_o(R) * G);
            const float wgt = a * T[j];
            cr[j] = fmaf(wgt, R_r(R), cr[j]);
            cg[j] = fmaf(wgt, R_g(R), cg[j]);
            cb[j] = fmaf(wgt, R_b(R), cb[j]);
            T[j] = T[j] * (1.0f - a);   // same rounding as the reference's cumprod of (1 - alpha)
        }
    };
    if (w.begin < w.end) {
        // two record buffers, alternating: while one is blended the other is in flight
        const uint32_t last = w.end - 1;
        uint32_t k = w.begin;
        Rec A, B;
        rec_request(stream_rec, k, A);
        for (;;) {
            rec_wait(A);
            rec_request(stream_rec, min(k + 1, last), B);
            blend(A);
            if (++k >= w.end) break;
            rec_wait(B);
            rec_request(stream_rec, min(k + 1, last), A);
            blend(B);
            if (++k >= w.end) break;
        }
        rec_wait(A);   // nothing may be in flight when the wave ends
        rec_wait(B);
    }
    if (!w.vx) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int y = w.y0 + j;
        if (y < w.ylim) {
            const size_t pix = (size_t)w.x * v.H + y;
            img[pix * 3 + 0] = cr[j]; img[pix * 3 + 1] = cg[j]; img[pix * 3 + 2] = cb[j];
            final_T[pix] = T[j];
        }
    }
}

// Sum over the 64 lanes with DPP row operations; the total lands in lane 63.
template <int CTRL, int ROW_MASK>
__device__ inline float dpp_add(float val)
{
    // lanes the control or the row mask leaves without a source receive 0
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(val), CTRL, ROW_MASK, 0xF, false);
    return val + __int_as_float(moved);
}

// Row-broadcast steps of the wave reduction.  hipcc fuses a DPP move into a float add only for full
// row masks, so these two steps are written as the single instruction they are.  DPP reads need two
// wait states after the VALU write of their source: one s_nop 1 ahead of each group; inside a
// group consecutive instructions touch different registers.
#define GSR_BCAST3(ctrl, mask, x)                                                                           \
    asm volatile("s_nop 1\n\t"                                                                             \
                 "v_add_f32_dpp %0, %0, %0 " ctrl " row_mask:" mask " bank_mask:0xf\n\t"                    \
                 "v_add_f32_dpp %1, %1, %1 " ctrl " row_mask:" mask " bank_mask:0xf\n\t"                    \
                 "v_add_f32_dpp %2, %2, %2 " ctrl " row_mask:" mask " bank_mask:0xf"                         \
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]))

// Sums 3*G values (G groups of three) over the 64 lanes; totals land in lane 63.  Each DPP step is
// applied to all values before the next step starts, so dependent steps are far enough apart.
template <int N>
__device__ inline void wave_sum_to_lane63(float (&x)[N])
{
    static_assert(N % 3 == 0, "values are reduced in groups of three");
#pragma unroll
    for (int c = 0; c < N; ++c) x[c] = dpp_add<0xB1, 0xF>(x[c]);   // quad_perm [1,0,3,2]
#pragma unroll
    for (int c = 0; c < N; ++c) x[c] = dpp_add<0x4E, 0xF>(x[c]);   // quad_perm [2,3,0,1]
#pragma unroll
    for (int c = 0; c < N; ++c) x[c] = dpp_add<0x141, 0xF>(x[c]);  // row_half_mirror
#pragma unroll
    for (int c = 0; c < N; ++c) x[c] = dpp_add<0x140, 0xF>(x[c]);  // row_mirror: every lane holds its row's sum
#pragma unroll
    for (int c = 0; c < N; c += 3) { float (&g)[3] = reinterpret_cast<float (&)[3]>(x[c]); GSR_BCAST3("row_bcast:15", "0xa", g); }
#pragma unroll
    for (int c = 0; c < N; c += 3) { float (&g)[3] = reinterpret_cast<float (&)[3]>(x[c]); GSR_BCAST3("row_bcast:31", "0xc", g); }
}

// One gradient record: plain stores when a single wave owns the tile, float atomics otherwise.
template <bool ATOMIC>
__device__ inline void store_pairgrad(float *dst, float v0, float v1, float v2, float v3, float v4, float v5, float v6,
                                      float v7, float v8)
{
    if (ATOMIC) {
        const float v[9] = {v0, v1, v2, v3, v4, v5, v6, v7, v8};
#pragma unroll
        for (int c = 0; c < 9; ++c)
            if (v[c] != 0.0f) atomicAdd(dst + c, v[c]);
    } else {
        reinterpret_cast<float4 *>(dst)[0] = make_float4(v0, v1, v2, v3);
        reinterpret_cast<float4 *>(dst)[1] = make_float4(v4, v5, v6, v7);
        reinterpret_cast<float4 *>(dst)[2] = make_float4(v8, 0.0f, 0.0f, 0.0f);
    }
}

// Reverse sweep of one list for the 4 pixels of a lane.  CHECK: pixel j only takes part in
// entries with index (k - begin) < nstop[j].
template <bool CHECK, bool ATOMIC>
__device__ inline void bwd_sweep(const WaveGeom &w, const float *__restrict__ stream_rec, const float pxf,
                                 const float (&pyf)[4], float (&T)[4], const float (&gr)[4], const float (&gg)[4],
                                 const float (&gb)[4], const uint32_t (&nstop)[4], float *__restrict__ pairgrad)
{
    float suffix[4] = {0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x & 63;
    auto entry = [&](const Rec &R, uint32_t k) {
        float *dst = pairgrad + (size_t)k * GSR_PAIRGRAD_FLOATS;
        if (rec_all_clamped_low(R)) {
            // alpha = 0.01 on every pixel: only the transmittance recurrence, the colour sums and the
            // suffix remain; the clamp blocks every geometric gradient
            const float om = 1.0f - kAlphaMin, rcp = 1.0f / (1.0f - kAlphaMin);
            float col[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float t0 = T[j] * rcp;
                float Tk = fmaf(fmaf(-om, t0, T[j]), rcp, t0);
                float wgt = kAlphaMin * Tk;
                const float cgd = fmaf(R_b(R), gb[j], fmaf(R_g(R), gg[j], R_r(R) * gr[j]));
                float suf = fmaf(wgt, cgd, suffix[j]);
                if (CHECK) {
                    const bool on = (k - w.begin) < nstop[j];
                    Tk = on ? Tk : T[j];
                    suf = on ? suf : suffix[j];
                    wgt = on ? wgt : 0.0f;
                }
                T[j] = Tk;
                suffix[j] = suf;
                col[0] = fmaf(wgt, gr[j], col[0]);
                col[1] = fmaf(wgt, gg[j], col[1]);
                col[2] = fmaf(wgt, gb[j], col[2]);
            }
            wave_sum_to_lane63(col);
            if (lane == 63) store_pairgrad<ATOMIC>(dst, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, col[0], col[1], col[2]);
            return;
        }
        const float dx = R_gx(R) - pxf;
        const float adx2 = R_A(R) * dx * dx;
        const float bdx = R_B(R) * dx;
        const float cadx = R_ca(R) * dx;   // with v = conic * d:  v_x = a dx + bh dy,  v_y = bh dx + d dy
        const float cbdx = R_cbh(R) * dx;
        float s_vx = 0.f, s_vy = 0.f, s_xx = 0.f, s_xy = 0.f, s_yy = 0.f, s_o = 0.f, s_r = 0.f, s_g = 0.f, s_b = 0.f;
        bool any_pass = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float dy = R_gy(R) - pyf[j];
            const float q = fmaf(fmaf(R_D(R), dy, bdx), dy, adx2);
            const float G = __builtin_amdgcn_exp2f(q);
            const float raw = R_o(R) * G;
            const float a = clamp_alpha(raw);
            // T_k = T_{k+1} / (1 - a), correctly rounded: quotient estimate, exact remainder (fma), one
            // correction.  A plain T * rcp(1 - a) is not good enough here: most entries of a long list
            // have the same a (0.01), so the reciprocal's rounding error repeats with the same sign
            // at every step and the transmittance drifts by ~1e-4 over a few thousand entries.
            const float om = 1.0f - a;
            const float rcp = __builtin_amdgcn_rcpf(om);
            const float t0 = T[j] * rcp;
            float Tk = fmaf(fmaf(-om, t0, T[j]), rcp, t0);   // transmittance in front of this entry
            float wgt = a * Tk;
            const float cgd = fmaf(R_b(R), gb[j], fmaf(R_g(R), gg[j], R_r(R) * gr[j]));
            float d_a = fmaf(Tk, cgd, -(suffix[j] * rcp));
            float suf = fmaf(wgt, cgd, suffix[j]);
            if (CHECK) {
                const bool on = (k - w.begin) < nstop[j];
                Tk = on ? Tk : T[j];
                suf = on ? suf : suffix[j];
                wgt = on ? wgt : 0.0f;
                d_a = on ? d_a : 0.0f;
            }
            T[j] = Tk;
            suffix[j] = suf;
            s_r = fmaf(wgt, gr[j], s_r);
            s_g = fmaf(wgt, gg[j], s_g);
            s_b = fmaf(wgt, gb[j], s_b);
            const bool pass = (a == raw);                  // clamp passes gradient on [0.01, 0.99]
            any_pass |= pass;
            const float d_raw = pass ? d_a : 0.0f;
            s_o = fmaf(d_raw, G, s_o);
            const float h = d_raw * a;                     // = -2 dL/d(exponent); a == raw when it matters
            // Sums are taken in the conic-weighted coordinates v: summing h*d and weighting by the
            // conic afterwards would cancel large coherent sums for elongated Gaussians.
            const float vx = fmaf(R_cbh(R), dy, cadx);
            const float vy = fmaf(R_cd(R), dy, cbdx);
            const float hvx = h * vx, hvy = h * vy;
            s_vx += hvx;
            s_vy += hvy;
            s_xx = fmaf(hvx, vx, s_xx);
            s_xy = fmaf(hvx, vy, s_xy);
            s_yy = fmaf(hvy, vy, s_yy);
        }
        const bool any_geom = __ballot(any_pass) != 0ull;
        if (any_geom) {
            float val[9] = {s_vx, s_vy, s_xx, s_xy, s_yy, s_o, s_r, s_g, s_b};
            wave_sum_to_lane63(val);
            if (lane == 63)
                store_pairgrad<ATOMIC>(dst, val[0], val[1], val[2], val[3], val[4], val[5], val[6], val[7], val[8]);
        } else {
            float col[3] = {s_r, s_g, s_b};
            wave_sum_to_lane63(col);
            if (lane == 63) store_pairgrad<ATOMIC>(dst, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, col[0], col[1], col[2]);
        }
    };
    // back to front, two record buffers alternating (see composite_fwd_kernel)
    uint32_t k = w.end - 1;   // the caller guarantees a non-empty list
    Rec A, B;
    rec_request(stream_rec, k, A);
    for (;;) {
        rec_wait(A);
        rec_request(stream_rec, k > w.begin ? k - 1 : k, B);
        entry(A, k);
        if (k == w.begin) break;
        --k;
        rec_wait(B);
        rec_request(stream_rec, k > w.begin ? k - 1 : k, A);
        entry(B, k);
        if (k == w.begin) break;
        --k;
    }
    rec_wait(A);
    rec_wait(B);
}

// ATOMIC = several waves share a tile (L > 16): their sums meet in pairgrad through float atomics
// (pairgrad zeroed by the launcher).  Otherwise each record is written once with plain stores.
template <bool ATOMIC>
__global__ __launch_bounds__(256) void composite_bwd_kernel(ViewK v, const uint32_t *__restrict__ tile_offsets,
                                                            const float *__restrict__ stream_rec,
                                                            const float *__restrict__ final_T,
                                                            const float *__restrict__ dimg,
                                                            float *__restrict__ pairgrad)
{
    WaveGeom w;
    if (!wave_setup(v, tile_offsets, w)) return;
    if (w.begin == w.end) return;
    const float pxf = (float)w.x;
    float pyf[4], T[4], gr[4], gg[4], gb[4];
    bool tiny = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int y = w.y0 + j;
        pyf[j] = (float)y;
        T[j] = 0.0f; gr[j] = gg[j] = gb[j] = 0.0f;   // a pixel outside the tile contributes exact zeros
        if (w.vx && y < w.ylim) {
            const size_t pix = (size_t)w.x * v.H + y;
            T[j] = final_T[pix];
            gr[j] = dimg[pix * 3 + 0]; gg[j] = dimg[pix * 3 + 1]; gb[j] = dimg[pix * 3 + 2];
            tiny |= !(T[j] >= kTinyT);
        }
    }
    uint32_t nstop[4] = {0u, 0u, 0u, 0u};
    if (__ballot(tiny) != 0ull) {
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
            const bool low = rec_all_clamped_low(R);
            const float dx = R_gx(R) - pxf;
            const float adx2 = R_A(R) * dx * dx;
            const float bdx = R_B(R) * dx;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dy = R_gy(R) - pyf[j];
                const float q = fmaf(fmaf(R_D(R), dy, bdx), dy, adx2);
                const float a = low ? kAlphaMin : clamp_alpha(R_o(R) * __builtin_amdgcn_exp2f(q));
                Tf[j] = Tf[j] * (1.0f - a);
                const bool hit = (Tf[j] < kTinyT) && (nstop[j] == len);
                nstop[j] = hit ? (k - w.begin + 1u) : nstop[j];
                Ts[j] = hit ? Tf[j] : Ts[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool inside = w.vx && (w.y0 + j) < w.ylim;
            if (nstop[j] != len) T[j] = inside ? Ts[j] : 0.0f;
        }
        bwd_sweep<true, ATOMIC>(w, stream_rec, pxf, pyf, T, gr, gg, gb, nstop, pairgrad);
    } else {
        bwd_sweep<false, ATOMIC>(w, stream_rec, pxf, pyf, T, gr, gg, gb, nstop, pairgrad);
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
                                 const float *final_T, const float *dimg, uint32_t num_pairs, float *pairgrad,
                                 void *stream)
{
    ViewK v;
    if (int e = gsr_make_view(view_host, &v)) return e;
    if (num_pairs == 0) return GSR_OK;
    if (!tile_offsets || !stream_rec || !final_T || !dimg || !pairgrad) {
        gsr_set_error("gsr_composite_bwd: null pointer");
        return GSR_ERR_ARG;
    }
    const int S = (v.L + 15) >> 4;
    const unsigned waves = (unsigned)v.T * S * S;
    const dim3 grid(gsr_div_up(waves, WAVES_PER_BLOCK)), block(64 * WAVES_PER_BLOCK);
    if (S > 1) {
        (void)hipMemsetAsync(pairgrad, 0, (size_t)num_pairs * GSR_PAIRGRAD_FLOATS * sizeof(float), (hipStream_t)stream);
        hipLaunchKernelGGL(composite_bwd_kernel<true>, grid, block, 0, (hipStream_t)stream, v, tile_offsets,
                           stream_rec, final_T, dimg, pairgrad);
    } else {
        hipLaunchKernelGGL(composite_bwd_kernel<false>, grid, block, 0, (hipStream_t)stream, v, tile_offsets,
                           stream_rec, final_T, dimg, pairgrad);
    }
    return gsr_check_launch("composite_bwd");
}
