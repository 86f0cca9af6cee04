// Stage 1 of the splat path: one thread per Gaussian.
//
// Forward: activations, 3-D covariance, projection + cull, 2-D covariance and its inverse,
// SH colour, radius, integer bounding box, tile count; writes the 48-byte splat record the
// compositing kernels stream.  Backward: sums a Gaussian's per-tile gradient records and chains
// them to the raw parameters.
//
// The discrete outputs (visibility mask, depth key, bounding box) must equal the reference's bit
// for bit, so every fp32 operation feeding them is written out one IEEE operation at a time in
// the order the reference's tensor ops apply them:
//   [N,4]@[4,4] and [2M,3]@[3,3] products (sgemm)  : acc = a0*b0; acc = fma(ak, bk, acc)
//   3x3 / 2x3 batched products (small-matrix loop)  : acc = 0;     acc = acc + ak*bk
//   F.normalize                                     : sum of squares left to right
//   exp of the log-scales                           : correctly rounded, (float)exp((double)x)
// This file is compiled with -ffp-contract=off; fused operations appear only as __fmaf_rn.
#include "gsr_common.hpp"

namespace {

__device__ __constant__ const float kC0 = 0.28209479177387814f;
__device__ __constant__ const float kC1 = 0.4886025119029199f;
__device__ __constant__ const float kC2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                              -1.0925484305920792f, 0.5462742152960396f};
__device__ __constant__ const float kC3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                              0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                              -0.5900435899266435f};

struct Geom {
    float s[3];
    float qn;
    float q[4];
    float R[3][3];
    float RS[3][3];
    float cov3[3][3];
    float c[4];     // clip-space position
    float ndc[3];
    float M[2][3];  // J @ W
    float cov2[4];
};

__device__ inline float dot4_fma(const float *v, const float *row)
{
    float acc = v[0] * row[0];
    acc = __fmaf_rn(v[1], row[1], acc);
    acc = __fmaf_rn(v[2], row[2], acc);
    acc = __fmaf_rn(v[3], row[3], acc);
    return acc;
}

// View-independent part: activations, rotation, 3-D covariance
// (GaussianModel.py:50-56,99-108; Utils/Transform.py:55-63,95-98).
__device__ inline void geometry_model(const float sr[3], const float qr[4], Geom &g)
{
#pragma unroll
    for (int k = 0; k < 3; ++k) g.s[k] = (float)exp((double)sr[k]);
    float ss = qr[0] * qr[0];
    ss = ss + qr[1] * qr[1];
    ss = ss + qr[2] * qr[2];
    ss = ss + qr[3] * qr[3];
    float nrm = sqrtf(ss);
    g.qn = nrm > 1e-12f ? nrm : 1e-12f;
#pragma unroll
    for (int k = 0; k < 4; ++k) g.q[k] = qr[k] / g.qn;
    const float w = g.q[0], x = g.q[1], y = g.q[2], z = g.q[3];
    g.R[0][0] = 2.0f * (w * w + x * x) - 1.0f; g.R[0][1] = 2.0f * (x * y - w * z); g.R[0][2] = 2.0f * (x * z + w * y);
    g.R[1][0] = 2.0f * (x * y + w * z); g.R[1][1] = 2.0f * (w * w + y * y) - 1.0f; g.R[1][2] = 2.0f * (y * z - w * x);
    g.R[2][0] = 2.0f * (x * z - w * y); g.R[2][1] = 2.0f * (y * z + w * x); g.R[2][2] = 2.0f * (w * w + z * z) - 1.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) g.RS[i][j] = g.R[i][j] * g.s[j];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) acc = acc + g.RS[i][k] * g.RS[j][k];
            g.cov3[i][j] = acc;
        }
}

// View-dependent part: clip position, NDC, M = J W, 2-D covariance (Utils/Camera.py:81-87,101-103).
// w2m and proj are the view's row-major 4x4 matrices.
__device__ inline void geometry_view(const float *w2m, const float *proj, const float p[3], Geom &g)
{
    const float ph[4] = {p[0], p[1], p[2], 1.0f};
    float m[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) m[j] = dot4_fma(ph, &w2m[4 * j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) g.c[j] = dot4_fma(m, &proj[4 * j]);
#pragma unroll
    for (int j = 0; j < 3; ++j) g.ndc[j] = g.c[j] / g.c[3];
    // J is built from the WORLD position; J01 = J10 = 0.
    float J[2][3];
    J[0][0] = 1.0f / p[2]; J[0][1] = 0.0f; J[0][2] = (-p[0]) / p[2];
    J[1][0] = 0.0f; J[1][1] = 1.0f / p[2]; J[1][2] = (-p[1]) / p[2];
    float t2[2][3], t3[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float acc = J[i][0] * w2m[4 * j + 0];
            acc = __fmaf_rn(J[i][1], w2m[4 * j + 1], acc);
            acc = __fmaf_rn(J[i][2], w2m[4 * j + 2], acc);
            g.M[i][j] = acc;
        }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) acc = acc + g.M[i][k] * g.cov3[k][j];
            t2[i][j] = acc;
        }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float acc = t2[i][0] * w2m[4 * 0 + j];
            acc = __fmaf_rn(t2[i][1], w2m[4 * 1 + j], acc);
            acc = __fmaf_rn(t2[i][2], w2m[4 * 2 + j], acc);
            t3[i][j] = acc;
        }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) acc = acc + t3[i][k] * J[j][k];
            g.cov2[2 * i + j] = acc;
        }
}

__device__ inline void geometry(const ViewK &v, const float p[3], const float sr[3], const float qr[4], Geom &g)
{
    geometry_model(sr, qr, g);
    geometry_view(v.w2m, v.proj, p, g);
}

// SH basis of an un-normalised direction (Utils/SphericalHarmonic.py:62-88), degree <= 3.
__device__ inline void sh_basis(int deg, float x, float y, float z, float b[16])
{
    b[0] = kC0;
    if (deg < 1) return;
    b[1] = -kC1 * y; b[2] = kC1 * z; b[3] = -kC1 * x;
    if (deg < 2) return;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    b[4] = kC2[0] * xy; b[5] = kC2[1] * yz; b[6] = kC2[2] * (2.0f * zz - xx - yy);
    b[7] = kC2[3] * xz; b[8] = kC2[4] * (xx - yy);
    if (deg < 3) return;
    b[9] = kC3[0] * y * (3.0f * xx - yy);
    b[10] = kC3[1] * xy * z;
    b[11] = kC3[2] * y * (4.0f * zz - xx - yy);
    b[12] = kC3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
    b[13] = kC3[4] * x * (4.0f * zz - xx - yy);
    b[14] = kC3[5] * z * (xx - yy);
    b[15] = kC3[6] * x * (xx - 3.0f * yy);
}

// d(sum_k co[k] basis_k)/d(x,y,z)
__device__ inline void sh_dir_grad(int deg, float x, float y, float z, const float co[16], float d[3])
{
    // (only the backward uses this: nothing discrete depends on its roundings, so multiply-adds may fuse here although the
    // file is compiled with contraction off for the forward's sake)
#pragma clang fp contract(fast)
    d[0] = d[1] = d[2] = 0.0f;
    if (deg < 1) return;
    d[1] += -kC1 * co[1]; d[2] += kC1 * co[2]; d[0] += -kC1 * co[3];
    if (deg < 2) return;
    const float xx = x * x, yy = y * y, zz = z * z;
    d[0] += kC2[0] * y * co[4] + kC2[2] * -2.0f * x * co[6] + kC2[3] * z * co[7] + kC2[4] * 2.0f * x * co[8];
    d[1] += kC2[0] * x * co[4] + kC2[1] * z * co[5] + kC2[2] * -2.0f * y * co[6] + kC2[4] * -2.0f * y * co[8];
    d[2] += kC2[1] * y * co[5] + kC2[2] * 4.0f * z * co[6] + kC2[3] * x * co[7];
    if (deg < 3) return;
    d[0] += kC3[0] * 6.0f * x * y * co[9] + kC3[1] * y * z * co[10] + kC3[2] * -2.0f * x * y * co[11] +
            kC3[3] * -6.0f * x * z * co[12] + kC3[4] * (4.0f * zz - 3.0f * xx - yy) * co[13] +
            kC3[5] * 2.0f * x * z * co[14] + kC3[6] * (3.0f * xx - 3.0f * yy) * co[15];
    d[1] += kC3[0] * (3.0f * xx - 3.0f * yy) * co[9] + kC3[1] * x * z * co[10] +
            kC3[2] * (4.0f * zz - xx - 3.0f * yy) * co[11] + kC3[3] * -6.0f * y * z * co[12] +
            kC3[4] * -2.0f * x * y * co[13] + kC3[5] * -2.0f * y * z * co[14] + kC3[6] * -6.0f * x * y * co[15];
    d[2] += kC3[1] * x * y * co[10] + kC3[2] * 8.0f * y * z * co[11] +
            kC3[3] * (6.0f * zz - 3.0f * xx - 3.0f * yy) * co[12] + kC3[4] * 8.0f * x * z * co[13] +
            kC3[5] * (xx - yy) * co[14];
}

// Zero-pivot test of a 2x2 LU with partial pivoting: the condition under which the reference's
// torch.linalg.inv raises (Utils/Probability.py:15).
__device__ inline bool lu2_singular(const float c[4])
{
    const float a = c[0], b = c[1], cc = c[2], d = c[3];
    if (fabsf(a) >= fabsf(cc)) {
        if (a == 0.0f) return true;
        return (d - (cc / a) * b) == 0.0f;
    }
    return (b - (a / cc) * d) == 0.0f;
}

__device__ inline void load_sh(const float *sh, int i, int K, float co[3][16])
{
    const float4 *src = reinterpret_cast<const float4 *>(sh + (size_t)i * 48);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int v4 = 0; v4 < 4; ++v4) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (v4 * 4 < K) t = src[c * 4 + v4];
            co[c][v4 * 4 + 0] = t.x; co[c][v4 * 4 + 1] = t.y; co[c][v4 * 4 + 2] = t.z; co[c][v4 * 4 + 3] = t.w;
        }
}

// Where stage 1 leaves what it computes for one view (every array is indexed by Gaussian).
// Sum of one value per lane over the wave, all 64 lanes active (every lane gets it): four adds with lane swizzles
// inside the rows of 16 (neighbour, pair, mirrored half row, mirrored row), then the four row sums meet on the scalar unit.
template <int CTRL>
__device__ inline uint32_t dpp_add_u32(uint32_t x)
{
    return x + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, false);
}
__device__ inline uint32_t wave_sum_u32(uint32_t x)
{
    x = dpp_add_u32<0x140>(dpp_add_u32<0x141>(dpp_add_u32<0x4E>(dpp_add_u32<0xB1>(x))));
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 0) + (uint32_t)__builtin_amdgcn_readlane((int)x, 16) +
           (uint32_t)__builtin_amdgcn_readlane((int)x, 32) + (uint32_t)__builtin_amdgcn_readlane((int)x, 48);
}

struct FwdOut {
    uint8_t *mask;
    uint32_t *depth_key;
    float *pos2d, *cov2d, *radius;
    int32_t *bbox;
    uint32_t *ntiles;
    float *rec;
};

// Gaussian i in view v, from the view-independent part in g (geometry_model): projection and cull, 2-D
// covariance, radius, integer bounding box, tile count, colour, splat record.  `co`: the Gaussian's sh
// coefficients, loaded on the first view that sees it (`co_loaded`).  Returns whether the Gaussian is visible.
__device__ inline bool project_to_view(const ViewK &v, int i, const float p[3], Geom &g, const float *__restrict__ opac_raw,
                                       const float *__restrict__ sh, const float *__restrict__ screen_coords, float (&co)[3][16],
                                       bool &co_loaded, const FwdOut &out, bool &singular, uint32_t &tiles)
{
    geometry_view(v.w2m, v.proj, p, g);
    // Utils/Camera.py:106-108 (NaN compares false: culled)
    const bool vis = (g.ndc[2] >= 0.1f) && (-1.3f <= g.ndc[0]) && (g.ndc[0] <= 1.3f) && (-1.3f <= g.ndc[1]) &&
                     (g.ndc[1] <= 1.3f);
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
    float px = 0.f, py = 0.f, pz = 0.f, rad = 0.f;
    float cv[4] = {0.f, 0.f, 0.f, 0.f};
    int bx[4] = {0, 0, 0, 0};
    uint32_t nt = 0, key = 0xFFFFFFFFu;
    if (vis) {
        float sx = 0.f, sy = 0.f, sz = 0.f;
        if (screen_coords) { sx = screen_coords[3 * i]; sy = screen_coords[3 * i + 1]; sz = screen_coords[3 * i + 2]; }
        // Utils/Camera.py:111-115
        px = sx + g.ndc[0]; py = sy + g.ndc[1]; pz = sz + g.ndc[2];
        px = (0.5f * (px + 1.0f)) * v.Wf;
        py = (1.0f - 0.5f * (py + 1.0f)) * v.Hf;
        key = __float_as_uint(pz);
#pragma unroll
        for (int k = 0; k < 4; ++k) cv[k] = g.cov2[k];
        // GaussianRenderer.py:40-44
        const float a = cv[0], b = cv[1], c = cv[2], d = cv[3];
        const float trace = a + d;
        const float det = a * d - b * c;
        const float term1 = 0.5f * trace;
        float disc = trace * trace - 4.0f * det;
        if (disc < 0.0f) disc = 0.0f;
        const float term2 = 0.5f * sqrtf(disc);
        const float lo = term1 - term2, hi = term1 + term2;
        rad = 3.0f * sqrtf(lo > hi ? lo : hi);
        // Utils/Screen.py:33-36: clamp, then truncate; a NaN edge gives an empty box
        const float e[4] = {px - rad, py - rad, px + rad, py + rad};
        const float lim[4] = {v.Wf, v.Hf, v.Wf, v.Hf};
        bool bad = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float t = e[k];
            if (t != t) { bad = true; t = 0.0f; }
            if (t < 0.0f) t = 0.0f;
            if (t > lim[k]) t = lim[k];
            bx[k] = (int)t;
        }
        if (bad) bx[0] = bx[1] = bx[2] = bx[3] = 0;
        int x0, x1, y0, y1;
        gsr_tile_span(bx[0], bx[2], v.L, x0, x1);
        gsr_tile_span(bx[1], bx[3], v.L, y0, y1);
        nt = (uint32_t)(x1 - x0 + 1) * (uint32_t)(y1 - y0 + 1);
        singular = singular || (nt > 0 && lu2_singular(cv));
        // inverse 2x2, pre-scaled so that G = exp2(A' dx^2 + B' dx dy + D' dy^2)
        const float idet = 1.0f / det;
        const float kk = -0.5f * 1.4426950408889634f;
        const float ia = d * idet, ib = -b * idet, ic = -c * idet, id = a * idet;
        // GaussianModel.py:58-60
        const float o = 1.0f / (1.0f + expf(-opac_raw[i]));
        // GaussianRenderer.py:71-74
        float bs[16];
        if (!co_loaded) { load_sh(sh, i, v.K, co); co_loaded = true; }
        sh_basis(v.deg, p[0] - v.cam[0], p[1] - v.cam[1], p[2] - v.cam[2], bs);
        float col[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (k < v.K) acc += bs[k] * co[ch][k];
            acc += 0.5f;
            col[ch] = acc < 0.0f ? 0.0f : acc;
        }
        // A NaN opacity makes alpha NaN in the reference (torch.clamp keeps NaN, GaussianRenderer.py:104)
        // and with it every pixel of the Gaussian's tiles.  The compositing kernels clamp with min/max
        // instructions, which drop a NaN operand, so the NaN is handed on through the colour instead:
        // the image comes out NaN in the same pixels and a diverged model stays visible in the loss.
        if (o != o) col[0] = col[1] = col[2] = o;
        r0 = make_float4(px, py, kk * ia, kk * (ib + ic));
        r1 = make_float4(kk * id, o, col[0], col[1]);
        r2 = make_float4(col[2], ia, 0.5f * (ib + ic), id);
    }
    out.mask[i] = vis ? 1 : 0;
    out.depth_key[i] = key;
    // Nothing downstream reads these two: they exist for the caller (uniform branches, a kernel argument).
    if (out.pos2d) { out.pos2d[3 * i] = px; out.pos2d[3 * i + 1] = py; out.pos2d[3 * i + 2] = pz; }
    if (out.cov2d) reinterpret_cast<float4 *>(out.cov2d)[i] = make_float4(cv[0], cv[1], cv[2], cv[3]);
    out.radius[i] = rad;
    reinterpret_cast<int4 *>(out.bbox)[i] = make_int4(bx[0], bx[1], bx[2], bx[3]);
    out.ntiles[i] = nt;
    tiles = nt;
    float4 *dst = reinterpret_cast<float4 *>(out.rec + (size_t)i * GSR_REC_FLOATS);
    dst[0] = r0; dst[1] = r1; dst[2] = r2;
    return vis;
}

__global__ __launch_bounds__(256) void preprocess_fwd_kernel(
    ViewK v, const float *__restrict__ coords, const float *__restrict__ scale_raw,
    const float *__restrict__ rot_raw, const float *__restrict__ opac_raw, const float *__restrict__ sh,
    const float *__restrict__ screen_coords, FwdOut out, gsr_counters *__restrict__ counters)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool vis = false;
    bool singular = false;
    uint32_t tiles = 0;
    if (i < v.n) {
        const float p[3] = {coords[3 * i], coords[3 * i + 1], coords[3 * i + 2]};
        const float sr[3] = {scale_raw[3 * i], scale_raw[3 * i + 1], scale_raw[3 * i + 2]};
        const float4 q4 = reinterpret_cast<const float4 *>(rot_raw)[i];
        const float qr[4] = {q4.x, q4.y, q4.z, q4.w};
        Geom g;
        geometry_model(sr, qr, g);
        float co[3][16];
        bool co_loaded = false;
        vis = project_to_view(v, i, p, g, opac_raw, sh, screen_coords, co, co_loaded, out, singular, tiles);
    }
    // One atomic per BLOCK for the visible count: atomics of all waves on one address serialise at
    // about 7 ns each, which at one per wave (15.6 k waves for a million Gaussians) costs more than the
    // rest of the kernel.
    // The same for the number of (tile, Gaussian) pairs of the view, the sum of the tile counts: it does not
    // depend on the depth order, so the host can read it - and size the per-pair buffers - while the sort runs.
    __shared__ uint32_t wave_visible[4], wave_tiles[4];
    const unsigned long long bal = __ballot(vis);
    const uint32_t wave_sum = wave_sum_u32(tiles);      // (at most 64 x 2^13 per wave)
    if ((threadIdx.x & 63) == 0) {
        wave_visible[threadIdx.x >> 6] = (uint32_t)__popcll(bal);
        wave_tiles[threadIdx.x >> 6] = wave_sum;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t c = wave_visible[0] + wave_visible[1] + wave_visible[2] + wave_visible[3];
        if (c) atomicAdd(&counters->num_visible, c);
        const uint32_t t = wave_tiles[0] + wave_tiles[1] + wave_tiles[2] + wave_tiles[3];
        if (t) atomicAdd(reinterpret_cast<unsigned long long *>(counters->pairs_wide), (unsigned long long)t);
    }
    if (singular) atomicOr(&counters->error, 1u);
}

// Stage 1 of several views of the same Gaussians in one pass: the parameters are read once (236 bytes per
// Gaussian, most of what the one-view kernel moves) and the view-independent part - activations, rotation,
// 3-D covariance - is computed once.  The outputs of view k start k * n elements (rows) into each array.
constexpr int kMaxFwdViews = 8;
struct FwdViews {
    ViewK v[kMaxFwdViews];
    int nv;
};

constexpr int kFwdViewsThreads = 512;
__global__ __launch_bounds__(kFwdViewsThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void preprocess_fwd_views_kernel(
    FwdViews views, const float *__restrict__ coords, const float *__restrict__ scale_raw,
    const float *__restrict__ rot_raw, const float *__restrict__ opac_raw, const float *__restrict__ sh, FwdOut out0,
    gsr_counters *__restrict__ counters)
{
    const int n = views.v[0].n;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ uint32_t visible[kMaxFwdViews], tiles_of[kMaxFwdViews];
    if (threadIdx.x < kMaxFwdViews) visible[threadIdx.x] = tiles_of[threadIdx.x] = 0;
    __syncthreads();
    float p[3] = {0.f, 0.f, 0.f};
    Geom g;
    float co[3][16];
    bool co_loaded = false;
    if (i < n) {
        p[0] = coords[3 * i]; p[1] = coords[3 * i + 1]; p[2] = coords[3 * i + 2];
        const float sr[3] = {scale_raw[3 * i], scale_raw[3 * i + 1], scale_raw[3 * i + 2]};
        const float4 q4 = reinterpret_cast<const float4 *>(rot_raw)[i];
        const float qr[4] = {q4.x, q4.y, q4.z, q4.w};
        geometry_model(sr, qr, g);
    }
    for (int k = 0; k < views.nv; ++k) {
        bool vis = false, singular = false;
        uint32_t tiles = 0;
        if (i < n) {
            const size_t at = (size_t)k * (size_t)n;
            const FwdOut out = {out0.mask + at, out0.depth_key + at, out0.pos2d ? out0.pos2d + 3 * at : nullptr,
                                out0.cov2d ? out0.cov2d + 4 * at : nullptr, out0.radius + at,
                                out0.bbox + 4 * at, out0.ntiles + at, out0.rec + GSR_REC_FLOATS * at};
            vis = project_to_view(views.v[k], i, p, g, opac_raw, sh, nullptr, co, co_loaded, out, singular, tiles);
        }
        const unsigned long long bal = __ballot(vis);
        const uint32_t wave_sum = wave_sum_u32(tiles);
        if ((threadIdx.x & 63) == 0 && bal) {
            atomicAdd(&visible[k], (uint32_t)__popcll(bal));
            atomicAdd(&tiles_of[k], wave_sum);
        }
        if (singular) atomicOr(&counters[k].error, 1u);
    }
    __syncthreads();
    if (threadIdx.x < views.nv && visible[threadIdx.x]) {
        atomicAdd(&counters[threadIdx.x].num_visible, visible[threadIdx.x]);
        if (tiles_of[threadIdx.x])
            atomicAdd(reinterpret_cast<unsigned long long *>(counters[threadIdx.x].pairs_wide), (unsigned long long)tiles_of[threadIdx.x]);
    }
}

template <bool ACC>
__device__ inline void put(float *dst, float v)
{
    if (ACC) *dst += v; else *dst = v;
}

// Sum of each Gaussian's per-tile gradient records -> gsum[N,12] (9 used; the last four floats of a
// record are partial sums of one quantity).  The records of
// Gaussian g are the contiguous slots slot_base[g] .. slot_base[g]+ntiles[g] of pairgrad.  One
// thread per Gaussian walks its slots in order; a Gaussian with many tiles is summed by the whole
// wave (strided partial sums, then a fixed butterfly), so one huge splat cannot serialise thousands
// of dependent loads in a single lane.  Both orders are fixed: the result is reproducible.
constexpr uint32_t kHeavyTiles = 48;   // (96: 0.079 ms, 32: 0.077, 48: 0.076 ms per view with the two-record turns below)

__device__ inline void add_values(double (&acc)[9], const float4 a, const float4 b, const float4 c)
{
    acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
    acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
    acc[8] += ((double)c.x + (double)c.y) + ((double)c.z + (double)c.w);   // blue arrives as four partial sums
}

__device__ inline void add_record(double (&acc)[9], const float *__restrict__ pairgrad, size_t slot)
{
    const float4 *src = reinterpret_cast<const float4 *>(pairgrad + (size_t)slot * GSR_PAIRGRAD_FLOATS);
    add_values(acc, src[0], src[1], src[2]);
}

// Four lanes per Gaussian: lane q of the four takes the records q, q+4, ... of its Gaussian (neighbouring
// lanes read neighbouring 48-byte records) and the four partial sums meet in two shuffle steps.  With
// one lane per Gaussian every lane walks its own records one after the other and the kernel waits on
// memory latency: a Gaussian has five records on average.
// (Walking the Gaussians in depth order, in which their records are consecutive in memory, was
// measured slower than index order: the scattered 48-byte result rows cost more than the reads gain.)
constexpr int kLanesPerGaussian = 4;   // (8 measured slower: 0.088 ms against 0.078 ms; 2 the same as 4)
// parts: rows of pairgrad per (tile, Gaussian) pair (gsr_pairgrad_parts): the rows of a Gaussian's pairs are
// contiguous, parts * ntiles of them from row parts * slot_base, and are added in row order.
__global__ __launch_bounds__(256) void pair_reduce_kernel(int n, uint32_t parts, const uint8_t *__restrict__ mask,
                                                          const uint32_t *__restrict__ slot_base,
                                                          const uint32_t *__restrict__ ntiles,
                                                          const float *__restrict__ pairgrad,
                                                          float *__restrict__ gsum)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = t / kLanesPerGaussian, q = t % kLanesPerGaussian;
    const int lane = threadIdx.x & 63;
    const bool live = i < n && mask[i];
    const uint32_t nt = live ? ntiles[i] * parts : 0u;
    const size_t e0 = live ? (size_t)slot_base[i] * parts : 0u;   // (rows: pairs x parts may pass 2^32)
    double acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.0;
    const bool heavy = nt >= kHeavyTiles;
    if (!heavy) {
        constexpr int kTurn = 2;
        uint32_t j = (uint32_t)q;
        // kTurn records per turn, all their loads issued before the first add (the sums keep their order): 0.085 ->
        // 0.077 ms per view; three or four per turn are slower again (0.100 ms: registers)
        for (; j + (kTurn - 1) * kLanesPerGaussian < nt; j += kTurn * kLanesPerGaussian) {
            float4 a[kTurn], b[kTurn], c[kTurn];
#pragma unroll
            for (int u = 0; u < kTurn; ++u) {
                const float4 *src = reinterpret_cast<const float4 *>(pairgrad + (e0 + j + u * kLanesPerGaussian) * GSR_PAIRGRAD_FLOATS);
                a[u] = src[0]; b[u] = src[1]; c[u] = src[2];
            }
#pragma unroll
            for (int u = 0; u < kTurn; ++u) add_values(acc, a[u], b[u], c[u]);
        }
        for (; j < nt; j += kLanesPerGaussian) add_record(acc, pairgrad, e0 + j);
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int d = 1; d < kLanesPerGaussian; d <<= 1) acc[k] += __shfl_xor(acc[k], d, 64);
    }
    // a Gaussian with very many tiles is summed by the whole wave (once: its first lane asks for it)
    unsigned long long todo = __ballot(heavy && q == 0);
    while (todo) {
        const int src_lane = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const size_t E0 = ((size_t)__shfl((uint32_t)(e0 >> 32), src_lane, 64) << 32) | __shfl((uint32_t)e0, src_lane, 64);
        const uint32_t NT = __shfl(nt, src_lane, 64);
        double part[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) part[k] = 0.0;
        for (uint32_t j = lane; j < NT; j += 64) add_record(part, pairgrad, E0 + j);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) part[k] += __shfl_xor(part[k], d, 64);
            if (lane == src_lane) acc[k] = part[k];
        }
    }
    if (i < n && q == 0) {
        float4 *dst = reinterpret_cast<float4 *>(gsum + (size_t)i * 12);
        dst[0] = make_float4((float)acc[0], (float)acc[1], (float)acc[2], (float)acc[3]);
        dst[1] = make_float4((float)acc[4], (float)acc[5], (float)acc[6], (float)acc[7]);
        dst[2] = make_float4((float)acc[8], 0.f, 0.f, 0.f);
    }
}

// Launch constants of one view for the backward of stage 1.
struct ViewCam {
    float Wf, Hf;
    float w2m[16];
    float proj[16];
    float cam[3];
};
constexpr int kMaxViews = GSR_MAX_VIEWS;
struct BwdViews {
    int nv;
    ViewCam cam[kMaxViews];
    const uint8_t *mask[kMaxViews];   // visibility of the view
    const float *gsum[kMaxViews];     // summed gradient records of the view, [N,12]
    float *g_screen[kMaxViews];       // gradient of the view's screen_coords leaf, [N,3]
    float *g_color[kMaxViews];        // (may be null) dL/d(colour) of the view as it reaches the clamp, [N,3]
};

// Backward of stage 1 for up to kMaxViews views of ONE parameter set: the gradients of all views are
// summed in registers and the dense parameter gradients (236 bytes per Gaussian) are written once.
// A view minibatch rendered one view at a time re-reads and re-writes them for every view.
// Everything that does not depend on the view - the chain from the 3-D covariance back to the raw
// scales and the quaternion - runs once on the summed dL/d(cov3).
template <bool ACC>
__global__ __launch_bounds__(256) void preprocess_bwd_kernel(
    BwdViews a, int n, int deg, int K, const float *__restrict__ coords, const float *__restrict__ scale_raw,
    const float *__restrict__ rot_raw, const float *__restrict__ opac_raw, const float *__restrict__ sh,
    float *__restrict__ g_coords, float *__restrict__ g_scale, float *__restrict__ g_rot, float *__restrict__ g_opac,
    float *__restrict__ g_sh)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float gc[3] = {0.f, 0.f, 0.f}, gs[3] = {0.f, 0.f, 0.f}, gr[4] = {0.f, 0.f, 0.f, 0.f};
    float go = 0.f;
    // g_sh == null: the sh gradient is left to gsr_sh_grad_views (it is a sum over the views of outer
    // products basis x dL/d(colour), which ranks that share the parameters can rebuild from the 3 colour
    // gradients per view instead of exchanging 48 floats)
    const bool write_sh = g_sh != nullptr;
    float4 *dsh = reinterpret_cast<float4 *>(g_sh + (size_t)i * 48);

    // The thread's loads are issued ahead of their use wherever the data allows it - the visibility bytes of all views
    // at once, the sums of view v + 1 while view v is chained, the colour gradient of view v + 1 while view v's basis is
    // evaluated: with one dependent load after the other (mask, then sums, then arithmetic, view by view, three more
    // times for the colours) the kernel waited out some thirty memory latencies per Gaussian at three waves per SIMD.
    uint32_t seen = 0;
#pragma unroll
    for (int vv = 0; vv < kMaxViews; ++vv)
        if (vv < a.nv) seen |= (a.mask[vv][i] != 0 ? 1u : 0u) << vv;
    if (!seen) {
        // Culled in every view: a zero cotangent, but the reference's autograd still divides it by the
        // clip-space w and multiplies it by the clip coordinates (the backward of Utils/Camera.py:103), so
        // coordinates that are not finite - which is why such a Gaussian is culled - give a NaN coordinate
        // gradient there, and here.
        {
            const float c0 = coords[3 * i], c1 = coords[3 * i + 1], c2 = coords[3 * i + 2];
            if (!(__builtin_isfinite(c0) && __builtin_isfinite(c1) && __builtin_isfinite(c2)))
                gc[0] = gc[1] = gc[2] = __builtin_nanf("");
        }
        for (int vv = 0; vv < a.nv; ++vv) {
            float *gsv = a.g_screen[vv] + 3 * (size_t)i;
            gsv[0] = 0.0f; gsv[1] = 0.0f; gsv[2] = 0.0f;
            if (a.g_color[vv]) {
                float *gcv = a.g_color[vv] + 3 * (size_t)i;
                gcv[0] = 0.0f; gcv[1] = 0.0f; gcv[2] = 0.0f;
            }
        }
        if (!ACC && write_sh) {
#pragma unroll
            for (int k = 0; k < 12; ++k) dsh[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else {
        const float p[3] = {coords[3 * i], coords[3 * i + 1], coords[3 * i + 2]};
        const float sr[3] = {scale_raw[3 * i], scale_raw[3 * i + 1], scale_raw[3 * i + 2]};
        const float4 q4 = reinterpret_cast<const float4 *>(rot_raw)[i];
        const float qr[4] = {q4.x, q4.y, q4.z, q4.w};
        Geom g;
        geometry_model(sr, qr, g);
        double gp[3] = {0.0, 0.0, 0.0};
        double dS[3][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};   // dL/d(cov3), summed over the views
        double d_o = 0.0;

        float4 n0 = make_float4(0.f, 0.f, 0.f, 0.f), n1 = n0;      // the sums of the next view
        auto fetch_sums = [&](int vv) {
            if (vv < a.nv && ((seen >> vv) & 1u)) {
                const float4 *gsrc = reinterpret_cast<const float4 *>(a.gsum[vv] + (size_t)i * 12);
                n0 = gsrc[0]; n1 = gsrc[1];
            }
        };
        fetch_sums(0);
        for (int vv = 0; vv < a.nv; ++vv) {
            float gsc[2] = {0.f, 0.f};
            const float4 s0 = n0, s1 = n1;
            fetch_sums(vv + 1);
            if ((seen >> vv) & 1u) {
                // (the chain below is the backward's own arithmetic: multiply-adds may fuse)
#pragma clang fp contract(fast)
                const ViewCam &cam = a.cam[vv];
                // With h = -2 dL/d(exponent) per pixel, d = centre - pixel, C = inv(cov2) (symmetric part) and
                // v = C d, the summed records hold: sum h v_x, sum h v_y, sum h v_x^2, sum h v_x v_y,
                // sum h v_y^2, dL/d(opacity), dL/d(colour).  exponent = d^T C d, so
                //   dL/d(centre) = sum (-h/2) (2 v) = -sum h v,   dL/d(cov2) = sum (-h/2) (-v v^T) = 1/2 sum h v v^T.
                const double d_px = -(double)s0.x, d_py = -(double)s0.y;
                const double D[2][2] = {{0.5 * (double)s0.z, 0.5 * (double)s0.w}, {0.5 * (double)s0.w, 0.5 * (double)s1.x}};
                d_o += (double)s1.y;
                geometry_view(cam.w2m, cam.proj, p, g);
                // pixel centre -> NDC (+ screen_coords) -> clip -> world  (Utils/Camera.py:102-115)
                {
                    const double dn[3] = {d_px * 0.5 * (double)cam.Wf, d_py * -0.5 * (double)cam.Hf, 0.0};
                    gsc[0] = (float)dn[0]; gsc[1] = (float)dn[1];
                    const double c3 = (double)g.c[3];
                    double dc[4];
                    dc[3] = 0.0;
#pragma unroll
                    for (int j = 0; j < 3; ++j) { dc[j] = dn[j] / c3; dc[3] -= dn[j] * (double)g.c[j] / (c3 * c3); }
                    double dm[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        dm[k] = 0.0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) dm[k] += dc[j] * (double)cam.proj[4 * j + k];
                    }
#pragma unroll
                    for (int k = 0; k < 3; ++k)
#pragma unroll
                        for (int j = 0; j < 4; ++j) gp[k] += dm[j] * (double)cam.w2m[4 * j + k];
                }
                // cov2 = M S M^T with M = J W
                double dM[2][3];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        double t = 0.0;
#pragma unroll
                        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                            for (int c = 0; c < 2; ++c) t += (double)g.M[rr][r] * D[rr][c] * (double)g.M[c][b];
                        dS[r][b] += t;
                    }
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        double t = 0.0;
#pragma unroll
                        for (int c = 0; c < 2; ++c)
#pragma unroll
                            for (int b = 0; b < 3; ++b)
                                t += D[r][c] * (double)g.M[c][b] * (double)g.cov3[k][b] +
                                     D[c][r] * (double)g.M[c][b] * (double)g.cov3[b][k];
                        dM[r][k] = t;
                    }
                {
                    // M = J W, W[k][j] = w2m[j][k]; J00 = J11 = 1/z, J02 = -x/z, J12 = -y/z (world x,y,z)
                    double dJ[2][3];
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            dJ[r][k] = 0.0;
#pragma unroll
                            for (int j = 0; j < 3; ++j) dJ[r][k] += dM[r][j] * (double)cam.w2m[4 * j + k];
                        }
                    const double x = p[0], y = p[1], z = p[2];
                    gp[0] += dJ[0][2] * (-1.0 / z);
                    gp[1] += dJ[1][2] * (-1.0 / z);
                    gp[2] += (dJ[0][0] + dJ[1][1]) * (-1.0 / (z * z)) + dJ[0][2] * (x / (z * z)) + dJ[1][2] * (y / (z * z));
                }
            }
            // screen_coords is a fresh leaf of every render (GaussianRenderer.py:64): its gradient belongs
            // to its view alone and is never accumulated
            float *gsv = a.g_screen[vv] + 3 * (size_t)i;
            gsv[0] = gsc[0]; gsv[1] = gsc[1]; gsv[2] = 0.0f;
        }
        // opacity = sigmoid(raw)
        const float o = 1.0f / (1.0f + expf(-opac_raw[i]));
        go = (float)(d_o * (double)(o * (1.0f - o)));

        // colour = max(sh . basis + 0.5, 0): gradient where the sum is >= 0 (torch.clamp rule).
        // One channel at a time keeps the register footprint small; the basis depends on the view.
        {
            const float4 *src = reinterpret_cast<const float4 *>(sh + (size_t)i * 48);
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                float co[16], acc[16];
#pragma unroll
                for (int v4 = 0; v4 < 4; ++v4) {
                    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (v4 * 4 < K) t = src[ch * 4 + v4];
                    co[v4 * 4 + 0] = t.x; co[v4 * 4 + 1] = t.y; co[v4 * 4 + 2] = t.z; co[v4 * 4 + 3] = t.w;
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[k] = 0.0f;
                float d_next = 0.0f;                  // the colour gradient of the next view
                auto fetch_col = [&](int vv) {
                    if (vv < a.nv && ((seen >> vv) & 1u)) d_next = a.gsum[vv][(size_t)i * 12 + 6 + ch];
                };
                fetch_col(0);
                for (int vv = 0; vv < a.nv; ++vv) {
                    const float d_col = d_next;
                    fetch_col(vv + 1);
                    if (!((seen >> vv) & 1u)) {
                        if (a.g_color[vv]) a.g_color[vv][3 * (size_t)i + ch] = 0.0f;
                        continue;
                    }
                    if (a.g_color[vv]) a.g_color[vv][3 * (size_t)i + ch] = d_col;
                    const float dx = p[0] - a.cam[vv].cam[0], dy = p[1] - a.cam[vv].cam[1], dz = p[2] - a.cam[vv].cam[2];
                    float bs[16];
                    sh_basis(deg, dx, dy, dz, bs);
                    float t = 0.0f;
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        if (k < K) t += bs[k] * co[k];
                    t += 0.5f;
                    if (t >= 0.0f) {
                        float dd[3];
                        sh_dir_grad(deg, dx, dy, dz, co, dd);
                        gp[0] += (double)d_col * (double)dd[0];
                        gp[1] += (double)d_col * (double)dd[1];
                        gp[2] += (double)d_col * (double)dd[2];
#pragma unroll
                        for (int k = 0; k < 16; ++k)
                            if (k < K) acc[k] = __fmaf_rn(d_col, bs[k], acc[k]);
                    }
                }
#pragma unroll
                for (int v4 = 0; v4 < 4; ++v4) {
                    if (!write_sh) break;
                    float4 o4 = make_float4(acc[4 * v4], acc[4 * v4 + 1], acc[4 * v4 + 2], acc[4 * v4 + 3]);
                    if (ACC) {
                        if (4 * v4 < K) {
                            const float4 t = dsh[ch * 4 + v4];
                            o4.x += t.x; o4.y += t.y; o4.z += t.z; o4.w += t.w;
                            dsh[ch * 4 + v4] = o4;
                        }
                    } else {
                        dsh[ch * 4 + v4] = o4;
                    }
                }
            }
        }
        {
#pragma clang fp contract(fast)
            // cov3 = RS RS^T, RS = R diag(s), s = exp(raw), R = R(q), q = raw / max(|raw|, eps)
            double dRS[3][3], dR[3][3], ds[3] = {0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    double t = 0.0;
#pragma unroll
                    for (int b = 0; b < 3; ++b) t += (dS[r][b] + dS[b][r]) * (double)g.RS[b][k];
                    dRS[r][k] = t;
                }
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    dR[r][k] = dRS[r][k] * (double)g.s[k];
                    ds[k] += dRS[r][k] * (double)g.R[r][k];
                }
#pragma unroll
            for (int k = 0; k < 3; ++k) gs[k] = (float)(ds[k] * (double)g.s[k]);
            const double w = g.q[0], x = g.q[1], y = g.q[2], z = g.q[3];
            double dq[4];
            dq[0] = 4 * w * (dR[0][0] + dR[1][1] + dR[2][2]) +
                    2 * (-z * dR[0][1] + y * dR[0][2] + z * dR[1][0] - x * dR[1][2] - y * dR[2][0] + x * dR[2][1]);
            dq[1] = 4 * x * dR[0][0] +
                    2 * (y * dR[0][1] + z * dR[0][2] + y * dR[1][0] - w * dR[1][2] + z * dR[2][0] + w * dR[2][1]);
            dq[2] = 4 * y * dR[1][1] +
                    2 * (x * dR[0][1] + w * dR[0][2] + x * dR[1][0] + z * dR[1][2] - w * dR[2][0] + z * dR[2][1]);
            dq[3] = 4 * z * dR[2][2] +
                    2 * (-w * dR[0][1] + x * dR[0][2] + w * dR[1][0] + y * dR[1][2] + x * dR[2][0] + y * dR[2][1]);
            const double nrm = g.qn;
            if (g.qn > 1e-12f) {
                const double qd = w * dq[0] + x * dq[1] + y * dq[2] + z * dq[3];
                const double qq[4] = {w, x, y, z};
#pragma unroll
                for (int k = 0; k < 4; ++k) gr[k] = (float)((dq[k] - qq[k] * qd) / nrm);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) gr[k] = (float)(dq[k] / nrm);
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) gc[k] = (float)gp[k];
    }

#pragma unroll
    for (int k = 0; k < 3; ++k) { put<ACC>(g_coords + 3 * i + k, gc[k]); put<ACC>(g_scale + 3 * i + k, gs[k]); }
    {
        float4 *dst = reinterpret_cast<float4 *>(g_rot) + i;
        float4 o4 = make_float4(gr[0], gr[1], gr[2], gr[3]);
        if (ACC) { const float4 t = *dst; o4.x += t.x; o4.y += t.y; o4.z += t.z; o4.w += t.w; }
        *dst = o4;
    }
    put<ACC>(g_opac + i, go);
}

// The sh gradient of a view minibatch from the per-view colour gradients: for every Gaussian and
// channel, sum over the views of dL/d(colour) * basis(direction of the view), where the colour passes
// its clamp (GaussianRenderer.py:71-74).  Same arithmetic and the same order over the views as the sh
// part of preprocess_bwd_kernel, so the result does not depend on how the views were spread over
// ranks.  A view in which the Gaussian is culled carries a zero colour gradient and is skipped.
struct ShViews {
    int nv;
    float cam[kMaxViews][3];
    const float *g_color[kMaxViews];   // [N,3] per view
};

template <bool ACC>
__global__ __launch_bounds__(256) void sh_grad_views_kernel(ShViews a, int n, int deg, int K,
                                                            const float *__restrict__ coords,
                                                            const float *__restrict__ sh, float *__restrict__ g_sh)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float p[3] = {coords[3 * i], coords[3 * i + 1], coords[3 * i + 2]};
    const float4 *src = reinterpret_cast<const float4 *>(sh + (size_t)i * 48);
    float4 *dsh = reinterpret_cast<float4 *>(g_sh + (size_t)i * 48);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float co[16], acc[16];
#pragma unroll
        for (int v4 = 0; v4 < 4; ++v4) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (v4 * 4 < K) t = src[ch * 4 + v4];
            co[v4 * 4 + 0] = t.x; co[v4 * 4 + 1] = t.y; co[v4 * 4 + 2] = t.z; co[v4 * 4 + 3] = t.w;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = 0.0f;
        for (int vv = 0; vv < a.nv; ++vv) {
            const float d_col = a.g_color[vv][3 * (size_t)i + ch];
            if (d_col == 0.0f) continue;
            const float dx = p[0] - a.cam[vv][0], dy = p[1] - a.cam[vv][1], dz = p[2] - a.cam[vv][2];
            float bs[16];
            sh_basis(deg, dx, dy, dz, bs);
            float t = 0.0f;
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (k < K) t += bs[k] * co[k];
            t += 0.5f;
            if (t >= 0.0f) {
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (k < K) acc[k] = __fmaf_rn(d_col, bs[k], acc[k]);
            }
        }
#pragma unroll
        for (int v4 = 0; v4 < 4; ++v4) {
            float4 o4 = make_float4(acc[4 * v4], acc[4 * v4 + 1], acc[4 * v4 + 2], acc[4 * v4 + 3]);
            if (ACC) {
                if (4 * v4 < K) {
                    const float4 t = dsh[ch * 4 + v4];
                    o4.x += t.x; o4.y += t.y; o4.z += t.z; o4.w += t.w;
                    dsh[ch * 4 + v4] = o4;
                }
            } else {
                dsh[ch * 4 + v4] = o4;
            }
        }
    }
}

}  // namespace

extern "C" int gsr_preprocess_fwd(const gsr_view *view_host, const float *coords, const float *scale_raw,
                                  const float *rot_raw, const float *opac_raw, const float *sh,
                                  const float *screen_coords, uint8_t *mask, uint32_t *depth_key, float *pos2d,
                                  float *cov2d, float *radius, int32_t *bbox, uint32_t *ntiles, float *rec,
                                  gsr_counters *counters, void *stream)
{
    ViewK v;
    if (int e = gsr_make_view(view_host, &v)) return e;
    if (!coords || !scale_raw || !rot_raw || !opac_raw || !sh || !mask || !depth_key || !pos2d || !cov2d ||
        !radius || !bbox || !ntiles || !rec || !counters) {
        gsr_set_error("gsr_preprocess_fwd: null pointer");
        return GSR_ERR_ARG;
    }
    if (v.n == 0) return GSR_OK;
    const FwdOut out = {mask, depth_key, pos2d, cov2d, radius, bbox, ntiles, rec};
    hipLaunchKernelGGL(preprocess_fwd_kernel, dim3(gsr_div_up(v.n, 256)), dim3(256), 0, (hipStream_t)stream, v,
                       coords, scale_raw, rot_raw, opac_raw, sh, screen_coords, out, counters);
    return gsr_check_launch("preprocess_fwd");
}

extern "C" int gsr_preprocess_fwd_views(const gsr_view *views_host, uint32_t num_views, const float *coords,
                                        const float *scale_raw, const float *rot_raw, const float *opac_raw, const float *sh,
                                        uint8_t *mask, uint32_t *depth_key, float *pos2d, float *cov2d, float *radius,
                                        int32_t *bbox, uint32_t *ntiles, float *rec, gsr_counters *counters, void *stream)
{
    if (!views_host || num_views == 0 || num_views > (uint32_t)kMaxFwdViews) {
        gsr_set_error("gsr_preprocess_fwd_views: %u views (1..%d expected)", num_views, kMaxFwdViews);
        return GSR_ERR_ARG;
    }
    FwdViews views;
    views.nv = (int)num_views;
    for (uint32_t k = 0; k < num_views; ++k) {
        if (int e = gsr_make_view(&views_host[k], &views.v[k])) return e;
        if (views.v[k].n != views.v[0].n || views.v[k].deg != views.v[0].deg) {
            gsr_set_error("gsr_preprocess_fwd_views: view %u differs from view 0 in the number of Gaussians or the sh degree", k);
            return GSR_ERR_ARG;
        }
    }
    for (uint32_t k = num_views; k < (uint32_t)kMaxFwdViews; ++k) views.v[k] = views.v[0];
    if (!coords || !scale_raw || !rot_raw || !opac_raw || !sh || !mask || !depth_key || !radius || !bbox || !ntiles ||
        !rec || !counters) {
        gsr_set_error("gsr_preprocess_fwd_views: null pointer");
        return GSR_ERR_ARG;
    }
    const int n = views.v[0].n;
    if (n == 0) return GSR_OK;
    const FwdOut out = {mask, depth_key, pos2d, cov2d, radius, bbox, ntiles, rec};
    hipLaunchKernelGGL(preprocess_fwd_views_kernel, dim3(gsr_div_up(n, kFwdViewsThreads)), dim3(kFwdViewsThreads), 0, (hipStream_t)stream, views,
                       coords, scale_raw, rot_raw, opac_raw, sh, out, counters);
    return gsr_check_launch("preprocess_fwd_views");
}

extern "C" size_t gsr_preprocess_bwd_workspace_bytes(uint32_t n)
{
    return ((size_t)(n ? n : 1) * 12 * sizeof(float) + 255) & ~(size_t)255;
}

extern "C" int gsr_pair_reduce(uint32_t n, uint32_t parts, const uint8_t *mask, const uint32_t *slot_base,
                               const uint32_t *ntiles, const float *pairgrad, float *gsum, void *stream)
{
    if (n == 0) return GSR_OK;
    if (parts == 0 || parts > 4) { gsr_set_error("gsr_pair_reduce: %u rows per pair (1..4 expected)", parts); return GSR_ERR_ARG; }
    if (!mask || !slot_base || !ntiles || !gsum) { gsr_set_error("gsr_pair_reduce: null pointer"); return GSR_ERR_ARG; }
    if ((unsigned long long)n * kLanesPerGaussian >= (1ull << 31)) { gsr_set_error("gsr_pair_reduce: too many Gaussians"); return GSR_ERR_ARG; }
    hipLaunchKernelGGL(pair_reduce_kernel, dim3(gsr_div_up(n * kLanesPerGaussian, 256)), dim3(256), 0, (hipStream_t)stream,
                       (int)n, parts, mask, slot_base, ntiles, pairgrad, gsum);
    return gsr_check_launch("pair_reduce");
}

extern "C" int gsr_preprocess_bwd_views(const gsr_view *views_host, uint32_t num_views, const float *coords,
                                        const float *scale_raw, const float *rot_raw, const float *opac_raw,
                                        const float *sh, const uint8_t *const *mask_host, const float *const *gsum_host,
                                        float *g_coords, float *g_scale, float *g_rot, float *g_opac, float *g_sh,
                                        float *const *g_screen_host, float *const *g_color_host, int accumulate,
                                        void *stream)
{
    if (num_views == 0 || !views_host || !mask_host || !gsum_host || !g_screen_host) {
        gsr_set_error("gsr_preprocess_bwd_views: no views or null view arrays");
        return GSR_ERR_ARG;
    }
    if (!coords || !scale_raw || !rot_raw || !opac_raw || !sh || !g_coords || !g_scale || !g_rot || !g_opac) {
        gsr_set_error("gsr_preprocess_bwd_views: null pointer");
        return GSR_ERR_ARG;
    }
    ViewK first;
    if (int e = gsr_make_view(&views_host[0], &first)) return e;
    if (first.n == 0) return GSR_OK;
    const dim3 grid(gsr_div_up(first.n, 256)), block(256);
    // at most GSR_MAX_VIEWS views per launch; further groups add to the result of the first
    for (uint32_t v0 = 0; v0 < num_views; v0 += kMaxViews) {
        BwdViews a;
        a.nv = (int)(num_views - v0 < (uint32_t)kMaxViews ? num_views - v0 : (uint32_t)kMaxViews);
        for (int k = 0; k < a.nv; ++k) {
            ViewK v;
            if (int e = gsr_make_view(&views_host[v0 + k], &v)) return e;
            if (v.n != first.n || v.deg != first.deg) {
                gsr_set_error("gsr_preprocess_bwd_views: view %u differs from view 0 in n or sh_degree", v0 + k);
                return GSR_ERR_ARG;
            }
            if (!mask_host[v0 + k] || !gsum_host[v0 + k] || !g_screen_host[v0 + k] || (g_color_host && !g_color_host[v0 + k])) {
                gsr_set_error("gsr_preprocess_bwd_views: null pointer for view %u", v0 + k);
                return GSR_ERR_ARG;
            }
            ViewCam &c = a.cam[k];
            c.Wf = v.Wf; c.Hf = v.Hf;
            for (int j = 0; j < 16; ++j) { c.w2m[j] = v.w2m[j]; c.proj[j] = v.proj[j]; }
            for (int j = 0; j < 3; ++j) c.cam[j] = v.cam[j];
            a.mask[k] = mask_host[v0 + k];
            a.gsum[k] = gsum_host[v0 + k];
            a.g_screen[k] = g_screen_host[v0 + k];
            a.g_color[k] = g_color_host ? g_color_host[v0 + k] : nullptr;
        }
        for (int k = a.nv; k < kMaxViews; ++k) { a.mask[k] = nullptr; a.gsum[k] = nullptr; a.g_screen[k] = nullptr; a.g_color[k] = nullptr; }
        if (accumulate || v0 > 0)
            hipLaunchKernelGGL(preprocess_bwd_kernel<true>, grid, block, 0, (hipStream_t)stream, a, first.n, first.deg,
                               first.K, coords, scale_raw, rot_raw, opac_raw, sh, g_coords, g_scale, g_rot, g_opac, g_sh);
        else
            hipLaunchKernelGGL(preprocess_bwd_kernel<false>, grid, block, 0, (hipStream_t)stream, a, first.n, first.deg,
                               first.K, coords, scale_raw, rot_raw, opac_raw, sh, g_coords, g_scale, g_rot, g_opac, g_sh);
    }
    return gsr_check_launch("preprocess_bwd_views");
}

extern "C" int gsr_preprocess_bwd(const gsr_view *view_host, const float *coords, const float *scale_raw,
                                  const float *rot_raw, const float *opac_raw, const float *sh,
                                  const uint8_t *mask, const uint32_t *slot_base, const uint32_t *ntiles,
                                  const float *pairgrad, float *g_coords, float *g_scale, float *g_rot,
                                  float *g_opac, float *g_sh, float *g_screen, int accumulate, void *workspace,
                                  size_t workspace_bytes, void *stream)
{
    ViewK v;
    if (int e = gsr_make_view(view_host, &v)) return e;
    if (!mask || !slot_base || !ntiles || !g_screen) { gsr_set_error("gsr_preprocess_bwd: null pointer"); return GSR_ERR_ARG; }
    if (!workspace || workspace_bytes < gsr_preprocess_bwd_workspace_bytes((uint32_t)v.n)) {
        gsr_set_error("gsr_preprocess_bwd: workspace too small");
        return GSR_ERR_ARG;
    }
    if (v.n == 0) return GSR_OK;
    float *gsum = (float *)workspace;
    if (int e = gsr_pair_reduce((uint32_t)v.n, gsr_pairgrad_parts(v.L), mask, slot_base, ntiles, pairgrad, gsum, stream)) return e;
    const float *gsum_c = gsum;
    return gsr_preprocess_bwd_views(view_host, 1, coords, scale_raw, rot_raw, opac_raw, sh, &mask, &gsum_c, g_coords,
                                    g_scale, g_rot, g_opac, g_sh, &g_screen, nullptr, accumulate, stream);
}

extern "C" int gsr_sh_grad_views(const gsr_view *views_host, uint32_t num_views, const float *coords, const float *sh,
                                 const float *const *g_color_host, float *g_sh, int accumulate, void *stream)
{
    if (num_views == 0 || !views_host || !g_color_host) {
        gsr_set_error("gsr_sh_grad_views: no views or null view arrays");
        return GSR_ERR_ARG;
    }
    if (!coords || !sh || !g_sh) { gsr_set_error("gsr_sh_grad_views: null pointer"); return GSR_ERR_ARG; }
    ViewK first;
    if (int e = gsr_make_view(&views_host[0], &first)) return e;
    if (first.n == 0) return GSR_OK;
    const dim3 grid(gsr_div_up(first.n, 256)), block(256);
    for (uint32_t v0 = 0; v0 < num_views; v0 += kMaxViews) {
        ShViews a;
        a.nv = (int)(num_views - v0 < (uint32_t)kMaxViews ? num_views - v0 : (uint32_t)kMaxViews);
        for (int k = 0; k < kMaxViews; ++k) { a.g_color[k] = nullptr; a.cam[k][0] = a.cam[k][1] = a.cam[k][2] = 0.0f; }
        for (int k = 0; k < a.nv; ++k) {
            ViewK v;
            if (int e = gsr_make_view(&views_host[v0 + k], &v)) return e;
            if (v.n != first.n || v.deg != first.deg) {
                gsr_set_error("gsr_sh_grad_views: view %u differs from view 0 in n or sh_degree", v0 + k);
                return GSR_ERR_ARG;
            }
            if (!g_color_host[v0 + k]) { gsr_set_error("gsr_sh_grad_views: null pointer for view %u", v0 + k); return GSR_ERR_ARG; }
            for (int j = 0; j < 3; ++j) a.cam[k][j] = v.cam[j];
            a.g_color[k] = g_color_host[v0 + k];
        }
        if (accumulate || v0 > 0)
            hipLaunchKernelGGL(sh_grad_views_kernel<true>, grid, block, 0, (hipStream_t)stream, a, first.n, first.deg, first.K,
                               coords, sh, g_sh);
        else
            hipLaunchKernelGGL(sh_grad_views_kernel<false>, grid, block, 0, (hipStream_t)stream, a, first.n, first.deg, first.K,
                               coords, sh, g_sh);
    }
    return gsr_check_launch("sh_grad_views");
}
