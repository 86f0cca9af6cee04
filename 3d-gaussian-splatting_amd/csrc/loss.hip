// Training loss of one view and its gradient with respect to the rendered image, in one pass over
// the image: loss = (1 - lambda) * mean|pred - gt| + lambda * (1 - mean SSIM)   (Trainer.py:35-38,
// Utils/Loss.py:4-5, Utils/Loss.py:59-97).
//
// The reference passes the renderer's [W,H,3] image to its SSIM, whose window code reads dimension
// -3 as the channel count (Utils/Loss.py:70): every image column x is an independent [H,3] plane
// and the 11x11 window slides along y and across the three colours, never along x.  A plane is a
// contiguous run of 3H floats, so one workgroup takes one run of rows of one plane, keeps it in LDS
// and does everything there: window sums of pred, gt, pred^2, gt^2, pred*gt, the SSIM map, its
// three partial derivatives, their window sums (the adjoint of a zero-padded symmetric window is the
// same window) and dL/dpred.  HBM traffic is one read of pred and gt and one write of dL/dpred.
//
// The 2-D window is the outer product of a 1-D window with itself (Utils/Loss.py:63-65), so a window
// sum is a blur along y followed by a 3x3 mix of the colours.  A thread owns two consecutive rows:
// it walks the 12 rows their windows cover once, and every staged row feeds both accumulators, which
// halves the LDS reads; the 15 (9 for the derivatives) blurred values per row are kept as pairs so
// that the blur runs on v_pk_fma_f32.  Rows are staged with even and odd rows in separate halves:
// lanes then read consecutive 24-byte (40-byte) records, which spreads over all LDS banks.
#include "gsr_common.hpp"

#include <math.h>

namespace {

constexpr int kHalf = 5;             // window_size // 2, Utils/Loss.py:80
constexpr int kWin = 2 * kHalf + 1;  // window_size = 11, Utils/Loss.py:69
constexpr int kRawDwords = 6;        // staged row: p0 p1 g0 g1 p2 g2
constexpr int kDDwords = 10;         // staged derivative row: Dm0 Dm1 Dm2 Ds0 Ds1 Ds2 Dc0 Dc1 Dc2 -
constexpr int kDPad = 6;             // derivative rows are staged at index (map row + kDPad)
constexpr int kFinishThreads = 1024;

typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

struct LossK {
    int W, H, chunks;
    float c_l1, c_ssim;  // factors of sign(pred - gt) and of d(sum SSIM)/dpred in dL/dpred
    float g[kWin];       // 1-D window, Utils/Loss.py:59-61
    f2 g2[kWin];         // the same, each weight twice (operand of the packed multiply-add)
};

// out[c] = sum over input colour cc of g[5 + cc - c] * q[cc]: the three window columns that overlap a
// 3-wide plane padded with zeros (Utils/Loss.py:80).
__device__ __forceinline__ void colour_mix(const LossK &k, float q0, float q1, float q2, float *out)
{
    out[0] = fmaf(k.g[kHalf + 2], q2, fmaf(k.g[kHalf + 1], q1, k.g[kHalf] * q0));
    out[1] = fmaf(k.g[kHalf + 1], q2, fmaf(k.g[kHalf], q1, k.g[kHalf - 1] * q0));
    out[2] = fmaf(k.g[kHalf], q2, fmaf(k.g[kHalf - 1], q1, k.g[kHalf - 2] * q0));
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// Geometry of a workgroup of THREADS threads: 2 map rows per thread, of which the outer 5 on each side
// only feed the derivative windows of the rows the workgroup owns.
template <int THREADS>
struct Geom {
    static constexpr int kMapRows = 2 * THREADS;
    static constexpr int kOutRows = kMapRows - 2 * kHalf;
    static constexpr int kInRows = kMapRows + 2 * kHalf;
    static constexpr int kRawHalf = (kInRows + 1) / 2;              // staged rows per parity
    static constexpr int kDHalf = (kMapRows + 2 * kDPad) / 2;
};

template <int THREADS, bool GRAD>
__global__ __launch_bounds__(THREADS) void loss_kernel(LossK k, const float *__restrict__ pred,
                                                       const float *__restrict__ gt, float *__restrict__ dpred,
                                                       float2 *__restrict__ partial)
{
    using G = Geom<THREADS>;
    // s_raw[parity][half index][6]; after the first blur the same memory stages dL/dpred of the owned rows
    __shared__ __attribute__((aligned(16))) float s_raw[2 * G::kRawHalf * kRawDwords];
    __shared__ __attribute__((aligned(16))) float s_d[GRAD ? 2 * G::kDHalf * kDDwords : 2];
    __shared__ float s_red[2 * ((THREADS + GSR_WAVE - 1) / GSR_WAVE)];
    static_assert(2 * G::kRawHalf * kRawDwords >= G::kOutRows * 3, "output staging fits in the raw rows");

    const int t = threadIdx.x;
    const int chunk = blockIdx.x % k.chunks, x = blockIdx.x / k.chunks;
    const int y0 = chunk * G::kOutRows;  // first row this workgroup owns
    const size_t plane = (size_t)x * (size_t)k.H * 3;

    // rows y0-10 .. y0+2*THREADS-1 of the plane, zero outside it; one thread stages one row
    for (int r = t; r < G::kInRows; r += THREADS) {
        const int y = y0 - 2 * kHalf + r;
        float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, g0 = 0.0f, g1 = 0.0f, g2 = 0.0f;
        if (y >= 0 && y < k.H) {
            const float *pp = pred + plane + (size_t)y * 3, *gp = gt + plane + (size_t)y * 3;
            p0 = pp[0], p1 = pp[1], p2 = pp[2];
            g0 = gp[0], g1 = gp[1], g2 = gp[2];
        }
        f2 *row = (f2 *)(s_raw + ((r & 1) * G::kRawHalf + (r >> 1)) * kRawDwords);
        row[0] = f2{p0, p1};
        row[1] = f2{g0, g1};
        row[2] = f2{p2, g2};
    }
    __syncthreads();

    // Thread t evaluates map rows 2t and 2t+1 (image rows y0-5+2t, +1); their windows cover staged rows
    // 2t .. 2t+11.  Blurred along y: A = (p0,p1) B = (g0,g1) C = (p2,g2), their squares, A*B, and p2*g2.
    f2 acc[2][7];
    float accs[2];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        accs[o] = 0.0f;
#pragma unroll
        for (int j = 0; j < 7; ++j) acc[o][j] = f2{0.0f, 0.0f};
    }
    f2 own[2][3];  // staged pred / gt of the thread's own two rows
    // Rows are fetched two iterations ahead, and an empty asm statement per iteration keeps the
    // compiler from hoisting all 12 row fetches (72 registers) to the top.
    const auto raw_row = [&](int kk, f2 (&dst)[3]) {
        const f2 *row = (const f2 *)(s_raw + ((kk & 1) * G::kRawHalf + t + (kk >> 1)) * kRawDwords);
        dst[0] = row[0];
        dst[1] = row[1];
        dst[2] = row[2];
    };
    f2 cur[3], nx1[3], nx2[3];
    raw_row(0, cur);
    raw_row(1, nx1);
#pragma unroll
    for (int kk = 0; kk < kWin + 1; ++kk) {
        if (kk + 2 < kWin + 1) raw_row(kk + 2, nx2);
        f2 q[7];
        q[0] = cur[0];
        q[1] = cur[1];
        q[2] = cur[2];
        q[3] = q[0] * q[0];
        q[4] = q[1] * q[1];
        q[5] = q[2] * q[2];
        q[6] = q[0] * q[1];
        const float qs = q[2].x * q[2].y;
        if (kk == kHalf || kk == kHalf + 1) {
            own[kk - kHalf][0] = q[0];
            own[kk - kHalf][1] = q[1];
            own[kk - kHalf][2] = q[2];
        }
        if (kk < kWin) {
#pragma unroll
            for (int j = 0; j < 7; ++j) acc[0][j] = pk_fma(k.g2[kk], q[j], acc[0][j]);
            accs[0] = fmaf(k.g[kk], qs, accs[0]);
        }
        if (kk >= 1) {
#pragma unroll
            for (int j = 0; j < 7; ++j) acc[1][j] = pk_fma(k.g2[kk - 1], q[j], acc[1][j]);
            accs[1] = fmaf(k.g[kk - 1], qs, accs[1]);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            cur[j] = nx1[j];
            nx1[j] = nx2[j];
        }
        // the accumulators are complete here and later fetches stay below this point
        asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[0][4]),
                          "+v"(acc[0][5]), "+v"(acc[0][6]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]),
                          "+v"(acc[1][3]), "+v"(acc[1][4]), "+v"(acc[1][5]), "+v"(acc[1][6]), "+v"(accs[0]), "+v"(accs[1])
                     :
                     : "memory");
    }

    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;  // Utils/Loss.py:91-92
    float ssim_sum = 0.0f, l1_sum = 0.0f;
    bool owned[2];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const int m = 2 * t + o;               // map row inside the workgroup
        const int ym = y0 - kHalf + m;         // image row
        const bool in_image = ym >= 0 && ym < k.H;
        owned[o] = m >= kHalf && m < kHalf + G::kOutRows && ym < k.H;
        float mu1[3], mu2[3], e11[3], e22[3], e12[3];
        colour_mix(k, acc[o][0].x, acc[o][0].y, acc[o][2].x, mu1);
        colour_mix(k, acc[o][1].x, acc[o][1].y, acc[o][2].y, mu2);
        colour_mix(k, acc[o][3].x, acc[o][3].y, acc[o][5].x, e11);
        colour_mix(k, acc[o][4].x, acc[o][4].y, acc[o][5].y, e22);
        colour_mix(k, acc[o][6].x, acc[o][6].y, accs[o], e12);
        float dm[3], ds[3], dc[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            // Utils/Loss.py:83-94
            const float mu1_sq = mu1[c] * mu1[c], mu2_sq = mu2[c] * mu2[c], mu12 = mu1[c] * mu2[c];
            const float s1 = e11[c] - mu1_sq, s2 = e22[c] - mu2_sq, s12 = e12[c] - mu12;
            const float A1 = 2.0f * mu12 + C1, A2 = 2.0f * s12 + C2;
            const float B1 = mu1_sq + mu2_sq + C1, B2 = s1 + s2 + C2;
            // v_rcp_f32 (1 ulp) instead of two IEEE divisions: B1, B2 >= C1, C2 > 0 and far from denormal
            const float inv1 = __builtin_amdgcn_rcpf(B1), inv2 = __builtin_amdgcn_rcpf(B2);
            const float inv = inv1 * inv2;
            const float S = A1 * A2 * inv;
            if (owned[o]) ssim_sum += S;
            // partial derivatives of S with respect to the window sums of pred, pred^2 and pred*gt
            const float d_mu = 2.0f * mu2[c] * (A2 - A1) * inv + 2.0f * mu1[c] * S * (inv2 - inv1);
            dm[c] = in_image ? d_mu : 0.0f;
            ds[c] = in_image ? -S * inv2 : 0.0f;
            dc[c] = in_image ? 2.0f * A1 * inv : 0.0f;
        }
        if (owned[o]) {  // Utils/Loss.py:5
            l1_sum += fabsf(own[o][0].x - own[o][1].x) + fabsf(own[o][0].y - own[o][1].y) +
                      fabsf(own[o][2].x - own[o][2].y);
        }
        if (GRAD) {
            const int r = m + kDPad;
            f2 *row = (f2 *)(s_d + ((r & 1) * G::kDHalf + (r >> 1)) * kDDwords);
            row[0] = f2{dm[0], dm[1]};
            row[1] = f2{dm[2], ds[0]};
            row[2] = f2{ds[1], ds[2]};
            row[3] = f2{dc[0], dc[1]};
            row[4] = f2{dc[2], 0.0f};
        }
    }

    if (GRAD) {
        __syncthreads();  // derivative rows complete; nobody reads the staged pred / gt any more
        // windows of map rows 2t, 2t+1 cover derivative rows 2t-5 .. 2t+6, staged at 2t+1 .. 2t+12
        f2 b[2][5];
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int j = 0; j < 5; ++j) b[o][j] = f2{0.0f, 0.0f};
        const auto d_row = [&](int kk, f2 (&dst)[5]) {
            const int odd = (kk + 1) & 1, half = (kk + 1) >> 1;
            const f2 *row = (const f2 *)(s_d + (odd * G::kDHalf + t + half) * kDDwords);
#pragma unroll
            for (int j = 0; j < 5; ++j) dst[j] = row[j];
        };
        f2 dcur[5], dnx1[5], dnx2[5];
        d_row(0, dcur);
        d_row(1, dnx1);
#pragma unroll
        for (int kk = 0; kk < kWin + 1; ++kk) {
            if (kk + 2 < kWin + 1) d_row(kk + 2, dnx2);
            if (kk < kWin) {
#pragma unroll
                for (int j = 0; j < 5; ++j) b[0][j] = pk_fma(k.g2[kk], dcur[j], b[0][j]);
            }
            if (kk >= 1) {
#pragma unroll
                for (int j = 0; j < 5; ++j) b[1][j] = pk_fma(k.g2[kk - 1], dcur[j], b[1][j]);
            }
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                dcur[j] = dnx1[j];
                dnx1[j] = dnx2[j];
            }
            asm volatile("" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[0][3]), "+v"(b[0][4]), "+v"(b[1][0]),
                              "+v"(b[1][1]), "+v"(b[1][2]), "+v"(b[1][3]), "+v"(b[1][4])
                         :
                         : "memory");
        }
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            if (!owned[o]) continue;
            float bm[3], bs[3], bc[3];
            colour_mix(k, b[o][0].x, b[o][0].y, b[o][1].x, bm);
            colour_mix(k, b[o][1].y, b[o][2].x, b[o][2].y, bs);
            colour_mix(k, b[o][3].x, b[o][3].y, b[o][4].x, bc);
            const float p[3] = {own[o][0].x, own[o][0].y, own[o][2].x};
            const float g[3] = {own[o][1].x, own[o][1].y, own[o][2].y};
            const int m = 2 * t + o;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float dssim = bm[c] + 2.0f * p[c] * bs[c] + g[c] * bc[c];
                const float diff = p[c] - g[c];
                const float sgn = (diff > 0.0f ? 1.0f : 0.0f) - (diff < 0.0f ? 1.0f : 0.0f);
                s_raw[(m - kHalf) * 3 + c] = k.c_l1 * sgn + k.c_ssim * dssim;
            }
        }
        __syncthreads();
        // the owned rows are one contiguous run of the plane: store it with consecutive lanes
        const int rows = min(G::kOutRows, k.H - y0);
        for (int i = t; i < rows * 3; i += THREADS) dpred[plane + (size_t)y0 * 3 + i] = s_raw[i];
    }

    l1_sum = wave_sum(l1_sum);
    ssim_sum = wave_sum(ssim_sum);
    const int wave = t / GSR_WAVE;
    if ((t & (GSR_WAVE - 1)) == 0) {
        s_red[wave * 2] = l1_sum;
        s_red[wave * 2 + 1] = ssim_sum;
    }
    __syncthreads();
    if (t == 0) {
        float a = 0.0f, b = 0.0f;
        for (int w = 0; w < (THREADS + GSR_WAVE - 1) / GSR_WAVE; ++w) {
            a += s_red[w * 2];
            b += s_red[w * 2 + 1];
        }
        partial[blockIdx.x] = make_float2(a, b);
    }
}

// Sums the per-workgroup partial sums in double and writes {loss, l1, ssim} (Trainer.py:36-38).
__global__ __launch_bounds__(kFinishThreads) void loss_finish_kernel(const float2 *__restrict__ partial, int blocks,
                                                                     double inv_count, float loss_lambda,
                                                                     float *__restrict__ out)
{
    __shared__ double s_a[kFinishThreads / GSR_WAVE], s_b[kFinishThreads / GSR_WAVE];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < blocks; i += kFinishThreads) {
        const float2 v = partial[i];
        a += (double)v.x;
        b += (double)v.y;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        a += __shfl_xor(a, m);
        b += __shfl_xor(b, m);
    }
    if ((threadIdx.x & (GSR_WAVE - 1)) == 0) {
        s_a[threadIdx.x / GSR_WAVE] = a;
        s_b[threadIdx.x / GSR_WAVE] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double l1_sum = 0.0, ssim_sum = 0.0;
        for (int w = 0; w < kFinishThreads / GSR_WAVE; ++w) {
            l1_sum += s_a[w];
            ssim_sum += s_b[w];
        }
        const float l1 = (float)(l1_sum * inv_count), ssim = (float)(ssim_sum * inv_count);
        out[0] = (1.0f - loss_lambda) * l1 + loss_lambda * (1.0f - ssim);
        out[1] = l1;
        out[2] = ssim;
    }
}

// Workgroup size for an image height: the one whose run of owned rows tiles the column with the fewest
// evaluated rows (a workgroup of n threads evaluates 2n rows and owns 2n - 10 of them).
struct Plan {
    int threads, chunks;
};

Plan loss_plan(int height)
{
    Plan best{0, 0};
    long best_rows = 0;
    for (int threads = 64; threads <= 256; threads += 64) {
        const int out_rows = 2 * threads - 2 * kHalf;
        const int chunks = (height + out_rows - 1) / out_rows;
        const long rows = (long)chunks * 2 * threads;
        if (best.threads == 0 || rows < best_rows) {
            best = Plan{threads, chunks};
            best_rows = rows;
        }
    }
    return best;
}

template <int THREADS>
void launch_loss(const LossK &k, unsigned blocks, const float *pred, const float *gt, float *dpred, float2 *partial,
                 hipStream_t s)
{
    if (dpred)
        hipLaunchKernelGGL((loss_kernel<THREADS, true>), dim3(blocks), dim3(THREADS), 0, s, k, pred, gt, dpred, partial);
    else
        hipLaunchKernelGGL((loss_kernel<THREADS, false>), dim3(blocks), dim3(THREADS), 0, s, k, pred, gt, dpred, partial);
}

}  // namespace

extern "C" size_t gsr_loss_workspace_bytes(int width, int height)
{
    if (width <= 0 || height <= 0) return 0;
    return (size_t)width * (size_t)loss_plan(height).chunks * sizeof(float2);
}

extern "C" int gsr_loss_fwd_bwd(const float *pred, const float *gt, int width, int height, float loss_lambda,
                                float grad_scale, float *out, float *dpred, void *workspace, size_t workspace_bytes,
                                void *stream)
{
    if (!pred || !gt || !out || !workspace) {
        gsr_set_error("gsr_loss_fwd_bwd: null pointer");
        return GSR_ERR_ARG;
    }
    if (width <= 0 || height <= 0 || (size_t)width * (size_t)height * 3 >= ((size_t)1 << 31)) {
        gsr_set_error("gsr_loss_fwd_bwd: image %d x %d out of range", width, height);
        return GSR_ERR_ARG;
    }
    if (!(loss_lambda >= 0.0f && loss_lambda <= 1.0f)) {
        gsr_set_error("gsr_loss_fwd_bwd: loss_lambda %g outside [0, 1]", (double)loss_lambda);
        return GSR_ERR_ARG;
    }
    if (workspace_bytes < gsr_loss_workspace_bytes(width, height)) {
        gsr_set_error("gsr_loss_fwd_bwd: workspace of %zu bytes, need %zu", workspace_bytes,
                      gsr_loss_workspace_bytes(width, height));
        return GSR_ERR_ARG;
    }
    const Plan plan = loss_plan(height);
    LossK k;
    k.W = width;
    k.H = height;
    k.chunks = plan.chunks;
    const double count = (double)width * (double)height * 3.0;
    k.c_l1 = (float)((double)grad_scale * (1.0 - (double)loss_lambda) / count);
    k.c_ssim = (float)(-(double)grad_scale * (double)loss_lambda / count);
    // Utils/Loss.py:59-61: fp32 values of the Gaussian, normalised by their fp32 sum
    float sum = 0.0f;
    for (int i = 0; i < kWin; ++i) {
        k.g[i] = (float)exp(-(double)((i - kHalf) * (i - kHalf)) / (2.0 * 1.5 * 1.5));
        sum += k.g[i];
    }
    for (int i = 0; i < kWin; ++i) {
        k.g[i] = k.g[i] / sum;
        k.g2[i] = f2{k.g[i], k.g[i]};
    }

    const unsigned blocks = (unsigned)width * (unsigned)plan.chunks;
    hipStream_t s = (hipStream_t)stream;
    float2 *partial = (float2 *)workspace;
    switch (plan.threads) {
    case 64: launch_loss<64>(k, blocks, pred, gt, dpred, partial, s); break;
    case 128: launch_loss<128>(k, blocks, pred, gt, dpred, partial, s); break;
    case 192: launch_loss<192>(k, blocks, pred, gt, dpred, partial, s); break;
    default: launch_loss<256>(k, blocks, pred, gt, dpred, partial, s); break;
    }
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(kFinishThreads), 0, s, partial, (int)blocks, 1.0 / count,
                       loss_lambda, out);
    return gsr_check_launch("gsr_loss_fwd_bwd");
}
