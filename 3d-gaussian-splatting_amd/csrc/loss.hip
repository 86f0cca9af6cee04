// Training loss of one view and its gradient with respect to the rendered image, in one pass over
// the image: loss = (1 - lambda) * mean|pred - gt| + lambda * (1 - mean SSIM)   (Trainer.py:35-38,
// Utils/Loss.py:4-5, Utils/Loss.py:59-97).
//
// The reference passes the renderer's [W,H,3] image to its SSIM, whose window code reads dimension
// -3 as the channel count (Utils/Loss.py:70): every image column x is an independent [H,3] plane
// and the 11x11 window slides along y and across the three colours, never along x.  A plane is a
// contiguous run of 3H floats, so one workgroup takes one run of rows of one plane, keeps it in LDS
// and does everything there: window sums of pred, gt, pred^2, gt^2, pred*gt (colour mix, then blur
// along y), the SSIM map, its three partial derivatives, their window sums (the adjoint of a
// zero-padded symmetric window is the same window) and dL/dpred.  HBM traffic is one read of pred
// and gt and one write of dL/dpred.
#include "gsr_common.hpp"

#include <math.h>

namespace {

constexpr int kHalf = 5;                        // window_size // 2, Utils/Loss.py:80
constexpr int kWin = 2 * kHalf + 1;             // window_size = 11, Utils/Loss.py:69
constexpr int kThreads = 256;
constexpr int kMapRows = kThreads;              // rows of the SSIM map one workgroup evaluates
constexpr int kOutRows = kMapRows - 2 * kHalf;  // rows of dL/dpred it owns
constexpr int kInRows = kMapRows + 2 * kHalf;   // rows of pred / gt it reads
constexpr int kMix = 15;                        // colour-mixed values per row: {p, g, pp, gg, pg} x 3
constexpr int kDMix = 9;                        // colour-mixed derivative values per row: 3 maps x 3

struct LossK {
    int W, H, chunks;
    float c_l1, c_ssim;   // factors of sign(pred - gt) and of d(sum SSIM)/dpred in dL/dpred
    float g[kWin];        // 1-D window, Utils/Loss.py:59-61
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

template <bool GRAD>
__global__ __launch_bounds__(kThreads) void loss_kernel(LossK k, const float *__restrict__ pred,
                                                        const float *__restrict__ gt, float *__restrict__ dpred,
                                                        float2 *__restrict__ partial)
{
    __shared__ float s_p[kInRows * 3];
    __shared__ float s_g[kInRows * 3];
    __shared__ float s_mix[kInRows * kMix];
    __shared__ float s_d[kMapRows * kDMix];
    __shared__ float s_red[2 * (kThreads / GSR_WAVE)];

    const int t = threadIdx.x;
    const int chunk = blockIdx.x % k.chunks, x = blockIdx.x / k.chunks;
    const int y0 = chunk * kOutRows;                 // first row this workgroup owns
    const size_t plane = (size_t)x * (size_t)k.H * 3;
    const int plane_len = k.H * 3;

    // rows y0-10 .. y0+255 of the plane, zero outside it
    const int first = (y0 - 2 * kHalf) * 3;
    for (int i = t; i < kInRows * 3; i += kThreads) {
        const int e = first + i;
        const bool in = e >= 0 && e < plane_len;
        s_p[i] = in ? pred[plane + e] : 0.0f;
        s_g[i] = in ? gt[plane + e] : 0.0f;
    }
    __syncthreads();

    for (int r = t; r < kInRows; r += kThreads) {
        const float p0 = s_p[r * 3], p1 = s_p[r * 3 + 1], p2 = s_p[r * 3 + 2];
        const float g0 = s_g[r * 3], g1 = s_g[r * 3 + 1], g2 = s_g[r * 3 + 2];
        float m[kMix];
        colour_mix(k, p0, p1, p2, m);
        colour_mix(k, g0, g1, g2, m + 3);
        colour_mix(k, p0 * p0, p1 * p1, p2 * p2, m + 6);
        colour_mix(k, g0 * g0, g1 * g1, g2 * g2, m + 9);
        colour_mix(k, p0 * g0, p1 * g1, p2 * g2, m + 12);
#pragma unroll
        for (int j = 0; j < kMix; ++j) s_mix[r * kMix + j] = m[j];
    }
    __syncthreads();

    // thread t evaluates map row ym = y0 - 5 + t, whose window covers staged rows t .. t+10
    float acc[kMix];
#pragma unroll
    for (int j = 0; j < kMix; ++j) acc[j] = 0.0f;
#pragma unroll
    for (int dy = 0; dy < kWin; ++dy) {
        const float w = k.g[dy];
#pragma unroll
        for (int j = 0; j < kMix; ++j) acc[j] = fmaf(w, s_mix[(t + dy) * kMix + j], acc[j]);
    }

    const int ym = y0 - kHalf + t;
    const bool in_image = ym >= 0 && ym < k.H;
    const bool own = t >= kHalf && t < kHalf + kOutRows && ym < k.H;
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;   // Utils/Loss.py:91-92
    float ssim_sum = 0.0f, dmap[kDMix];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        // Utils/Loss.py:83-94
        const float mu1 = acc[c], mu2 = acc[3 + c];
        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
        const float s1 = acc[6 + c] - mu1_sq, s2 = acc[9 + c] - mu2_sq, s12 = acc[12 + c] - mu12;
        const float A1 = 2.0f * mu12 + C1, A2 = 2.0f * s12 + C2;
        const float B1 = mu1_sq + mu2_sq + C1, B2 = s1 + s2 + C2;
        const float inv = 1.0f / (B1 * B2);
        const float S = A1 * A2 * inv;
        if (own) ssim_sum += S;
        // partial derivatives of S with respect to the window sums of pred, pred^2 and pred*gt
        const float d_mu = 2.0f * mu2 * (A2 - A1) * inv + 2.0f * mu1 * S * (1.0f / B2 - 1.0f / B1);
        dmap[c] = in_image ? d_mu : 0.0f;
        dmap[3 + c] = in_image ? -S / B2 : 0.0f;
        dmap[6 + c] = in_image ? 2.0f * A1 * inv : 0.0f;
    }

    const int rr = (t + kHalf) * 3;   // this thread's own row in the staged pred / gt
    const float p[3] = {s_p[rr], s_p[rr + 1], s_p[rr + 2]};
    const float g[3] = {s_g[rr], s_g[rr + 1], s_g[rr + 2]};
    float l1_sum = 0.0f;
    if (own) l1_sum = fabsf(p[0] - g[0]) + fabsf(p[1] - g[1]) + fabsf(p[2] - g[2]);   // Utils/Loss.py:5

    if (GRAD) {
        float m[kDMix];
        colour_mix(k, dmap[0], dmap[1], dmap[2], m);
        colour_mix(k, dmap[3], dmap[4], dmap[5], m + 3);
        colour_mix(k, dmap[6], dmap[7], dmap[8], m + 6);
#pragma unroll
        for (int j = 0; j < kDMix; ++j) s_d[t * kDMix + j] = m[j];
        __syncthreads();   // also: every thread has taken its row out of s_p, which is reused below
        if (own) {
            float b[kDMix];
#pragma unroll
            for (int j = 0; j < kDMix; ++j) b[j] = 0.0f;
#pragma unroll
            for (int dy = 0; dy < kWin; ++dy) {
                const float w = k.g[dy];
#pragma unroll
                for (int j = 0; j < kDMix; ++j) b[j] = fmaf(w, s_d[(t - kHalf + dy) * kDMix + j], b[j]);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float ds = b[c] + 2.0f * p[c] * b[3 + c] + g[c] * b[6 + c];
                const float diff = p[c] - g[c];
                const float sgn = (diff > 0.0f ? 1.0f : 0.0f) - (diff < 0.0f ? 1.0f : 0.0f);
                s_p[(t - kHalf) * 3 + c] = k.c_l1 * sgn + k.c_ssim * ds;
            }
        }
        __syncthreads();
        // the owned rows are one contiguous run of the plane: store it with consecutive lanes
        const int rows = min(kOutRows, k.H - y0);
        for (int i = t; i < rows * 3; i += kThreads) dpred[plane + (size_t)y0 * 3 + i] = s_p[i];
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
        for (int w = 0; w < kThreads / GSR_WAVE; ++w) {
            a += s_red[w * 2];
            b += s_red[w * 2 + 1];
        }
        partial[blockIdx.x] = make_float2(a, b);
    }
}

// Sums the per-workgroup partial sums in double and writes {loss, l1, ssim} (Trainer.py:36-38).
__global__ __launch_bounds__(kThreads) void loss_finish_kernel(const float2 *__restrict__ partial, int blocks, double inv_count,
                                                               float loss_lambda, float *__restrict__ out)
{
    __shared__ double s_a[kThreads], s_b[kThreads];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < blocks; i += kThreads) {
        const float2 v = partial[i];
        a += (double)v.x;
        b += (double)v.y;
    }
    s_a[threadIdx.x] = a;
    s_b[threadIdx.x] = b;
    __syncthreads();
    for (int step = kThreads / 2; step >= 1; step >>= 1) {
        if ((int)threadIdx.x < step) {
            s_a[threadIdx.x] += s_a[threadIdx.x + step];
            s_b[threadIdx.x] += s_b[threadIdx.x + step];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float l1 = (float)(s_a[0] * inv_count), ssim = (float)(s_b[0] * inv_count);
        out[0] = (1.0f - loss_lambda) * l1 + loss_lambda * (1.0f - ssim);
        out[1] = l1;
        out[2] = ssim;
    }
}

int loss_chunks(int height) { return (height + kOutRows - 1) / kOutRows; }

}  // namespace

extern "C" size_t gsr_loss_workspace_bytes(int width, int height)
{
    if (width <= 0 || height <= 0) return 0;
    return (size_t)width * (size_t)loss_chunks(height) * sizeof(float2);
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
    LossK k;
    k.W = width;
    k.H = height;
    k.chunks = loss_chunks(height);
    const double count = (double)width * (double)height * 3.0;
    k.c_l1 = (float)((double)grad_scale * (1.0 - (double)loss_lambda) / count);
    k.c_ssim = (float)(-(double)grad_scale * (double)loss_lambda / count);
    // Utils/Loss.py:59-61: fp32 values of the Gaussian, normalised by their fp32 sum
    float sum = 0.0f;
    for (int i = 0; i < kWin; ++i) {
        k.g[i] = (float)exp(-(double)((i - kHalf) * (i - kHalf)) / (2.0 * 1.5 * 1.5));
        sum += k.g[i];
    }
    for (int i = 0; i < kWin; ++i) k.g[i] = k.g[i] / sum;

    const unsigned blocks = (unsigned)width * (unsigned)k.chunks;
    hipStream_t s = (hipStream_t)stream;
    float2 *partial = (float2 *)workspace;
    if (dpred)
        hipLaunchKernelGGL(loss_kernel<true>, dim3(blocks), dim3(kThreads), 0, s, k, pred, gt, dpred, partial);
    else
        hipLaunchKernelGGL(loss_kernel<false>, dim3(blocks), dim3(kThreads), 0, s, k, pred, gt, dpred, partial);
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(kThreads), 0, s, partial, (int)blocks, 1.0 / count, loss_lambda, out);
    return gsr_check_launch("gsr_loss_fwd_bwd");
}
