// optimizer.step() of the torch.optim.Adam the reference trainer builds (Trainer.py:24: lr, betas,
// eps 1e-8, no weight decay, no amsgrad; stepped at Trainer.py:80) for up to GSR_ADAM_MAX_TENSORS
// parameter tensors in ONE launch.  Per element, in PyTorch's operation order:
//   m   = lerp(m, g, 1 - beta1)
//   v   = v * beta2 + ((1 - beta2) * g) * g
//   den = sqrt(v) / sqrt(1 - beta2^t) + eps
//   p   = p + (-lr / (1 - beta1^t)) * (m / den)
// The scalars are formed in double precision on the host, as PyTorch forms them, and applied in
// fp32.  The kernel is a pure stream: 16 bytes read (p, g, m, v) and 12 written per element, no
// reuse, so it is bound by HBM bandwidth; tensors whose four pointers are 16-byte aligned move as
// float4.
#include "gsr_common.hpp"

#include <math.h>

namespace {

constexpr int kThreads = 256;
constexpr int kVecPerThread = 2;                               // float4 per thread
constexpr uint32_t kBlockElems = kThreads * 4 * kVecPerThread; // 2048 elements per workgroup

struct AdamK {
    float *p[GSR_ADAM_MAX_TENSORS];
    const float *g[GSR_ADAM_MAX_TENSORS];
    float *m[GSR_ADAM_MAX_TENSORS];
    float *v[GSR_ADAM_MAX_TENSORS];
    unsigned long long numel[GSR_ADAM_MAX_TENSORS];
    uint32_t block_begin[GSR_ADAM_MAX_TENSORS + 1];   // first workgroup of each tensor
    float neg_step_size[GSR_ADAM_MAX_TENSORS];        // -lr / (1 - beta1^t)
    float bias2_sqrt[GSR_ADAM_MAX_TENSORS];           // sqrt(1 - beta2^t)
    int num;
    float w, one_minus_w, beta2, one_minus_beta2, eps;
    int lerp_low;                                     // |w| < 0.5: the other branch of lerp
};

__device__ __forceinline__ void adam_element(const AdamK &k, int t, float &p, float g, float &m, float &v)
{
    const float diff = g - m;
    m = k.lerp_low ? m + k.w * diff : g - diff * k.one_minus_w;
    v = v * k.beta2 + (k.one_minus_beta2 * g) * g;
    const float den = sqrtf(v) / k.bias2_sqrt[t] + k.eps;
    p = p + k.neg_step_size[t] * (m / den);
}

__global__ __launch_bounds__(kThreads) void adam_kernel(AdamK k)
{
    int t = 0;
#pragma unroll
    for (int i = 1; i < GSR_ADAM_MAX_TENSORS; ++i)
        if (i < k.num && blockIdx.x >= k.block_begin[i]) t = i;
    float *__restrict__ p = k.p[t];
    const float *__restrict__ g = k.g[t];
    float *__restrict__ m = k.m[t];
    float *__restrict__ v = k.v[t];
    const unsigned long long n = k.numel[t];
    const unsigned long long base = (unsigned long long)(blockIdx.x - k.block_begin[t]) * kBlockElems;
    const bool aligned = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15u) == 0;
    if (aligned && base + kBlockElems <= n) {
#pragma unroll
        for (int j = 0; j < kVecPerThread; ++j) {
            const unsigned long long e = base + 4ull * (threadIdx.x + (unsigned)j * kThreads);
            float4 pp = *reinterpret_cast<const float4 *>(p + e);
            const float4 gg = *reinterpret_cast<const float4 *>(g + e);
            float4 mm = *reinterpret_cast<const float4 *>(m + e);
            float4 vv = *reinterpret_cast<const float4 *>(v + e);
            adam_element(k, t, pp.x, gg.x, mm.x, vv.x);
            adam_element(k, t, pp.y, gg.y, mm.y, vv.y);
            adam_element(k, t, pp.z, gg.z, mm.z, vv.z);
            adam_element(k, t, pp.w, gg.w, mm.w, vv.w);
            *reinterpret_cast<float4 *>(p + e) = pp;
            *reinterpret_cast<float4 *>(m + e) = mm;
            *reinterpret_cast<float4 *>(v + e) = vv;
        }
    } else {
        for (uint32_t j = threadIdx.x; j < kBlockElems; j += kThreads) {
            const unsigned long long e = base + j;
            if (e >= n) break;
            float pp = p[e], mm = m[e], vv = v[e];
            adam_element(k, t, pp, g[e], mm, vv);
            p[e] = pp; m[e] = mm; v[e] = vv;
        }
    }
}

}  // namespace

extern "C" int gsr_adam_step(uint32_t num_tensors, float *const *param_host, const float *const *grad_host,
                             float *const *exp_avg_host, float *const *exp_avg_sq_host, const uint64_t *numel_host,
                             const uint32_t *step_host, double lr, double beta1, double beta2, double eps, void *stream)
{
    if (num_tensors == 0) return GSR_OK;
    if (num_tensors > GSR_ADAM_MAX_TENSORS) {
        gsr_set_error("gsr_adam_step: %u tensors, at most %d per call", num_tensors, GSR_ADAM_MAX_TENSORS);
        return GSR_ERR_ARG;
    }
    if (!param_host || !grad_host || !exp_avg_host || !exp_avg_sq_host || !numel_host || !step_host) {
        gsr_set_error("gsr_adam_step: null pointer");
        return GSR_ERR_ARG;
    }
    if (!(lr >= 0.0) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0)) {
        gsr_set_error("gsr_adam_step: lr %g, betas (%g, %g), eps %g out of range", lr, beta1, beta2, eps);
        return GSR_ERR_ARG;
    }
    AdamK k;
    k.num = 0;
    unsigned long long blocks = 0;
    for (uint32_t i = 0; i < num_tensors; ++i) {
        if (numel_host[i] == 0) continue;
        if (!param_host[i] || !grad_host[i] || !exp_avg_host[i] || !exp_avg_sq_host[i]) {
            gsr_set_error("gsr_adam_step: tensor %u has a null pointer", i);
            return GSR_ERR_ARG;
        }
        if (step_host[i] == 0) { gsr_set_error("gsr_adam_step: tensor %u: the step count starts at 1", i); return GSR_ERR_ARG; }
        const int t = k.num++;
        k.p[t] = param_host[i]; k.g[t] = grad_host[i]; k.m[t] = exp_avg_host[i]; k.v[t] = exp_avg_sq_host[i];
        k.numel[t] = numel_host[i];
        k.block_begin[t] = (uint32_t)blocks;
        blocks += (numel_host[i] + kBlockElems - 1) / kBlockElems;
        const double step = (double)step_host[i];
        const double bias1 = 1.0 - pow(beta1, step), bias2 = 1.0 - pow(beta2, step);
        k.neg_step_size[t] = (float)(-(lr / bias1));
        k.bias2_sqrt[t] = (float)sqrt(bias2);
    }
    if (blocks >= (1ull << 31)) { gsr_set_error("gsr_adam_step: too many elements"); return GSR_ERR_ARG; }
    if (k.num == 0) return GSR_OK;
    for (int t = k.num; t < GSR_ADAM_MAX_TENSORS; ++t) {
        k.p[t] = nullptr; k.g[t] = nullptr; k.m[t] = nullptr; k.v[t] = nullptr; k.numel[t] = 0;
        k.neg_step_size[t] = 0.0f; k.bias2_sqrt[t] = 1.0f;
    }
    for (int t = k.num; t <= GSR_ADAM_MAX_TENSORS; ++t) k.block_begin[t] = (uint32_t)blocks;
    k.w = (float)(1.0 - beta1);
    k.one_minus_w = 1.0f - k.w;
    k.lerp_low = fabsf(k.w) < 0.5f;
    k.beta2 = (float)beta2;
    k.one_minus_beta2 = (float)(1.0 - beta2);
    k.eps = (float)eps;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, k);
    return gsr_check_launch("adam_step");
}
