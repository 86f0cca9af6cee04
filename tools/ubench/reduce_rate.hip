// Micro-benchmark: the wave reduction of the nine per-entry gradient sums of the compositing backward.
// MODE 0: merged cross-lane reduction as in csrc/composite.hip (permlane swaps + DPP), one store.
// MODE 1: through LDS - every lane deposits its nine partial sums, after three entries the 27 sums are
//         taken by two lanes each reading 32 deposits (transposed read), one store per sum.
// Prints cycles per entry per SIMD at four waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int CTRL>
__device__ inline float dpp_add(float val)
{
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(val), CTRL, 0xF, 0xF, false);
    return val + __int_as_float(moved);
}
__device__ inline float merge_halves(float a, float b)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ inline float merge_rows(float a, float b)
{
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ inline float merge_half_rows(float a, float b)
{
    const float sa = dpp_add<0x128>(a), sb = dpp_add<0x128>(b);
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(sa), __float_as_int(sb), 0xE4, 0xF, 0xC, false));
}
__device__ inline float sum8(float x) { return dpp_add<0x141>(dpp_add<0x4E>(dpp_add<0xB1>(x))); }
__device__ inline float sum16(float x) { return dpp_add<0x140>(sum8(x)); }

constexpr int kBatch = 3;
constexpr int kRow = 64 + 4;   // floats per deposit row: 64 lanes + a skew of 16 bytes

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k(int iters, float *out)
{
    __shared__ __attribute__((aligned(16))) float lds[4 * kBatch * 9 * kRow];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *mine = lds + wave * (kBatch * 9 * kRow);
    float v[9];
    for (int j = 0; j < 9; ++j) v[j] = 0.001f * (lane + 3 * j + 1);
    float *dst = out + ((size_t)blockIdx.x * 4 + wave) * 64;
    float keep = 0.0f;
    for (int it = 0; it < iters; ++it) {
        // new values per entry (cheap, equal in both modes)
        for (int j = 0; j < 9; ++j) v[j] = v[j] * 1.0001f + 0.5f;
        if (MODE == 0) {
            const float m0 = merge_halves(v[0], v[4]), m1 = merge_halves(v[2], v[6]), m2 = merge_halves(v[1], v[5]),
                        m3 = merge_halves(v[3], v[7]);
            const float n0 = merge_rows(m0, m1), n1 = merge_rows(m2, m3);
            const float all8 = sum8(merge_half_rows(n0, n1));
            const float blue = sum16(v[8]);
            const bool main = (lane & 7) == 0;
            const float val = main ? all8 : blue;
            if (main || (lane & 15) == 1) dst[(lane >> 3) + (main ? 0 : 8)] = val;
        } else if (MODE == 2) {     // as MODE 0 without the ninth value
            const float m0 = merge_halves(v[0], v[4]), m1 = merge_halves(v[2], v[6]), m2 = merge_halves(v[1], v[5]),
                        m3 = merge_halves(v[3], v[7]);
            const float n0 = merge_rows(m0, m1), n1 = merge_rows(m2, m3);
            const float all8 = sum8(merge_half_rows(n0, n1));
            if ((lane & 7) == 0) dst[lane >> 3] = all8;
        } else if (MODE == 3) {     // as MODE 0 without the store
            const float m0 = merge_halves(v[0], v[4]), m1 = merge_halves(v[2], v[6]), m2 = merge_halves(v[1], v[5]),
                        m3 = merge_halves(v[3], v[7]);
            const float n0 = merge_rows(m0, m1), n1 = merge_rows(m2, m3);
            const float all8 = sum8(merge_half_rows(n0, n1));
            const float blue = sum16(v[8]);
            keep += all8 + blue;
        } else if (MODE == 4) {     // only the per-entry value update (the floor of this benchmark)
            keep += v[3];
        } else {
            const int e = it % kBatch;
            float *row = mine + e * 9 * kRow + lane;
#pragma unroll
            for (int j = 0; j < 9; ++j) row[j * kRow] = v[j];
            if (e == kBatch - 1) {
                // 27 sums, two lanes each: lane 2 s + h reads deposits 32 h .. 32 h + 31 of sum s
                const int s = lane >> 1, h = lane & 1;
                float acc = 0.0f;
                if (s < kBatch * 9) {
                    const float4 *src = reinterpret_cast<const float4 *>(mine + s * kRow + 32 * h);
                    float4 a = src[0];
#pragma unroll
                    for (int q = 1; q < 8; ++q) { const float4 b = src[q]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
                    acc = (a.x + a.y) + (a.z + a.w);
                }
                acc = dpp_add<0xB1>(acc);   // the neighbour's half
                if (h == 0 && s < kBatch * 9) dst[s] = acc;
            }
        }
        keep += v[0];
    }
    if (keep == 123.0f) dst[0] = keep;
}

int main()
{
    const int blocks = 256 * 4, iters = 3000;    // 16 waves per CU = 4 per SIMD
    float *out;
    hipMalloc(&out, (size_t)blocks * 4 * 64 * sizeof(float));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 5; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, iters, out);
            else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, iters, out);
            else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, iters, out);
            else if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, iters, out);
            else hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, iters, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            // 4 waves per SIMD share the SIMD: cycles per entry per SIMD = time * clock / (iters * 4)
            if (rep) printf("mode %d: %.3f ms, %.1f cycles per entry per SIMD at 2.4 GHz\n", mode, ms, ms * 1e-3 * 2.4e9 / (iters * 4.0));
        }
    }
    return 0;
}
