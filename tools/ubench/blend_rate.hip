// Micro-benchmark: the per-entry arithmetic of the forward blend (no memory traffic), every SIMD
// full (8 waves).  Record values live in SGPRs and change per iteration through scalar integer
// ops only, as in the real kernel where they arrive by scalar loads.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ inline f2 splat(float s) { return f2{s, s}; }
__device__ inline f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ inline float clamp_alpha(float raw) { return fminf(fmaxf(raw, 0.01f), 0.99f); }
struct RecS { float gx, gy, A, B, D, o, r, g, b; };

template <int MODE, int NSALU, int NBR>
__global__ __launch_bounds__(256) void k(RecS R0, int iters, float *out)
{
    const int lane = threadIdx.x & 63;
    const float pxf = (float)(lane >> 2);
    f2 pyf[2], T[2], cr[2], cg[2], cb[2];
    for (int h = 0; h < 2; ++h) {
        pyf[h] = f2{(float)((lane & 3) * 4 + 2 * h), (float)((lane & 3) * 4 + 2 * h + 1)};
        T[h] = splat(1.0f); cr[h] = cg[h] = cb[h] = splat(0.0f);
    }
    for (int it = 0; it < iters; ++it) {
        RecS R = R0;
        const int tw = __builtin_amdgcn_readfirstlane(it) & 3;
        R.gx = __int_as_float(__float_as_int(R0.gx) ^ tw);
        R.gy = __int_as_float(__float_as_int(R0.gy) ^ (tw << 1));
        R.o = __int_as_float(__float_as_int(R0.o) ^ tw);
        R.r = __int_as_float(__float_as_int(R0.r) ^ tw);
        // extra scalar work per entry: NSALU dependent scalar adds, NBR taken scalar branches
        if (NSALU > 0) {
            int acc = it;
#pragma unroll
            for (int q = 0; q < NSALU; ++q) asm volatile("s_add_i32 %0, %0, 3" : "+s"(acc));
            R.g = __int_as_float(__float_as_int(R.g) ^ (acc & 1));
        }
        if (NBR > 0) {
#pragma unroll
            for (int q = 0; q < NBR; ++q) asm volatile("s_cmp_eq_u32 0, 0\n\ts_cbranch_scc1 1f\n\ts_nop 0\n\ts_nop 0\n1:" ::: "scc");
        }
        if (MODE == 0) {   // full path
            const float dx = R.gx - pxf;
            const float adx2 = R.A * dx * dx;
            const float bdx = R.B * dx;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f2 dy = splat(R.gy) - pyf[h];
                const f2 q = pk_fma(pk_fma(splat(R.D), dy, splat(bdx)), dy, splat(adx2));
                const f2 G = f2{__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};
                const f2 raw = splat(R.o) * G;
                const f2 a = f2{clamp_alpha(raw.x), clamp_alpha(raw.y)};
                const f2 wgt = a * T[h];
                cr[h] = pk_fma(wgt, splat(R.r), cr[h]);
                cg[h] = pk_fma(wgt, splat(R.g), cg[h]);
                cb[h] = pk_fma(wgt, splat(R.b), cb[h]);
                T[h] = T[h] * (splat(1.0f) - a);
            }
        } else {           // all-clamped-low path
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f2 wgt = splat(0.01f) * T[h];
                cr[h] = pk_fma(wgt, splat(R.r), cr[h]);
                cg[h] = pk_fma(wgt, splat(R.g), cg[h]);
                cb[h] = pk_fma(wgt, splat(R.b), cb[h]);
                T[h] = T[h] * splat(0.99f);
            }
        }
    }
    float s = 0;
    for (int h = 0; h < 2; ++h) s += T[h].x + T[h].y + cr[h].x + cr[h].y + cg[h].x + cg[h].y + cb[h].x + cb[h].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int NSALU, int NBR>
void run(const char *name, float *out)
{
    const int blocks = 256 * 8, iters = 20000;
    RecS R = {7.3f, 5.1f, -0.02f, 0.003f, -0.015f, 0.7f, 0.5f, 0.4f, 0.3f};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NSALU, NBR>), dim3(blocks), dim3(256), 0, 0, R, iters, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NSALU, NBR>), dim3(blocks), dim3(256), 0, 0, R, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-18s %8.3f ms  %7.1f cycles per wave-entry per SIMD (at 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / (iters * 8.0));
}
int main()
{
    float *out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    run<0, 0, 0>("blend full", out); run<1, 0, 0>("blend clamped-low", out);
    run<0, 30, 0>("full + 30 salu", out); run<1, 30, 0>("low + 30 salu", out);
    run<0, 0, 4>("full + 4 branches", out); run<1, 0, 4>("low + 4 branches", out);
    run<0, 30, 4>("full + 30 salu + 4 br", out); run<1, 30, 4>("low + 30 salu + 4 br", out);
    return 0;
}
