// Micro-benchmark: the per-entry arithmetic of the compositing backward (no memory traffic), as in
// csrc/composite.hip: two full halves + the merged wave reduction, or two clamped-low halves + the
// colour reduction.  Record values live in SGPRs and change per iteration through scalar integer ops.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ inline f2 splat(float s) { return f2{s, s}; }
__device__ inline f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ inline float clamp_alpha(float raw) { return fminf(fmaxf(raw, 0.01f), 0.99f); }
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

struct RecS { float gx, gy, A, B, D, o, r, g, b, ca, cbh, cd; };

template <int MODE>
__global__ __launch_bounds__(256) void k(RecS R0, int iters, float *out)
{
    const int lane = threadIdx.x & 63;
    const float pxf = (float)(lane >> 2);
    f2 pyf[2], T[2], gr[2], gg[2], gb[2], suffix[2];
    for (int h = 0; h < 2; ++h) {
        pyf[h] = f2{(float)((lane & 3) + 8 * h), (float)((lane & 3) + 8 * h + 4)};
        T[h] = splat(0.5f); suffix[h] = splat(0.f);
        gr[h] = splat(0.1f + lane); gg[h] = splat(0.2f + lane); gb[h] = splat(0.3f);
    }
    float sink = 0.f;
    for (int it = 0; it < iters; ++it) {
        RecS R = R0;
        const int tw = __builtin_amdgcn_readfirstlane(it) & 3;
        R.gx = __int_as_float(__float_as_int(R0.gx) ^ tw);
        R.gy = __int_as_float(__float_as_int(R0.gy) ^ (tw << 1));
        R.o = __int_as_float(__float_as_int(R0.o) ^ tw);
        R.r = __int_as_float(__float_as_int(R0.r) ^ tw);
        const f2 z = splat(0.f);
        f2 s_vx = z, s_vy = z, s_xx = z, s_xy = z, s_yy = z, s_o = z, s_r = z, s_g = z, s_b = z;
        if (MODE == 0) {
            const float dx = R.gx - pxf;
            const float adx2 = R.A * dx * dx;
            const float bdx = R.B * dx;
            const float cadx = R.ca * dx, cbdx = R.cbh * dx;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f2 dy = splat(R.gy) - pyf[h];
                const f2 q = pk_fma(pk_fma(splat(R.D), dy, splat(bdx)), dy, splat(adx2));
                const f2 G = f2{__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};
                const f2 raw = splat(R.o) * G;
                const f2 a = f2{clamp_alpha(raw.x), clamp_alpha(raw.y)};
                const f2 om = splat(1.0f) - a;
                const f2 rcp = f2{__builtin_amdgcn_rcpf(om.x), __builtin_amdgcn_rcpf(om.y)};
                const f2 t0 = T[h] * rcp;
                f2 Tk = pk_fma(pk_fma(-om, t0, T[h]), rcp, t0);
                f2 wgt = a * Tk;
                const f2 cgd = pk_fma(splat(R.b), gb[h], pk_fma(splat(R.g), gg[h], splat(R.r) * gr[h]));
                f2 d_a = pk_fma(Tk, cgd, -(suffix[h] * rcp));
                suffix[h] = pk_fma(wgt, cgd, suffix[h]);
                T[h] = Tk;
                s_r = pk_fma(wgt, gr[h], s_r); s_g = pk_fma(wgt, gg[h], s_g); s_b = pk_fma(wgt, gb[h], s_b);
                const bool p0 = (a.x == raw.x), p1 = (a.y == raw.y);
                const f2 d_raw = f2{p0 ? d_a.x : 0.0f, p1 ? d_a.y : 0.0f};
                s_o = pk_fma(d_raw, G, s_o);
                const f2 hh = d_raw * a;
                const f2 vx = pk_fma(splat(R.cbh), dy, splat(cadx));
                const f2 vy = pk_fma(splat(R.cd), dy, splat(cbdx));
                const f2 hvx = hh * vx, hvy = hh * vy;
                s_vx += hvx; s_vy += hvy;
                s_xx = pk_fma(hvx, vx, s_xx); s_xy = pk_fma(hvx, vy, s_xy); s_yy = pk_fma(hvy, vy, s_yy);
            }
            const float m0 = merge_halves(s_vx.x + s_vx.y, s_yy.x + s_yy.y), m1 = merge_halves(s_xx.x + s_xx.y, s_r.x + s_r.y);
            const float m2 = merge_halves(s_vy.x + s_vy.y, s_o.x + s_o.y), m3 = merge_halves(s_xy.x + s_xy.y, s_g.x + s_g.y);
            const float all8 = sum8(merge_half_rows(merge_rows(m0, m1), merge_rows(m2, m3)));
            const float blue = sum16(s_b.x + s_b.y);
            sink += (lane & 7) == 0 ? all8 : blue;
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float om = 0.99f, rcp = 1.0f / 0.99f;
                const f2 t0 = T[h] * splat(rcp);
                f2 Tk = pk_fma(pk_fma(splat(-om), t0, T[h]), splat(rcp), t0);
                f2 wgt = splat(0.01f) * Tk;
                const f2 cgd = pk_fma(splat(R.b), gb[h], pk_fma(splat(R.g), gg[h], splat(R.r) * gr[h]));
                suffix[h] = pk_fma(wgt, cgd, suffix[h]);
                T[h] = Tk * splat(0.5f);
                s_r = pk_fma(wgt, gr[h], s_r); s_g = pk_fma(wgt, gg[h], s_g); s_b = pk_fma(wgt, gb[h], s_b);
            }
            sink += sum16(merge_rows(merge_halves(s_r.x + s_r.y, s_g.x + s_g.y), s_b.x + s_b.y));
        }
    }
    float s = sink;
    for (int h = 0; h < 2; ++h) s += T[h].x + T[h].y + suffix[h].x + suffix[h].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, float *out, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd, iters = 10000;
    RecS R = {7.3f, 5.1f, -0.02f, 0.003f, -0.015f, 0.7f, 0.5f, 0.4f, 0.3f, 0.03f, -0.002f, 0.02f};
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, R, iters, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, R, iters, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-22s %d waves/SIMD %8.3f ms  %7.1f cycles per wave-entry per SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / ((double)iters * waves_per_simd));
}
int main()
{
    float *out; (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    for (int w : {8, 5, 4}) { run<0>("bwd full+reduce", out, w); run<1>("bwd low+colour reduce", out, w); }
    return 0;
}
