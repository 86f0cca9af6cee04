// Micro-benchmark: how the record stream of a view can be written.
//   A  sorted order (one thread per sorted position p): word[p] -> Gaussian g -> gather the 48-byte record rec[g] ->
//      64-byte stream record at stream[p] (consecutive).  The gathers are random in a 48 MB table.
//   B  emission order (one thread per emission slot e; the slots of a Gaussian are consecutive): rec[owner[e]] is read
//      once per Gaussian by neighbouring threads, the stream record goes to stream[pos[e]] (64 aligned bytes at a
//      random place).  Needs pos[] = the inverse of the sort's permutation:
//   C  pos[slot[p]] = p (4-byte writes at random).
// Sizes of the headline workload: 1 M Gaussians, 5.15 M pairs.  Prints the average time of each kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <numeric>
#include <random>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ inline void store16(float4 *dst, float4 v)
{
    if (NT) __builtin_nontemporal_store(v4f{v.x, v.y, v.z, v.w}, reinterpret_cast<v4f *>(dst));
    else *dst = v;
}

// A as the library does it: the 64 records of a wave are staged in LDS and leave as four fully coalesced non-temporal
// stores of 1 KiB; STRIDE = floats between two rows of the table (12: packed, 16: one 64-byte sector per record)
template <int STRIDE>
__global__ __launch_bounds__(256) void sorted_staged(const uint32_t *__restrict__ gauss_at, const float *__restrict__ rec,
                                                     uint32_t n, float *__restrict__ stream)
{
    __shared__ float4 tile[256 * 4];
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    float4 *wave_tile = tile + (threadIdx.x - lane) * 4;
    const uint32_t wave_first = p - lane;
    if (wave_first >= n) return;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a, c = a;
    if (p < n) {
        const float4 *src = reinterpret_cast<const float4 *>(rec + (size_t)gauss_at[p] * STRIDE);
        a = src[0]; b = src[1]; c = src[2];
    }
    wave_tile[lane * 4 + 0] = a; wave_tile[lane * 4 + 1] = b; wave_tile[lane * 4 + 2] = c; wave_tile[lane * 4 + 3] = make_float4(0.f, 0.f, 0.f, 0.f);
    __builtin_amdgcn_wave_barrier();
    float4 *dst = reinterpret_cast<float4 *>(stream + (size_t)wave_first * 16);
    const uint32_t wave_f4 = min(64u, n - wave_first) * 4u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t f = q * 64 + lane;
        if (f < wave_f4) store16<true>(dst + f, wave_tile[f]);
    }
}

template <bool NT>
__global__ __launch_bounds__(256) void sorted_order(const uint32_t *__restrict__ gauss_at, const float *__restrict__ rec,
                                                    uint32_t n, float *__restrict__ stream)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t g = gauss_at[p];
    const float4 *src = reinterpret_cast<const float4 *>(rec + (size_t)g * 12);
    const float4 a = src[0], b = src[1], c = src[2];
    float4 *dst = reinterpret_cast<float4 *>(stream + (size_t)p * 16);
    store16<NT>(dst + 0, a); store16<NT>(dst + 1, b); store16<NT>(dst + 2, c); store16<NT>(dst + 3, make_float4(0.f, 0.f, 0.f, 0.f));
}

template <bool NT>
__global__ __launch_bounds__(256) void emission_order(const uint32_t *__restrict__ owner, const uint32_t *__restrict__ pos,
                                                      const float *__restrict__ rec, uint32_t n, float *__restrict__ stream)
{
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const uint32_t g = owner[e], p = pos[e];
    const float4 *src = reinterpret_cast<const float4 *>(rec + (size_t)g * 12);
    const float4 a = src[0], b = src[1], c = src[2];
    float4 *dst = reinterpret_cast<float4 *>(stream + (size_t)p * 16);
    store16<NT>(dst + 0, a); store16<NT>(dst + 1, b); store16<NT>(dst + 2, c); store16<NT>(dst + 3, make_float4(0.f, 0.f, 0.f, 0.f));
}

__global__ __launch_bounds__(256) void inverse(const uint32_t *__restrict__ slot_at, uint32_t n, uint32_t *__restrict__ pos)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) pos[slot_at[p]] = p;
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <class F>
static float timed(F launch, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main()
{
    const uint32_t N = 1000000;
    std::mt19937 rng(1);
    // tiles per Gaussian: 1 + geometric-like, mean about 5.15
    std::vector<uint32_t> owner;
    owner.reserve(5300000);
    std::exponential_distribution<double> ex(1.0 / 4.15);
    for (uint32_t g = 0; g < N; ++g) {
        const uint32_t nt = 1u + (uint32_t)ex(rng);
        for (uint32_t k = 0; k < nt; ++k) owner.push_back(g);   // emission order: depth order = index order here
    }
    const uint32_t I = (uint32_t)owner.size();
    // the depth order is unrelated to the index: give the Gaussians of the emission order random table rows
    std::vector<uint32_t> row(N);
    std::iota(row.begin(), row.end(), 0u);
    std::shuffle(row.begin(), row.end(), rng);
    for (auto &o : owner) o = row[o];
    std::vector<uint32_t> slot_at(I), pos(I), gauss_at(I);
    std::iota(slot_at.begin(), slot_at.end(), 0u);
    std::shuffle(slot_at.begin(), slot_at.end(), rng);        // sorted position p holds emission slot slot_at[p]
    for (uint32_t p = 0; p < I; ++p) { pos[slot_at[p]] = p; gauss_at[p] = owner[slot_at[p]]; }
    printf("%u Gaussians, %u pairs\n", N, I);
    uint32_t *d_owner, *d_pos, *d_slot_at, *d_gauss_at, *d_pos_out;
    float *d_rec, *d_stream;
    CHECK(hipMalloc(&d_owner, I * 4)); CHECK(hipMalloc(&d_pos, I * 4)); CHECK(hipMalloc(&d_slot_at, I * 4));
    CHECK(hipMalloc(&d_gauss_at, I * 4)); CHECK(hipMalloc(&d_pos_out, I * 4));
    CHECK(hipMalloc(&d_rec, (size_t)N * 64)); CHECK(hipMalloc(&d_stream, (size_t)I * 64));
    CHECK(hipMemcpy(d_owner, owner.data(), I * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_pos, pos.data(), I * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_slot_at, slot_at.data(), I * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_gauss_at, gauss_at.data(), I * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(d_rec, 0, (size_t)N * 64));
    const dim3 grid((I + 255) / 256), block(256);
    printf("A  sorted order, staged, 48-byte rows  %.4f ms\n", timed([&] { hipLaunchKernelGGL(sorted_staged<12>, grid, block, 0, 0, d_gauss_at, d_rec, I, d_stream); }, 20));
    printf("A  sorted order, staged, 64-byte rows  %.4f ms\n", timed([&] { hipLaunchKernelGGL(sorted_staged<16>, grid, block, 0, 0, d_gauss_at, d_rec, I, d_stream); }, 20));
    printf("A  sorted order, plain stores          %.4f ms\n", timed([&] { hipLaunchKernelGGL(sorted_order<false>, grid, block, 0, 0, d_gauss_at, d_rec, I, d_stream); }, 20));
    printf("A  sorted order, non-temporal stores   %.4f ms\n", timed([&] { hipLaunchKernelGGL(sorted_order<true>, grid, block, 0, 0, d_gauss_at, d_rec, I, d_stream); }, 20));
    printf("B  emission order, plain stores        %.4f ms\n", timed([&] { hipLaunchKernelGGL(emission_order<false>, grid, block, 0, 0, d_owner, d_pos, d_rec, I, d_stream); }, 20));
    printf("B  emission order, non-temporal stores %.4f ms\n", timed([&] { hipLaunchKernelGGL(emission_order<true>, grid, block, 0, 0, d_owner, d_pos, d_rec, I, d_stream); }, 20));
    printf("C  inverse permutation (4-byte scatter) %.4f ms\n", timed([&] { hipLaunchKernelGGL(inverse, grid, block, 0, 0, d_slot_at, I, d_pos_out); }, 20));
    CHECK(hipDeviceSynchronize());
    return 0;
}
