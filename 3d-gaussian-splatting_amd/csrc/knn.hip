// Mean squared distance of every point to its three nearest neighbours: the quantity the reference
// takes from simple_knn's distCUDA2 to size the initial Gaussians (GaussianModel.py:82-83).  That
// extension (Inria's simple-knn, not part of the reference tree and not pinned by it) finds the exact
// three nearest OTHER points (the point's own index is skipped, coincident points count with distance
// zero) with a Morton-order search and returns (d1 + d2 + d3) / 3 of the squared distances.
//
// Here the search is exhaustive: a point cloud from structure-from-motion has 10^4..10^6 points and
// this runs once per training run.  One thread per query point; the candidates stream through LDS in
// tiles of 1024 points as float4 (x, y, z, -), every lane reading the same candidate in a step
// (an LDS broadcast), three running minima per thread.  2 N^2 subtract/multiply-add triples:
// compute-bound on the vector ALU (0.2 s for a million points).
#include "gsr_common.hpp"

#include <float.h>

namespace {

constexpr int kThreads = 256;
constexpr int kTile = 1024;

__global__ __launch_bounds__(kThreads) void knn_dist2_kernel(uint32_t n, const float *__restrict__ points,
                                                            float *__restrict__ out)
{
    __shared__ float4 tile[kTile];
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    const bool live = i < n;
    const float qx = live ? points[3 * (size_t)i] : 0.0f, qy = live ? points[3 * (size_t)i + 1] : 0.0f,
                qz = live ? points[3 * (size_t)i + 2] : 0.0f;
    float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;   // ascending
    for (uint32_t base = 0; base < n; base += kTile) {
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < kTile; t += kThreads) {
            const uint32_t j = base + t;
            // padding far away: never among the three nearest of a real point
            tile[t] = j < n ? make_float4(points[3 * (size_t)j], points[3 * (size_t)j + 1], points[3 * (size_t)j + 2], 0.0f)
                            : make_float4(FLT_MAX, FLT_MAX, FLT_MAX, 0.0f);
        }
        __syncthreads();
        const uint32_t count = min((uint32_t)kTile, n - base);
        // the query's own slot in this tile (if any) is skipped by index, as simple_knn does
        const uint32_t self = i - base;   // wraps to a huge value when the query is not in this tile
#pragma unroll 8
        for (uint32_t t = 0; t < count; ++t) {
            const float4 c = tile[t];
            const float dx = c.x - qx, dy = c.y - qy, dz = c.z - qz;
            float d = dx * dx + dy * dy + dz * dz;
            d = t == self ? FLT_MAX : d;
            // insert into the sorted triple
            const float m0 = fminf(b0, d), r0 = fmaxf(b0, d);
            const float m1 = fminf(b1, r0), r1 = fmaxf(b1, r0);
            b0 = m0; b1 = m1; b2 = fminf(b2, r1);
        }
    }
    if (live) out[i] = (b0 + b1 + b2) / 3.0f;
}

}  // namespace

extern "C" int gsr_knn_dist2(uint32_t n, const float *points, float *dist2, void *stream)
{
    if (n == 0) return GSR_OK;
    if (!points || !dist2) { gsr_set_error("gsr_knn_dist2: null pointer"); return GSR_ERR_ARG; }
    if (n < 4) { gsr_set_error("gsr_knn_dist2: %u points, at least 4 are needed for three neighbours", n); return GSR_ERR_ARG; }
    hipLaunchKernelGGL(knn_dist2_kernel, dim3(gsr_div_up(n, kThreads)), dim3(kThreads), 0, (hipStream_t)stream, n, points,
                       dist2);
    return gsr_check_launch("knn_dist2");
}
