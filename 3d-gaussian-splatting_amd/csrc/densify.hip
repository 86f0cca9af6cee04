// Densification bookkeeping of the Gaussian set on the device: the statistics gathered after every
// backward (GaussianModel.update_densify_stats, GaussianModel.py:230-232), the clone / split pass
// (GaussianModel.densify, clone, split: GaussianModel.py:179-245) and the in-place remove
// (GaussianModel.py:148-164).  All of it is row bookkeeping bound by HBM bandwidth: flags, an
// exclusive scan for the destination rows (scan_sort.hip) and row copies.  Rows are the reference's
// parameter rows: coords 3, scale 3, rotation 4, opacity 1, sh 48 floats.
//
// The reference's behaviour is kept as it is, including where it differs from the published
// method: the scale test compares the RAW (log) scale with 0.01 * scene radius
// (GaussianModel.py:240,244); the centres of a split Gaussian are the rotated samples alone - the
// 4x4 rotation has no translation, so nothing is added at GaussianModel.py:212; remove writes
// ACTIVATED opacity / rotation / scale into the raw tensors and compacts only the tail.
#include "gsr_common.hpp"

#include <math.h>

namespace {

constexpr int kThreads = 256;
constexpr uint8_t kNone = 0, kClone = 1, kSplit = 2;
constexpr int kShVec = 12;   // float4 per row of sh[.,3,16]

__global__ __launch_bounds__(kThreads) void stats_kernel(uint32_t n, const uint8_t *__restrict__ mask,
                                                         const float *__restrict__ g_screen,
                                                         float *__restrict__ coords_grads, int32_t *__restrict__ denom)
{
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n || !mask[i]) return;
    const float gx = g_screen[3 * (size_t)i], gy = g_screen[3 * (size_t)i + 1];
    coords_grads[i] += sqrtf(gx * gx + gy * gy);   // torch.norm(grad[:, :2], dim=-1)
    denom[i] += 1;
}

// stats_kernel for the views of one step in one launch: a thread adds its row's contributions in the order of the
// views, which is the order of the float additions of one launch per view.
struct StatsViews {
    int nv;
    const uint8_t *mask[GSR_MAX_VIEWS];
    const float *g_screen[GSR_MAX_VIEWS];
};
__global__ __launch_bounds__(kThreads) void stats_views_kernel(uint32_t n, StatsViews a, float *__restrict__ coords_grads,
                                                               int32_t *__restrict__ denom)
{
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    float sum = 0.0f;
    int32_t seen = 0;
    bool any = false;
    for (int v = 0; v < a.nv; ++v) {
        if (!a.mask[v][i]) continue;
        const float gx = a.g_screen[v][3 * (size_t)i], gy = a.g_screen[v][3 * (size_t)i + 1];
        if (!any) { sum = coords_grads[i]; any = true; }
        sum += sqrtf(gx * gx + gy * gy);
        ++seen;
    }
    if (any) { coords_grads[i] = sum; denom[i] += seen; }
}

// GaussianModel.py:236-244 for the rows below the size.  A row is cloned when its mean gradient
// reaches the threshold and its largest raw scale is at most the scale threshold, split when the
// scale is larger: the two sets are disjoint and a cloned copy can never be split, so one pass over
// the rows present before the clone decides both.
__global__ __launch_bounds__(kThreads) void classify_kernel(uint32_t size, const float *__restrict__ coords_grads,
                                                            const int32_t *__restrict__ denom,
                                                            const float *__restrict__ scale_raw, float grad_thr,
                                                            float scale_thr, uint8_t *__restrict__ cls,
                                                            uint32_t *__restrict__ clone_flag,
                                                            uint32_t *__restrict__ split_flag)
{
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= size) return;
    float g = coords_grads[i] / (float)denom[i];
    if (g != g) g = 0.0f;                                           // grads[grads.isnan()] = 0
    const float *s = scale_raw + 3 * (size_t)i;
    const float smax = fmaxf(fmaxf(s[0], s[1]), s[2]);              // torch.max(..., dim=1).values
    const bool hot = fabsf(g) >= grad_thr;
    const bool has_nan = (s[0] != s[0]) || (s[1] != s[1]) || (s[2] != s[2]);   // torch.max propagates nan: both tests fail
    uint8_t c = kNone;
    if (hot && !has_nan) c = (smax <= scale_thr) ? kClone : kSplit;
    cls[i] = c;
    clone_flag[i] = c == kClone;
    split_flag[i] = c == kSplit;
}

struct Rows {
    float *coords, *scale, *rot, *opac, *sh;
};

// Utils/Transform.py:55-63 on the normalised quaternion (GaussianModel.py:56), row-major 3x3.
__device__ inline void rotation_of(const float *q_raw, float *R)
{
    const float a = q_raw[0], b = q_raw[1], c = q_raw[2], d = q_raw[3];
    const float nrm = fmaxf(sqrtf(((a * a + b * b) + c * c) + d * d), 1e-12f);
    const float w = a / nrm, x = b / nrm, y = c / nrm, z = d / nrm;
    R[0] = 2.0f * (w * w + x * x) - 1.0f; R[1] = 2.0f * (x * y - w * z);        R[2] = 2.0f * (x * z + w * y);
    R[3] = 2.0f * (x * y + w * z);        R[4] = 2.0f * (w * w + y * y) - 1.0f; R[5] = 2.0f * (y * z - w * x);
    R[6] = 2.0f * (x * z - w * y);        R[7] = 2.0f * (y * z + w * x);        R[8] = 2.0f * (w * w + z * z) - 1.0f;
}

// One thread per row present before the pass: the 11 small floats of a cloned or split row.
__global__ __launch_bounds__(kThreads) void densify_rows_kernel(uint32_t size, uint32_t num_clone, uint32_t num_split,
                                                                const uint8_t *__restrict__ cls,
                                                                const uint32_t *__restrict__ clone_rank,
                                                                const uint32_t *__restrict__ split_rank,
                                                                const float *__restrict__ noise, Rows r)
{
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= size) return;
    const uint8_t c = cls[i];
    if (c == kNone) return;
    const size_t src = i;
    if (c == kClone) {                                              // GaussianModel.py:189-193
        const size_t dst = (size_t)size + clone_rank[i];
        for (int j = 0; j < 3; ++j) { r.coords[3 * dst + j] = r.coords[3 * src + j]; r.scale[3 * dst + j] = r.scale[3 * src + j]; }
        for (int j = 0; j < 4; ++j) r.rot[4 * dst + j] = r.rot[4 * src + j];
        r.opac[dst] = r.opac[src];
        return;
    }
    // split, GaussianModel.py:209-224: two samples of N(0, diag(std^2)) rotated by R; sample k of the
    // reference's [2 * num_split, 3] draw belongs to split row k mod num_split
    const uint32_t k = split_rank[i];
    const size_t dst = (size_t)size + num_clone + k;
    float R[9], std[3], lg[3];
    rotation_of(r.rot + 4 * src, R);
    for (int j = 0; j < 3; ++j) {
        std[j] = (float)exp((double)r.scale[3 * src + j]);          // the activated scale
        lg[j] = logf(std[j] / 1.6f);                                // log(scale / (0.8 * split_factor)), :213
    }
    for (int h = 0; h < 2; ++h) {
        const float *z = noise + 3 * ((size_t)h * num_split + k);
        const float s0 = z[0] * std[0], s1 = z[1] * std[1], s2 = z[2] * std[2];
        const size_t row = h == 0 ? src : dst;
        for (int j = 0; j < 3; ++j) {
            float acc = 0.0f;                                       // row vector times R^T
            acc += s0 * R[3 * j + 0];
            acc += s1 * R[3 * j + 1];
            acc += s2 * R[3 * j + 2];
            r.coords[3 * row + j] = acc + 0.0f;
            r.scale[3 * row + j] = lg[j];
        }
    }
    for (int j = 0; j < 4; ++j) r.rot[4 * dst + j] = r.rot[4 * src + j];
    r.opac[dst] = r.opac[src];
}

// The 48 sh floats of a cloned or split row: one thread per float4.
__global__ __launch_bounds__(kThreads) void densify_sh_kernel(uint32_t size, uint32_t num_clone,
                                                              const uint8_t *__restrict__ cls,
                                                              const uint32_t *__restrict__ clone_rank,
                                                              const uint32_t *__restrict__ split_rank, float4 *sh)
{
    const unsigned long long t = (unsigned long long)blockIdx.x * kThreads + threadIdx.x;
    const unsigned long long i = t / kShVec;
    if (i >= size) return;
    const uint8_t c = cls[i];
    if (c == kNone) return;
    const uint32_t j = (uint32_t)(t - i * kShVec);
    const size_t dst = c == kClone ? (size_t)size + clone_rank[i] : (size_t)size + num_clone + split_rank[i];
    sh[dst * kShVec + j] = sh[i * kShVec + j];
}

// ---- remove (GaussianModel.py:148-164) -----------------------------------------------------------
constexpr int kRowFloats = 61;   // 59 parameter floats + coords_grads + denom (as bits)

__global__ __launch_bounds__(kThreads) void keep_flag_kernel(uint32_t size, const uint8_t *__restrict__ mask,
                                                             uint32_t *__restrict__ keep_flag)
{
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i < size) keep_flag[i] = mask[i] ? 0u : 1u;
}

// Kept row with rank k among the kept rows goes to row k when k >= size - 2 r (the last r kept rows land
// in rows [size - 2r, size - r)).  Sources and destinations overlap, so the rows are staged first.
__global__ __launch_bounds__(kThreads) void remove_stage_kernel(uint32_t size, uint32_t first_dst,
                                                                const uint8_t *__restrict__ mask,
                                                                const uint32_t *__restrict__ keep_rank, Rows r,
                                                                const float *__restrict__ coords_grads,
                                                                const int32_t *__restrict__ denom,
                                                                float *__restrict__ staging)
{
    const unsigned long long t = (unsigned long long)blockIdx.x * kThreads + threadIdx.x;
    const unsigned long long i = t / 64;      // one wave per row, lane = float of the row (61 of 64 active)
    if (i >= size || mask[i]) return;
    const uint32_t k = keep_rank[i];
    if (k < first_dst) return;
    const int e = (int)(t & 63);
    if (e >= kRowFloats) return;
    float val;
    if (e < 3) val = r.coords[3 * i + e];
    else if (e < 6) val = (float)exp((double)r.scale[3 * i + (e - 3)]);            // self.scale: exp
    else if (e < 10) {                                                             // self.rotation: normalised
        const float *q = r.rot + 4 * i;
        const float nrm = fmaxf(sqrtf(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]), 1e-12f);
        val = q[e - 6] / nrm;
    } else if (e == 10) val = 1.0f / (1.0f + expf(-r.opac[i]));                    // self.opacity: sigmoid
    else if (e < 59) val = r.sh[48 * i + (e - 11)];
    else if (e == 59) val = coords_grads[i];
    else val = __int_as_float(denom[i]);
    staging[(size_t)(k - first_dst) * kRowFloats + e] = val;
}

__global__ __launch_bounds__(kThreads) void remove_commit_kernel(uint32_t num_rows, uint32_t first_dst,
                                                                 const float *__restrict__ staging, Rows r,
                                                                 float *__restrict__ coords_grads,
                                                                 int32_t *__restrict__ denom)
{
    const unsigned long long t = (unsigned long long)blockIdx.x * kThreads + threadIdx.x;
    const unsigned long long j = t / 64;
    if (j >= num_rows) return;
    const int e = (int)(t & 63);
    if (e >= kRowFloats) return;
    const size_t i = (size_t)first_dst + j;
    const float val = staging[j * kRowFloats + e];
    if (e < 3) r.coords[3 * i + e] = val;
    else if (e < 6) r.scale[3 * i + (e - 3)] = val;
    else if (e < 10) r.rot[4 * i + (e - 6)] = val;
    else if (e == 10) r.opac[i] = val;
    else if (e < 59) r.sh[48 * i + (e - 11)] = val;
    else if (e == 59) coords_grads[i] = val;
    else denom[i] = __float_as_int(val);
}

}  // namespace

extern "C" int gsr_densify_stats(uint32_t n, const uint8_t *mask, const float *g_screen, float *coords_grads,
                                 int32_t *denom, void *stream)
{
    if (n == 0) return GSR_OK;
    if (!mask || !g_screen || !coords_grads || !denom) { gsr_set_error("gsr_densify_stats: null pointer"); return GSR_ERR_ARG; }
    hipLaunchKernelGGL(stats_kernel, dim3(gsr_div_up(n, kThreads)), dim3(kThreads), 0, (hipStream_t)stream, n, mask,
                       g_screen, coords_grads, denom);
    return gsr_check_launch("densify_stats");
}

extern "C" int gsr_densify_stats_views(uint32_t n, uint32_t num_views, const uint8_t *const *mask_host,
                                       const float *const *g_screen_host, float *coords_grads, int32_t *denom, void *stream)
{
    if (n == 0 || num_views == 0) return GSR_OK;
    if (!mask_host || !g_screen_host || !coords_grads || !denom) { gsr_set_error("gsr_densify_stats_views: null pointer"); return GSR_ERR_ARG; }
    for (uint32_t v0 = 0; v0 < num_views; v0 += GSR_MAX_VIEWS) {
        StatsViews a;
        a.nv = (int)((num_views - v0 < (uint32_t)GSR_MAX_VIEWS) ? num_views - v0 : (uint32_t)GSR_MAX_VIEWS);
        for (int k = 0; k < GSR_MAX_VIEWS; ++k) {
            a.mask[k] = k < a.nv ? mask_host[v0 + k] : nullptr;
            a.g_screen[k] = k < a.nv ? g_screen_host[v0 + k] : nullptr;
            if (k < a.nv && (!a.mask[k] || !a.g_screen[k])) {
                gsr_set_error("gsr_densify_stats_views: null pointer for view %u", v0 + k);
                return GSR_ERR_ARG;
            }
        }
        hipLaunchKernelGGL(stats_views_kernel, dim3(gsr_div_up(n, kThreads)), dim3(kThreads), 0, (hipStream_t)stream, n, a,
                           coords_grads, denom);
    }
    return gsr_check_launch("densify_stats_views");
}

extern "C" size_t gsr_densify_select_workspace_bytes(uint32_t size) { return gsr_scan_workspace_bytes(size); }

extern "C" int gsr_densify_select(uint32_t size, const float *coords_grads, const int32_t *denom, const float *scale_raw,
                                  float grad_threshold, float scale_threshold, uint8_t *cls, uint32_t *clone_rank,
                                  uint32_t *split_rank, uint32_t *counts, void *workspace, size_t workspace_bytes,
                                  void *stream)
{
    if (!counts) { gsr_set_error("gsr_densify_select: null pointer"); return GSR_ERR_ARG; }
    if (size == 0) {
        (void)hipMemsetAsync(counts, 0, 2 * sizeof(uint32_t), (hipStream_t)stream);
        return GSR_OK;
    }
    if (!coords_grads || !denom || !scale_raw || !cls || !clone_rank || !split_rank) {
        gsr_set_error("gsr_densify_select: null pointer");
        return GSR_ERR_ARG;
    }
    hipLaunchKernelGGL(classify_kernel, dim3(gsr_div_up(size, kThreads)), dim3(kThreads), 0, (hipStream_t)stream, size,
                       coords_grads, denom, scale_raw, grad_threshold, scale_threshold, cls, clone_rank, split_rank);
    if (int e = gsr_check_launch("densify_select")) return e;
    if (int e = gsr_exclusive_scan(clone_rank, clone_rank, size, counts, workspace, workspace_bytes, stream)) return e;
    return gsr_exclusive_scan(split_rank, split_rank, size, counts + 1, workspace, workspace_bytes, stream);
}

static int check_rows(const char *what, float *coords, float *scale_raw, float *rot_raw, float *opac_raw, float *sh)
{
    if (!coords || !scale_raw || !rot_raw || !opac_raw || !sh) { gsr_set_error("%s: null pointer", what); return GSR_ERR_ARG; }
    if (((uintptr_t)sh) & 15u) { gsr_set_error("%s: sh must be 16-byte aligned", what); return GSR_ERR_ARG; }
    return GSR_OK;
}

extern "C" int gsr_densify_apply(uint32_t size, uint32_t num_clone, uint32_t num_split, const uint8_t *cls,
                                 const uint32_t *clone_rank, const uint32_t *split_rank, const float *noise,
                                 float *coords, float *scale_raw, float *rot_raw, float *opac_raw, float *sh,
                                 void *stream)
{
    if (size == 0 || (num_clone == 0 && num_split == 0)) return GSR_OK;
    if (int e = check_rows("gsr_densify_apply", coords, scale_raw, rot_raw, opac_raw, sh)) return e;
    if (!cls || !clone_rank || !split_rank || (num_split && !noise)) {
        gsr_set_error("gsr_densify_apply: null pointer");
        return GSR_ERR_ARG;
    }
    if ((unsigned long long)size + num_clone + num_split >= (1ull << 31)) {
        gsr_set_error("gsr_densify_apply: %u + %u + %u rows exceed the 31-bit row index", size, num_clone, num_split);
        return GSR_ERR_ARG;
    }
    const Rows r{coords, scale_raw, rot_raw, opac_raw, sh};
    hipLaunchKernelGGL(densify_rows_kernel, dim3(gsr_div_up(size, kThreads)), dim3(kThreads), 0, (hipStream_t)stream, size,
                       num_clone, num_split, cls, clone_rank, split_rank, noise, r);
    const unsigned long long threads = (unsigned long long)size * kShVec;
    hipLaunchKernelGGL(densify_sh_kernel, dim3((unsigned)((threads + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       (hipStream_t)stream, size, num_clone, cls, clone_rank, split_rank, reinterpret_cast<float4 *>(sh));
    return gsr_check_launch("densify_apply");
}

extern "C" size_t gsr_remove_select_workspace_bytes(uint32_t size) { return gsr_scan_workspace_bytes(size); }

extern "C" int gsr_remove_select(uint32_t size, const uint8_t *mask, uint32_t *keep_rank, uint32_t *num_kept,
                                 void *workspace, size_t workspace_bytes, void *stream)
{
    if (!num_kept) { gsr_set_error("gsr_remove_select: null pointer"); return GSR_ERR_ARG; }
    if (size == 0) {
        (void)hipMemsetAsync(num_kept, 0, sizeof(uint32_t), (hipStream_t)stream);
        return GSR_OK;
    }
    if (!mask || !keep_rank) { gsr_set_error("gsr_remove_select: null pointer"); return GSR_ERR_ARG; }
    hipLaunchKernelGGL(keep_flag_kernel, dim3(gsr_div_up(size, kThreads)), dim3(kThreads), 0, (hipStream_t)stream, size,
                       mask, keep_rank);
    if (int e = gsr_check_launch("remove_select")) return e;
    return gsr_exclusive_scan(keep_rank, keep_rank, size, num_kept, workspace, workspace_bytes, stream);
}

extern "C" size_t gsr_remove_apply_workspace_bytes(uint32_t num_removed)
{
    return (size_t)num_removed * kRowFloats * sizeof(float);
}

extern "C" int gsr_remove_apply(uint32_t size, uint32_t num_removed, const uint8_t *mask, const uint32_t *keep_rank,
                                float *coords, float *scale_raw, float *rot_raw, float *opac_raw, float *sh,
                                float *coords_grads, int32_t *denom, void *workspace, size_t workspace_bytes,
                                void *stream)
{
    if (num_removed == 0) return GSR_OK;                            // GaussianModel.py:150-151
    if ((unsigned long long)num_removed * 2ull > size) {
        gsr_set_error("gsr_remove_apply: %u of %u rows to remove: more than half (the reference fails on its slice shapes there)",
                      num_removed, size);
        return GSR_ERR_ARG;
    }
    if (int e = check_rows("gsr_remove_apply", coords, scale_raw, rot_raw, opac_raw, sh)) return e;
    if (!mask || !keep_rank || !coords_grads || !denom || !workspace) {
        gsr_set_error("gsr_remove_apply: null pointer");
        return GSR_ERR_ARG;
    }
    if (workspace_bytes < gsr_remove_apply_workspace_bytes(num_removed)) {
        gsr_set_error("gsr_remove_apply: workspace of %zu bytes, %zu needed", workspace_bytes,
                      gsr_remove_apply_workspace_bytes(num_removed));
        return GSR_ERR_ARG;
    }
    const Rows r{coords, scale_raw, rot_raw, opac_raw, sh};
    const uint32_t first_dst = size - 2 * num_removed;
    float *staging = static_cast<float *>(workspace);
    const unsigned long long t1 = (unsigned long long)size * 64ull, t2 = (unsigned long long)num_removed * 64ull;
    hipLaunchKernelGGL(remove_stage_kernel, dim3((unsigned)((t1 + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       (hipStream_t)stream, size, first_dst, mask, keep_rank, r, coords_grads, denom, staging);
    hipLaunchKernelGGL(remove_commit_kernel, dim3((unsigned)((t2 + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       (hipStream_t)stream, num_removed, first_dst, staging, r, coords_grads, denom);
    return gsr_check_launch("remove_apply");
}
