/*
 * CPU oracle for the Gaussian-splat render path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file restates, in plain C, the algorithm of the reference renderer
 * (GaussianRenderer.py:52-113 and the helpers it pulls in).  It exists to
 * check the HIP kernels and to be timed as the CPU baseline; the shipped
 * library never links, loads or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every stage of
 * this file against fixtures produced by running the reference itself
 * (tests/golden/make_golden.py).
 *
 * Arithmetic: fp32, one IEEE operation per reference tensor op, in the
 * reference's order, so that the discrete results (visibility mask, integer
 * bounding boxes, tile lists, sort order) come out bit-identical:
 *   - matmuls that ATen lowers to sgemm ([N,4]@[4,4], [M*2,3]@[3,3]) are a
 *     k-ordered fused-multiply-add chain   acc = a0*b0; acc = fma(ak,bk,acc);
 *   - matmuls that ATen lowers to its small-matrix bmm loop (3x3 / 2x3
 *     batched) are   acc = 0; acc = acc + ak*bk   with separate roundings;
 *   - F.normalize sums the squares left to right;
 *   - exp() of the log-scales is the correctly rounded fp32 value
 *     ((float)exp((double)x)); the reference's vendor exp differs from it by
 *     one ulp in about 1 % of inputs, which moves a bounding-box edge with
 *     probability ~1e-8.
 * Compile with -ffp-contract=off so the compiler adds no fusions of its own.
 * Per-Gaussian gradient sums over pixels are accumulated in double so the
 * oracle's own rounding stays far below the 1e-4 comparison tolerance.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int32_t n;            /* Gaussians (model size, not capacity) */
    int32_t sh_degree;    /* 0..3 */
    int32_t width;        /* slow image axis (x) */
    int32_t height;       /* fast image axis (y) */
    int32_t tile_length;  /* edge of a reference tile in pixels */
    float w2m[16];        /* world->model, row-major */
    float proj[16];       /* projection, row-major */
    float cam_pos[3];
} gso_view;

#define GSO_OK 0
#define GSO_ERR_SINGULAR (-2)
#define GSO_ERR_ARG (-1)

/* ---- SH constants (Utils/SphericalHarmonic.py:14-31) ---- */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

static inline float exp_cr(float x) { return (float)exp((double)x); }

/* SH basis values b[0..K) for an un-normalised direction (Utils/SphericalHarmonic.py:62-88). */
static void sh_basis(int deg, float x, float y, float z, float *b)
{
    b[0] = SH_C0;
    if (deg < 1) return;
    b[1] = -SH_C1 * y; b[2] = SH_C1 * z; b[3] = -SH_C1 * x;
    if (deg < 2) return;
    float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    b[4] = SH_C2[0] * xy; b[5] = SH_C2[1] * yz; b[6] = SH_C2[2] * (2.0f * zz - xx - yy);
    b[7] = SH_C2[3] * xz; b[8] = SH_C2[4] * (xx - yy);
    if (deg < 3) return;
    b[9] = SH_C3[0] * y * (3.0f * xx - yy);
    b[10] = SH_C3[1] * xy * z;
    b[11] = SH_C3[2] * y * (4.0f * zz - xx - yy);
    b[12] = SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
    b[13] = SH_C3[4] * x * (4.0f * zz - xx - yy);
    b[14] = SH_C3[5] * z * (xx - yy);
    b[15] = SH_C3[6] * x * (xx - 3.0f * yy);
}

/* d(basis_k)/d(x,y,z) */
static void sh_basis_grad(int deg, float x, float y, float z, float (*g)[3])
{
    for (int k = 0; k < 16; ++k) g[k][0] = g[k][1] = g[k][2] = 0.0f;
    if (deg < 1) return;
    g[1][1] = -SH_C1; g[2][2] = SH_C1; g[3][0] = -SH_C1;
    if (deg < 2) return;
    float xx = x * x, yy = y * y, zz = z * z;
    g[4][0] = SH_C2[0] * y; g[4][1] = SH_C2[0] * x;
    g[5][1] = SH_C2[1] * z; g[5][2] = SH_C2[1] * y;
    g[6][0] = SH_C2[2] * -2.0f * x; g[6][1] = SH_C2[2] * -2.0f * y; g[6][2] = SH_C2[2] * 4.0f * z;
    g[7][0] = SH_C2[3] * z; g[7][2] = SH_C2[3] * x;
    g[8][0] = SH_C2[4] * 2.0f * x; g[8][1] = SH_C2[4] * -2.0f * y;
    if (deg < 3) return;
    g[9][0] = SH_C3[0] * 6.0f * x * y; g[9][1] = SH_C3[0] * (3.0f * xx - 3.0f * yy);
    g[10][0] = SH_C3[1] * y * z; g[10][1] = SH_C3[1] * x * z; g[10][2] = SH_C3[1] * x * y;
    g[11][0] = SH_C3[2] * -2.0f * x * y; g[11][1] = SH_C3[2] * (4.0f * zz - xx - 3.0f * yy);
    g[11][2] = SH_C3[2] * 8.0f * y * z;
    g[12][0] = SH_C3[3] * -6.0f * x * z; g[12][1] = SH_C3[3] * -6.0f * y * z;
    g[12][2] = SH_C3[3] * (6.0f * zz - 3.0f * xx - 3.0f * yy);
    g[13][0] = SH_C3[4] * (4.0f * zz - 3.0f * xx - yy); g[13][1] = SH_C3[4] * -2.0f * x * y;
    g[13][2] = SH_C3[4] * 8.0f * x * z;
    g[14][0] = SH_C3[5] * 2.0f * x * z; g[14][1] = SH_C3[5] * -2.0f * y * z; g[14][2] = SH_C3[5] * (xx - yy);
    g[15][0] = SH_C3[6] * (3.0f * xx - 3.0f * yy); g[15][1] = SH_C3[6] * -6.0f * x * y;
}

/* Intermediate per-Gaussian geometry shared by forward and backward. */
typedef struct {
    float s[3];        /* exp(scale_raw) */
    float qn;          /* max(|q|, eps) */
    float q[4];        /* normalised quaternion */
    float R[3][3];
    float RS[3][3];
    float cov3[3][3];
    float m[4];        /* ph @ w2m^T */
    float c[4];        /* m @ proj^T  (clip space) */
    float ndc[3];
    float J[2][3];
    float M[2][3];     /* J @ W */
    float t2[2][3];    /* M @ cov3 */
    float t3[2][3];    /* t2 @ W^T */
    float cov2[4];     /* row-major 2x2 */
} geom_t;

/* row i of  v[4] @ A^T  as sgemm does it: k-ordered fma chain */
static inline float dot4_fma(const float *v, const float *row)
{
    float acc = v[0] * row[0];
    acc = fmaf(v[1], row[1], acc);
    acc = fmaf(v[2], row[2], acc);
    acc = fmaf(v[3], row[3], acc);
    return acc;
}

static void geometry(const gso_view *v, const float *p, const float *sr, const float *qr, geom_t *g)
{
    /* GaussianModel.py:50-56: scale = exp(_scale), rotation = F.normalize(_rotation) */
    for (int k = 0; k < 3; ++k) g->s[k] = exp_cr(sr[k]);
    float ss = qr[0] * qr[0];
    ss = ss + qr[1] * qr[1];
    ss = ss + qr[2] * qr[2];
    ss = ss + qr[3] * qr[3];
    float nrm = sqrtf(ss);
    g->qn = nrm > 1e-12f ? nrm : 1e-12f;
    for (int k = 0; k < 4; ++k) g->q[k] = qr[k] / g->qn;
    float w = g->q[0], x = g->q[1], y = g->q[2], z = g->q[3];
    /* Utils/Transform.py:55-63 */
    g->R[0][0] = 2.0f * (w * w + x * x) - 1.0f; g->R[0][1] = 2.0f * (x * y - w * z); g->R[0][2] = 2.0f * (x * z + w * y);
    g->R[1][0] = 2.0f * (x * y + w * z); g->R[1][1] = 2.0f * (w * w + y * y) - 1.0f; g->R[1][2] = 2.0f * (y * z - w * x);
    g->R[2][0] = 2.0f * (x * z - w * y); g->R[2][1] = 2.0f * (y * z + w * x); g->R[2][2] = 2.0f * (w * w + z * z) - 1.0f;
    /* GaussianModel.py:107-108: (R S)(R S)^T through the small-matrix bmm loop */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) g->RS[i][j] = g->R[i][j] * g->s[j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float acc = 0.0f;
            for (int k = 0; k < 3; ++k) acc = acc + g->RS[i][k] * g->RS[j][k];
            g->cov3[i][j] = acc;
        }
    /* Utils/Camera.py:101-103 */
    float ph[4] = {p[0], p[1], p[2], 1.0f};
    for (int j = 0; j < 4; ++j) g->m[j] = dot4_fma(ph, &v->w2m[4 * j]);
    for (int j = 0; j < 4; ++j) g->c[j] = dot4_fma(g->m, &v->proj[4 * j]);
    for (int j = 0; j < 3; ++j) g->ndc[j] = g->c[j] / g->c[3];
    /* Utils/Camera.py:81-87: J from the WORLD position, W = w2m[:3,:3]^T */
    memset(g->J, 0, sizeof(g->J));
    g->J[0][0] = 1.0f / p[2]; g->J[0][2] = (-p[0]) / p[2];
    g->J[1][1] = 1.0f / p[2]; g->J[1][2] = (-p[1]) / p[2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j) {  /* sgemm: W[k][j] = w2m[j][k] */
            float acc = g->J[i][0] * v->w2m[4 * j + 0];
            acc = fmaf(g->J[i][1], v->w2m[4 * j + 1], acc);
            acc = fmaf(g->J[i][2], v->w2m[4 * j + 2], acc);
            g->M[i][j] = acc;
        }
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j) {  /* bmm loop */
            float acc = 0.0f;
            for (int k = 0; k < 3; ++k) acc = acc + g->M[i][k] * g->cov3[k][j];
            g->t2[i][j] = acc;
        }
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j) {  /* sgemm: W^T[k][j] = w2m[k][j] */
            float acc = g->t2[i][0] * v->w2m[4 * 0 + j];
            acc = fmaf(g->t2[i][1], v->w2m[4 * 1 + j], acc);
            acc = fmaf(g->t2[i][2], v->w2m[4 * 2 + j], acc);
            g->t3[i][j] = acc;
        }
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) {  /* bmm loop: J^T[k][j] = J[j][k] */
            float acc = 0.0f;
            for (int k = 0; k < 3; ++k) acc = acc + g->t3[i][k] * g->J[j][k];
            g->cov2[2 * i + j] = acc;
        }
}

/* Zero-pivot test of a 2x2 LU with partial pivoting (what torch.linalg.inv raises on,
 * Utils/Probability.py:15). */
static int lu2_singular(const float *c)
{
    float a = c[0], b = c[1], cc = c[2], d = c[3];
    if (fabsf(a) >= fabsf(cc)) {
        if (a == 0.0f) return 1;
        return (d - (cc / a) * b) == 0.0f;
    }
    return (b - (a / cc) * d) == 0.0f;
}

/*
 * Stage 1: activations, projection, cull mask, 2-D covariance, colour, radius, integer bbox.
 * Everything is N-indexed; rows of culled Gaussians are zero.
 *   mask[N] u8, pos2d[N,3] (pixel x, pixel y, NDC depth), cov2d[N,4], conic[N,4] (inverse),
 *   opacity[N], color[N,3], radius[N], bbox[N,4] = xmin,ymin,xmax,ymax.
 * screen_coords may be NULL (treated as zeros; GaussianRenderer.py:64).
 * Returns the number of visible Gaussians, or GSO_ERR_SINGULAR when a Gaussian that
 * lands in at least one tile has a singular 2-D covariance.
 */
int64_t gso_preprocess_fwd(const gso_view *v, const float *coords, const float *scale_raw,
                           const float *rot_raw, const float *opac_raw, const float *sh,
                           const float *screen_coords, uint8_t *mask, float *pos2d, float *cov2d,
                           float *conic, float *opacity, float *color, float *radius, int32_t *bbox)
{
    const int n = v->n;
    const float Wf = (float)v->width, Hf = (float)v->height;
    const int K = (v->sh_degree + 1) * (v->sh_degree + 1);
    int64_t visible = 0;
    int singular = 0;
    if (v->sh_degree < 0 || v->sh_degree > 3) return GSO_ERR_ARG;
#pragma omp parallel for schedule(static) reduction(+ : visible) reduction(| : singular)
    for (int i = 0; i < n; ++i) {
        geom_t g;
        const float *p = coords + 3 * i;
        geometry(v, p, scale_raw + 3 * i, rot_raw + 4 * i, &g);
        mask[i] = 0;
        pos2d[3 * i] = pos2d[3 * i + 1] = pos2d[3 * i + 2] = 0.0f;
        for (int k = 0; k < 4; ++k) { cov2d[4 * i + k] = 0.0f; conic[4 * i + k] = 0.0f; bbox[4 * i + k] = 0; }
        opacity[i] = 0.0f; radius[i] = 0.0f;
        color[3 * i] = color[3 * i + 1] = color[3 * i + 2] = 0.0f;
        /* Utils/Camera.py:106-108 */
        int vis = (g.ndc[2] >= 0.1f) && (-1.3f <= g.ndc[0]) && (g.ndc[0] <= 1.3f) &&
                  (-1.3f <= g.ndc[1]) && (g.ndc[1] <= 1.3f);
        if (!vis) continue;
        mask[i] = 1;
        ++visible;
        float sx = 0.f, sy = 0.f, sz = 0.f;
        if (screen_coords) { sx = screen_coords[3 * i]; sy = screen_coords[3 * i + 1]; sz = screen_coords[3 * i + 2]; }
        /* Utils/Camera.py:111-115 */
        float px = sx + g.ndc[0], py = sy + g.ndc[1], pz = sz + g.ndc[2];
        px = (0.5f * (px + 1.0f)) * Wf;
        py = (1.0f - 0.5f * (py + 1.0f)) * Hf;
        pos2d[3 * i] = px; pos2d[3 * i + 1] = py; pos2d[3 * i + 2] = pz;
        for (int k = 0; k < 4; ++k) cov2d[4 * i + k] = g.cov2[k];
        /* GaussianModel.py:58-60 */
        opacity[i] = 1.0f / (1.0f + expf(-opac_raw[i]));
        /* GaussianRenderer.py:71-74 */
        float b[16];
        float dx = p[0] - v->cam_pos[0], dy = p[1] - v->cam_pos[1], dz = p[2] - v->cam_pos[2];
        sh_basis(v->sh_degree, dx, dy, dz, b);
        for (int c = 0; c < 3; ++c) {
            const float *co = sh + (size_t)i * 48 + 16 * c;
            float acc = 0.0f;
            for (int k = 0; k < K; ++k) acc += b[k] * co[k];
            acc += 0.5f;
            color[3 * i + c] = acc < 0.0f ? 0.0f : acc;  /* NaN passes through like torch.clamp */
        }
        /* GaussianRenderer.py:40-44 */
        float a = g.cov2[0], bb = g.cov2[1], cc = g.cov2[2], d = g.cov2[3];
        float trace = a + d;
        float det = a * d - bb * cc;
        float term1 = 0.5f * trace;
        float disc = trace * trace - 4.0f * det;
        if (disc < 0.0f) disc = 0.0f;
        float term2 = 0.5f * sqrtf(disc);
        float lo = term1 - term2, hi = term1 + term2;
        float r = 3.0f * sqrtf(lo > hi ? lo : hi);
        radius[i] = r;
        /* Utils/Screen.py:33-36: clamp then truncate; NaN becomes an empty box */
        float e[4] = {px - r, py - r, px + r, py + r};
        float lim[4] = {Wf, Hf, Wf, Hf};
        int32_t bx[4];
        int bad = 0;
        for (int k = 0; k < 4; ++k) {
            float t = e[k];
            if (t != t) { bad = 1; t = 0.0f; }
            if (t < 0.0f) t = 0.0f;
            if (t > lim[k]) t = lim[k];
            bx[k] = (int32_t)t;
        }
        if (bad) bx[0] = bx[1] = bx[2] = bx[3] = 0;
        for (int k = 0; k < 4; ++k) bbox[4 * i + k] = bx[k];
        int in_tiles = (bx[2] > bx[0]) && (bx[3] > bx[1]);
        /* analytic inverse of the (slightly asymmetric) 2x2 */
        float idet = 1.0f / det;
        conic[4 * i + 0] = d * idet; conic[4 * i + 1] = -bb * idet;
        conic[4 * i + 2] = -cc * idet; conic[4 * i + 3] = a * idet;
        if (in_tiles && lu2_singular(g.cov2)) singular |= 1;
    }
    return singular ? GSO_ERR_SINGULAR : visible;
}

static inline void tile_span(int lo, int hi, int L, int *t0, int *t1)
{
    /* Utils/Screen.py:58-62 in closed form: tiles lo/L .. (hi-1)/L when hi > lo, else none */
    if (hi > lo) { *t0 = lo / L; *t1 = (hi - 1) / L; }
    else { *t0 = 0; *t1 = -1; }
}

int64_t gso_num_tiles(const gso_view *v)
{
    int L = v->tile_length;
    int64_t nbw = (v->width + L - 1) / L, nbh = (v->height + L - 1) / L;
    return nbw * nbh;
}

/* Per-tile membership counts; tile id = yi + nbh*xi (Utils/Screen.py:38-39).  Returns total. */
int64_t gso_bin_count(const gso_view *v, const uint8_t *mask, const int32_t *bbox, int64_t *tile_counts)
{
    int L = v->tile_length;
    int nbh = (v->height + L - 1) / L;
    int64_t T = gso_num_tiles(v), total = 0;
    memset(tile_counts, 0, sizeof(int64_t) * T);
    for (int i = 0; i < v->n; ++i) {
        if (!mask[i]) continue;
        int x0, x1, y0, y1;
        tile_span(bbox[4 * i], bbox[4 * i + 2], L, &x0, &x1);
        tile_span(bbox[4 * i + 1], bbox[4 * i + 3], L, &y0, &y1);
        for (int xi = x0; xi <= x1; ++xi)
            for (int yi = y0; yi <= y1; ++yi) { tile_counts[yi + (int64_t)nbh * xi]++; total++; }
    }
    return total;
}

typedef struct { float depth; int32_t idx; } sort_item;
static int cmp_item(const void *a, const void *b)
{
    const sort_item *x = (const sort_item *)a, *y = (const sort_item *)b;
    if (x->depth < y->depth) return -1;
    if (x->depth > y->depth) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);  /* stable: ascending Gaussian index */
}

/*
 * Per-tile lists of Gaussian indices, ascending NDC depth, ties by ascending index
 * (GaussianRenderer.py:85-89 with a stable argsort).  tile_offsets[T+1], tile_lists[total].
 */
int gso_bin_sort(const gso_view *v, const uint8_t *mask, const int32_t *bbox, const float *pos2d,
                 int64_t *tile_offsets, int32_t *tile_lists)
{
    int L = v->tile_length;
    int nbh = (v->height + L - 1) / L;
    int64_t T = gso_num_tiles(v);
    int64_t *cnt = (int64_t *)malloc(sizeof(int64_t) * (T + 1));
    int64_t total = gso_bin_count(v, mask, bbox, cnt);
    tile_offsets[0] = 0;
    for (int64_t t = 0; t < T; ++t) tile_offsets[t + 1] = tile_offsets[t] + cnt[t];
    sort_item *items = (sort_item *)malloc(sizeof(sort_item) * (total > 0 ? total : 1));
    int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (T + 1));
    memcpy(cur, tile_offsets, sizeof(int64_t) * (T + 1));
    for (int i = 0; i < v->n; ++i) {
        if (!mask[i]) continue;
        int x0, x1, y0, y1;
        tile_span(bbox[4 * i], bbox[4 * i + 2], L, &x0, &x1);
        tile_span(bbox[4 * i + 1], bbox[4 * i + 3], L, &y0, &y1);
        for (int xi = x0; xi <= x1; ++xi)
            for (int yi = y0; yi <= y1; ++yi) {
                int64_t t = yi + (int64_t)nbh * xi;
                items[cur[t]].depth = pos2d[3 * i + 2];
                items[cur[t]].idx = i;
                cur[t]++;
            }
    }
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t t = 0; t < T; ++t) {
        int64_t b = tile_offsets[t], e = tile_offsets[t + 1];
        qsort(items + b, (size_t)(e - b), sizeof(sort_item), cmp_item);
        for (int64_t k = b; k < e; ++k) tile_lists[k] = items[k].idx;
    }
    free(items); free(cnt); free(cur);
    return GSO_OK;
}

/* Sensitivity probe for the tests: the exponent of every pair is multiplied by this (default 1).
 * A value like 1 + 1e-7 mimics another correct fp32 evaluation of G and shows which differences
 * come from discrete clamp decisions rather than from rounding. */
static double gso_exponent_scale = 1.0;
void gso_set_exponent_scale(double s) { gso_exponent_scale = s; }
/* Second probe: uniform relative noise of this amplitude on every transmittance the backward uses. */
static double gso_t_noise = 0.0;
void gso_set_t_noise(double s) { gso_t_noise = s; }

#define GSO_CAT_(a, b) a##b
#define GSO_CAT(a, b) GSO_CAT_(a, b)

#define REAL float
#define GSO_EXP expf
#define GSO_NAME(x) GSO_CAT(x, _f32)
#include "gso_composite.inc"
#undef REAL
#undef GSO_EXP
#undef GSO_NAME

#define REAL double
#define GSO_EXP exp
#define GSO_NAME(x) GSO_CAT(x, _f64)
#include "gso_composite.inc"
#undef REAL
#undef GSO_EXP
#undef GSO_NAME

/* precise != 0 selects the double-precision pair arithmetic */
int gso_composite_fwd(const gso_view *v, const int64_t *tile_offsets, const int32_t *tile_lists,
                      const float *pos2d, const float *conic, const float *opacity, const float *color,
                      float *img, int precise)
{
    return precise ? gso_composite_fwd_f64(v, tile_offsets, tile_lists, pos2d, conic, opacity, color, img)
                   : gso_composite_fwd_f32(v, tile_offsets, tile_lists, pos2d, conic, opacity, color, img);
}

int gso_composite_bwd(const gso_view *v, const int64_t *tile_offsets, const int32_t *tile_lists,
                      const float *pos2d, const float *conic, const float *opacity, const float *color,
                      const float *dimg, double *d_pos2d, double *d_conic, double *d_opacity,
                      double *d_color, int precise)
{
    return precise ? gso_composite_bwd_f64(v, tile_offsets, tile_lists, pos2d, conic, opacity, color, dimg,
                                           d_pos2d, d_conic, d_opacity, d_color)
                   : gso_composite_bwd_f32(v, tile_offsets, tile_lists, pos2d, conic, opacity, color, dimg,
                                           d_pos2d, d_conic, d_opacity, d_color);
}

/*
 * Stage 1 backward: chain the stage-3 gradients to the raw parameters.
 * Outputs (fp32, overwritten): g_coords[N,3], g_scale[N,3], g_rot[N,4], g_opac[N],
 * g_sh[N,3,16] (only the first K coefficients non-zero), g_screen[N,3] (NDC units; column 2 and
 * rows of culled Gaussians exactly 0).
 */
int gso_preprocess_bwd(const gso_view *v, const float *coords, const float *scale_raw,
                       const float *rot_raw, const float *opac_raw, const float *sh,
                       const float *screen_coords, const uint8_t *mask, const double *d_pos2d,
                       const double *d_conic, const double *d_opacity, const double *d_color,
                       float *g_coords, float *g_scale, float *g_rot, float *g_opac, float *g_sh,
                       float *g_screen)
{
    const int n = v->n;
    const int K = (v->sh_degree + 1) * (v->sh_degree + 1);
    const float Wf = (float)v->width, Hf = (float)v->height;
    (void)screen_coords;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        float *gc = g_coords + 3 * i, *gs = g_scale + 3 * i, *gr = g_rot + 4 * i, *gsc = g_screen + 3 * i;
        gc[0] = gc[1] = gc[2] = 0.f; gs[0] = gs[1] = gs[2] = 0.f; gr[0] = gr[1] = gr[2] = gr[3] = 0.f;
        gsc[0] = gsc[1] = gsc[2] = 0.f; g_opac[i] = 0.f;
        memset(g_sh + (size_t)i * 48, 0, 48 * sizeof(float));
        if (!mask[i]) {
            /* A culled Gaussian has a zero cotangent, but the reference's autograd still divides it by the
             * clip-space w and multiplies it by the clip coordinates (the backward of Utils/Camera.py:103):
             * coordinates that are not finite turn its whole coordinate gradient into NaN. */
            const float *pc = coords + 3 * i;
            if (!(isfinite(pc[0]) && isfinite(pc[1]) && isfinite(pc[2]))) gc[0] = gc[1] = gc[2] = NAN;
            continue;
        }
        geom_t g;
        const float *p = coords + 3 * i;
        geometry(v, p, scale_raw + 3 * i, rot_raw + 4 * i, &g);
        double gp[3] = {0, 0, 0};
        /* opacity = sigmoid(raw) */
        float o = 1.0f / (1.0f + expf(-opac_raw[i]));
        g_opac[i] = (float)(d_opacity[i] * (double)(o * (1.0f - o)));
        /* colour = clamp(sh . basis + 0.5, min 0): gradient where the sum is >= 0 */
        {
            float b[16], bg[16][3];
            float dx = p[0] - v->cam_pos[0], dy = p[1] - v->cam_pos[1], dz = p[2] - v->cam_pos[2];
            sh_basis(v->sh_degree, dx, dy, dz, b);
            sh_basis_grad(v->sh_degree, dx, dy, dz, bg);
            for (int c = 0; c < 3; ++c) {
                const float *co = sh + (size_t)i * 48 + 16 * c;
                float acc = 0.0f;
                for (int k = 0; k < K; ++k) acc += b[k] * co[k];
                acc += 0.5f;
                if (!(acc >= 0.0f)) continue;
                double dc = d_color[3 * i + c];
                for (int k = 0; k < K; ++k) {
                    g_sh[(size_t)i * 48 + 16 * c + k] = (float)(dc * (double)b[k]);
                    for (int a = 0; a < 3; ++a) gp[a] += dc * (double)co[k] * (double)bg[k][a];
                }
            }
        }
        /* pixel centre -> NDC -> clip -> world */
        {
            double dn[3] = {d_pos2d[2 * i] * 0.5 * (double)Wf, d_pos2d[2 * i + 1] * -0.5 * (double)Hf, 0.0};
            gsc[0] = (float)dn[0]; gsc[1] = (float)dn[1]; gsc[2] = 0.0f;
            double c3 = (double)g.c[3];
            double dc[4];
            dc[3] = 0.0;
            for (int j = 0; j < 3; ++j) { dc[j] = dn[j] / c3; dc[3] -= dn[j] * (double)g.c[j] / (c3 * c3); }
            double dm[4];
            for (int k = 0; k < 4; ++k) {
                dm[k] = 0.0;
                for (int j = 0; j < 4; ++j) dm[k] += dc[j] * (double)v->proj[4 * j + k];
            }
            for (int k = 0; k < 3; ++k)
                for (int j = 0; j < 4; ++j) gp[k] += dm[j] * (double)v->w2m[4 * j + k];
        }
        /* conic = inv(cov2): dL/dcov2 = -C^T G C^T */
        double D[2][2];
        {
            float a = g.cov2[0], bb = g.cov2[1], cc = g.cov2[2], d = g.cov2[3];
            double det = (double)(a * d - bb * cc);
            double C[2][2] = {{(double)d / det, -(double)bb / det}, {-(double)cc / det, (double)a / det}};
            double Gm[2][2] = {{d_conic[4 * i], d_conic[4 * i + 1]}, {d_conic[4 * i + 2], d_conic[4 * i + 3]}};
            double tmp[2][2];
            for (int r = 0; r < 2; ++r)
                for (int c = 0; c < 2; ++c) tmp[r][c] = C[0][r] * Gm[0][c] + C[1][r] * Gm[1][c]; /* C^T G */
            for (int r = 0; r < 2; ++r)
                for (int c = 0; c < 2; ++c) D[r][c] = -(tmp[r][0] * C[c][0] + tmp[r][1] * C[c][1]); /* . C^T */
            /* The reference inverts the covariances of the Gaussians of a tile's list only
             * (GaussianRenderer.py:89-97, Utils/Probability.py:15): a Gaussian in no list receives an exact
             * zero here even when its covariance is singular (a zero scale), not 0 * inf. */
            if (Gm[0][0] == 0.0 && Gm[0][1] == 0.0 && Gm[1][0] == 0.0 && Gm[1][1] == 0.0)
                D[0][0] = D[0][1] = D[1][0] = D[1][1] = 0.0;
        }
        /* cov2 = M S M^T, M = J W */
        double dS[3][3], dM[2][3];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                double acc = 0.0;
                for (int r = 0; r < 2; ++r)
                    for (int c = 0; c < 2; ++c) acc += (double)g.M[r][a] * D[r][c] * (double)g.M[c][b];
                dS[a][b] = acc;
            }
        for (int r = 0; r < 2; ++r)
            for (int a = 0; a < 3; ++a) {
                double acc = 0.0;
                for (int c = 0; c < 2; ++c)
                    for (int b = 0; b < 3; ++b)
                        acc += D[r][c] * (double)g.M[c][b] * (double)g.cov3[a][b] +
                               D[c][r] * (double)g.M[c][b] * (double)g.cov3[b][a];
                dM[r][a] = acc;
            }
        /* M = J W, W[k][j] = w2m[j][k]  ->  dJ[i][k] = sum_j dM[i][j] w2m[j][k] */
        {
            double dJ[2][3];
            for (int r = 0; r < 2; ++r)
                for (int k = 0; k < 3; ++k) {
                    dJ[r][k] = 0.0;
                    for (int j = 0; j < 3; ++j) dJ[r][k] += dM[r][j] * (double)v->w2m[4 * j + k];
                }
            double x = p[0], y = p[1], z = p[2];
            gp[0] += dJ[0][2] * (-1.0 / z);
            gp[1] += dJ[1][2] * (-1.0 / z);
            gp[2] += (dJ[0][0] + dJ[1][1]) * (-1.0 / (z * z)) + dJ[0][2] * (x / (z * z)) + dJ[1][2] * (y / (z * z));
        }
        /* cov3 = RS RS^T, RS = R diag(s) */
        {
            double dRS[3][3], dR[3][3], ds[3] = {0, 0, 0};
            for (int a = 0; a < 3; ++a)
                for (int k = 0; k < 3; ++k) {
                    double acc = 0.0;
                    for (int b = 0; b < 3; ++b) acc += (dS[a][b] + dS[b][a]) * (double)g.RS[b][k];
                    dRS[a][k] = acc;
                }
            for (int a = 0; a < 3; ++a)
                for (int k = 0; k < 3; ++k) {
                    dR[a][k] = dRS[a][k] * (double)g.s[k];
                    ds[k] += dRS[a][k] * (double)g.R[a][k];
                }
            for (int k = 0; k < 3; ++k) gs[k] = (float)(ds[k] * (double)g.s[k]);
            double w = g.q[0], x = g.q[1], y = g.q[2], z = g.q[3];
            double dq[4];
            dq[0] = 4 * w * (dR[0][0] + dR[1][1] + dR[2][2]) +
                    2 * (-z * dR[0][1] + y * dR[0][2] + z * dR[1][0] - x * dR[1][2] - y * dR[2][0] + x * dR[2][1]);
            dq[1] = 4 * x * dR[0][0] +
                    2 * (y * dR[0][1] + z * dR[0][2] + y * dR[1][0] - w * dR[1][2] + z * dR[2][0] + w * dR[2][1]);
            dq[2] = 4 * y * dR[1][1] +
                    2 * (x * dR[0][1] + w * dR[0][2] + x * dR[1][0] + z * dR[1][2] - w * dR[2][0] + z * dR[2][1]);
            dq[3] = 4 * z * dR[2][2] +
                    2 * (-w * dR[0][1] + x * dR[0][2] + w * dR[1][0] + y * dR[1][2] + x * dR[2][0] + y * dR[2][1]);
            /* q = raw / max(|raw|, eps) */
            double nrm = g.qn;
            float ss = rot_raw[4 * i] * rot_raw[4 * i] + rot_raw[4 * i + 1] * rot_raw[4 * i + 1] +
                       rot_raw[4 * i + 2] * rot_raw[4 * i + 2] + rot_raw[4 * i + 3] * rot_raw[4 * i + 3];
            if (sqrtf(ss) > 1e-12f) {
                double qd = w * dq[0] + x * dq[1] + y * dq[2] + z * dq[3];
                double qq[4] = {w, x, y, z};
                for (int k = 0; k < 4; ++k) gr[k] = (float)((dq[k] - qq[k] * qd) / nrm);
            } else {
                for (int k = 0; k < 4; ++k) gr[k] = (float)(dq[k] / nrm);
            }
        }
        for (int k = 0; k < 3; ++k) gc[k] = (float)gp[k];
    }
    return GSO_OK;
}

int gso_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void gso_set_threads(int t)
{
#ifdef _OPENMP
    omp_set_num_threads(t);
#else
    (void)t;
#endif
}
