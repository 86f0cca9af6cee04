// Shared declarations of the HIP sources of libgsr.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gsr.h"

#define GSR_WAVE 64

// Launch constants of one view, passed to kernels by value (kernarg segment -> SGPRs).
struct ViewK {
    int n, deg, K, W, H, L;
    int nbw, nbh, T;       // tile grid: tile id = yi + nbh * xi  (Utils/Screen.py:26-28,38-39)
    float Wf, Hf;
    float w2m[16];
    float proj[16];
    float cam[3];
};

void gsr_set_error(const char *fmt, ...);
int gsr_make_view(const gsr_view *v, ViewK *out);  // validates, returns GSR_OK / GSR_ERR_ARG
int gsr_check_launch(const char *what);

static inline unsigned gsr_div_up(unsigned a, unsigned b) { return (a + b - 1) / b; }

// Tiles lo/L .. (hi-1)/L overlap the integer interval [lo, hi) when hi > lo, none otherwise.
// Closed form of the membership test in Utils/Screen.py:58-62.
__host__ __device__ static inline void gsr_tile_span(int lo, int hi, int L, int &t0, int &t1)
{
    if (hi > lo) { t0 = lo / L; t1 = (hi - 1) / L; }
    else { t0 = 0; t1 = -1; }
}
