// Argument validation, error text and version of libgsr.so.
#include <stdarg.h>
#include <stdio.h>

#include "gsr_common.hpp"

static thread_local char g_err[512] = "";

void gsr_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *gsr_last_error(void) { return g_err; }
extern "C" int gsr_version(void) { return GSR_VERSION_WORD; }

int gsr_check_launch(const char *what)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        gsr_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return GSR_ERR_LAUNCH;
    }
    return GSR_OK;
}

int gsr_make_view(const gsr_view *v, ViewK *out)
{
    if (!v || !out) { gsr_set_error("null view"); return GSR_ERR_ARG; }
    if (v->n < 0) { gsr_set_error("view.n = %d is negative", v->n); return GSR_ERR_ARG; }
    // eval_sh asserts 0 <= deg <= 4 (Utils/SphericalHarmonic.py:58); the model stores 16
    // coefficients per channel, i.e. degree <= 3 (GaussianModel.py:15,79)
    if (v->sh_degree < 0 || v->sh_degree > 3) {
        gsr_set_error("view.sh_degree = %d outside 0..3", v->sh_degree);
        return GSR_ERR_ARG;
    }
    if (v->width <= 0 || v->height <= 0 || v->width > 32768 || v->height > 32768) {
        gsr_set_error("view size %d x %d outside 1..32768", v->width, v->height);
        return GSR_ERR_ARG;
    }
    if (v->tile_length <= 0) { gsr_set_error("view.tile_length = %d must be positive", v->tile_length); return GSR_ERR_ARG; }
    out->n = v->n;
    out->deg = v->sh_degree;
    out->K = (v->sh_degree + 1) * (v->sh_degree + 1);
    out->W = v->width;
    out->H = v->height;
    out->L = v->tile_length;
    out->nbw = (v->width + v->tile_length - 1) / v->tile_length;
    out->nbh = (v->height + v->tile_length - 1) / v->tile_length;
    const long long T = (long long)out->nbw * out->nbh;
    if (T > (1ll << 24)) { gsr_set_error("too many tiles (%lld)", T); return GSR_ERR_ARG; }
    out->T = (int)T;
    out->Wf = (float)v->width;
    out->Hf = (float)v->height;
    for (int i = 0; i < 16; ++i) { out->w2m[i] = v->w2m[i]; out->proj[i] = v->proj[i]; }
    for (int i = 0; i < 3; ++i) out->cam[i] = v->cam_pos[i];
    return GSR_OK;
}
