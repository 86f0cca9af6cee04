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

// scan_sort.hip, used by binning.hip: radix sort that ping-pongs between two buffer pairs without a
// final copy (*result_in_first: sorted data is in the *_a buffers), and a scan along a permutation.
// positions_as_values: the values are the positions 0 .. n-1 of each sequence (an argsort); vals_a need not hold them
int gsr_sort_pingpong32(uint32_t *keys_a, uint32_t *vals_a, uint32_t *keys_b, uint32_t *vals_b, uint32_t n, int begin_bit,
                        int end_bit, void *workspace, size_t workspace_bytes, void *stream, bool *result_in_first,
                        bool positions_as_values = false);
int gsr_sort_pingpong64(uint32_t *keys_a, uint64_t *vals_a, uint32_t *keys_b, uint64_t *vals_b, uint32_t n, int begin_bit,
                        int end_bit, void *workspace, size_t workspace_bytes, void *stream, bool *result_in_first);
int gsr_sort_words64(uint64_t *words_a, uint64_t *words_b, uint32_t n, int begin_bit, int end_bit, void *workspace,
                     size_t workspace_bytes, void *stream, bool *result_in_first);   // 64-bit words by their own bits
#define GSR_MAX_SEGS 8
size_t gsr_sort_words64_segs_bytes(const uint32_t *n, uint32_t count);
int gsr_sort_words64_segs(uint64_t *words_a, uint64_t *words_b, const uint32_t *n, const uint32_t *base, uint32_t count,
                          int begin_bit, int end_bit, void *workspace, size_t workspace_bytes, void *stream,
                          bool *result_in_first);   // sequences of different lengths, each from a multiple of 4096
int gsr_sort_pingpong32_batch(uint32_t *keys_a, uint32_t *vals_a, uint32_t *keys_b, uint32_t *vals_b, uint32_t n, uint32_t nseq,
                              int begin_bit, int end_bit, void *workspace, size_t workspace_bytes, void *stream,
                              bool *result_in_first, bool positions_as_values = false);   // nseq sequences of n pairs, one after the other
int gsr_exclusive_scan_along_batch(const uint32_t *in, const uint32_t *idx, uint32_t *out, uint32_t n, uint32_t nseq,
                                   uint32_t *total, uint32_t *overflow_flag, uint32_t counter_stride, void *workspace,
                                   size_t workspace_bytes, void *stream, uint32_t *out_seq,
                                   uint32_t *gathered /* n * nseq words of scratch for in[idx[i]] in sequence order; may be null */);
int gsr_exclusive_scan_along(const uint32_t *in, const uint32_t *idx, uint32_t *out, uint32_t n, uint32_t *total,
                             void *workspace, size_t workspace_bytes, void *stream, uint32_t *out_seq,
                             uint32_t *overflow_flag /* bit 1 raised when the total does not fit 32 bits; may be null */,
                             uint32_t *gathered /* n words of scratch, may be null */);

extern "C" uint32_t gsr_pairgrad_parts(int tile_length);

static inline unsigned gsr_div_up(unsigned a, unsigned b) { return (a + b - 1) / b; }

// Tiles lo/L .. (hi-1)/L overlap the integer interval [lo, hi) when hi > lo, none otherwise.
// Closed form of the membership test in Utils/Screen.py:58-62.
__host__ __device__ static inline void gsr_tile_span(int lo, int hi, int L, int &t0, int &t1)
{
    if (hi > lo) { t0 = lo / L; t1 = (hi - 1) / L; }
    else { t0 = 0; t1 = -1; }
}
