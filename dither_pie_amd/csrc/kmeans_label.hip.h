// Device helpers shared by the Lloyd-pass kernels (kmeans.hip: passes over pixels; kmeans_hist.hip: passes over the colour
// histogram): the float64 decision behind every float32 ranking, and the biased float32 score whose bits order like keys.
#pragma once
#include "dp_internal.h"

namespace dp {
namespace {

__device__ __forceinline__ int med3_s32(const int a, const int b, const int c)
{
    return max(min(a, b), min(max(a, b), c));  // selected as one v_med3_i32
}

// ---------------------------------------------------------------------------------------------------------------
// The float64 decision behind the float32 ranking (near ties only).  Centre records in LDS: 4 doubles {c0, c1, c2, cn}.
//   mean == nullptr: direct distances ((x0-c0)^2 + (x1-c1)^2) + (x2-c2)^2, lowest index on exact ties.
//   mean != nullptr: sklearn's own expression (KMeans.fit -> _k_means_lloyd.pyx::_update_chunk_dense, as
//     dithering_lib.py:1854-1856 runs it): data and centres mean-centred in float64, v_j = |c'_j|^2 - 2 x'.c'_j with
//     |c'|^2 = (fl(c0'^2) + fl(c2'^2)) + fl(c1'^2) (numpy einsum over three elements in 512-bit lanes) and
//     x'.c' = fma(x2', c2', fma(x1', c1', fl(x0' c0'))) (the OpenBLAS dgemm micro-kernel), first minimum.  It orders like
//     the distance except where two centres are EXACTLY equidistant (k-means++ seeds are data points: integer centres
//     tie on up to 1 % of a structured image's pixels in the first pass), where its rounding decides -- reproducibly,
//     and that is what the reference's palette then depends on (tests/golden kmx_*).  Rounding error < 1e-9 against a
//     near-tie window of 0.25: every sample whose label the expression could change comes through here.
__device__ __forceinline__ void stage_centre_f64(double *rec, const double c0, const double c1, const double c2, const double *mean)
{
    if (mean) {
        const double a = __dsub_rn(c0, mean[0]), b = __dsub_rn(c1, mean[1]), c = __dsub_rn(c2, mean[2]);
        rec[0] = a;
        rec[1] = b;
        rec[2] = c;
        rec[3] = __dadd_rn(__dadd_rn(__dmul_rn(a, a), __dmul_rn(c, c)), __dmul_rn(b, b));
    } else {
        rec[0] = c0;
        rec[1] = c1;
        rec[2] = c2;
        rec[3] = 0.0;
    }
}

__device__ __forceinline__ int label_f64(const double *s_c, const int K, const double *mean, const uint32_t r, const uint32_t g,
                                         const uint32_t b, int lab)
{
    double bd = __longlong_as_double(0x7ff0000000000000LL);
    if (mean) {
        const double y0 = __dsub_rn((double)r, mean[0]), y1 = __dsub_rn((double)g, mean[1]), y2 = __dsub_rn((double)b, mean[2]);
        for (int j = 0; j < K; ++j) {
            double acc = __dmul_rn(y0, s_c[4 * j]);
            acc = __fma_rn(y1, s_c[4 * j + 1], acc);
            acc = __fma_rn(y2, s_c[4 * j + 2], acc);
            const double v = __dsub_rn(s_c[4 * j + 3], __dmul_rn(2.0, acc));
            if (v < bd) {
                bd = v;
                lab = j;
            }
        }
    } else {
        const double x0 = (double)r, x1 = (double)g, x2 = (double)b;
        for (int j = 0; j < K; ++j) {
            const double a = __dsub_rn(x0, s_c[4 * j]), c = __dsub_rn(x1, s_c[4 * j + 1]), e = __dsub_rn(x2, s_c[4 * j + 2]);
            const double d = __dadd_rn(__dadd_rn(__dmul_rn(a, a), __dmul_rn(c, c)), __dmul_rn(e, e));
            if (d < bd) {
                bd = d;
                lab = j;
            }
        }
    }
    return lab;
}


// KEYS (K <= 256): the scores carry a bias of 2^19 + 195076, which puts every one of them into [2^19, 2^20) -- one
// exponent, so the float bits shifted left by 8 order like the scores and leave room for the centre's index:
// key = bits << 8 | j (v_lshl_or_b32), the two smallest keys by v_med3_i32 + v_min_i32 -- 6 instructions per pixel and
// centre instead of 7 (no compare + select for the label).  The bias costs precision (scores within 0.15 of their exact
// value instead of 0.11): the float64 scan decides below a gap of 6 ulp (0.375).
constexpr float kScoreBias = 524288.0f + 195076.0f;

}  // namespace
}  // namespace dp
