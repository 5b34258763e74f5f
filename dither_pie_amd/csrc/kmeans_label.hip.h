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

// status words (float64) of the device-side Lloyd loop
enum { kStDone = 0, kStIter = 1, kStInertia = 2, kStShift = 3, kStTolAbs = 4, kStQTotal = 5, kStWords = 8 };

// One workgroup.  totals: 5K int64, planar as dp_kmeans_step_u8 writes them into one buffer: sums [K][3] | counts [K] |
// squared norms [K] (the last part only has to be valid in the first iteration: its total is a constant of the data).  centers: K*3 float64, updated in place.  prev: [K][4]
// int64 scratch (the previous iteration's sums and counts).  status: kStWords float64.
//   done = 0 running; 1 assignments unchanged (centres kept: converged strictly); 2 shift <= tol or max_iter reached
//   (centres updated; the inertia of these final centres needs one more pass, flagged by done = 2 -> 3 below)
// An iteration that finds done != 0 on entry only refreshes the inertia once (done 2 -> 3) and changes nothing else.
// (a device function of one 256-thread workgroup: kmeans_update_kernel is it alone, hist_pass_kernel's last workgroup calls it
// at the end of a fused iteration -- there through ATOMIC loads, the totals having been written by other workgroups' atomics)
template <bool ATOMIC_LOADS>
__device__ __forceinline__ void lloyd_update_block(const long long *__restrict__ totals_in, double *__restrict__ centers,
                                                   long long *__restrict__ prev, double *__restrict__ status, const int K,
                                                   const double tol, const int max_iter, double *s_red)
{
    struct Totals {
        const long long *p;
        __device__ __forceinline__ long long operator[](const int i) const
        {
            if (ATOMIC_LOADS) return __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return p[i];
        }
    };
    const Totals totals{totals_in};
    const int t = threadIdx.x;
    const int done = (int)status[kStDone];
    auto block_sum = [&](double v) -> double {
        s_red[t] = v;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (t < off) s_red[t] = __dadd_rn(s_red[t], s_red[t + off]);
            __syncthreads();
        }
        const double r = s_red[0];
        __syncthreads();
        return r;
    };
    // inertia of the CURRENT centres with these totals: sum_k (q_k - 2 c_k.s_k + n_k |c_k|^2), q summed separately
    double part = 0.0, qpart = 0.0, npart = 0.0, s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int k = t; k < K; k += 256) {
        const double a = (double)totals[3 * k], b = (double)totals[3 * k + 1], c = (double)totals[3 * k + 2],
                     nn = (double)totals[3 * K + k];
        const double c0 = centers[3 * k], c1 = centers[3 * k + 1], c2 = centers[3 * k + 2];
        part += -2.0 * (c0 * a + c1 * b + c2 * c) + nn * (c0 * c0 + c1 * c1 + c2 * c2);
        qpart += (double)totals[4 * K + k];
        npart += nn;
        s0 += a;
        s1 += b;
        s2 += c;
    }
    const double cross = block_sum(part);
    if (done == 3 || done == 1) return;  // nothing left to do
    const int iter = (int)status[kStIter] + (done == 0 ? 1 : 0);
    double qtot = status[kStQTotal];
    if (iter == 1 && done == 0) {
        qtot = block_sum(qpart);
        const double N = block_sum(npart), m0 = block_sum(s0) / N, m1 = block_sum(s1) / N, m2 = block_sum(s2) / N;
        if (t == 0) {
            status[kStQTotal] = qtot;
            // sklearn: tol * mean of the per-channel variances
            status[kStTolAbs] = tol * ((qtot / N - (m0 * m0 + m1 * m1 + m2 * m2)) / 3.0);
        }
    }
    __syncthreads();
    if (done == 2) {  // the extra pass after the last update: the inertia of the final centres
        if (t == 0) {
            status[kStInertia] = qtot + cross;
            status[kStDone] = 3.0;
        }
        return;
    }
    // same assignments as in the previous iteration?  (sums and counts all equal)
    double diff = 0.0, shift = 0.0;
    for (int k = t; k < K; k += 256) {
        for (int c = 0; c < 4; ++c) {
            const long long cur = c < 3 ? totals[3 * k + c] : totals[3 * K + k];
            if (iter > 1 && prev[4 * k + c] != cur) diff += 1.0;
            if (iter == 1) diff += 1.0;
        }
        const double nn = (double)totals[3 * K + k];
        if (nn > 0.0) {
            for (int c = 0; c < 3; ++c) {
                const double nw = (double)totals[3 * k + c] / nn, d = nw - centers[3 * k + c];
                shift += d * d;
            }
        }
    }
    const double ndiff = block_sum(diff);
    const double tshift = block_sum(shift);
    const bool same = ndiff == 0.0;
    const double tol_abs = status[kStTolAbs];
    __syncthreads();
    for (int k = t; k < K; k += 256) {
        for (int c = 0; c < 4; ++c) prev[4 * k + c] = c < 3 ? totals[3 * k + c] : totals[3 * K + k];
        const double nn = (double)totals[3 * K + k];
        if (!same && nn > 0.0)
            for (int c = 0; c < 3; ++c) centers[3 * k + c] = (double)totals[3 * k + c] / nn;
    }
    if (t == 0) {
        status[kStIter] = (double)iter;
        status[kStInertia] = qtot + cross;  // of the centres this pass was run with
        status[kStShift] = tshift;
        if (same) status[kStDone] = 1.0;
        else if (tshift <= tol_abs || iter >= max_iter) status[kStDone] = 2.0;
    }
}

}  // namespace
}  // namespace dp
