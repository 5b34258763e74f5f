// Device emulation of scipy.spatial.KDTree.query(x, k) for k in {1,2} (p=2, eps=0, no distance
// bound), the call behind every palette lookup of the reference (dithering_lib.py:340, 360, 556,
// 672).  Only pixels whose answer depends on scipy's visiting order take this path (exact distance
// ties) -- plus every pixel of error diffusion, where the query point is a float32 triple.
//
// The traversal keeps scipy's two binary heaps exactly: `neighbors` (the k best so far, furthest on
// top) and `q` (pending far children keyed on their box distance), with scipy's sift rules (strict
// `<` going up; going down the left child wins unless the right one is strictly smaller).
#pragma once
#include "dp_internal.h"

namespace dp {

struct QItem {
    double prio;     // min_dist of the pending node
    double side[3];  // per-axis squared distances to its box
    int node;
};

// all arithmetic below must round exactly once per operation (no FMA contraction)
__device__ __forceinline__ double sq_dist3(const double *__restrict__ p, const double x0, const double x1,
                                           const double x2)
{
    double s = 0.0, d;
    d = __dsub_rn(p[0], x0);
    s = __dadd_rn(s, __dmul_rn(d, d));
    d = __dsub_rn(p[1], x1);
    s = __dadd_rn(s, __dmul_rn(d, d));
    d = __dsub_rn(p[2], x2);
    s = __dadd_rn(s, __dmul_rn(d, d));
    return s;
}

template <int KQ, int CAP>
__device__ void tree_query(const PalDev &pal, const double x0, const double x1, const double x2, double *d2_out,
                           int *i_out)
{
    const double xs[3] = {x0, x1, x2};
    QItem q[CAP];
    int qn = 0;
    double nb_prio[2];
    int nb_idx[2];
    int nbn = 0;
    double ub = __longlong_as_double(0x7ff0000000000000LL);  // +inf

    QItem cur;
    cur.node = 0;
    cur.prio = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double a = __dsub_rn(xs[i], pal.maxes[i]), b = __dsub_rn(pal.mins[i], xs[i]);
        double s = a > b ? a : b;
        if (s < 0.0) s = 0.0;
        cur.side[i] = __dmul_rn(s, s);
        cur.prio = __dadd_rn(cur.prio, cur.side[i]);
    }
    for (;;) {
        const int node = cur.node;
        const int sd = pal.split_dim[node];
        if (sd < 0) {
            const int e = pal.end[node];
            for (int j = pal.start[node]; j < e; ++j) {
                const int pi = pal.indices[j];
                const double s = sq_dist3(pal.pts + 3 * pi, x0, x1, x2);
                if (s < ub) {
                    if (nbn == KQ) {  // drop the furthest
                        nb_prio[0] = nb_prio[nbn - 1];
                        nb_idx[0] = nb_idx[nbn - 1];
                        --nbn;
                    }
                    nb_prio[nbn] = -s;
                    nb_idx[nbn] = pi;
                    if (nbn == 1 && nb_prio[1] < nb_prio[0]) {
                        const double tp = nb_prio[0];
                        const int ti = nb_idx[0];
                        nb_prio[0] = nb_prio[1];
                        nb_idx[0] = nb_idx[1];
                        nb_prio[1] = tp;
                        nb_idx[1] = ti;
                    }
                    ++nbn;
                    if (nbn == KQ) ub = -nb_prio[0];
                }
            }
            if (qn == 0) break;
            // pop the nearest pending node
            cur = q[0];
            q[0] = q[qn - 1];
            --qn;
            int i = 0, l = 1, r = 2;
            while ((l < qn && q[i].prio > q[l].prio) || (r < qn && q[i].prio > q[r].prio)) {
                const int c = (r < qn && q[l].prio > q[r].prio) ? r : l;
                const QItem t = q[c];
                q[c] = q[i];
                q[i] = t;
                i = c;
                l = 2 * i + 1;
                r = 2 * i + 2;
            }
        } else {
            if (cur.prio > ub) break;
            const double split = pal.split[node];
            QItem far = cur;
            const double xv = sd == 0 ? x0 : (sd == 1 ? x1 : x2);
            if (xv < split) {
                cur.node = pal.less[node];
                far.node = pal.greater[node];
            } else {
                cur.node = pal.greater[node];
                far.node = pal.less[node];
            }
            const double df = __dsub_rn(xv, split);
            const double ns = __dmul_rn(df, df);
            const double old = sd == 0 ? far.side[0] : (sd == 1 ? far.side[1] : far.side[2]);
            far.prio = __dadd_rn(far.prio, __dsub_rn(ns, old));
            if (sd == 0)
                far.side[0] = ns;
            else if (sd == 1)
                far.side[1] = ns;
            else
                far.side[2] = ns;
            if (far.prio <= ub && qn < CAP) {
                int i = qn++;
                q[i] = far;
                while (i > 0 && q[i].prio < q[(i - 1) / 2].prio) {
                    const QItem t = q[(i - 1) / 2];
                    q[(i - 1) / 2] = q[i];
                    q[i] = t;
                    i = (i - 1) / 2;
                }
            }
        }
    }
    // heap-sort the neighbours: furthest comes off first and fills from the back
    const double inf = __longlong_as_double(0x7ff0000000000000LL);
    d2_out[0] = inf;
    i_out[0] = pal.K;
    if (KQ == 2) {
        d2_out[1] = inf;
        i_out[1] = pal.K;
    }
    if (nbn == 2) {
        d2_out[1] = -nb_prio[0];
        i_out[1] = nb_idx[0];
        d2_out[0] = -nb_prio[1];
        i_out[0] = nb_idx[1];
    } else if (nbn == 1) {
        d2_out[0] = -nb_prio[0];
        i_out[0] = nb_idx[0];
    }
}

// dithering_lib.py:361-365, 376: factor from re-squared sqrt distances, compared with the f32 threshold
__device__ __forceinline__ bool ordered_use_nearest(const double d2_0, const double d2_1, const float t)
{
    const double r0 = __dsqrt_rn(d2_0), r1 = __dsqrt_rn(d2_1);
    const double s0 = __dmul_rn(r0, r0), s1 = __dmul_rn(r1, r1);
    const double tot = __dadd_rn(s0, s1);
    const double f = (tot == 0.0) ? 0.0 : __ddiv_rn(s0, tot);
    return f <= (double)t;
}

// dithering_lib.py:539-549; float32, one rounding per operation
__device__ __forceinline__ float ign_threshold(const int gx, const int gy, const float sx, const float sy,
                                               const float sc)
{
    const float xv = __fmul_rn(__fadd_rn((float)gx, sx), sc);
    const float yv = __fmul_rn(__fadd_rn((float)gy, sy), sc);
    const float s = __fadd_rn(__fmul_rn(xv, 0.06711056f), __fmul_rn(yv, 0.00583715f));
    const float u = __fsub_rn(s, floorf(s));
    const float v = __fmul_rn(u, 52.9829189f);
    return __fsub_rn(v, floorf(v));
}

// The same as a real function (arguments and result in registers): for kernels that reach the float64 chain only on
// exact equality of the integer decision and would otherwise carry a copy of its ~70 instructions at every inlined site.
__device__ __noinline__ bool ordered_use_nearest_call(const double d2_0, const double d2_1, const float t)
{
    return ordered_use_nearest(d2_0, d2_1, t);
}

}  // namespace dp
