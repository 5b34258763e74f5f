/*
 * dp_oracle.c -- CPU restatement of dither_pie's per-pixel hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under dither_pie_amd/ (the product) may
 * import, link or call this file.  Allowed users: tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg, and there only as the checker / the timed
 * CPU baseline.
 *
 * Parity status: PINNED by outputs of the reference itself generated in the
 * build container (tests/golden/make_golden.py imports /root/reference and
 * writes tests/golden/ small.npz + kat.json); the reference ships no tests of its
 * own, so there are no upstream golden vectors.
 *
 * What is restated, and the reference lines each function follows:
 *   orc_tree_build / orc_tree_query  scipy.spatial.KDTree (leafsize 10) as
 *       called at dithering_lib.py:339-340, 358-360, 554-556, 655+672.  scipy
 *       is a third-party dependency that is not under /root/reference (version
 *       unpinned by the reference, 1.15.3 in the build container); its
 *       published algorithm (ckdtree build.cxx / query.cxx, libstdc++
 *       std::nth_element) is restated here from SURVEY.md Appendix A.1.
 *   orc_ordered_u8        NoDitherStrategy.dither          dithering_lib.py:337-341
 *                         MatrixDitherStrategy.dither      dithering_lib.py:355-378
 *                         IGN strategy                     dithering_lib.py:539-568
 *                         (with the uint8/gamma wrapper of apply_dithering :1952-1992)
 *   orc_ign_thresholds    _generate_thresholds             dithering_lib.py:539-549
 *   orc_blue_noise        generate_blue_noise              dithering_lib.py:381-399
 *                         (numpy legacy RandomState.shuffle = MT19937, restated)
 *   orc_error_diffusion_numba_u8  the same strategy's numba branch (:213-308) -- parity unpinned, see the function
 *   orc_hybrid_numba_u8   HybridDitherStrategy's numba branch (_hybrid_numba, :1396-1494) -- parity unpinned, see the function
 *   orc_error_diffusion_u8 ErrorDiffusionDitherStrategy.dither, pure-Python
 *                         branch                           dithering_lib.py:655-690
 *   orc_kmeans_step       one Lloyd assignment + accumulation pass of
 *                         sklearn KMeans as used at        dithering_lib.py:1854-1856
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (see oracle/Makefile).
 * -ffp-contract=off matters: every f32/f64 product is rounded before it is added.
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_KMAX 1024
#define ORC_NMAX (2 * ORC_KMAX)
#define ORC_LEAFSIZE 10

typedef struct {
    int K;
    int nnodes;
    double pts[ORC_KMAX * 3];
    int idx[ORC_KMAX];
    int split_dim[ORC_NMAX]; /* -1 = leaf */
    double split[ORC_NMAX];
    int start[ORC_NMAX], end[ORC_NMAX];
    int less[ORC_NMAX], greater[ORC_NMAX];
    double mins[3], maxes[3];
} orc_tree;

/* ------------------------------------------------------------------ */
/* libstdc++ std::nth_element (introselect) on an index array with the */
/* ckdtree comparator data[a][d] < data[b][d].                          */
/* ------------------------------------------------------------------ */
typedef struct {
    const double *pts;
    int d;
} cmp_ctx;

static inline int lt(const cmp_ctx *c, int a, int b)
{
    return c->pts[a * 3 + c->d] < c->pts[b * 3 + c->d];
}

static inline void iswap(int *a, int *b)
{
    int t = *a;
    *a = *b;
    *b = t;
}

static void move_median_to_first(const cmp_ctx *c, int *result, int *a, int *b, int *cc)
{
    if (lt(c, *a, *b)) {
        if (lt(c, *b, *cc))
            iswap(result, b);
        else if (lt(c, *a, *cc))
            iswap(result, cc);
        else
            iswap(result, a);
    } else if (lt(c, *a, *cc))
        iswap(result, a);
    else if (lt(c, *b, *cc))
        iswap(result, cc);
    else
        iswap(result, b);
}

static int *unguarded_partition(const cmp_ctx *c, int *first, int *last, int *pivot)
{
    for (;;) {
        while (lt(c, *first, *pivot))
            ++first;
        --last;
        while (lt(c, *pivot, *last))
            --last;
        if (!(first < last))
            return first;
        iswap(first, last);
        ++first;
    }
}

static void insertion_sort(const cmp_ctx *c, int *first, int *last)
{
    if (first == last)
        return;
    for (int *i = first + 1; i != last; ++i) {
        int val = *i;
        if (lt(c, val, *first)) {
            memmove(first + 1, first, (size_t)(i - first) * sizeof(int));
            *first = val;
        } else {
            int *hole = i;
            int *next = i - 1;
            while (lt(c, val, *next)) {
                *hole = *next;
                hole = next;
                --next;
            }
            *hole = val;
        }
    }
}

/* std::__adjust_heap + __push_heap (max-heap under lt) */
static void adjust_heap(const cmp_ctx *c, int *first, long hole, long len, int value)
{
    const long top = hole;
    long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (lt(c, first[child], first[child - 1]))
            child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    long parent = (hole - 1) / 2;
    while (hole > top && lt(c, first[parent], value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}

static void heap_select(const cmp_ctx *c, int *first, int *middle, int *last)
{
    long len = middle - first;
    if (len >= 2) {
        long parent = (len - 2) / 2;
        for (;;) {
            int v = first[parent];
            adjust_heap(c, first, parent, len, v);
            if (parent == 0)
                break;
            parent--;
        }
    }
    for (int *i = middle; i < last; ++i) {
        if (lt(c, *i, *first)) {
            int v = *i;
            *i = *first;
            adjust_heap(c, first, 0, len, v);
        }
    }
}

static void nth_element_idx(const cmp_ctx *c, int *first, int *nth, int *last)
{
    if (first == last || nth == last)
        return;
    long n = last - first;
    int lg = 0;
    while ((n >> (lg + 1)) != 0)
        lg++;
    long depth = 2L * lg;
    while (last - first > 3) {
        if (depth == 0) {
            heap_select(c, first, nth + 1, last);
            iswap(first, nth);
            return;
        }
        --depth;
        int *mid = first + (last - first) / 2;
        move_median_to_first(c, first, first + 1, mid, last - 1);
        int *cut = unguarded_partition(c, first + 1, last, first);
        if (cut <= nth)
            first = cut;
        else
            last = cut;
    }
    insertion_sort(c, first, last);
}

/* ------------------------------------------------------------------ */
/* ckdtree build (balanced_tree=True, compact_nodes=True, leafsize 10) */
/* ------------------------------------------------------------------ */
static int hoare_partition(orc_tree *t, int s, int e, int d, double split)
{
    int p = s, q = e - 1;
    while (p <= q) {
        if (t->pts[t->idx[p] * 3 + d] < split)
            ++p;
        else if (t->pts[t->idx[q] * 3 + d] >= split)
            --q;
        else {
            iswap(&t->idx[p], &t->idx[q]);
            ++p;
            --q;
        }
    }
    return p;
}

static int build_rec(orc_tree *t, int s, int e)
{
    int node = t->nnodes++;
    t->start[node] = s;
    t->end[node] = e;
    t->split_dim[node] = -1;
    t->split[node] = 0.0;
    t->less[node] = t->greater[node] = -1;
    if (e - s <= ORC_LEAFSIZE)
        return node;

    double mins[3], maxes[3];
    for (int i = 0; i < 3; i++)
        mins[i] = maxes[i] = t->pts[t->idx[s] * 3 + i];
    for (int j = s + 1; j < e; j++)
        for (int i = 0; i < 3; i++) {
            double v = t->pts[t->idx[j] * 3 + i];
            maxes[i] = maxes[i] > v ? maxes[i] : v;
            mins[i] = mins[i] < v ? mins[i] : v;
        }
    int d = 0;
    double size = 0;
    for (int i = 0; i < 3; i++)
        if (maxes[i] - mins[i] > size) {
            d = i;
            size = maxes[i] - mins[i];
        }
    if (maxes[d] == mins[d])
        return node; /* all points identical -> leaf */

    cmp_ctx c = {t->pts, d};
    int half = (e - s) / 2;
    nth_element_idx(&c, t->idx + s, t->idx + s + half, t->idx + e);
    double split = t->pts[t->idx[s + half] * 3 + d];
    int p = hoare_partition(t, s, e, d, split);
    if (p == s) {
        /* no point strictly below the median value: slide just above the minimum */
        double mn = t->pts[t->idx[s] * 3 + d];
        for (int j = s + 1; j < e; j++) {
            double v = t->pts[t->idx[j] * 3 + d];
            if (v < mn)
                mn = v;
        }
        split = nextafter(mn, HUGE_VAL);
        p = hoare_partition(t, s, e, d, split);
    } else if (p == e) {
        /* cannot happen with a median split (the median itself is >= split) */
        double mx = t->pts[t->idx[s] * 3 + d];
        for (int j = s + 1; j < e; j++) {
            double v = t->pts[t->idx[j] * 3 + d];
            if (v > mx)
                mx = v;
        }
        split = mx;
        p = hoare_partition(t, s, e, d, split);
    }
    t->split_dim[node] = d;
    t->split[node] = split;
    int l = build_rec(t, s, p);
    int g = build_rec(t, p, e);
    t->less[node] = l;
    t->greater[node] = g;
    return node;
}

int orc_tree_build(orc_tree *t, const double *pts, int K)
{
    if (K < 1 || K > ORC_KMAX)
        return -1;
    t->K = K;
    t->nnodes = 0;
    memcpy(t->pts, pts, sizeof(double) * 3 * (size_t)K);
    for (int i = 0; i < K; i++)
        t->idx[i] = i;
    for (int i = 0; i < 3; i++)
        t->mins[i] = t->maxes[i] = pts[i];
    for (int j = 1; j < K; j++)
        for (int i = 0; i < 3; i++) {
            double v = pts[j * 3 + i];
            if (v > t->maxes[i])
                t->maxes[i] = v;
            if (v < t->mins[i])
                t->mins[i] = v;
        }
    build_rec(t, 0, K);
    return 0;
}

size_t orc_tree_sizeof(void) { return sizeof(orc_tree); }

/* worker threads of the row-parallel loops (bench.py's cpu_baseline states how many it used) */
int orc_set_threads(int n)
{
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
}

/* flat export for tests: arrays must hold nnodes (<= 2K) entries */
int orc_tree_export(const orc_tree *t, int *idx, int *split_dim, double *split, int *start, int *end,
                    int *less, int *greater)
{
    memcpy(idx, t->idx, sizeof(int) * (size_t)t->K);
    for (int i = 0; i < t->nnodes; i++) {
        split_dim[i] = t->split_dim[i];
        split[i] = t->split[i];
        start[i] = t->start[i];
        end[i] = t->end[i];
        less[i] = t->less[i];
        greater[i] = t->greater[i];
    }
    return t->nnodes;
}

/* ------------------------------------------------------------------ */
/* ckdtree query, k in {1,2}, p=2, eps=0, no upper bound               */
/* ------------------------------------------------------------------ */
typedef struct {
    double prio;
    int item;
} hitem;

static void heap_push(hitem *h, int *n, hitem it)
{
    int i = (*n)++;
    h[i] = it;
    while (i > 0 && h[i].prio < h[(i - 1) / 2].prio) {
        hitem t = h[(i - 1) / 2];
        h[(i - 1) / 2] = h[i];
        h[i] = t;
        i = (i - 1) / 2;
    }
}

static hitem heap_pop(hitem *h, int *n)
{
    hitem top = h[0];
    h[0] = h[*n - 1];
    (*n)--;
    int nn = *n, i = 0, j = 1, k = 2;
    while ((j < nn && h[i].prio > h[j].prio) || (k < nn && h[i].prio > h[k].prio)) {
        int l = (k < nn && h[j].prio > h[k].prio) ? k : j;
        hitem t = h[l];
        h[l] = h[i];
        h[i] = t;
        i = l;
        j = 2 * i + 1;
        k = 2 * i + 2;
    }
    return top;
}

typedef struct {
    int node;
    double side[3];
    double min_dist;
} ninfo;

/* returns number of neighbours found (<= k); d2 = squared distances, ascending */
int orc_tree_query(const orc_tree *t, const double *x, int k, double *d2_out, int *i_out)
{
    ninfo pool[ORC_NMAX];
    int npool = 0;
    hitem q[ORC_NMAX];
    int qn = 0;
    hitem nb[2];
    int nbn = 0;
    double ub = INFINITY;

    ninfo *cur = &pool[npool++];
    cur->node = 0;
    cur->min_dist = 0.0;
    for (int i = 0; i < 3; i++) {
        double a = x[i] - t->maxes[i], b = t->mins[i] - x[i];
        double s = a > b ? a : b;
        if (s < 0)
            s = 0;
        cur->side[i] = s * s;
        cur->min_dist += cur->side[i];
    }
    for (;;) {
        int node = cur->node;
        if (t->split_dim[node] == -1) {
            for (int j = t->start[node]; j < t->end[node]; j++) {
                int pi = t->idx[j];
                const double *p = &t->pts[pi * 3];
                double s = 0.0, df;
                df = p[0] - x[0];
                s += df * df;
                df = p[1] - x[1];
                s += df * df;
                df = p[2] - x[2];
                s += df * df;
                if (s < ub) {
                    if (nbn == k)
                        (void)heap_pop(nb, &nbn);
                    hitem it = {-s, pi};
                    heap_push(nb, &nbn, it);
                    if (nbn == k)
                        ub = -nb[0].prio;
                }
            }
            if (qn == 0)
                break;
            hitem it = heap_pop(q, &qn);
            cur = &pool[it.item];
        } else {
            if (cur->min_dist > ub)
                break;
            int d = t->split_dim[node];
            double split = t->split[node];
            ninfo *far = &pool[npool];
            *far = *cur;
            if (x[d] < split) {
                cur->node = t->less[node];
                far->node = t->greater[node];
            } else {
                cur->node = t->greater[node];
                far->node = t->less[node];
            }
            double sd = x[d] - split;
            double ns = sd * sd;
            far->min_dist += ns - far->side[d];
            far->side[d] = ns;
            if (far->min_dist <= ub) {
                hitem it = {far->min_dist, npool};
                npool++;
                heap_push(q, &qn, it);
            }
        }
    }
    int found = nbn;
    for (int i = nbn - 1; i >= 0; --i) {
        hitem it = heap_pop(nb, &nbn);
        d2_out[i] = -it.prio;
        i_out[i] = it.item;
    }
    return found;
}

/* batch query for tests: x is n x 3 doubles */
void orc_tree_query_batch(const orc_tree *t, const double *x, long n, int k, double *d2, int *ii)
{
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; i++) {
        double dd[2] = {INFINITY, INFINITY};
        int jj[2] = {t->K, t->K};
        orc_tree_query(t, x + 3 * i, k, dd, jj);
        for (int c = 0; c < k; c++) {
            d2[i * k + c] = dd[c];
            ii[i * k + c] = jj[c];
        }
    }
}

/* ------------------------------------------------------------------ */
/* thresholds                                                          */
/* ------------------------------------------------------------------ */
static inline float fractf_(float v) { return v - floorf(v); }

/* dithering_lib.py:539-549; every f32 op individually rounded */
static inline float ign_threshold(int gx, int gy, float sx, float sy, float scale)
{
    float xv = ((float)gx + sx) * scale;
    float yv = ((float)gy + sy) * scale;
    float a = xv * 0.06711056f;
    float b = yv * 0.00583715f;
    float u = fractf_(a + b);
    return fractf_(u * 52.9829189f);
}

void orc_ign_thresholds(int h, int w, int y0, int x0, double scale, int seed, float *out)
{
    float sx = (float)((double)seed * 0.37), sy = (float)((double)seed * 0.73);
    float sc = (float)scale;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            out[(size_t)y * w + x] = ign_threshold(x0 + x, y0 + y, sx, sy, sc);
}

/* ------------------------------------------------------------------ */
/* ordered / nearest, uint8 in -> uint8 out                            */
/* mode: 0 = nearest only (k=1); 1 = threshold matrix; 2 = IGN         */
/* pal: K x 3 f32 as seen by the KD-tree; out_colors: K x 3 uint8      */
/* lut_in: NULL or 256-entry uint8 LUT applied to every input channel  */
/* (y0,x0): global coordinates of in[0] (tile sharding)                */
/* idx_out: NULL or h*w int32 chosen palette indices                   */
/* ------------------------------------------------------------------ */
int orc_ordered_u8(const uint8_t *in, uint8_t *out, int32_t *idx_out, int h, int w, int y0, int x0,
                   const float *pal, int K, const uint8_t *out_colors, const uint8_t *lut_in, int mode,
                   const float *thr, int th_h, int th_w, double ign_scale, int ign_seed)
{
    orc_tree *t = (orc_tree *)malloc(sizeof(orc_tree));
    double pts[ORC_KMAX * 3];
    if (!t || K < 1 || K > ORC_KMAX) {
        free(t);
        return -1;
    }
    for (int i = 0; i < 3 * K; i++)
        pts[i] = (double)pal[i];
    orc_tree_build(t, pts, K);
    float sx = (float)((double)ign_seed * 0.37), sy = (float)((double)ign_seed * 0.73);
    float sc = (float)ign_scale;
    int k = (mode == 0) ? 1 : 2;
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            const uint8_t *px = in + ((size_t)y * w + x) * 3;
            double q[3];
            for (int c = 0; c < 3; c++)
                q[c] = (double)(float)(lut_in ? lut_in[px[c]] : px[c]);
            double d2[2] = {INFINITY, INFINITY};
            int ii[2] = {K, K};
            orc_tree_query(t, q, k, d2, ii);
            int pick = ii[0];
            if (mode != 0) {
                /* dist**2 of the returned sqrt distances, dithering_lib.py:361 */
                double r0 = sqrt(d2[0]), r1 = sqrt(d2[1]);
                double s0 = r0 * r0, s1 = r1 * r1;
                double tot = s0 + s1;
                double f = (tot == 0.0) ? 0.0 : s0 / tot;
                float tf;
                if (mode == 1)
                    tf = thr[(size_t)((y0 + y) % th_h) * th_w + (size_t)((x0 + x) % th_w)];
                else
                    tf = ign_threshold(x0 + x, y0 + y, sx, sy, sc);
                if (!(f <= (double)tf))
                    pick = ii[1];
            }
            if (pick >= K)
                pick = ii[0]; /* K==1: the reference would raise IndexError; not reachable for t>=0 */
            uint8_t *o = out + ((size_t)y * w + x) * 3;
            o[0] = out_colors[pick * 3 + 0];
            o[1] = out_colors[pick * 3 + 1];
            o[2] = out_colors[pick * 3 + 2];
            if (idx_out)
                idx_out[(size_t)y * w + x] = pick;
        }
    }
    free(t);
    return 0;
}

/* ------------------------------------------------------------------ */
/* error diffusion (pure-Python branch), dithering_lib.py:655-690      */
/* taps: dx[], dy[], wq[] = f32(weight/divisor) in list order          */
/* ------------------------------------------------------------------ */
int orc_error_diffusion_u8(const uint8_t *in, uint8_t *out, int h, int w, const float *pal, int K,
                           const uint8_t *out_colors, const uint8_t *lut_in, const int *dx, const int *dy,
                           const double *wq, int ntaps, int serpentine)
{
    orc_tree *t = (orc_tree *)malloc(sizeof(orc_tree));
    float *W = (float *)malloc(sizeof(float) * 3 * (size_t)h * w);
    int32_t *pick = (int32_t *)malloc(sizeof(int32_t) * (size_t)h * w);
    double pts[ORC_KMAX * 3];
    if (!t || !W || !pick || K < 1 || K > ORC_KMAX) {
        free(t);
        free(W);
        free(pick);
        return -1;
    }
    for (int i = 0; i < 3 * K; i++)
        pts[i] = (double)pal[i];
    orc_tree_build(t, pts, K);
    for (size_t i = 0; i < (size_t)h * w * 3; i++)
        W[i] = (float)(lut_in ? lut_in[in[i]] : in[i]);
    for (int y = 0; y < h; y++) {
        int rev = serpentine && (y & 1);
        int dir = rev ? -1 : 1;
        for (int step = 0; step < w; step++) {
            int x = rev ? (w - 1 - step) : step;
            float *p = W + ((size_t)y * w + x) * 3;
            float old[3];
            double q[3];
            for (int c = 0; c < 3; c++) {
                float v = p[c];
                v = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
                old[c] = v;
                q[c] = (double)v;
            }
            double d2[1];
            int ii[1] = {0};
            orc_tree_query(t, q, 1, d2, ii);
            int j = ii[0];
            pick[(size_t)y * w + x] = j;
            float err[3];
            for (int c = 0; c < 3; c++) {
                p[c] = pal[j * 3 + c];
                err[c] = old[c] - pal[j * 3 + c];
            }
            for (int k = 0; k < ntaps; k++) {
                int nx = x + dx[k] * dir, ny = y + dy[k];
                if (nx >= 0 && nx < w && ny >= 0 && ny < h) {
                    /* numpy: float32 array * python float -> the scalar is cast to f32 */
                    float wf = (float)wq[k];
                    float *tp = W + ((size_t)ny * w + nx) * 3;
                    for (int c = 0; c < 3; c++) {
                        float prod = err[c] * wf;
                        tp[c] = tp[c] + prod;
                    }
                }
            }
        }
    }
    for (size_t i = 0; i < (size_t)h * w; i++) {
        int j = pick[i];
        out[i * 3 + 0] = out_colors[j * 3 + 0];
        out[i * 3 + 1] = out_colors[j * 3 + 1];
        out[i * 3 + 2] = out_colors[j * 3 + 2];
    }
    free(t);
    free(W);
    free(pick);
    return 0;
}

/* ------------------------------------------------------------------ */
/* error diffusion, the numba branch: _error_diffusion_numba,          */
/* dithering_lib.py:213-308 (dispatch :638-653, arrays built :640-642) */
/* work float32; palette float32; weights float32; divisor float64.    */
/* TYPED PER NUMBA'S UNIFICATION RULE (fixtures pending): a variable   */
/* has ONE type, the unification of every assignment to it.            */
/*   r = work_2d[y, x, 0]   float32      (:239)                        */
/*   r = 0.0 / r = 255.0    float64      (:242-245)                    */
/* => r, g, b are float64 (holding the float32 value or a clamp bound) */
/* => dr = r - palette_arr[i, 0] is float64 (float64 - float32),       */
/*    dist = (dr*dr + dg*dg) + db*db in float64, no contraction        */
/*    (numba emits separate fmul / fadd without fastmath),             */
/*    strict `dist < best_dist`, best_dist = 1e20: first minimum;      */
/* => err0 = r - chosen0 is float64 (NOT rounded to float32);          */
/*    wgt = weights[k] / divisor  float32 / float64 = float64;         */
/*    work[ny, nx] += err0 * wgt: float64 product, float64 sum with    */
/*    the float32 element, ONE rounding to float32 on the store.       */
/* Rounds 1-3 read the scan and the error as float32; no numba release */
/* is known to type them so, and that reading is gone.                 */
/* PARITY UNPINNED: numba cannot be installed in the build image (no   */
/* network), so no reference output exists for this function; it is    */
/* checked against an independent numpy transcription of the same      */
/* lines (oracle.py: error_diffusion_numba_numpy) only.                */
/* ------------------------------------------------------------------ */
int orc_error_diffusion_numba_u8(const uint8_t *in, uint8_t *out, int h, int w, const float *pal, int K,
                                 const uint8_t *out_colors, const uint8_t *lut_in, const int *dx, const int *dy,
                                 const float *weights, double divisor, int ntaps, int serpentine)
{
    float *W = (float *)malloc(sizeof(float) * 3 * (size_t)h * w);
    int32_t *pick = (int32_t *)malloc(sizeof(int32_t) * (size_t)h * w);
    if (!W || !pick || K < 1) {
        free(W);
        free(pick);
        return -1;
    }
    for (size_t i = 0; i < (size_t)h * w * 3; i++)
        W[i] = (float)(lut_in ? lut_in[in[i]] : in[i]);
    for (int y = 0; y < h; y++) {
        int rev = serpentine && (y & 1);
        int dir = rev ? -1 : 1;
        for (int step = 0; step < w; step++) {
            int x = rev ? (w - 1 - step) : step;
            float *p = W + ((size_t)y * w + x) * 3;
            double v[3];  /* r, g, b: float64 by unification */
            for (int c = 0; c < 3; c++) {
                double t = (double)p[c];
                v[c] = t < 0.0 ? 0.0 : (t > 255.0 ? 255.0 : t);
            }
            int best = 0;
            double best_dist = 1e20;
            for (int i = 0; i < K; i++) {
                volatile double dr = v[0] - (double)pal[i * 3 + 0], dg = v[1] - (double)pal[i * 3 + 1], db = v[2] - (double)pal[i * 3 + 2];
                volatile double rr = dr * dr, gg = dg * dg, bb = db * db;  /* (volatile: no contraction into fma) */
                volatile double s1 = rr + gg;
                volatile double dist = s1 + bb;
                if (dist < best_dist) {
                    best_dist = dist;
                    best = i;
                }
            }
            pick[(size_t)y * w + x] = best;
            double err[3];
            for (int c = 0; c < 3; c++) {
                volatile double e = v[c] - (double)pal[best * 3 + c];
                p[c] = pal[best * 3 + c];
                err[c] = e;
            }
            for (int k = 0; k < ntaps; k++) {
                int nx = x + dx[k] * dir, ny = y + dy[k];
                if (nx >= 0 && nx < w && ny >= 0 && ny < h) {
                    volatile double wgt = (double)weights[k] / divisor;
                    float *tp = W + ((size_t)ny * w + nx) * 3;
                    for (int c = 0; c < 3; c++) {
                        volatile double prod = err[c] * wgt;
                        volatile double sum = (double)tp[c] + prod;
                        tp[c] = (float)sum;
                    }
                }
            }
        }
    }
    for (size_t i = 0; i < (size_t)h * w; i++) {
        int j = pick[i];
        out[i * 3 + 0] = out_colors[j * 3 + 0];
        out[i * 3 + 1] = out_colors[j * 3 + 1];
        out[i * 3 + 2] = out_colors[j * 3 + 2];
    }
    free(W);
    free(pick);
    return 0;
}

/* ------------------------------------------------------------------ */
/* blue noise, dithering_lib.py:381-399                                */
/* ------------------------------------------------------------------ */
typedef struct {
    uint32_t mt[624];
    int pos;
} mt19937;

static void mt_seed(mt19937 *s, uint32_t seed)
{
    /* numpy legacy RandomState(int) -> init_genrand */
    s->mt[0] = seed;
    for (int i = 1; i < 624; i++)
        s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->pos = 624;
}

static uint32_t mt_next(mt19937 *s)
{
    if (s->pos == 624) {
        uint32_t *mt = s->mt;
        for (int i = 0; i < 624; i++) {
            uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
            mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        s->pos = 0;
    }
    uint32_t y = s->mt[s->pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/* legacy rk_interval(max): masked rejection sampling on 32-bit words */
static uint32_t mt_interval(mt19937 *s, uint32_t max)
{
    if (max == 0)
        return 0;
    uint32_t mask = max;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    uint32_t v;
    while ((v = (mt_next(s) & mask)) > max)
        ;
    return v;
}

int orc_blue_noise(int size, uint32_t seed, float *out)
{
    int n = size * size;
    int *coords = (int *)malloc(sizeof(int) * (size_t)n); /* r*size+c */
    float *md = (float *)malloc(sizeof(float) * (size_t)n);
    if (!coords || !md) {
        free(coords);
        free(md);
        return -1;
    }
    for (int i = 0; i < n; i++) {
        coords[i] = i;
        md[i] = INFINITY;
    }
    mt19937 rng;
    mt_seed(&rng, seed);
    for (int i = n - 1; i >= 1; i--) {
        int j = (int)mt_interval(&rng, (uint32_t)i);
        int tmp = coords[i];
        coords[i] = coords[j];
        coords[j] = tmp;
    }
    int remaining = n;
    double denom = (double)(n - 1) + 1e-9;
    for (int i = 0; i < n; i++) {
        int bp = 0;
        float bv = md[coords[0]];
        for (int p = 1; p < remaining; p++) {
            float v = md[coords[p]];
            if (v > bv) {
                bv = v;
                bp = p;
            }
        }
        int best = coords[bp];
        out[best] = (float)((double)i / denom);
        memmove(coords + bp, coords + bp + 1, sizeof(int) * (size_t)(remaining - bp - 1));
        remaining--;
        int br = best / size, bc = best % size;
        for (int p = 0; p < remaining; p++) {
            int rr = coords[p] / size, cc = coords[p] % size;
            int d2 = (rr - br) * (rr - br) + (cc - bc) * (cc - bc);
            if ((float)d2 < md[coords[p]])
                md[coords[p]] = (float)d2;
        }
    }
    free(coords);
    free(md);
    return 0;
}

/* ------------------------------------------------------------------ */
/* one Lloyd pass on uint8 pixels: nearest centre in f64 (lowest index */
/* on ties, as sklearn's argmin), exact integer sums and counts        */
/*                                                                      */
/* mean == NULL: distances as ((x0-c0)^2 + (x1-c1)^2) + (x2-c2)^2.     */
/* mean != NULL: the label of a sample is decided as sklearn decides   */
/* it (sklearn/cluster/_k_means_lloyd.pyx, _update_chunk_dense, reached */
/* from KMeans.fit, dithering_lib.py:1854-1856): the data and the      */
/* centres are mean-centred in float64 (KMeans.fit: X -= X.mean(0)),   */
/* v_j = |c'_j|^2 - 2 x'.c'_j with                                      */
/*   |c'|^2 = (fl(c0'^2) + fl(c2'^2)) + fl(c1'^2)   numpy einsum        */
/*            "ij,ij->i" on three elements with 512-bit lanes           */
/*            (row_norms): products rounded, halves added first         */
/*   x'.c'  = fma(x2', c2', fma(x1', c1', fl(x0' c0')))   the OpenBLAS  */
/*            dgemm micro-kernels (Haswell / SkylakeX): one fused       */
/*            multiply-add chain over the three features                */
/*   v      = fl(|c'|^2 - 2 acc)   (alpha = -2 is exact)                */
/* and the first minimum wins.  In exact arithmetic v_j orders like the */
/* distance, so this only matters for samples equidistant from two      */
/* centres (k-means++ seeds are data points: integer centres tie on 0.3 */
/* to 1 % of the pixels of a structured image in the first iteration),  */
/* where rounding decides -- and decides identically on every x86 with  */
/* FMA and AVX-512, which is what generated tests/golden (kmx_*: 213    */
/* tied samples, all reproduced).  The formula's rounding error is      */
/* below 1e-9, so it is evaluated only where the two best exact         */
/* distances are within 1e-6 of each other; sklearn evaluates it        */
/* everywhere, with the same argmin.                                    */
/* ------------------------------------------------------------------ */
void orc_kmeans_step_sk(const uint8_t *px, long n, const double *centers, int K, const double *mean, int64_t *sums,
                        int64_t *counts, double *inertia)
{
    memset(sums, 0, sizeof(int64_t) * 3 * (size_t)K);
    memset(counts, 0, sizeof(int64_t) * (size_t)K);
    double *cc = NULL, *cn = NULL;
    if (mean) {
        cc = (double *)malloc(sizeof(double) * 3 * (size_t)K);
        cn = (double *)malloc(sizeof(double) * (size_t)K);
        for (int j = 0; j < K; j++) {
            for (int c = 0; c < 3; c++) cc[3 * j + c] = centers[3 * j + c] - mean[c];
            const double p0 = cc[3 * j] * cc[3 * j], p1 = cc[3 * j + 1] * cc[3 * j + 1], p2 = cc[3 * j + 2] * cc[3 * j + 2];
            cn[j] = (p0 + p2) + p1;
        }
    }
    double tot = 0.0;
    for (long i = 0; i < n; i++) {
        double x0 = px[3 * i], x1 = px[3 * i + 1], x2 = px[3 * i + 2];
        int best = 0;
        double bd = INFINITY, sd = INFINITY;
        for (int j = 0; j < K; j++) {
            double a = x0 - centers[3 * j], b = x1 - centers[3 * j + 1], c = x2 - centers[3 * j + 2];
            double d = (a * a + b * b) + c * c;
            if (d < bd) {
                sd = bd;
                bd = d;
                best = j;
            } else if (d < sd) {
                sd = d;
            }
        }
        if (mean && sd - bd <= 1e-6) {
            const double y0 = x0 - mean[0], y1 = x1 - mean[1], y2 = x2 - mean[2];
            double bv = INFINITY;
            for (int j = 0; j < K; j++) {
                double acc = y0 * cc[3 * j];
                acc = fma(y1, cc[3 * j + 1], acc);
                acc = fma(y2, cc[3 * j + 2], acc);
                const double v = cn[j] - 2.0 * acc;
                if (v < bv) {
                    bv = v;
                    best = j;
                }
            }
            const double a = x0 - centers[3 * best], b = x1 - centers[3 * best + 1], c = x2 - centers[3 * best + 2];
            bd = (a * a + b * b) + c * c;
        }
        sums[3 * best] += px[3 * i];
        sums[3 * best + 1] += px[3 * i + 1];
        sums[3 * best + 2] += px[3 * i + 2];
        counts[best]++;
        tot += bd;
    }
    free(cc);
    free(cn);
    *inertia = tot;
}

void orc_kmeans_step(const uint8_t *px, long n, const double *centers, int K, int64_t *sums,
                     int64_t *counts, double *inertia)
{
    orc_kmeans_step_sk(px, n, centers, K, NULL, sums, counts, inertia);
}

/* ------------------------------------------------------------------ */


/* ------------------------------------------------------------------ */
/* variable-weight diffusers (pure-Python branches of the reference)   */
/*   model 1  PerceptualDitherStrategy        dithering_lib.py:1030-1066 */
/*   model 2  HybridDitherStrategy            dithering_lib.py:1111-1155 (p0 = lum_factor, p1 = col_factor) */
/*   model 3  AdaptiveVarianceDitherStrategy  dithering_lib.py:984-1017 (gate[y*w+x] = var_map >= threshold) */
/*   model 4  OstromoukhovDitherStrategy      dithering_lib.py:1229-1266 (coef[256][3] = f32(c_k/divisor)) */
/* All are raster scans that push fl32(err * coefficient) into the      */
/* not-yet-visited neighbours; only model 4 clamps before the search    */
/* and supports serpentine.                                             */
/* ------------------------------------------------------------------ */
int orc_var_diffusion_u8(const uint8_t *in, uint8_t *out, int h, int w, const float *pal, int K,
                         const uint8_t *out_colors, const uint8_t *lut_in, int model, double p0, double p1,
                         int serpentine, const uint8_t *gate, const float *coef)
{
    static const int fs_dx[4] = {1, -1, 0, 1}, fs_dy[4] = {0, 1, 1, 1};
    static const float fs_w[4] = {7.0f / 16, 3.0f / 16, 5.0f / 16, 1.0f / 16};
    orc_tree *t = (orc_tree *)malloc(sizeof(orc_tree));
    float *W = (float *)malloc(sizeof(float) * 3 * (size_t)h * w);
    float *gray = (float *)malloc(sizeof(float) * (size_t)h * w);
    int32_t *pick = (int32_t *)malloc(sizeof(int32_t) * (size_t)h * w);
    double pts[ORC_KMAX * 3];
    if (!t || !W || !gray || !pick || K < 1 || K > ORC_KMAX || model < 1 || model > 4) {
        free(t); free(W); free(gray); free(pick);
        return -1;
    }
    for (int i = 0; i < 3 * K; i++) pts[i] = (double)pal[i];
    orc_tree_build(t, pts, K);
    for (size_t i = 0; i < (size_t)h * w * 3; i++) W[i] = (float)(lut_in ? lut_in[in[i]] : in[i]);
    for (size_t i = 0; i < (size_t)h * w; i++) {
        /* 0.299*R + 0.587*G + 0.114*B in float32, left to right */
        float a = 0.299f * W[3 * i], b = 0.587f * W[3 * i + 1], c = 0.114f * W[3 * i + 2];
        gray[i] = (a + b) + c;
    }
    const float lf = (float)p0, cf = (float)p1;
    for (int y = 0; y < h; y++) {
        const int rev = (model == 4) && serpentine && (y & 1);
        const int dir = rev ? -1 : 1;
        for (int step = 0; step < w; step++) {
            const int x = rev ? (w - 1 - step) : step;
            float *p = W + ((size_t)y * w + x) * 3;
            float old[3];
            double q[3];
            for (int c = 0; c < 3; c++) {
                float v = p[c];
                if (model == 4) v = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
                old[c] = v;
                q[c] = (double)v;
            }
            double d2[1];
            int ii[1] = {0};
            orc_tree_query(t, q, 1, d2, ii);
            const int j = ii[0];
            pick[(size_t)y * w + x] = j;
            float err[3];
            for (int c = 0; c < 3; c++) {
                p[c] = pal[j * 3 + c];
                err[c] = old[c] - pal[j * 3 + c];
            }
            if (model == 4) {
                float lum = (0.299f * old[0] + 0.587f * old[1]) + 0.114f * old[2];
                lum = lum < 0.0f ? 0.0f : (lum > 255.0f ? 255.0f : lum);
                const float *ck = coef + 3 * (int)lum;
                int nx = x + dir;
                if (nx >= 0 && nx < w)
                    for (int c = 0; c < 3; c++) p[3 * dir + c] = p[3 * dir + c] + err[c] * ck[0];
                if (y + 1 < h) {
                    nx = x - dir;
                    float *r1 = W + ((size_t)(y + 1) * w) * 3;
                    if (nx >= 0 && nx < w)
                        for (int c = 0; c < 3; c++) r1[3 * nx + c] = r1[3 * nx + c] + err[c] * ck[1];
                    for (int c = 0; c < 3; c++) r1[3 * x + c] = r1[3 * x + c] + err[c] * ck[2];
                }
                continue;
            }
            float fe[3] = {err[0], err[1], err[2]};
            float scale = 1.0f;
            if (model == 1) {
                const float lum = gray[(size_t)y * w + x];
                scale = 0.5f + 0.5f * (lum / 255.0f);
            } else if (model == 2) {
                const float lv = (0.299f * err[0] + 0.587f * err[1]) + 0.114f * err[2];
                const float el[3] = {0.299f * lv, 0.587f * lv, 0.114f * lv};
                for (int c = 0; c < 3; c++) {
                    const float ec = err[c] - el[c];
                    fe[c] = lf * el[c] + cf * ec;
                }
            } else if (model == 3) {
                if (!gate[(size_t)y * w + x]) continue;
            }
            for (int k = 0; k < 4; k++) {
                const int nx = x + fs_dx[k], ny = y + fs_dy[k];
                if (nx < 0 || nx >= w || ny < 0 || ny >= h) continue;
                const float wk = (model == 1) ? fs_w[k] * scale : fs_w[k];
                float *tp = W + ((size_t)ny * w + nx) * 3;
                for (int c = 0; c < 3; c++) tp[c] = tp[c] + fe[c] * wk;
            }
        }
    }
    for (size_t i = 0; i < (size_t)h * w; i++) {
        const int j = pick[i];
        out[i * 3 + 0] = out_colors[j * 3 + 0];
        out[i * 3 + 1] = out_colors[j * 3 + 1];
        out[i * 3 + 2] = out_colors[j * 3 + 2];
    }
    free(t); free(W); free(gray); free(pick);
    return 0;
}

/* scipy.ndimage.uniform_filter1d (NI_UniformFilter1D) along one axis, mode='nearest', origin 0:    */
/* float32 in/out, double running sum tmp += (entering - leaving), out = tmp / size                  */
void orc_uniform_filter1d_f32(const float *in, float *out, int n_lines, int length, long line_stride,
                              long elem_stride, int size)
{
    const int s1 = size / 2;
    double *buf = (double *)malloc(sizeof(double) * (size_t)(length + size));
    for (int l = 0; l < n_lines; l++) {
        const float *src = in + (size_t)l * line_stride;
        float *dst = out + (size_t)l * line_stride;
        for (int i = 0; i < length + size - 1; i++) {
            int k = i - s1;
            k = k < 0 ? 0 : (k >= length ? length - 1 : k);
            buf[i] = (double)src[(size_t)k * elem_stride];
        }
        double tmp = 0.0;
        for (int i = 0; i < size; i++) tmp += buf[i];
        dst[0] = (float)(tmp / (double)size);
        for (int i = 1; i < length; i++) {
            tmp += buf[i + size - 1] - buf[i - 1];
            dst[(size_t)i * elem_stride] = (float)(tmp / (double)size);
        }
    }
    free(buf);
}

/* ------------------------------------------------------------------ */
/* HybridDitherStrategy, the numba branch: _hybrid_numba,              */
/* dithering_lib.py:1396-1494 (dispatch :1114-1125).                   */
/* work float32; palette float32; lum_factor / col_factor float64.     */
/* TYPED PER NUMBA'S UNIFICATION RULE (fixtures pending), exactly as   */
/* orc_error_diffusion_numba_u8 above: r, g, b are assigned a float32  */
/* element (:1408-1410) and float64 literals (:1411-1422) => float64;  */
/* the scan, err = r - chosen, and everything computed from it are     */
/* float64, every product and sum rounded on its own (no fastmath):    */
/*   lum_err_val = (0.299*err0 + 0.587*err1) + 0.114*err2   (:1445)    */
/*   lum_c = w_c * lum_err_val                              (:1446-48) */
/*   fe_c = lum_factor*lum_c + col_factor*(err_c - lum_c)   (:1449-51) */
/*   work[.., c] += fe_c * (7.0/16.0 | 3.0/16.0 | 5.0/16.0 | 1.0/16.0) */
/*      -- float64 product, float64 sum with the float32 element, one  */
/*      rounding to float32 on the store                  (:1453-1468) */
/* Unlike the pure-Python branch of the same strategy (:1127-1152) the */
/* value IS clamped to [0, 255] before the search (:1411-1422).        */
/* PARITY UNPINNED: numba cannot be installed in the build image.      */
/* ------------------------------------------------------------------ */
int orc_hybrid_numba_u8(const uint8_t *in, uint8_t *out, int h, int w, const float *pal, int K, const uint8_t *out_colors,
                        const uint8_t *lut_in, double lum_factor, double col_factor)
{
    float *W = (float *)malloc(sizeof(float) * 3 * (size_t)h * w);
    int32_t *pick = (int32_t *)malloc(sizeof(int32_t) * (size_t)h * w);
    if (!W || !pick || K < 1) {
        free(W);
        free(pick);
        return -1;
    }
    for (size_t i = 0; i < (size_t)h * w * 3; i++)
        W[i] = (float)(lut_in ? lut_in[in[i]] : in[i]);
    static const int tdx[4] = {1, -1, 0, 1}, tdy[4] = {0, 1, 1, 1};
    const double tw[4] = {7.0 / 16.0, 3.0 / 16.0, 5.0 / 16.0, 1.0 / 16.0};
    static const double lw[3] = {0.299, 0.587, 0.114};
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            float *p = W + ((size_t)y * w + x) * 3;
            double v[3];
            for (int c = 0; c < 3; c++) {
                double t = (double)p[c];
                v[c] = t < 0.0 ? 0.0 : (t > 255.0 ? 255.0 : t);
            }
            int best = 0;
            double best_dist = 1e20;
            for (int i = 0; i < K; i++) {
                volatile double dr = v[0] - (double)pal[i * 3 + 0], dg = v[1] - (double)pal[i * 3 + 1], db = v[2] - (double)pal[i * 3 + 2];
                volatile double rr = dr * dr, gg = dg * dg, bb = db * db;
                volatile double s1 = rr + gg;
                volatile double dist = s1 + bb;
                if (dist < best_dist) {
                    best_dist = dist;
                    best = i;
                }
            }
            pick[(size_t)y * w + x] = best;
            double err[3];
            for (int c = 0; c < 3; c++) {
                volatile double e = v[c] - (double)pal[best * 3 + c];
                p[c] = pal[best * 3 + c];
                err[c] = e;
            }
            volatile double a0 = lw[0] * err[0], a1 = lw[1] * err[1], a2 = lw[2] * err[2];
            volatile double a01 = a0 + a1;
            volatile double lum = a01 + a2;
            double fe[3];
            for (int c = 0; c < 3; c++) {
                volatile double lc = lw[c] * lum;
                volatile double rest = err[c] - lc;
                volatile double t1 = lum_factor * lc, t2 = col_factor * rest;
                volatile double f = t1 + t2;
                fe[c] = f;
            }
            for (int k = 0; k < 4; k++) {
                int nx = x + tdx[k], ny = y + tdy[k];
                if (nx >= 0 && nx < w && ny >= 0 && ny < h) {
                    float *tp = W + ((size_t)ny * w + nx) * 3;
                    for (int c = 0; c < 3; c++) {
                        volatile double prod = fe[c] * tw[k];
                        volatile double sum = (double)tp[c] + prod;
                        tp[c] = (float)sum;
                    }
                }
            }
        }
    }
    for (size_t i = 0; i < (size_t)h * w; i++) {
        int j = pick[i];
        out[i * 3 + 0] = out_colors[j * 3 + 0];
        out[i * 3 + 1] = out_colors[j * 3 + 1];
        out[i * 3 + 2] = out_colors[j * 3 + 2];
    }
    free(W);
    free(pick);
    return 0;
}
