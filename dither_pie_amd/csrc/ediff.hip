// Error diffusion (ErrorDiffusionDitherStrategy.dither, pure-Python branch,
// dithering_lib.py:655-690) for packed uint8 RGB frames.
//
// The reference pushes err*w into the not-yet-visited neighbours in raster order, in float32, one
// rounding for the product and one for the add.  Both kernels below use the equivalent PULL form:
// when pixel (y,x) is reached, its value is rebuilt as  pix + sum_k fl(err(src_k) * wq_k)  with the
// source pixels taken in the order the reference visited them (earlier rows first, then the scan
// order inside a row), which reproduces the same float32 sums bit for bit.  The taps arrive
// pre-sorted in that order (dy descending, dx descending) from the launcher.
//
// ed_wavefront_kernel (serpentine off): one 64-lane wave per frame, lane = image row inside a
//   64-row band, anti-diagonal schedule: lane L works on x = t - skew*L at step t, so every source
//   pixel of a row above was finished `skew` steps earlier.  Errors live in a 16-deep per-row ring
//   in LDS; the two rows that cross a band boundary go through a small global buffer.
// ed_serial_kernel (any scan, used for serpentine): rows are strictly sequential under a
//   serpentine scan (the first pixel of row y+1 needs the last pixel of row y), so parallelism
//   comes from frames only: lane = frame, error rows interleaved across lanes in global memory.
//
// Nearest colour: brute force in float64 with the KD-tree's arithmetic; an exact tie between the
// two smallest distances (they do occur: diffused errors are dyadic) is resolved by replaying
// scipy's traversal (tree_query<1>) unless the palette fits one leaf (then the lowest index wins).
#include "dp_internal.h"
#include "tree_query.cuh"

namespace dp {
namespace {

constexpr int kMaxTaps = 16;
constexpr int kRing = 16;  // per-row error ring depth (positions), power of two

struct Taps {
    int n;
    int skew;
    int dx[kMaxTaps], dy[kMaxTaps];
    float wq[kMaxTaps];
};

__device__ __forceinline__ int nearest_f64(const PalDev &pal, const float o0, const float o1, const float o2)
{
    const double x0 = (double)o0, x1 = (double)o1, x2 = (double)o2;
    const double inf = __longlong_as_double(0x7ff0000000000000LL);
    double b0 = inf, b1 = inf;
    int i0 = 0;
    const int K = pal.K;
    for (int j = 0; j < K; ++j) {
        const double d = sq_dist3(pal.pts + 3 * j, x0, x1, x2);
        if (d < b0) {
            b1 = b0;
            b0 = d;
            i0 = j;
        } else if (d < b1) {
            b1 = d;
        }
    }
    if (b0 == b1 && K > kLeafSize) {
        double d2[1];
        int ii[1];
        tree_query<1>(pal, x0, x1, x2, d2, ii);
        i0 = ii[0];
    }
    return i0;
}

__device__ __forceinline__ float clamp255(const float v) { return v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v); }

__global__ __launch_bounds__(64) void ed_wavefront_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                          const int h, const int w, const PalDev pal, const Taps taps,
                                                          float *__restrict__ bnd_all)
{
    __shared__ float s_ring[64][kRing][3];
    __shared__ uint8_t s_lut[256];
    const int L = threadIdx.x;
    const size_t f = blockIdx.x;
    for (int i = L; i < 256; i += 64) s_lut[i] = pal.lut_in ? pal.lut_in[i] : (uint8_t)i;
    const uint8_t *fin = in + f * (size_t)h * w * 3;
    uint8_t *fout = out + f * (size_t)h * w * 3;
    // boundary rows: [2 buffers][2 rows][w][3] floats per frame
    float *bnd = bnd_all + f * (size_t)4 * w * 3;
    const int skew = taps.skew;
    const int n_bands = (h + 63) / 64;
    __syncthreads();

    for (int band = 0; band < n_bands; ++band) {
        const int r = band * 64 + L;
        const float *bprev = bnd + (size_t)((band + 1) & 1) * 2 * w * 3;  // written by the previous band
        float *bnext = bnd + (size_t)(band & 1) * 2 * w * 3;
        const int rows_here = min(64, h - band * 64);
        const int steps = w + skew * (rows_here - 1);
        for (int t = 0; t < steps; ++t) {
            const int x = t - skew * L;
            const bool act = (r < h) && x >= 0 && x < w;
            float e0 = 0.f, e1 = 0.f, e2 = 0.f;
            if (act) {
                const uint8_t *p = fin + ((size_t)r * w + x) * 3;
                float a0 = (float)s_lut[p[0]], a1 = (float)s_lut[p[1]], a2 = (float)s_lut[p[2]];
                for (int k = 0; k < taps.n; ++k) {
                    const int sxp = x - taps.dx[k];
                    const int sr = r - taps.dy[k];
                    if (sxp < 0 || sxp >= w || sr < 0) continue;
                    const int rel = L - taps.dy[k];
                    float s0, s1, s2;
                    if (rel >= 0) {
                        s0 = s_ring[rel][sxp & (kRing - 1)][0];
                        s1 = s_ring[rel][sxp & (kRing - 1)][1];
                        s2 = s_ring[rel][sxp & (kRing - 1)][2];
                    } else {
                        const float *b = bprev + ((size_t)(rel + 2) * w + sxp) * 3;
                        s0 = b[0];
                        s1 = b[1];
                        s2 = b[2];
                    }
                    const float wq = taps.wq[k];
                    a0 = __fadd_rn(a0, __fmul_rn(s0, wq));
                    a1 = __fadd_rn(a1, __fmul_rn(s1, wq));
                    a2 = __fadd_rn(a2, __fmul_rn(s2, wq));
                }
                const float o0 = clamp255(a0), o1 = clamp255(a1), o2 = clamp255(a2);
                const int j = nearest_f64(pal, o0, o1, o2);
                e0 = __fsub_rn(o0, (float)pal.pts[3 * j]);
                e1 = __fsub_rn(o1, (float)pal.pts[3 * j + 1]);
                e2 = __fsub_rn(o2, (float)pal.pts[3 * j + 2]);
                const uint32_t c = pal.out_rgb[j];
                uint8_t *o = fout + ((size_t)r * w + x) * 3;
                o[0] = (uint8_t)c;
                o[1] = (uint8_t)(c >> 8);
                o[2] = (uint8_t)(c >> 16);
            }
            __syncthreads();  // every pull of this step is done before any ring slot is overwritten
            if (act) {
                s_ring[L][x & (kRing - 1)][0] = e0;
                s_ring[L][x & (kRing - 1)][1] = e1;
                s_ring[L][x & (kRing - 1)][2] = e2;
                if (L >= 62) {
                    float *b = bnext + ((size_t)(L - 62) * w + x) * 3;
                    b[0] = e0;
                    b[1] = e1;
                    b[2] = e2;
                }
            }
            __syncthreads();
        }
        __threadfence();
        __syncthreads();
    }
}

// lane = frame; err rows: ring[3][w][3][n_frames] floats (frame index fastest => coalesced)
__global__ __launch_bounds__(64) void ed_serial_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                       const int64_t n_frames, const int h, const int w,
                                                       const PalDev pal, const Taps taps, const int serpentine,
                                                       float *__restrict__ ring)
{
    const int64_t f = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (f >= n_frames) return;
    const uint8_t *fin = in + (size_t)f * h * w * 3;
    uint8_t *fout = out + (size_t)f * h * w * 3;
    const size_t nf = (size_t)n_frames;
    for (int y = 0; y < h; ++y) {
        const bool rev = serpentine && (y & 1);
        for (int step = 0; step < w; ++step) {
            const int x = rev ? (w - 1 - step) : step;
            const uint8_t *p = fin + ((size_t)y * w + x) * 3;
            uint32_t c0 = p[0], c1 = p[1], c2 = p[2];
            if (pal.lut_in) {
                c0 = pal.lut_in[c0];
                c1 = pal.lut_in[c1];
                c2 = pal.lut_in[c2];
            }
            float a0 = (float)c0, a1 = (float)c1, a2 = (float)c2;
            for (int k = 0; k < taps.n; ++k) {
                const int sr = y - taps.dy[k];
                if (sr < 0) continue;
                const int sdir = (serpentine && (sr & 1)) ? -1 : 1;
                const int sxp = x - taps.dx[k] * sdir;
                if (sxp < 0 || sxp >= w) continue;
                const float *e = ring + (((size_t)(sr % 3) * w + sxp) * 3) * nf + f;
                const float wq = taps.wq[k];
                a0 = __fadd_rn(a0, __fmul_rn(e[0], wq));
                a1 = __fadd_rn(a1, __fmul_rn(e[nf], wq));
                a2 = __fadd_rn(a2, __fmul_rn(e[2 * nf], wq));
            }
            const float o0 = clamp255(a0), o1 = clamp255(a1), o2 = clamp255(a2);
            const int j = nearest_f64(pal, o0, o1, o2);
            float *e = ring + (((size_t)(y % 3) * w + x) * 3) * nf + f;
            e[0] = __fsub_rn(o0, (float)pal.pts[3 * j]);
            e[nf] = __fsub_rn(o1, (float)pal.pts[3 * j + 1]);
            e[2 * nf] = __fsub_rn(o2, (float)pal.pts[3 * j + 2]);
            const uint32_t c = pal.out_rgb[j];
            uint8_t *o = fout + ((size_t)y * w + x) * 3;
            o[0] = (uint8_t)c;
            o[1] = (uint8_t)(c >> 8);
            o[2] = (uint8_t)(c >> 16);
        }
    }
}

}  // namespace

size_t error_diffusion_ws_bytes(int64_t n_frames, int h, int w)
{
    (void)h;
    // wavefront: 4 boundary rows per frame; serial: 3 error rows per frame; take the larger
    return (size_t)n_frames * (size_t)w * 3 * sizeof(float) * 4 + 256;
}

int launch_error_diffusion(const uint8_t *in, uint8_t *out, int64_t n_frames, int h, int w, const PalDev &pal,
                           const int32_t *dx, const int32_t *dy, const float *wq, int ntaps, int serpentine,
                           void *ws, size_t ws_bytes, hipStream_t s)
{
    (void)ws_bytes;
    Taps t;
    t.n = ntaps;
    // reference visiting order of the source pixels: earlier rows first (dy descending), and inside a
    // source row in its scan order, which is dx descending for both scan directions
    int order[kMaxTaps];
    for (int i = 0; i < ntaps; ++i) order[i] = i;
    for (int i = 1; i < ntaps; ++i) {  // stable insertion sort
        const int v = order[i];
        int j = i - 1;
        while (j >= 0 && (dy[order[j]] < dy[v] || (dy[order[j]] == dy[v] && dx[order[j]] < dx[v]))) {
            order[j + 1] = order[j];
            --j;
        }
        order[j + 1] = v;
    }
    int skew = 1;
    for (int i = 0; i < ntaps; ++i) {
        t.dx[i] = dx[order[i]];
        t.dy[i] = dy[order[i]];
        t.wq[i] = wq[order[i]];
        // a source on row y-dy at x-dx must be finished strictly before step t: skew*dy > -dx
        if (t.dy[i] > 0) {
            const int need = (-t.dx[i]) / t.dy[i] + 1;
            if (need > skew) skew = need;
        }
    }
    t.skew = skew;
    for (int i = ntaps; i < kMaxTaps; ++i) {
        t.dx[i] = t.dy[i] = 0;
        t.wq[i] = 0.f;
    }
    ProfMark *pm = prof_begin(s);
    if (!serpentine && skew * 2 + 2 < kRing) {
        if (n_frames > 0x7fffffff) {
            set_error("dp_error_diffusion_u8: too many frames for one launch");
            return DP_EINVAL;
        }
        hipLaunchKernelGGL(ed_wavefront_kernel, dim3((unsigned)n_frames), dim3(64), 0, s, in, out, h, w, pal, t,
                           reinterpret_cast<float *>(ws));
    } else {
        const int64_t blocks = (n_frames + 63) / 64;
        hipLaunchKernelGGL(ed_serial_kernel, dim3((unsigned)blocks), dim3(64), 0, s, in, out, n_frames, h, w, pal, t,
                           serpentine, reinterpret_cast<float *>(ws));
    }
    prof_end(pm, s);
    DP_HIP(hipGetLastError());
    return DP_OK;
}

}  // namespace dp
