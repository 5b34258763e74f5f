// Variable-weight error diffusers of the reference, frame-parallel version:
//   model 1  PerceptualDitherStrategy        dithering_lib.py:1030-1066   weight_k * (0.5 + 0.5*lum/255) of the SOURCE pixel
//   model 2  HybridDitherStrategy            dithering_lib.py:1111-1155   the error is split into luminance/colour parts
//   model 3  AdaptiveVarianceDitherStrategy  dithering_lib.py:984-1017    sources diffuse only where the local variance is high
//   model 4  OstromoukhovDitherStrategy      dithering_lib.py:1229-1266   3 taps, coefficients from a 256-entry table indexed by
//                                                                          the source's luminance; clamps; optional serpentine
// Same pull formulation as ediff.hip (pixel + sum of fl32(err_src * coefficient) in the reference's visiting
// order), with a fourth float per stored error that carries what the coefficient depends on (sensitivity, gate or
// table row).  var_wavefront_kernel runs them on the anti-diagonal schedule of ediff.hip (serpentine off);
// var_serial_kernel (lane = frame) covers Ostromoukhov's serpentine scan, whose rows are strictly sequential.
//
// variance_gate: scipy.ndimage.uniform_filter(size, mode='nearest') restated (float32 passes along axis 0 then
// axis 1, each a double running sum `tmp += entering - leaving`, out = tmp/size) on gray and gray^2, then
// max(0, mean_sq - sq_mean^2) >= threshold  (dithering_lib.py:988-992, 1019-1025).
#include <algorithm>
#include <cstdlib>

#include "dp_internal.h"
#include "ed_nearest.hip.h"
#include "tree_query.hip.h"
#include "wave_util.hip.h"

namespace dp {
namespace {

__device__ __forceinline__ float clamp255f(const float v) { return v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v); }

// near ties of the float32 scan: float64 scan with the KD-tree's arithmetic, scipy's traversal on exact ties
template <int CAP>
__device__ __forceinline__ int nearest_any_f64(const PalDev &pal, const float o0, const float o1, const float o2)
{
    int i0 = 0;
    const int K = pal.K;
    const double x0 = (double)o0, x1 = (double)o1, x2 = (double)o2;
    const double inf = __longlong_as_double(0x7ff0000000000000LL);
    double e0 = inf, e1 = inf;
    for (int j = 0; j < K; ++j) {
        const double d = sq_dist3(pal.pts + 3 * j, x0, x1, x2);
        if (d < e0) {
            e1 = e0;
            e0 = d;
            i0 = j;
        } else if (d < e1) {
            e1 = d;
        }
    }
    if (e0 == e1 && K > kLeafSize) {
        double d2[1];
        int ii[1];
        tree_query<1, CAP>(pal, x0, x1, x2, d2, ii);
        i0 = ii[0];
    }
    return i0;
}

template <int CAP>
__device__ __forceinline__ int nearest_any(const PalDev &pal, const float4 *__restrict__ cand, const float o0,
                                           const float o1, const float o2)
{
    // float32 prefilter with a relative AND absolute margin (values are not clamped here, distances can be tiny).
    // `cand` = {x, y, z, out_rgb}: in LDS for the wavefront kernel (broadcast reads, four in flight), global otherwise
    float b0 = __int_as_float(0x7f800000), b1 = b0;
    int i0 = 0;
    const int K = pal.K;
    auto visit = [&](const float4 c, const int j) {
        keep_record_whole(c);  // (ds_read_b128 instead of the narrowed ds_read_b96: ed_nearest.hip.h)
        const float a = c.x - o0, b = c.y - o1, cc = c.z - o2;
        const float d = __fmaf_rn(a, a, __fmaf_rn(b, b, cc * cc));
        const bool lt0 = d < b0;
        b1 = lt0 ? b0 : (d < b1 ? d : b1);
        i0 = lt0 ? j : i0;
        b0 = lt0 ? d : b0;
    };
    int jj = 0;
    for (; jj + 4 <= K; jj += 4) {
        const float4 c0 = cand[jj], c1 = cand[jj + 1], c2 = cand[jj + 2], c3 = cand[jj + 3];
        visit(c0, jj);
        visit(c1, jj + 1);
        visit(c2, jj + 2);
        visit(c3, jj + 3);
    }
    for (; jj < K; ++jj) visit(cand[jj], jj);
    if (b1 > b0 * 1.000002f) return i0;
    if (b0 == b1) {
        const int tied = integer_tie_choice(pal, cand, o0, o1, o2, b0, i0);
        if (tied >= 0) return tied;
    }
    return nearest_any_f64<CAP>(pal, o0, o1, o2);
}

// Palettes of up to 16 colours, any query point: the key scan of nearest_color_cells (ed_nearest.hip.h) over the list of
// the point's cell in the EXTENDED 16^3 table (ediff.hip, build_ed_cells: the outermost cells stand for the half-spaces
// beyond the cube, a point is looked up by its clamped coordinates).  These diffusers do not clamp their values, and a
// wave nearly always holds a point outside the cube: without this every step scanned the whole palette.
template <int CAP>
__device__ __forceinline__ int nearest_ext(const PalDev &pal, const float4 *__restrict__ cand, const uint32_t *__restrict__ ext,
                                           const float o0, const float o1, const float o2)
{
    const int c0 = min(max((int)o0 >> 4, 0), 15), c1 = min(max((int)o1 >> 4, 0), 15), c2 = min(max((int)o2 >> 4, 0), 15);
    const uint32_t e = ext[c0 | (c1 << 4) | (c2 << 8)];
    const int n = (int)(e & 15u);
    if (n <= 7) {
        const int j1 = (e >> 4) & 15, j2 = (e >> 8) & 15, j3 = (e >> 12) & 15, j4 = (e >> 16) & 15;
        const float4 q1 = cand[j1], q2 = cand[j2], q3 = cand[j3], q4 = cand[j4];
        int k1 = ed_key(q1, o0, o1, o2, 1u), k2 = ed_key(q2, o0, o1, o2, 2u), k3 = ed_key(q3, o0, o1, o2, 3u),
            k4 = ed_key(q4, o0, o1, o2, 4u);
        int m0 = min(min(k1, k2), k3), m1 = ed_med3(k1, k2, k3);
        m1 = ed_med3(m0, m1, k4);
        m0 = min(m0, k4);
        if (n > 4) {
            const int j5 = (e >> 20) & 15, j6 = (e >> 24) & 15, j7 = (e >> 28) & 15;
            const float4 q5 = cand[j5], q6 = cand[j6], q7 = cand[j7];
            const int k5 = ed_key(q5, o0, o1, o2, 5u), k6 = ed_key(q6, o0, o1, o2, 6u), k7 = ed_key(q7, o0, o1, o2, 7u);
            m1 = ed_med3(m0, m1, k5);
            m0 = min(m0, k5);
            m1 = ed_med3(m0, m1, k6);
            m0 = min(m0, k6);
            m1 = ed_med3(m0, m1, k7);
            m0 = min(m0, k7);
        }
        const float f0 = __int_as_float(m0 & ~7), f1 = __int_as_float(m1 & ~7);
        if (f1 > f0 * 1.000003f) return (int)((e >> (4 * (m0 & 7))) & 15u);
    }
    return nearest_any<CAP>(pal, cand, o0, o1, o2);  // long lists, near ties, exact ties: the full scan and its validation
}

// Palettes of 17..256 colours, any query point: the key scan over the byte list of the point's cell in the extended 16^3 table
// (host_logic.h: EdTables::ext16 -- the outermost cells are unbounded, a point is looked up by its clamped coordinates).  Lists of
// up to twelve entries are padded to a multiple of four positions by the builder; longer ones, near ties and exact ties go to the full
// scan and its validation.  Until round 5 every step of these diffusers scanned the whole palette above 16 colours.
template <int CAP>
__device__ __forceinline__ int nearest_ext16(const PalDev &pal, const float4 *__restrict__ cand, const uint4 *__restrict__ ext16,
                                             const float o0, const float o1, const float o2, const bool inside)
{
    const int c0 = min(max((int)o0 >> 4, 0), 15), c1 = min(max((int)o1 >> 4, 0), 15), c2 = min(max((int)o2 >> 4, 0), 15);
    uint4 blk = ext16[c0 | (c1 << 4) | (c2 << 8)];
    int n = (int)(blk.x & 255u);
    if (n == 254) {
        // an outermost cell with too long a list: down its octree by the CLAMPED integer coordinates (a point beyond the cube falls
        // into the outer children, which are unbounded on that side); sizes 8, 4, 2, 1 -- a unit child is a list or 255
        const uint32_t i0 = (uint32_t)min(max((int)o0, 0), 255), i1 = (uint32_t)min(max((int)o1, 0), 255), i2 = (uint32_t)min(max((int)o2, 0), 255);
        for (int bit = 3; n == 254 && bit >= 0; --bit) {
            const uint32_t sub = ((i0 >> bit) & 1u) | (((i1 >> bit) & 1u) << 1) | (((i2 >> bit) & 1u) << 2);
            blk = pal.ed_ext_nodes[(size_t)(blk.x >> 8) * 8 + sub];
            n = (int)(blk.x & 255u);
        }
    }
    // A lane whose cell has no usable list and whose point lies beyond the cube must scan the whole palette -- and then the wave
    // executes that scan anyway: all its lanes take it, instead of the scan PLUS the list paths of the others (a palette crowded at a
    // face of the cube -- any palette under use_gamma, at the dark end -- has such a lane in most waves: 256 colours 59 -> 87 ms per
    // 1080p frame with the three paths side by side).
    if (__ballot((n < 1 || n > 15) && !(inside && pal.ed_cells != nullptr)) != 0ull) return nearest_any<CAP>(pal, cand, o0, o1, o2);
    if (pal.K > 256) {
        // 257..1024 colours: up to twelve ten-bit entries (ed_nearest.hip.h: ed_wide_entry); the same two stages as below
        if (n >= 1 && n <= 12) {
            int m0 = 0x7fffffff, m1 = 0x7fffffff;
            for (uint32_t g = 0; (int)g < n; g += 4u) {
                const int j1 = ed_wide_entry(blk, g), j2 = ed_wide_entry(blk, g + 1u), j3 = ed_wide_entry(blk, g + 2u), j4 = ed_wide_entry(blk, g + 3u);
                const float4 q1 = cand[j1], q2 = cand[j2], q3 = cand[j3], q4 = cand[j4];
                const int k1 = ed_key16(q1, o0, o1, o2, g), k2 = ed_key16(q2, o0, o1, o2, g + 1u), k3 = ed_key16(q3, o0, o1, o2, g + 2u),
                          k4 = ed_key16(q4, o0, o1, o2, g + 3u);
                m1 = ed_med3(m0, m1, k1);
                m0 = min(m0, k1);
                m1 = ed_med3(m0, m1, k2);
                m0 = min(m0, k2);
                m1 = ed_med3(m0, m1, k3);
                m0 = min(m0, k3);
                m1 = ed_med3(m0, m1, k4);
                m0 = min(m0, k4);
            }
            const float f0 = __int_as_float(m0 & ~15), f1 = __int_as_float(m1 & ~15);
            if (f1 > f0 * 1.000004f) return ed_wide_entry(blk, (uint32_t)m0 & 15u);
            float b0 = __int_as_float(0x7f800000), b1 = b0;
            int i0 = 0;
            for (int i = 0; i < n; ++i) {
                const int j = ed_wide_entry(blk, (uint32_t)i);
                const float4 c = cand[j];
                const float a0 = c.x - o0, a1 = c.y - o1, a2 = c.z - o2;
                const float d = __fmaf_rn(a0, a0, __fmaf_rn(a1, a1, a2 * a2));
                const bool lt0 = d < b0;
                b1 = lt0 ? b0 : (d < b1 ? d : b1);
                i0 = lt0 ? j : i0;
                b0 = lt0 ? d : b0;
            }
            if (b1 > b0 * 1.000002f) return i0;
        }
        if (n > 12 && inside && pal.ed_cells) return nearest_color_cells<CAP, false, CAP == kQueueLarge>(pal, cand, nullptr, o0, o1, o2, nullptr, nullptr, nullptr);
        return nearest_any<CAP>(pal, cand, o0, o1, o2);
    }
    if (n >= 1 && n <= 12) {
        uint4 b = blk;
        b.x = __funnelshift_r(b.x, b.y, 8);  // drop the count byte
        b.y = __funnelshift_r(b.y, b.z, 8);
        b.z = __funnelshift_r(b.z, b.w, 8);
        int m0 = 0x7fffffff, m1 = 0x7fffffff;
        uint32_t tag = 0u;
        for (int left = n; left > 0; left -= 4, tag += 4u) {
            const int j1 = (int)(b.x & 255u), j2 = (int)((b.x >> 8) & 255u), j3 = (int)((b.x >> 16) & 255u), j4 = (int)(b.x >> 24);
            const float4 q1 = cand[j1], q2 = cand[j2], q3 = cand[j3], q4 = cand[j4];
            const int k1 = ed_key16(q1, o0, o1, o2, tag), k2 = ed_key16(q2, o0, o1, o2, tag + 1u), k3 = ed_key16(q3, o0, o1, o2, tag + 2u),
                      k4 = ed_key16(q4, o0, o1, o2, tag + 3u);
            m1 = ed_med3(m0, m1, k1);
            m0 = min(m0, k1);
            m1 = ed_med3(m0, m1, k2);
            m0 = min(m0, k2);
            m1 = ed_med3(m0, m1, k3);
            m0 = min(m0, k3);
            m1 = ed_med3(m0, m1, k4);
            m0 = min(m0, k4);
            b.x = b.y;
            b.y = b.z;
            b.z = 0u;
        }
        const float f0 = __int_as_float(m0 & ~15), f1 = __int_as_float(m1 & ~15);
        if (f1 > f0 * 1.000004f) {
            const uint32_t pos = ((uint32_t)m0 & 15u) + 1u;  // byte 1..12 of the block
            const uint32_t wsel = pos < 4u ? blk.x : (pos < 8u ? blk.y : (pos < 12u ? blk.z : blk.w));
            return (int)((wsel >> ((pos & 3u) * 8u)) & 255u);
        }
    }
    if (n >= 1 && n <= 15) {
        // longer lists and near ties of the key scan: the float32 scan of nearest_any over the LISTED entries only (a wave pays for
        // the slowest of its 64 lanes: sending it through the whole palette for one lane's 13-entry list cost most of the gain)
        float b0 = __int_as_float(0x7f800000), b1 = b0;
        int i0 = 0;
        uint4 b = blk;
        b.x = __funnelshift_r(b.x, b.y, 8);
        b.y = __funnelshift_r(b.y, b.z, 8);
        b.z = __funnelshift_r(b.z, b.w, 8);
        b.w >>= 8;
        for (int i = 0; i < n; ++i) {
            const int j = (int)(b.x & 255u);
            const float4 c = cand[j];
            const float a0 = c.x - o0, a1 = c.y - o1, a2 = c.z - o2;
            const float d = __fmaf_rn(a0, a0, __fmaf_rn(a1, a1, a2 * a2));
            const bool lt0 = d < b0;
            b1 = lt0 ? b0 : (d < b1 ? d : b1);
            i0 = lt0 ? j : i0;
            b0 = lt0 ? d : b0;
            b.x = __funnelshift_r(b.x, b.y, 8);
            b.y = __funnelshift_r(b.y, b.z, 8);
            b.z = __funnelshift_r(b.z, b.w, 8);
            b.w >>= 8;
        }
        if (b1 > b0 * 1.000002f) return i0;
    }
    // a 16-wide cell with more than 15 possible nearest entries (a crowded palette -- e.g. any palette under use_gamma, whose dark
    // entries crowd the low end of the linear scale): a point INSIDE the cube has the 8^3 lists and their octree (error diffusion's
    // path); only points beyond the cube in such a cell scan the whole palette
    if (n > 15 && inside && pal.ed_cells) return nearest_color_cells<CAP, false, false>(pal, cand, nullptr, o0, o1, o2, nullptr, nullptr, nullptr);
    return nearest_any<CAP>(pal, cand, o0, o1, o2);
}

struct VarParams {
    int model;
    int serpentine;
    float lum_factor, col_factor;
    const uint8_t *gate;   // model 3: n_frames*h*w bytes
    const float *coef;     // model 4: 256*3 float32
};

// taps in the reference's visiting order of the SOURCES (earlier row first, inside a row dx descending)
__constant__ int kFsDx[4] = {1, 0, -1, 1};
__constant__ int kFsDy[4] = {1, 1, 1, 0};
__constant__ float kFsW[4] = {1.0f / 16, 5.0f / 16, 3.0f / 16, 7.0f / 16};
// Ostromoukhov: (dx,dy) relative to the scan direction and the table column of each tap
__constant__ int kOsDx[3] = {0, -1, 1};
__constant__ int kOsDy[3] = {1, 1, 0};
__constant__ int kOsCol[3] = {2, 1, 0};

template <int CAP>
__global__ __launch_bounds__(64) void var_serial_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                        const int64_t n_frames, const int h, const int w,
                                                        const PalDev pal, const VarParams vp, float *__restrict__ ring)
{
    const int64_t f = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (f >= n_frames) return;
    const uint8_t *fin = in + (size_t)f * h * w * 3;
    uint8_t *fout = out + (size_t)f * h * w * 3;
    const size_t nf = (size_t)n_frames;
    const int model = vp.model;
    const int ntaps = model == 4 ? 3 : 4;
    for (int y = 0; y < h; ++y) {
        const bool rev = model == 4 && vp.serpentine && (y & 1);
        for (int step = 0; step < w; ++step) {
            const int x = rev ? (w - 1 - step) : step;
            const uint8_t *p = fin + ((size_t)y * w + x) * 3;
            uint32_t c0 = p[0], c1 = p[1], c2 = p[2];
            if (pal.lut_in) {
                c0 = pal.lut_in[c0];
                c1 = pal.lut_in[c1];
                c2 = pal.lut_in[c2];
            }
            const float g0 = (float)c0, g1 = (float)c1, g2 = (float)c2;  // the original (linearised) pixel
            float a0 = g0, a1 = g1, a2 = g2;
            for (int k = 0; k < ntaps; ++k) {
                const int dy = model == 4 ? kOsDy[k] : kFsDy[k];
                const int sr = y - dy;
                if (sr < 0) continue;
                const int sdir = (model == 4 && vp.serpentine && (sr & 1)) ? -1 : 1;
                const int dx = model == 4 ? kOsDx[k] : kFsDx[k];
                const int sxp = x - dx * sdir;
                if (sxp < 0 || sxp >= w) continue;
                const float *e = ring + (((size_t)(sr % 3) * w + sxp) * 4) * nf + f;
                const float aux = e[3 * nf];
                float wk;
                if (model == 1)
                    wk = __fmul_rn(kFsW[k], aux);
                else if (model == 3) {
                    if (aux == 0.0f) continue;
                    wk = kFsW[k];
                } else if (model == 4)
                    wk = vp.coef[3 * (int)aux + kOsCol[k]];
                else
                    wk = kFsW[k];
                a0 = __fadd_rn(a0, __fmul_rn(e[0], wk));
                a1 = __fadd_rn(a1, __fmul_rn(e[nf], wk));
                a2 = __fadd_rn(a2, __fmul_rn(e[2 * nf], wk));
            }
            float o0 = a0, o1 = a1, o2 = a2;
            if (model == 4) {
                o0 = clamp255f(o0);
                o1 = clamp255f(o1);
                o2 = clamp255f(o2);
            }
            const int j = nearest_any<CAP>(pal, pal.fcand, o0, o1, o2);
            float e0 = __fsub_rn(o0, pal.pts_f32[3 * j]), e1 = __fsub_rn(o1, pal.pts_f32[3 * j + 1]),
                  e2 = __fsub_rn(o2, pal.pts_f32[3 * j + 2]);
            float aux = 1.0f;
            if (model == 1) {
                const float lum = __fadd_rn(__fadd_rn(__fmul_rn(0.299f, g0), __fmul_rn(0.587f, g1)), __fmul_rn(0.114f, g2));
                aux = __fadd_rn(0.5f, __fmul_rn(0.5f, __fdiv_rn(lum, 255.0f)));
            } else if (model == 2) {
                const float lv = __fadd_rn(__fadd_rn(__fmul_rn(0.299f, e0), __fmul_rn(0.587f, e1)), __fmul_rn(0.114f, e2));
                const float l0 = __fmul_rn(0.299f, lv), l1 = __fmul_rn(0.587f, lv), l2 = __fmul_rn(0.114f, lv);
                e0 = __fadd_rn(__fmul_rn(vp.lum_factor, l0), __fmul_rn(vp.col_factor, __fsub_rn(e0, l0)));
                e1 = __fadd_rn(__fmul_rn(vp.lum_factor, l1), __fmul_rn(vp.col_factor, __fsub_rn(e1, l1)));
                e2 = __fadd_rn(__fmul_rn(vp.lum_factor, l2), __fmul_rn(vp.col_factor, __fsub_rn(e2, l2)));
            } else if (model == 3) {
                aux = vp.gate[((size_t)f * h + y) * w + x] ? 1.0f : 0.0f;
            } else if (model == 4) {
                float lum = __fadd_rn(__fadd_rn(__fmul_rn(0.299f, o0), __fmul_rn(0.587f, o1)), __fmul_rn(0.114f, o2));
                lum = clamp255f(lum);
                aux = (float)(int)lum;
            }
            float *e = ring + (((size_t)(y % 3) * w + x) * 4) * nf + f;
            e[0] = e0;
            e[nf] = e1;
            e[2 * nf] = e2;
            e[3 * nf] = aux;
            const uint32_t c = pal.out_rgb[j];
            uint8_t *o = fout + ((size_t)y * w + x) * 3;
            o[0] = (uint8_t)c;
            o[1] = (uint8_t)(c >> 8);
            o[2] = (uint8_t)(c >> 16);
        }
    }
}

// ---- Ostromoukhov with a serpentine scan: one wave per frame (the scheme of ed_rowserial_kernel, ediff.hip) --------
// Rows are strictly sequential, so only the latency of one pixel step counts.  The two newest error rows {e0,e1,e2,
// table row} live in LDS next to the coefficient table; each lane prepares, 64 pixels ahead of the walk, the input
// plus the two contributions from the row above for "its" pixel; the walk adds the same-row tap (the previous
// pixel's error, coefficient by its table row: one LDS broadcast read), evaluates the palette lane-parallel and
// reduces with DPP.  ~250 ns per pixel step instead of ~2 us for the lane = frame kernel.
template <int CAP, int M>  // M: palette entries per lane (K <= 64 * M)
__global__ __launch_bounds__(64) void os_rowserial_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                          const int h, const int w, const PalDev pal, const VarParams vp)
{
    extern __shared__ __align__(16) float s_dyn[];  // err[2][w][4], coef[768], then (M > 1) K x {x, y, z, out_rgb}
    __shared__ uint8_t s_lut[256];
    float4 *s_err = reinterpret_cast<float4 *>(s_dyn);
    float *s_coef = s_dyn + (size_t)8 * w;
    const float4 *s_pal = reinterpret_cast<const float4 *>(s_coef + 768);
    const int lane = threadIdx.x;
    const size_t f = blockIdx.x;
    const uint8_t *fin = in + f * (size_t)h * w * 3;
    uint8_t *fout = out + f * (size_t)h * w * 3;
    const float inf = __int_as_float(0x7f800000);
    const int K = pal.K;
    const bool serp = vp.serpentine != 0;

    float4 pc[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const int j = lane + 64 * m;
        pc[m] = j < K ? pal.fcand[j] : make_float4(1e30f, 1e30f, 1e30f, 0.f);  // distance overflows to +inf
    }
    if (M > 1)
        for (int j = lane; j < K; j += 64) const_cast<float4 *>(s_pal)[j] = pal.fcand[j];
    for (int i = lane; i < 768; i += 64) s_coef[i] = vp.coef[i];
    for (int i = lane; i < 256; i += 64) s_lut[i] = pal.lut_in ? pal.lut_in[i] : (uint8_t)i;
    __syncthreads();

    for (int y = 0; y < h; ++y) {
        const bool rev = serp && (y & 1);
        float e1x = 0.f, e1y = 0.f, e1z = 0.f;  // error of the previous pixel of this row
        int row1 = 0;                            // ... and its table row
        for (int step0 = 0; step0 < w; step0 += 64) {
            // input + the two taps from the row above (sources in the reference's visiting order), lane-parallel
            float p0 = 0.f, p1 = 0.f, p2 = 0.f;
            if (step0 + lane < w) {
                const int x = rev ? (w - 1 - (step0 + lane)) : (step0 + lane);
                const uint8_t *px = fin + ((size_t)y * w + x) * 3;
                p0 = (float)s_lut[px[0]];
                p1 = (float)s_lut[px[1]];
                p2 = (float)s_lut[px[2]];
                if (y > 0) {
                    const int sdir = (serp && ((y - 1) & 1)) ? -1 : 1;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int sxp = x - kOsDx[k] * sdir;
                        if (sxp < 0 || sxp >= w) continue;
                        const float4 e = s_err[(size_t)((y - 1) & 1) * w + sxp];
                        const float wk = s_coef[3 * (int)e.w + kOsCol[k]];
                        p0 = __fadd_rn(p0, __fmul_rn(e.x, wk));
                        p1 = __fadd_rn(p1, __fmul_rn(e.y, wk));
                        p2 = __fadd_rn(p2, __fmul_rn(e.z, wk));
                    }
                }
            }
            float my_e0 = 0.f, my_e1 = 0.f, my_e2 = 0.f, my_aux = 0.f;
            uint32_t my_c = 0;
            const int nstep = min(64, w - step0);
            for (int i = 0; i < nstep; ++i) {
                float a0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p0), i));
                float a1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p1), i));
                float a2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p2), i));
                // same-row tap: the previous pixel (none at the start of a row: its error is 0, which adds nothing)
                const float w1 = s_coef[3 * row1 + kOsCol[2]];
                a0 = __fadd_rn(a0, __fmul_rn(e1x, w1));
                a1 = __fadd_rn(a1, __fmul_rn(e1y, w1));
                a2 = __fadd_rn(a2, __fmul_rn(e1z, w1));
                const float o0 = clamp255f(a0), o1 = clamp255f(a1), o2 = clamp255f(a2);
                float b0 = inf, b1 = inf;
                int i0 = lane;
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const float da = pc[m].x - o0, db = pc[m].y - o1, dc = pc[m].z - o2;
                    const float d = __fmaf_rn(da, da, __fmaf_rn(db, db, dc * dc));
                    const bool lt0 = d < b0;
                    b1 = lt0 ? b0 : (d < b1 ? d : b1);
                    i0 = lt0 ? lane + 64 * m : i0;
                    b0 = lt0 ? d : b0;
                }
                const float B0 = wave_min_to_all(b0);
                const unsigned long long wm = __ballot(b0 == B0);
                const int winner = __ffsll((long long)wm) - 1;
                const float lim = B0 * 1.000002f;
                const unsigned long long close = __ballot(b0 <= lim);
                const bool near_tie = (close & (close - 1ull)) != 0ull || (M > 1 && __ballot(b1 <= lim) != 0ull);
                int j = M == 1 ? winner : __builtin_amdgcn_readlane(i0, winner);
                float cx, cy, cz;
                uint32_t cw;
                if (!near_tie && M == 1) {
                    cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pc[0].x), winner));
                    cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pc[0].y), winner));
                    cz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pc[0].z), winner));
                    cw = (uint32_t)__builtin_amdgcn_readlane(__float_as_int(pc[0].w), winner);
                } else {
                    if (near_tie) j = nearest_any_f64<CAP>(pal, o0, o1, o2);
                    const float4 c = (M > 1 && !near_tie) ? s_pal[j] : pal.fcand[j];
                    cx = c.x;
                    cy = c.y;
                    cz = c.z;
                    cw = __float_as_uint(c.w);
                }
                const float ex = __fsub_rn(o0, cx), ey = __fsub_rn(o1, cy), ez = __fsub_rn(o2, cz);
                float lum = __fadd_rn(__fadd_rn(__fmul_rn(0.299f, o0), __fmul_rn(0.587f, o1)), __fmul_rn(0.114f, o2));
                const int row = (int)clamp255f(lum);
                if (lane == i) {
                    my_e0 = ex;
                    my_e1 = ey;
                    my_e2 = ez;
                    my_aux = (float)row;
                    my_c = cw;
                }
                e1x = ex;
                e1y = ey;
                e1z = ez;
                row1 = row;
            }
            if (lane < nstep) {
                const int x = rev ? (w - 1 - (step0 + lane)) : (step0 + lane);
                s_err[(size_t)(y & 1) * w + x] = make_float4(my_e0, my_e1, my_e2, my_aux);
                uint8_t *o = fout + ((size_t)y * w + x) * 3;
                o[0] = (uint8_t)my_c;
                o[1] = (uint8_t)(my_c >> 8);
                o[2] = (uint8_t)(my_c >> 16);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---- anti-diagonal wavefront version (serpentine off): same schedule as ed_wavefront_kernel in ediff.hip ------------
// (one workgroup per frame, lane = row of a 64-row band, bands pipelined across the waves through a progress word,
// global traffic in 16-step bursts), with four floats per stored error (the fourth is `aux`) and at most 12 waves so
// that the wider rings fit LDS.  All four models have skew 2 (their only upward-left tap is dx=-1, dy=1).
constexpr int kVWaves = 12;
constexpr int kVRing = 8;
constexpr int kVRingStride = kVRing * 4 + 4;
constexpr int kVPeriod = 16;
constexpr int kVProgWords = 64;  // progress words per frame when a frame's bands are spread over workgroups (ediff.hip)

template <int CAP, int MODEL>  // MODEL: 1 perceptual, 2 hybrid, 3 adaptive variance, 4 Ostromoukhov (vp.model)
__global__ __launch_bounds__(64 * kVWaves) void var_wavefront_kernel(const uint8_t *__restrict__ in,
                                                                     uint8_t *__restrict__ out, const int h, const int w,
                                                                     const PalDev pal, const VarParams vp,
                                                                     float *__restrict__ bnd_all, const int G,
                                                                     uint32_t *__restrict__ gprog_all, const int test_giveup,
                                                                     const uint32_t n_frames)
{
    // G == 1 with progress words given: the repair launch behind a G > 1 launch (see ed_wavefront_kernel) -- only frames
    // whose give-up flag is set are done again
    if (G == 1 && gprog_all != nullptr &&
        __hip_atomic_load(&gprog_all[(size_t)blockIdx.x * kVProgWords + kVProgWords - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
        return;
    if (G != 1 && test_giveup) {
        if (threadIdx.x == 0)
            __hip_atomic_store(&gprog_all[(size_t)(blockIdx.x / (unsigned)G) * kVProgWords + kVProgWords - 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // G > 1: the bands of a frame are spread over G workgroups (few frames in flight), see ed_wavefront_kernel
    // a row's ring padded from 32 to 36 words: with 32, sixteen lanes meet in one LDS bank on every ring access (the
    // slot a lane touches depends on the lane: (t - 2L - dx) & 7); 36 keeps the 16-byte alignment and is conflict-free
    __shared__ __align__(16) float s_ring[kVWaves][64][kVRingStride];
    __shared__ float s_vring[kVWaves][2][64][4];
    __shared__ float s_bout[kVWaves][2][kVPeriod][4];
    __shared__ uint8_t s_lut[256];
    __shared__ volatile uint32_t s_prog[kVWaves];
    __shared__ __align__(16) float s_zero4[4];  // the "error" of rows that do not exist (weight read as 1.0, error 0)
    // {x, y, z, out_rgb bits}; palettes of 9..16 colours keep the candidate lists of the 16^3 cells behind their 16 entries (ediff.hip)
    __shared__ float4 s_pal[DP_MAX_COLORS + 16];
    uint32_t *s_coarse = reinterpret_cast<uint32_t *>(s_pal + 16);
    const int L = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int NW = blockDim.x >> 6;
    // PERSISTENT (plain one-workgroup launches): the workgroup does every gridDim.x-th frame, its waves taking the bands of all of
    // them round-robin by a running band number (ediff.hip, ed_wavefront_kernel)
    const bool persist = G == 1 && gprog_all == nullptr;
    const size_t f0 = blockIdx.x / (unsigned)G;
    const int NWT = NW * G;
    const int gw = (int)(blockIdx.x % (unsigned)G) * NW + wv;
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_lut[i] = pal.lut_in ? pal.lut_in[i] : (uint8_t)i;
    for (int i = threadIdx.x; i < pal.K; i += blockDim.x) s_pal[i] = pal.fcand[i];
    // (model 4 clamps its values: the plain table; the others look their unclamped values up in the extended one)
    const uint32_t *coarse_src = MODEL == 4 ? pal.ed_coarse : pal.ed_coarse_ext;
    if (coarse_src)
        for (int i = threadIdx.x; i < 4096; i += blockDim.x) s_coarse[i] = coarse_src[i];
    const uint32_t *coarse = coarse_src ? s_coarse : nullptr;
    if (threadIdx.x < kVWaves) s_prog[threadIdx.x] = 0;
    if (threadIdx.x < 4) s_zero4[threadIdx.x] = threadIdx.x == 3 ? 1.0f : 0.0f;
    const long frame_bytes = (long)h * w * 3;
    const long gate_bytes = (long)h * w;
    constexpr int model = MODEL;
    constexpr int ntaps = model == 4 ? 3 : 4;
    constexpr int skew = 2;
    const int n_bands = (h + 63) / 64;
    const int n_mine = persist ? (int)((n_frames - (uint32_t)f0 + gridDim.x - 1u) / gridDim.x) : 1;  // frames of this workgroup
    const int n_gbands = n_mine * n_bands;  // (< 65536: the launcher sizes the grid for it)
    __syncthreads();

    for (int gb = gw; gb < n_gbands; gb += NWT) {  // gb: running band number over this workgroup's frames
        const int fi = gb / n_bands, band = gb - fi * n_bands;
        const size_t f = f0 + (size_t)fi * gridDim.x;
        const uint8_t *fin = in + f * (size_t)h * w * 3;
        uint8_t *fout = out + f * (size_t)h * w * 3;
        const uint8_t *fgate = vp.gate ? vp.gate + f * (size_t)h * w : nullptr;
        float *bnd = bnd_all + f * (size_t)4 * w * 4;  // [2 buffers][2 rows][w][4]
        uint32_t *gprog = gprog_all + f * (size_t)kVProgWords;
        const int r = band * 64 + L;
        const float *bprev = bnd + (size_t)((band + 1) & 1) * 2 * w * 4;
        float *bnext = bnd + (size_t)(band & 1) * 2 * w * 4;
        const int rows_here = min(64, h - band * 64);
        const int steps = w + skew * (rows_here - 1);
        const int pw = (gw + NWT - 1) % NWT;
        const bool row_ok = r < h;
        const long row_byte = (long)r * w * 3;
        const long row_gate = (long)r * w;

        // the taps as immediates (Floyd-Steinberg's four, or Ostromoukhov's three), in the reference's visiting order
        constexpr int kDx[4] = {MODEL == 4 ? 0 : 1, MODEL == 4 ? -1 : 0, MODEL == 4 ? 1 : -1, 1};
        constexpr int kDy[4] = {1, 1, MODEL == 4 ? 0 : 1, 0};
        constexpr float kW[4] = {1.0f / 16, 5.0f / 16, 3.0f / 16, 7.0f / 16};
        constexpr int kCol[3] = {2, 1, 0};
        // where each tap of this lane's row finds its source row: the ring of a row of the band, the ring of the two rows
        // above the band, or -- the row does not exist -- the zero slot; the rings hold zero errors for the two columns
        // either side of the image (below), so a tap needs neither a row nor a column test
        typedef const __attribute__((address_space(3))) float lds_float_t;
        lds_float_t *tb[4];
        uint32_t tm[4];
#pragma unroll
        for (int k = 0; k < ntaps; ++k) {
            const int rel = L - kDy[k];
            const bool exists = r - kDy[k] >= 0;
            const float *row = rel >= 0 ? &s_ring[wv][rel][0] : &s_vring[wv][rel + 2][0][0];
            tb[k] = exists ? (lds_float_t *)row : (lds_float_t *)s_zero4;
            tm[k] = exists ? (rel >= 0 ? (uint32_t)(kVRing - 1) : 63u) : 0u;
        }
        uint32_t pix[12], cur[13], outb[13], gpx[4], gcur[5];
        float pb0 = 0.f, pb1 = 0.f, pb2 = 0.f, pb3 = 1.0f;
        int pb_col = 0;
        bool pb_valid = false;
#pragma unroll
        for (int k = 0; k < 12; ++k) pix[k] = 0;
#pragma unroll
        for (int k = 0; k < 13; ++k) cur[k] = outb[k] = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) gpx[k] = gcur[k] = 0;
        gcur[4] = 0;

        for (int t0 = -kVPeriod; t0 < steps + kVPeriod; t0 += kVPeriod) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (L == 0) {
                int ack = t0 - kVPeriod - 63 * skew + 1024;
                ack = ack < 0 ? 0 : ack;
                const uint32_t word = ((uint32_t)gb << 16) | (uint32_t)ack;
                if (G == 1) s_prog[wv] = word;
                else __hip_atomic_store(&gprog[gw], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (pb_valid) {
                float *dst = &s_vring[wv][L >> 5][pb_col & 63][0];
                dst[0] = pb0;
                dst[1] = pb1;
                dst[2] = pb2;
                dst[3] = pb3;
            }
            {
                const int xs = (t0 - kVPeriod) - skew * L;
                const int lo = xs < 0 ? 0 : xs, hi = xs + kVPeriod > w ? w : xs + kVPeriod;
                if (row_ok && hi > lo) {
                    const long B = row_byte + (long)xs * 3;
                    if (lo == xs && hi == xs + kVPeriod) {
#pragma unroll
                        for (int k = 0; k < 12; ++k) *reinterpret_cast<uint32_t *>(fout + B + 4 * k) = outb[k];
                    } else {
#pragma unroll
                        for (int i = 0; i < kVPeriod; ++i) {
                            if (xs + i >= 0 && xs + i < w) {
                                const int bo = 3 * i;
                                const uint32_t c = __funnelshift_r(outb[bo >> 2], outb[(bo >> 2) + 1], (bo & 3) * 8);
                                uint8_t *o = fout + B + bo;
                                o[0] = (uint8_t)c;
                                o[1] = (uint8_t)(c >> 8);
                                o[2] = (uint8_t)(c >> 16);
                            }
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 13; ++k) outb[k] = 0;
                if (L >= 62 && row_ok && hi > lo) {
                    for (int i = lo - xs; i < hi - xs; ++i) {
                        float *b = bnext + ((size_t)(L - 62) * w + (xs + i)) * 4;
                        if (G == 1) {
                            b[0] = s_bout[wv][L - 62][i][0];
                            b[1] = s_bout[wv][L - 62][i][1];
                            b[2] = s_bout[wv][L - 62][i][2];
                            b[3] = s_bout[wv][L - 62][i][3];
                        } else {
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                __hip_atomic_store(b + c, s_bout[wv][L - 62][i][c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 12; ++k) cur[k] = pix[k];
#pragma unroll
            for (int k = 0; k < 4; ++k) gcur[k] = gpx[k];
            {
                const int xn = (t0 + kVPeriod) - skew * L;
                const long B = row_byte + (long)xn * 3;
                if (row_ok && xn + kVPeriod > 0 && xn < w) {
                    if (B >= 0 && B + 48 <= frame_bytes) {
#pragma unroll
                        for (int k = 0; k < 12; ++k) pix[k] = *reinterpret_cast<const uint32_t *>(fin + B + 4 * k);
                    } else {
#pragma unroll
                        for (int k = 0; k < 12; ++k) {
                            uint32_t v = 0;
#pragma unroll
                            for (int bb = 0; bb < 4; ++bb) {
                                const long a = B + 4 * k + bb;
                                if (a >= 0 && a < frame_bytes) v |= (uint32_t)fin[a] << (8 * bb);
                            }
                            pix[k] = v;
                        }
                    }
                    if (model == 3) {  // the next 16 gate bytes of this row
                        const long G = row_gate + xn;
                        if (G >= 0 && G + 16 <= gate_bytes) {  // four (unaligned) dword loads instead of sixteen byte loads
#pragma unroll
                            for (int k = 0; k < 4; ++k) gpx[k] = *reinterpret_cast<const uint32_t *>(fgate + G + 4 * k);
                        } else
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            uint32_t v = 0;
#pragma unroll
                            for (int bb = 0; bb < 4; ++bb) {
                                const long a = G + 4 * k + bb;
                                if (a >= 0 && a < gate_bytes) v |= (uint32_t)fgate[a] << (8 * bb);
                            }
                            gpx[k] = v;
                        }
                    }
                }
            }
            pb_valid = false;
            if (band > 0) {
                const int x0n = t0 + kVPeriod;
                int need = x0n + 18;
                need = need > w ? w : need;
                if (x0n - 14 < w && need > 0) {
                    const uint32_t want = (uint32_t)(need + 1024);
                    for (uint32_t spins = 0;; ++spins) {
                        const uint32_t v = G == 1 ? s_prog[pw] : __hip_atomic_load(&gprog[pw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((int)(v >> 16) > gb - 1 || ((int)(v >> 16) == gb - 1 && (v & 0xffffu) >= want)) break;
                        __builtin_amdgcn_s_sleep(4);
                        // another workgroup produces this: never wait forever (a give-up flag stays in the workspace)
                        if (G != 1 && (spins > (1u << 24) || (spins % 1024u == 1023u &&
                                                               __hip_atomic_load(&gprog[kVProgWords - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u))) {
                            if (L == 0) __hip_atomic_store(&gprog[kVProgWords - 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            return;
                        }
                    }
                    const int col = x0n - 14 + (L & 31);
                    if (col >= 0 && col < w) {
                        const float *b = bprev + ((size_t)(L >> 5) * w + col) * 4;
                        pb0 = __hip_atomic_load(b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        pb1 = __hip_atomic_load(b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        pb2 = __hip_atomic_load(b + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        pb3 = __hip_atomic_load(b + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        pb_col = col;
                        pb_valid = true;
                    } else if (col >= -2 && col < w + 2) {  // the two columns either side of the image: zero errors
                        pb0 = pb1 = pb2 = 0.f;
                        pb3 = 1.0f;
                        pb_col = col;
                        pb_valid = true;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();

            // four steps per round (see ed_wavefront_kernel): the period buffers move by whole registers
            for (int i4 = 0; i4 < kVPeriod; i4 += 4) {
              uint32_t cb[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int i = i4 + q;
                const int t = t0 + i;
                const int x = t - skew * L;
                const bool act = row_ok && (uint32_t)x < (uint32_t)w;  // (0 <= t < steps follows for the rows of the band)
                float e0 = 0.f, e1 = 0.f, e2 = 0.f, aux = 1.0f;
                uint32_t cbytes = 0;
                if (act) {
                    const uint32_t pxv = q == 0 ? cur[0] : (q == 1 ? __funnelshift_r(cur[0], cur[1], 24)
                                                               : (q == 2 ? __funnelshift_r(cur[1], cur[2], 16) : (cur[2] >> 8)));
                    const float g0 = (float)s_lut[pxv & 255u], g1 = (float)s_lut[(pxv >> 8) & 255u],
                                g2 = (float)s_lut[(pxv >> 16) & 255u];
                    float a0 = g0, a1 = g1, a2 = g2;
                    // the reads of every tap first (and, Ostromoukhov, the coefficient reads that depend on them), then the sums in
                    // the reference's order: written as one loop the compiler waits for each tap's read before it issues the next
                    // tap's -- one LDS round trip after the other on the step's critical path (ediff.hip, round 4)
                    float s0[4], s1[4], s2[4], s3[4], wt[4];
#pragma unroll
                    for (int k = 0; k < ntaps; ++k) {
                        lds_float_t *src = tb[k] + ((x - kDx[k]) & (int)tm[k]) * 4;
                        s0[k] = src[0];
                        s1[k] = src[1];
                        s2[k] = src[2];
                        s3[k] = src[3];
                    }
#pragma unroll
                    for (int k = 0; k < ntaps; ++k) {
                        const float sa = s3[k];
                        if (model == 1)
                            wt[k] = __fmul_rn(kW[k], sa);
                        else if (model == 3)
                            wt[k] = sa == 0.0f ? 0.0f : kW[k];  // a closed gate: the source adds +-0, which changes nothing (a is never -0)
                        else if (model == 4)
                            wt[k] = vp.coef[3 * (int)sa + kCol[k]];
                        else
                            wt[k] = kW[k];
                    }
#pragma unroll
                    for (int k = 0; k < ntaps; ++k) {
                        a0 = __fadd_rn(a0, __fmul_rn(s0[k], wt[k]));
                        a1 = __fadd_rn(a1, __fmul_rn(s1[k], wt[k]));
                        a2 = __fadd_rn(a2, __fmul_rn(s2[k], wt[k]));
                    }
                    float o0 = a0, o1 = a1, o2 = a2;
                    if (model == 4) {
                        o0 = clamp255f(o0);
                        o1 = clamp255f(o1);
                        o2 = clamp255f(o2);
                    }
                    // points inside the colour cube (always, once clamped: Ostromoukhov) search only the candidate list of
                    // their cell, as error diffusion does; a wave with a point outside scans the palette
                    const bool inside = model == 4 || (o0 >= 0.0f && o0 <= 255.0f && o1 >= 0.0f && o1 <= 255.0f && o2 >= 0.0f && o2 <= 255.0f);
                    // (perceptual: its waves nearly always hold a point outside the cube -- measured, the lists only cost there)
                    int j;
                    if (model != 4 && coarse) j = nearest_ext<CAP>(pal, s_pal, coarse, o0, o1, o2);
                    else if (model != 4 && pal.ed_ext16) j = nearest_ext16<CAP>(pal, s_pal, pal.ed_ext16, o0, o1, o2, inside);
                    else if (model != 1 && pal.ed_cells && __ballot(!inside) == 0ull) j = nearest_color_cells<CAP, false, CAP == kQueueLarge>(pal, s_pal, coarse, o0, o1, o2, nullptr, nullptr, pal.ed_h4);
                    else j = nearest_any<CAP>(pal, s_pal, o0, o1, o2);
                    const float4 pj = s_pal[j];
                    e0 = __fsub_rn(o0, pj.x);
                    e1 = __fsub_rn(o1, pj.y);
                    e2 = __fsub_rn(o2, pj.z);
                    if (model == 1) {
                        const float lum = __fadd_rn(__fadd_rn(__fmul_rn(0.299f, g0), __fmul_rn(0.587f, g1)), __fmul_rn(0.114f, g2));
                        aux = __fadd_rn(0.5f, __fmul_rn(0.5f, __fdiv_rn(lum, 255.0f)));
                    } else if (model == 2) {
                        const float lv = __fadd_rn(__fadd_rn(__fmul_rn(0.299f, e0), __fmul_rn(0.587f, e1)), __fmul_rn(0.114f, e2));
                        const float l0 = __fmul_rn(0.299f, lv), l1 = __fmul_rn(0.587f, lv), l2 = __fmul_rn(0.114f, lv);
                        e0 = __fadd_rn(__fmul_rn(vp.lum_factor, l0), __fmul_rn(vp.col_factor, __fsub_rn(e0, l0)));
                        e1 = __fadd_rn(__fmul_rn(vp.lum_factor, l1), __fmul_rn(vp.col_factor, __fsub_rn(e1, l1)));
                        e2 = __fadd_rn(__fmul_rn(vp.lum_factor, l2), __fmul_rn(vp.col_factor, __fsub_rn(e2, l2)));
                    } else if (model == 3) {
                        aux = ((gcur[0] >> (8 * q)) & 255u) ? 1.0f : 0.0f;
                    } else if (model == 4) {
                        float lum = __fadd_rn(__fadd_rn(__fmul_rn(0.299f, o0), __fmul_rn(0.587f, o1)), __fmul_rn(0.114f, o2));
                        aux = (float)(int)clamp255f(lum);
                    }
                    cbytes = __float_as_uint(pj.w);
                }
                cb[q] = cbytes;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // columns -2, -1, w and w+1 are written too, with zero errors (e is 0, aux 1 without a pixel)
                if (row_ok && x >= -2 && x < w + 2) {
                    float *dst = &s_ring[wv][L][(x & (kVRing - 1)) * 4];
                    dst[0] = e0;
                    dst[1] = e1;
                    dst[2] = e2;
                    dst[3] = aux;
                    if (act && L >= 62) {
                        float *bo = &s_bout[wv][L - 62][i][0];
                        bo[0] = e0;
                        bo[1] = e1;
                        bo[2] = e2;
                        bo[3] = aux;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
              }
#pragma unroll
              for (int k = 0; k < 9; ++k) {
                  cur[k] = cur[k + 3];
                  outb[k] = outb[k + 3];
              }
              cur[9] = cur[10] = cur[11] = 0u;
              outb[9] = cb[0] | (cb[1] << 24);
              outb[10] = (cb[1] >> 8) | (cb[2] << 16);
              outb[11] = (cb[2] >> 16) | (cb[3] << 8);
#pragma unroll
              for (int k = 0; k < 4; ++k) gcur[k] = gcur[k + 1];  // four gate bytes consumed
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (L == 0) {
            if (G == 1) s_prog[wv] = ((uint32_t)(gb + 1) << 16);
            else __hip_atomic_store(&gprog[gw], (uint32_t)(gb + 1) << 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- variance gate -------------------------------------------------------------------------------------------
__device__ __forceinline__ float gray_of(const uint8_t *p, const uint8_t *lut)
{
    uint32_t c0 = p[0], c1 = p[1], c2 = p[2];
    if (lut) {
        c0 = lut[c0];
        c1 = lut[c1];
        c2 = lut[c2];
    }
    return __fadd_rn(__fadd_rn(__fmul_rn(0.299f, (float)c0), __fmul_rn(0.587f, (float)c1)), __fmul_rn(0.114f, (float)c2));
}

// scipy accumulates each line as a double running sum (tmp += entering - leaving).  Every term is a float32 in
// [0, 65025] whose bits span less than 53 binary places together with the partial sums (gray = f32 sums of
// 0.299*r etc.: >= 2^-27 granularity, <= 2^8; gray^2: >= 2^-30, <= 2^16; windows of <= 129 terms), so every partial
// sum is exact and the running sum equals the plain window sum in any order: one thread per pixel reproduces it.
// (float)(t / size) as scipy computes it -- the double quotient correctly rounded, then rounded to float32 -- without the
// ~40 instructions of a float64 division: q = t * (1/size) is within 2 ulp of the rounded quotient, and both round to the
// same float32 unless a float32 rounding boundary (low 29 mantissa bits = 0x10000000) lies that close; then divide.
__device__ __forceinline__ float mean_f32(const double t, const double size, const double rcp)
{
    double q = __dmul_rn(t, rcp);
    const uint32_t lo = (uint32_t)__double2loint(q) & 0x1fffffffu;
    if (lo - 0x0ffffffcu <= 8u) q = __ddiv_rn(t, size);
    return (float)q;
}

// Both axes in one kernel for windows up to 9 x 9: a workgroup takes a tile of 64 x 16 pixels, stages gray and gray^2 of
// the tile and its halo in LDS (edge pixels repeated: mode='nearest'), runs axis 0 into a second LDS plane (the
// float32 values scipy stores between the axes) and axis 1 from there.  3 B read and 1 B written per pixel instead of
// 3 + 8 written + 8 read + 1, and four multiplications instead of four float64 divisions.
constexpr int kGateTW = 64, kGateTH = 16, kGateMaxR = 4;

template <int RADIUS>  // compiled per window radius: the tile widths are constants (index arithmetic without divisions)
__global__ __launch_bounds__(256) void var_gate_fused_kernel(const uint8_t *__restrict__ in, const uint8_t *__restrict__ lut,
                                                              uint8_t *__restrict__ gate, const int h, const int w,
                                                              const float thr)
{
    constexpr int radius = RADIUS;
    constexpr int kMaxW = kGateTW + 2 * kGateMaxR, kMaxH = kGateTH + 2 * kGateMaxR;
    __shared__ float s_g[kMaxH][kMaxW], s_q[kMaxH][kMaxW];        // gray, gray^2 of the tile + halo
    __shared__ float s_tg[kGateTH][kMaxW], s_tq[kGateTH][kMaxW];  // after axis 0
    const int64_t f = blockIdx.z;
    const uint8_t *fin = in + (size_t)f * h * w * 3;
    uint8_t *fgate = gate + (size_t)f * h * w;
    const int x0 = blockIdx.x * kGateTW, y0 = blockIdx.y * kGateTH;
    constexpr int size = 2 * radius + 1, tw = kGateTW + 2 * radius, th = kGateTH + 2 * radius;
    const double dsize = (double)size, rcp = 1.0 / dsize;
    for (int i = threadIdx.x; i < tw * th; i += 256) {
        const int ty = i / tw, tx = i - ty * tw;
        int y = y0 + ty - radius, x = x0 + tx - radius;
        y = y < 0 ? 0 : (y >= h ? h - 1 : y);
        x = x < 0 ? 0 : (x >= w ? w - 1 : x);
        // one (unaligned) dword per pixel instead of three byte loads; the frame's very last pixel has no fourth byte
        const size_t at = ((size_t)y * w + x) * 3;
        float g;
        if (at + 4 <= (size_t)h * w * 3) {
            const uint32_t v = *reinterpret_cast<const uint32_t *>(fin + at);
            uint32_t c0 = v & 255u, c1 = (v >> 8) & 255u, c2 = (v >> 16) & 255u;
            if (lut) {
                c0 = lut[c0];
                c1 = lut[c1];
                c2 = lut[c2];
            }
            g = __fadd_rn(__fadd_rn(__fmul_rn(0.299f, (float)c0), __fmul_rn(0.587f, (float)c1)), __fmul_rn(0.114f, (float)c2));
        } else {
            g = gray_of(fin + at, lut);
        }
        s_g[ty][tx] = g;
        s_q[ty][tx] = __fmul_rn(g, g);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < tw * kGateTH; i += 256) {
        const int ty = i / tw, tx = i - ty * tw;
        double tg = 0.0, ts = 0.0;
        for (int k = 0; k < size; ++k) {
            tg = __dadd_rn(tg, (double)s_g[ty + k][tx]);
            ts = __dadd_rn(ts, (double)s_q[ty + k][tx]);
        }
        s_tg[ty][tx] = mean_f32(tg, dsize, rcp);
        s_tq[ty][tx] = mean_f32(ts, dsize, rcp);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kGateTW * kGateTH; i += 256) {
        const int ty = i / kGateTW, tx = i - ty * kGateTW;
        const int y = y0 + ty, x = x0 + tx;
        if (y >= h || x >= w) continue;
        double tg = 0.0, ts = 0.0;
        for (int k = 0; k < size; ++k) {
            tg = __dadd_rn(tg, (double)s_tg[ty][tx + k]);
            ts = __dadd_rn(ts, (double)s_tq[ty][tx + k]);
        }
        const float mean_sq = mean_f32(ts, dsize, rcp), mean = mean_f32(tg, dsize, rcp);
        float var = __fsub_rn(mean_sq, __fmul_rn(mean, mean));
        var = var > 0.0f ? var : 0.0f;
        fgate[(size_t)y * w + x] = var >= thr ? 1 : 0;
    }
}

// pass 1: along axis 0 on gray^2 (t_sq) and gray (t_g), float32 results as scipy stores them between the axes
__global__ void var_axis0_kernel(const uint8_t *__restrict__ in, const uint8_t *__restrict__ lut, float *__restrict__ t_sq,
                                 float *__restrict__ t_g, const int64_t n_frames, const int h, const int w, const int size)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int64_t f = blockIdx.z;
    if (x >= w) return;
    const uint8_t *fin = in + (size_t)f * h * w * 3;
    const int s1 = size / 2;
    double tg = 0.0, ts = 0.0;
    for (int i = 0; i < size; ++i) {
        int k = y + i - s1;
        k = k < 0 ? 0 : (k >= h ? h - 1 : k);
        const float g = gray_of(fin + ((size_t)k * w + x) * 3, lut);
        tg = __dadd_rn(tg, (double)g);
        ts = __dadd_rn(ts, (double)__fmul_rn(g, g));
    }
    const size_t o = ((size_t)f * h + y) * w + x;
    t_g[o] = (float)__ddiv_rn(tg, (double)size);
    t_sq[o] = (float)__ddiv_rn(ts, (double)size);
}

// pass 2: along axis 1 on the float32 results of pass 1, then the gate
__global__ void var_axis1_kernel(const float *__restrict__ t_sq, const float *__restrict__ t_g, uint8_t *__restrict__ gate,
                                 const int64_t n_frames, const int h, const int w, const int size, const float thr)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int64_t f = blockIdx.z;
    if (x >= w) return;
    const size_t row = ((size_t)f * h + y) * w;
    const int s1 = size / 2;
    double tg = 0.0, ts = 0.0;
    for (int i = 0; i < size; ++i) {
        int k = x + i - s1;
        k = k < 0 ? 0 : (k >= w ? w - 1 : k);
        tg = __dadd_rn(tg, (double)t_g[row + k]);
        ts = __dadd_rn(ts, (double)t_sq[row + k]);
    }
    const float mean_sq = (float)__ddiv_rn(ts, (double)size), mean = (float)__ddiv_rn(tg, (double)size);
    float var = __fsub_rn(mean_sq, __fmul_rn(mean, mean));
    var = var > 0.0f ? var : 0.0f;
    gate[row + x] = var >= thr ? 1 : 0;
}

}  // namespace

// The two float planes between the passes are kept for a chunk of frames only (about a gigabyte), whatever the batch
static int64_t variance_gate_chunk(int64_t n_frames, int h, int w)
{
    const int64_t per_frame = (int64_t)h * w * (int64_t)sizeof(float) * 2;
    int64_t budget = (int64_t)1 << 30;
    if (const char *e = exp_env("DP_GATE_CHUNK_BYTES")) budget = std::max<int64_t>(1, atoll(e));  // tests: force several chunks
    int64_t c = per_frame > 0 ? budget / per_frame : n_frames;
    c = c < 1 ? 1 : c;
    return c < n_frames ? c : (n_frames < 1 ? 1 : n_frames);
}

size_t variance_gate_ws_bytes(int64_t n_frames, int h, int w)
{
    return (size_t)variance_gate_chunk(n_frames, h, w) * h * w * sizeof(float) * 2 + 256;
}

int launch_variance_gate(const uint8_t *in, uint8_t *gate, int64_t n_frames, int h, int w, const PalDev &pal, float thr,
                         int radius, void *ws, hipStream_t s)
{
    const int size = 2 * radius + 1;
    if (h > 65535) {
        set_error("dp_variance_gate_u8: h > 65535 not supported");
        return DP_EUNSUPPORTED;
    }
    if (radius <= kGateMaxR && !exp_env("DP_GATE_TWO_PASS")) {
        for (int64_t f0 = 0; f0 < n_frames; f0 += 65535) {
            const int64_t nf = std::min<int64_t>(65535, n_frames - f0);
            const dim3 grid((w + kGateTW - 1) / kGateTW, (h + kGateTH - 1) / kGateTH, (unsigned)nf);
#define DP_GATE(R)                                                                                                       \
    hipLaunchKernelGGL(var_gate_fused_kernel<R>, grid, dim3(256), 0, s, in + (size_t)f0 * h * w * 3, pal.lut_in,           \
                       gate + (size_t)f0 * h * w, h, w, thr)
            switch (radius) {
            case 0: DP_GATE(0); break;
            case 1: DP_GATE(1); break;
            case 2: DP_GATE(2); break;
            case 3: DP_GATE(3); break;
            default: DP_GATE(4); break;
            }
#undef DP_GATE
        }
        DP_HIP(hipGetLastError());
        return DP_OK;
    }
    const int64_t chunk = std::min<int64_t>(variance_gate_chunk(n_frames, h, w), 65535);
    float *t_sq = reinterpret_cast<float *>(ws);
    float *t_g = t_sq + (size_t)chunk * h * w;
    for (int64_t f0 = 0; f0 < n_frames; f0 += chunk) {  // chunks run back to back on the stream and reuse the planes
        const int64_t nf = std::min<int64_t>(chunk, n_frames - f0);
        const uint8_t *in_c = in + (size_t)f0 * h * w * 3;
        uint8_t *gate_c = gate + (size_t)f0 * h * w;
        const dim3 grid((w + 255) / 256, h, (unsigned)nf);
        hipLaunchKernelGGL(var_axis0_kernel, grid, dim3(256), 0, s, in_c, pal.lut_in, t_sq, t_g, nf, h, w, size);
        hipLaunchKernelGGL(var_axis1_kernel, grid, dim3(256), 0, s, t_sq, t_g, gate_c, nf, h, w, size, thr);
    }
    DP_HIP(hipGetLastError());
    return DP_OK;
}

int launch_variable_diffusion(const uint8_t *in, uint8_t *out, int64_t n_frames, int h, int w, const PalDev &pal, int model,
                              float p0, float p1, int serpentine, const uint8_t *gate, const float *coef, void *ws,
                              hipStream_t s)
{
    VarParams vp;
    vp.model = model;
    vp.serpentine = serpentine;
    vp.lum_factor = p0;
    vp.col_factor = p1;
    vp.gate = gate;
    vp.coef = coef;
    ProfMark *pm = prof_begin(s);
    if (!(model == 4 && serpentine) && w < 60000 && n_frames <= 0x7fffffff) {
        const int n_bands = (h + 63) / 64;
        int nw = n_bands < kVWaves ? n_bands : kVWaves;
        // few frames in flight: a frame's bands over G workgroups (see launch_error_diffusion); the progress words sit
        // behind the boundary rows in the workspace (error_diffusion_ws_bytes reserves them)
        int G = 1, cus = 0, dev_id = 0;
        if (hipGetDevice(&dev_id) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev_id) != hipSuccess) cus = 0;
        uint32_t *gprog = nullptr;
        if (n_frames * 2 <= cus && n_bands >= 4 && w >= 64 && !exp_env("DP_ED_ONE_WG")) {
            const int nwt = n_bands < 2 * kVWaves ? n_bands : 2 * kVWaves;
            while (G * 2 <= 16 && n_frames * (G * 2) <= cus && G * 2 <= nwt) G *= 2;
            if (G > 1) {
                nw = (nwt + G - 1) / G;
                const size_t prog_off = ((size_t)n_frames * (size_t)w * 64 + 255) & ~(size_t)255;
                gprog = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(ws) + prog_off);
                DP_HIP(hipMemsetAsync(gprog, 0, (size_t)n_frames * kVProgWords * sizeof(uint32_t), s));
            }
        }
        const int test_giveup = exp_env("DP_ED_TEST_GIVEUP") ? 1 : 0;
        const int nw1 = n_bands < kVWaves ? n_bands : kVWaves;
        // (behind a G > 1 launch: the repair launch for frames whose workgroups gave up waiting for each other)
        // batches larger than the device: one persistent workgroup per CU (launch_error_diffusion, ediff.hip)
        int64_t pgrid = n_frames * G;
        if (G == 1 && nw > 4 && cus > 0 && n_frames > cus && n_bands <= 4096 && !exp_env("DP_ED_NO_PERSIST")) {
            pgrid = cus;
            const int64_t need = (n_frames * n_bands + 59999) / 60000;
            if (pgrid < need) pgrid = need;
        }
        if (const char *e = exp_env("DP_ED_GRID")) {
            const int64_t v = atoll(e);
            if (G == 1 && v >= 1 && v <= n_frames && (n_frames + v - 1) / v * n_bands < 60000) pgrid = v;
        }
        const uint32_t nfr = (uint32_t)n_frames;
#define DP_VARW(C, M)                                                                                                     \
    do {                                                                                                                 \
        hipLaunchKernelGGL((var_wavefront_kernel<C, M>), dim3((unsigned)pgrid), dim3(64 * nw), 0, s, in, out, h, w, pal, vp, \
                           reinterpret_cast<float *>(ws), G, gprog, test_giveup, nfr);                                   \
        if (G > 1)                                                                                                       \
            hipLaunchKernelGGL((var_wavefront_kernel<C, M>), dim3((unsigned)n_frames), dim3(64 * nw1), 0, s, in, out, h, w, pal, vp, \
                               reinterpret_cast<float *>(ws), 1, gprog, 0, nfr);                                         \
    } while (0)
#define DP_VARW_M(C)                   \
    do {                               \
        if (model == 1) DP_VARW(C, 1); \
        else if (model == 2) DP_VARW(C, 2); \
        else if (model == 3) DP_VARW(C, 3); \
        else DP_VARW(C, 4);            \
    } while (0)
        if (pal.n_inner > kQueueSmall || pal.K > 256) DP_VARW_M(kQueueLarge);   // (wide lists of 257..1024 colours: these instances)
        else DP_VARW_M(kQueueSmall);
#undef DP_VARW_M
#undef DP_VARW
        prof_end(pm, s);
        DP_HIP(hipGetLastError());
        return DP_OK;
    }
    if (model == 4 && n_frames <= 0x7fffffff &&
        ((size_t)8 * w + 768) * sizeof(float) + (pal.K > 64 ? (size_t)pal.K * 16 : 0) + 512 <= (size_t)158 * 1024) {
        // Ostromoukhov, serpentine: one wave per frame (two error rows + the coefficient table in LDS)
        const size_t lds = ((size_t)8 * w + 768) * sizeof(float) + (pal.K > 64 ? (size_t)pal.K * 16 : 0);
        const bool big = pal.n_inner > kQueueSmall;
#define DP_OSR(C, MM)                                                                                                   \
    do {                                                                                                               \
        auto kern = os_rowserial_kernel<C, MM>;                                                                        \
        DP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                   (int)lds));                                                                         \
        hipLaunchKernelGGL(kern, dim3((unsigned)n_frames), dim3(64), lds, s, in, out, h, w, pal, vp);                   \
    } while (0)
        if (pal.K <= 64) {
            if (big) DP_OSR(kQueueLarge, 1); else DP_OSR(kQueueSmall, 1);
        } else if (pal.K <= 256) {
            if (big) DP_OSR(kQueueLarge, 4); else DP_OSR(kQueueSmall, 4);
        } else {
            if (big) DP_OSR(kQueueLarge, 16); else DP_OSR(kQueueSmall, 16);
        }
#undef DP_OSR
        prof_end(pm, s);
        DP_HIP(hipGetLastError());
        return DP_OK;
    }
    const int64_t blocks = (n_frames + 63) / 64;
    if (pal.n_inner > kQueueSmall)
        hipLaunchKernelGGL(var_serial_kernel<kQueueLarge>, dim3((unsigned)blocks), dim3(64), 0, s, in, out, n_frames, h, w,
                           pal, vp, reinterpret_cast<float *>(ws));
    else
        hipLaunchKernelGGL(var_serial_kernel<kQueueSmall>, dim3((unsigned)blocks), dim3(64), 0, s, in, out, n_frames, h, w,
                           pal, vp, reinterpret_cast<float *>(ws));
    prof_end(pm, s);
    DP_HIP(hipGetLastError());
    return DP_OK;
}

}  // namespace dp
