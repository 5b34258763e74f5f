// Lloyd passes of the k-means palette extractor (ColorReducer.generate_kmeans_palette,
// dithering_lib.py:1845-1857 -> sklearn KMeans) over packed uint8 RGB pixels.
//
// kmeans_step_kernel: 3 B/pixel in, nothing out but K*(3+1[+1]) int64 totals.  Each lane owns 4 consecutive pixels
// (one dwordx3 load); the centres sit in LDS as float32 {-2c_r, -2c_g, -2c_b, |c|^2} and are read as wave-wide
// broadcasts.  The label comes from a float32 scan of  score_j = |c_j|^2 - 2 c_j.x  (three v_fma_f32 per pixel and
// centre: the f32 add/mul/fma instructions are the ones that issue at 2-3 cycles on gfx950, see
// profiles/microbench) that keeps the two smallest scores (v_cmp, v_cndmask, v_med3_f32, v_min_f32); when they are
// closer than the float32 error bound, the float64 scan over (x-c)^2 decides (lowest index on exact ties), so labels
// are the float64 labels of the reference.  Per-cluster totals accumulate in LDS as packed 64-bit words
// (r | g<<28, b | count<<28[, sum of squares]; one set per wave, a workgroup sees < 2^16 pixels between flushes, so
// every field fits) and leave the workgroup as int64 atomics.  All totals are integers: any rank count and any
// reduction order give identical results.
//
// kmeans_cells_build[16]_kernel + kmeans_cells_kernel (K <= 256, images of 2^19 pixels and more): the same pass over
// per-cell candidate lists rebuilt from the current centres in front of every pass -- a pixel scores the 1..7 (K > 64: 1..15) centres
// that can be nearest somewhere in its 16x16x16 cell instead of all K; totals through wave-level sums where a wave's
// pixels share a label and one packed LDS atomic per pixel elsewhere.  Same labels, same int64 totals.
//
// kmeans_update_kernel: the centre update of one Lloyd iteration on the device (means, squared shift, sklearn's
// tolerance test, "assignments unchanged" test, inertia), so that the host loop launches iterations back to back and
// reads the status back only every few iterations.
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

#include "dp_internal.h"
#include "wave_util.hip.h"
#include "kmeans_label.hip.h"

namespace dp {
namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / 64;
constexpr int kGroupsPerFlush = 1 << 14;  // groups of 4 pixels a workgroup accumulates before it flushes (2^16 pixels)

template <bool SQ, bool KEYS>
__global__ __launch_bounds__(kBlock) void kmeans_step_kernel(const uint8_t *__restrict__ px, const int64_t n,
                                                             const double *__restrict__ centers, const double *__restrict__ mean, const int K,
                                                             unsigned long long *__restrict__ sums,
                                                             unsigned long long *__restrict__ counts,
                                                             unsigned long long *__restrict__ sumsq)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float4 *s_c4 = reinterpret_cast<float4 *>(smem);                                    // K: {-2c, |c|^2} float32
    double *s_c = reinterpret_cast<double *>(s_c4 + K);                                  // K*3 float64 (near ties)
    unsigned long long *s_acc = reinterpret_cast<unsigned long long *>(s_c + 4 * K);     // [waves][K][2 or 3]
    constexpr int kW = SQ ? 3 : 2;
    for (int i = threadIdx.x; i < K; i += kBlock) {
        const double c0 = centers[3 * i], c1 = centers[3 * i + 1], c2 = centers[3 * i + 2];
        stage_centre_f64(s_c + 4 * i, c0, c1, c2, mean);
        s_c4[i] = make_float4((float)(-2.0 * c0), (float)(-2.0 * c1), (float)(-2.0 * c2),
                              (float)(c0 * c0 + c1 * c1 + c2 * c2 + (KEYS ? (double)kScoreBias : 0.0)));
    }
    for (int i = threadIdx.x; i < kWavesPerBlock * K * kW; i += kBlock) s_acc[i] = 0;
    __syncthreads();
    unsigned long long *acc = s_acc + (size_t)(threadIdx.x >> 6) * K * kW;  // this wave's totals

    const int64_t n_groups = (n + 3) / 4;
    const bool aligned = ((uintptr_t)px & 3) == 0;
    uint32_t since_flush = 0;
    auto flush = [&]() {
        __syncthreads();
        for (int i = threadIdx.x; i < K; i += kBlock) {
            unsigned long long rg = 0, bn = 0, sq = 0;
#pragma unroll
            for (int w = 0; w < kWavesPerBlock; ++w) {
                unsigned long long *a = s_acc + ((size_t)w * K + i) * kW;
                rg += a[0];
                bn += a[1];
                a[0] = a[1] = 0;
                if (SQ) {
                    sq += a[2];
                    a[2] = 0;
                }
            }
            // (the per-wave fields hold < 2^16 pixels each: summed over 4 waves r and g stay below 2^26 -- unpack first)
            if (bn >> 28) {
                atomicAdd(&sums[3 * i], rg & 0xfffffffull);
                atomicAdd(&sums[3 * i + 1], rg >> 28);
                atomicAdd(&sums[3 * i + 2], bn & 0xfffffffull);
                atomicAdd(&counts[i], bn >> 28);
                if (SQ) atomicAdd(&sumsq[i], sq);
            }
        }
        __syncthreads();
    };
    for (int64_t g0 = (int64_t)blockIdx.x * kBlock; g0 < n_groups; g0 += (int64_t)gridDim.x * kBlock) {
        const int64_t gi = g0 + threadIdx.x;
        const int64_t p0 = gi * 4;
        const int cnt = gi < n_groups ? (int)min<int64_t>(4, n - p0) : 0;
        uint32_t v[4] = {0u, 0u, 0u, 0u};
        if (aligned && cnt == 4) {
            const uint3 w = reinterpret_cast<const uint3 *>(px)[gi];
            v[0] = w.x & 0xffffffu;
            v[1] = __builtin_amdgcn_perm(w.y, w.x, 0x0c050403u);
            v[2] = __builtin_amdgcn_perm(w.z, w.y, 0x0c040302u);
            v[3] = w.z >> 8;
        } else {
            for (int q = 0; q < cnt; ++q) {
                const uint8_t *b = px + (p0 + q) * 3;
                v[q] = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16);
            }
        }
        // centre loop outermost: each centre is read from LDS once for the lane's four pixels
        float fr[4], fg[4], fb[4], b0[4], b1[4];
        int best[4], k0[4], k1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            fr[q] = (float)(v[q] & 255u);
            fg[q] = (float)((v[q] >> 8) & 255u);
            fb[q] = (float)(v[q] >> 16);
            b0[q] = b1[q] = __int_as_float(0x7f800000);
            best[q] = 0;
            k0[q] = k1[q] = 0x7fffffff;
        }
#pragma unroll 4
        for (int j = 0; j < K; ++j) {
            const float4 c = s_c4[j];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float t = __fmaf_rn(fb[q], c.z, __fmaf_rn(fg[q], c.y, __fmaf_rn(fr[q], c.x, c.w)));
                if (KEYS) {
                    const int key = (int)((__float_as_uint(t) << 8) + (uint32_t)j);  // one v_lshl_add_u32
                    k1[q] = med3_s32(k0[q], k1[q], key);  // second smallest of {k0 <= k1, key}
                    k0[q] = min(k0[q], key);
                } else {
                    best[q] = t < b0[q] ? j : best[q];
                    b1[q] = __builtin_amdgcn_fmed3f(b0[q], b1[q], t);  // second smallest of {b0 <= b1, t}
                    b0[q] = fminf(b0[q], t);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q < cnt) {
                const uint32_t r = v[q] & 255u, g = (v[q] >> 8) & 255u, b = v[q] >> 16;
                int lab = KEYS ? (k0[q] & 255) : best[q];
                // float32 evaluation of |c|^2 - 2 c.x: the rounded coefficients are off by <= 2^-24 relative (|2c| <= 510,
                // x <= 255: 0.008 per term), three fma roundings of values below 2^19.6 (<= 0.024 each), |c|^2 by 0.012:
                // each score is within 0.11 of its exact value, so a gap of more than 0.25 (+ 1e-6 relative) settles the
                // order; anything closer -- and K == 1 leaves b1 infinite -- is decided in float64 as the reference does
                // KEYS: values below 2^20 (three roundings <= 0.031 each, the biased |c|^2 by 0.031, coefficients 0.024:
                // within 0.15), keys 256 apart per ulp of 0.0625: a gap of more than 6 ulp settles the order (K == 1
                // leaves k1 at its initial value, far away, and the label is 0 either way)
                const bool near_tie = KEYS ? (k1[q] - k0[q] <= (6 << 8) + 255) : !(b1[q] - b0[q] > 0.25f + 1e-6f * fabsf(b1[q]));
                if (near_tie) lab = label_f64(s_c, K, mean, r, g, b, lab);
                atomicAdd(&acc[lab * kW], (unsigned long long)r | ((unsigned long long)g << 28));
                atomicAdd(&acc[lab * kW + 1], (unsigned long long)b | (1ull << 28));
                if (SQ) atomicAdd(&acc[lab * kW + 2], (unsigned long long)(r * r + g * g + b * b));
            }
        }
        since_flush += kBlock;
        if (since_flush >= (uint32_t)kGroupsPerFlush) {  // block-uniform
            flush();
            since_flush = 0;
        }
    }
    flush();
}

// ---------------------------------------------------------------------------------------------------------------
// kmeans_mfma_kernel (K <= 256): the scores on the matrix cores.  NOT the default: measured slower than the VALU kernel
// at every K (table below); kept selectable (DP_KMEANS_MFMA=1) with its own parity test as the evidence for that choice.
//   score[centre i][pixel j] = (|c_i|^2 + BIAS) - 2 c_i . x_j  as two chained v_mfma_f32_32x32x2_f32 per 32 centres x 32
//   pixels (A = {-2c_r | -2c_g} then {-2c_b | 0}, B = {r | g} then {b | .}, C = |c|^2 + BIAS), so the VALU is left with
//   the reduction only: 3 instructions per pixel and centre instead of 7.  BIAS = 2^19 + 195076 puts every score into
//   [2^19, 2^20): one exponent, so the float bits shifted left by 8 still order like the scores and leave room for a
//   tag -- key = bits << 8 | (block of 32 centres << 4 | register number) (v_lshl_or_b32), the two smallest keys so far
//   kept by v_med3_i32 + v_min_i32, the two halves of a pixel's scores (lanes j and j + 32) merged at the end through
//   v_permlane32_swap_b32.  A wave takes 4 tiles of 32 pixels per round so that a block's A and C registers are
//   loaded once per 128 pixels.
//   Exactness: a score is within 0.3 of its exact value (coefficients rounded to float32: 0.03 + 0.023; three products
//   and three accumulations at < 2^20, whatever the matrix core's internal rounding: 6 x 0.0625 / 2 = 0.19, 0.29 if it
//   truncates); a gap of more than 12 ulp (0.75) between the two smallest proves the float64 order, anything closer is
//   decided by the float64 scan of kmeans_step_kernel.  The int64 totals are therefore the reference's, bit for bit.
//   D layout (CDNA3/4 ISA, V_MFMA_F32_32X32X2_F32): lane l, register r holds row (r & 3) + 4 (l >> 5) + 8 (r >> 2) of
//   column l & 31.
//   Measured per 8K pass (33 M pixels, MI355X, profiles/experiments/r02_kmeans_mfma_scores.md), VALU / MFMA kernel:
//   K = 32: 0.233 / 0.307 ms, 64: 0.397 / 0.499, 128: 0.770 / 0.940, 256: 1.557 / 1.974.  The reduction alone is
//   3 instructions per score, but with the accumulator traffic of the matrix instruction in the same register file the
//   VALU issues one instruction per ~7 cycles instead of 4.3 (SQ_WAIT_INST_ANY 58 % of the wave cycles), and the
//   kernel has a larger fixed part (one dword load per pixel and lane pair, LDS atomics from half-filled waves).
// ---------------------------------------------------------------------------------------------------------------
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int kNearKeys = (12 << 8) | 255;
constexpr int kMfmaMaxK = 256;
constexpr int kTiles = 4;  // tiles of 32 pixels a wave takes per round

template <bool SQ>
__global__ __launch_bounds__(kBlock) void kmeans_mfma_kernel(const uint8_t *__restrict__ px, const int64_t n,
                                                             const double *__restrict__ centers, const double *__restrict__ mean, const int K,
                                                             unsigned long long *__restrict__ sums,
                                                             unsigned long long *__restrict__ counts,
                                                             unsigned long long *__restrict__ sumsq)
{
    constexpr int kW = SQ ? 3 : 2;
    extern __shared__ __align__(16) unsigned char smem[];
    const int KB = (K + 31) >> 5, KP = KB * 32;  // blocks of 32 centre rows (rows >= K: a score no pixel reaches)
    float4 *s_c4 = reinterpret_cast<float4 *>(smem);                                  // KP: {-2c, |c|^2 + BIAS}
    double *s_c = reinterpret_cast<double *>(s_c4 + KP);                               // 4 KP float64 (near ties)
    unsigned long long *s_acc = reinterpret_cast<unsigned long long *>(s_c + 4 * KP);  // [waves][KP][2 or 3]
    for (int i = threadIdx.x; i < KP; i += kBlock) {
        if (i < K) {
            const double c0 = centers[3 * i], c1 = centers[3 * i + 1], c2 = centers[3 * i + 2];
            stage_centre_f64(s_c + 4 * i, c0, c1, c2, mean);
            s_c4[i] = make_float4((float)(-2.0 * c0), (float)(-2.0 * c1), (float)(-2.0 * c2),
                                  (float)(c0 * c0 + c1 * c1 + c2 * c2 + (double)kScoreBias));
        } else {
            s_c4[i] = make_float4(0.f, 0.f, 0.f, 1048000.0f);  // above every real score (< 2^19 + 390152), below 2^20
        }
    }
    for (int i = threadIdx.x; i < kWavesPerBlock * KP * kW; i += kBlock) s_acc[i] = 0;
    __syncthreads();
    unsigned long long *acc = s_acc + (size_t)(threadIdx.x >> 6) * KP * kW;
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;

    auto flush = [&]() {
        __syncthreads();
        for (int i = threadIdx.x; i < K; i += kBlock) {
            unsigned long long rg = 0, bn = 0, sq = 0;
#pragma unroll
            for (int w = 0; w < kWavesPerBlock; ++w) {
                unsigned long long *a = s_acc + ((size_t)w * KP + i) * kW;
                rg += a[0];
                bn += a[1];
                a[0] = a[1] = 0;
                if (SQ) {
                    sq += a[2];
                    a[2] = 0;
                }
            }
            if (bn >> 28) {
                atomicAdd(&sums[3 * i], rg & 0xfffffffull);
                atomicAdd(&sums[3 * i + 1], rg >> 28);
                atomicAdd(&sums[3 * i + 2], bn & 0xfffffffull);
                atomicAdd(&counts[i], bn >> 28);
                if (SQ) atomicAdd(&sumsq[i], sq);
            }
        }
        __syncthreads();
    };
    // one pixel per lane pair: lanes j and j + 32 load the same (unaligned) dword {r, g, b, next r}
    auto load_px = [&](const int64_t p) -> uint32_t {
        if (p + 1 < n) {
            uint32_t v;
            __builtin_memcpy(&v, px + p * 3, 4);
            return v;
        }
        if (p < n) return (uint32_t)px[p * 3] | ((uint32_t)px[p * 3 + 1] << 8) | ((uint32_t)px[p * 3 + 2] << 16);
        return 0u;
    };
    constexpr int kRoundPx = kWavesPerBlock * kTiles * 32;  // pixels a workgroup takes per round
    uint32_t since_flush = 0;
    const int64_t bstride = (int64_t)gridDim.x * kRoundPx;
    const int64_t lane_px = (threadIdx.x >> 6) * (kTiles * 32) + j;
    uint32_t wn[kTiles];
#pragma unroll
    for (int t = 0; t < kTiles; ++t) wn[t] = load_px((int64_t)blockIdx.x * kRoundPx + lane_px + 32 * t);
    // (block-uniform bounds: every wave of a workgroup runs the same number of rounds, for the flush barriers)
    for (int64_t b0 = (int64_t)blockIdx.x * kRoundPx; b0 < n; b0 += bstride) {
        const int64_t p0 = b0 + lane_px;
        uint32_t w[kTiles];
        float x1[kTiles], x2[kTiles];
        int m0[kTiles], m1[kTiles];
        const uint32_t sh = 8u * (uint32_t)h;
#pragma unroll
        for (int t = 0; t < kTiles; ++t) {
            w[t] = wn[t];
            wn[t] = load_px(p0 + bstride + 32 * t);  // the next round's pixels, in flight during this round
            x1[t] = (float)((w[t] >> sh) & 255u);
            x2[t] = (float)((w[t] >> 16) & 255u);
            m0[t] = m1[t] = 0x7fffffff;
        }
        for (int blk = 0; blk < KB; ++blk) {
            const float4 cj = s_c4[32 * blk + j];
            const float a1 = h ? cj.y : cj.x, a2 = h ? 0.0f : cj.z;
            floatx16 cbias;
#pragma unroll
            for (int r = 0; r < 16; ++r) cbias[r] = s_c4[32 * blk + (r & 3) + 4 * h + 8 * (r >> 2)].w;
            const uint32_t tag0 = (uint32_t)blk << 4;
#pragma unroll
            for (int t = 0; t < kTiles; ++t) {
                floatx16 d = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, x1[t], cbias, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, x2[t], d, 0, 0, 0);
                int a = m0[t], b = m1[t];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = (int)((__float_as_uint(d[r]) << 8) | (tag0 + (uint32_t)r));
                    b = med3_s32(a, b, key);
                    a = min(a, key);
                }
                m0[t] = a;
                m1[t] = b;
            }
        }
#pragma unroll
        for (int t = 0; t < kTiles; ++t) {
            // lanes 0..31 hold the rows with h = 0, lanes 32..63 those with h = 1: after the swap x[0] is the lower half's
            // value in every lane and x[1] the upper half's
            const auto sa = __builtin_amdgcn_permlane32_swap((unsigned)m0[t], (unsigned)m0[t], false, false);
            const auto sb = __builtin_amdgcn_permlane32_swap((unsigned)m1[t], (unsigned)m1[t], false, false);
            const int a_lo = (int)sa[0], a_hi = (int)sa[1], b_lo = (int)sb[0], b_hi = (int)sb[1];
            const int k0 = min(a_lo, a_hi), k1 = min(min(max(a_lo, a_hi), b_lo), b_hi);
            const int hh = a_hi < a_lo ? 1 : 0;  // which half owns the smallest (equal keys: the margin test below)
            const int r = k0 & 15, blk = (k0 >> 4) & 15;
            int lab = (r & 3) + 4 * hh + 8 * (r >> 2) + 32 * blk;
            const int64_t p = p0 + 32 * t;
            const uint32_t cr = w[t] & 255u, cg = (w[t] >> 8) & 255u, cbl = (w[t] >> 16) & 255u;
            if (p < n) {
                // a near tie (or K == 1): float64
                if (k1 - k0 <= kNearKeys || lab >= K) lab = label_f64(s_c, K, mean, cr, cg, cbl, 0);
                // lanes j add r | g << 28, lanes j + 32 add b | count << 28 (and nobody else touches this pixel)
                const unsigned long long v = h ? ((unsigned long long)cbl | (1ull << 28)) : ((unsigned long long)cr | ((unsigned long long)cg << 28));
                atomicAdd(&acc[lab * kW + h], v);
                if (SQ && h == 0) atomicAdd(&acc[lab * kW + 2], (unsigned long long)(cr * cr + cg * cg + cbl * cbl));
            }
        }
        since_flush += kRoundPx;
        if (since_flush >= (uint32_t)(kGroupsPerFlush * 4)) {  // block-uniform
            flush();
            since_flush = 0;
        }
    }
    flush();
}

// ---------------------------------------------------------------------------------------------------------------
// Lloyd pass over per-cell candidate lists (K <= 256, large n): the product path for big images.
// kmeans_cells_build_kernel (one launch in front of every pass, four lanes per cell): for every cell of the 16x16x16 colour grid, the centres
// that can be nearest to SOME point of the cell -- first those whose smallest distance to the cell's box does not exceed
// the smallest "largest distance" of any centre, then a pairwise test: j leaves the list if another listed k is closer
// on the whole box (|x-c_k|^2 - |x-c_j|^2 is linear in x: its maximum over the box is at a corner).  float32 with a
// slack of 1.0 on quantities below 4e5 (errors < 0.25): a superset of the exact sets, 1.6 entries on average for 32
// centres of uniform data, 6 at most.  8 bytes per cell: {n, e0 .. e6}, unused positions name entry K, a dummy whose
// score is above every real one; a cell with more than 7 candidates holds n = 2 and dummies only, which the pass below
// reads as a near tie.
// kmeans_cells_kernel: as kmeans_step_kernel, but a pixel scores only its cell's candidates -- the wave runs to the
// longest list among its lanes (4 rounds on uniform random pixels, 1-2 on images), 7 instructions per pixel and round
// (byte -> LDS offset, three v_fma_f32, key, v_med3_i32, v_min_i32; the key's tag is the list position).  Two keys
// within the float32 error bound, or a dummy-only cell, go to the float64 scan over all K as before: labels and int64
// totals are the reference's.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kCellsMaxK8 = 64;     // 8-byte lists up to here (K <= 255 would work; longer lists pay off above)
constexpr int kCellsMaxK = 256;
constexpr uint32_t kC4Bytes16 = 16u * 257u;  // the centre records in front of the 16-byte lists
constexpr int kCellsGrid = 4096;
constexpr int64_t kCellsMinPixels = 1 << 19;  // below this the table build and its 32 KB copy per workgroup do not pay
constexpr float kDummyScore = 1048000.0f;  // above every real score (< 2^19 + 390152), below 2^20

// (four lanes per cell: each takes every fourth centre in the two bound passes and every fourth survivor in the
// pairwise pass; the survivors meet in a bit mask in LDS, so the list comes out in ascending order whoever found what)
constexpr int kBuildThreads = 256;
constexpr int kBuildCells = kBuildThreads / 4;
__global__ __launch_bounds__(kBuildThreads) void kmeans_cells_build_kernel(const double *__restrict__ centers, const int K,
                                                                           uint32_t *__restrict__ cells)
{
    __shared__ float4 s_c[kCellsMaxK];
    __shared__ uint32_t s_mask[kBuildCells][8];  // survivors of the bound test, by centre
    __shared__ uint32_t s_keep[kBuildCells];     // survivors of the pairwise test, by position among the former
    for (int i = threadIdx.x; i < K; i += kBuildThreads) {
        const float x = (float)centers[3 * i], y = (float)centers[3 * i + 1], z = (float)centers[3 * i + 2];
        s_c[i] = make_float4(x, y, z, x * x + y * y + z * z);
    }
    const int cl = threadIdx.x >> 2, sub = threadIdx.x & 3;
    if (sub == 0) {
#pragma unroll
        for (int w = 0; w < 8; ++w) s_mask[cl][w] = 0u;
    }
    __syncthreads();
    const int cell = blockIdx.x * kBuildCells + cl;  // g' | r' << 4 | b' << 8 (what kmeans_cells_kernel's multiply yields)
    const float lo0 = (float)(((cell >> 4) & 15) << 4), lo1 = (float)((cell & 15) << 4), lo2 = (float)(((cell >> 8) & 15) << 4);
    const float hi0 = lo0 + 15.f, hi1 = lo1 + 15.f, hi2 = lo2 + 15.f;
    float U = __int_as_float(0x7f800000);
    for (int j = sub; j < K; j += 4) {
        const float4 c = s_c[j];
        const float f0 = fmaxf(fabsf(c.x - lo0), fabsf(c.x - hi0)), f1 = fmaxf(fabsf(c.y - lo1), fabsf(c.y - hi1)),
                    f2 = fmaxf(fabsf(c.z - lo2), fabsf(c.z - hi2));
        U = fminf(U, f0 * f0 + f1 * f1 + f2 * f2);
    }
    U = fminf(U, __shfl_xor(U, 1));
    U = fminf(U, __shfl_xor(U, 2));
    U += 1.0f;
    for (int j = sub; j < K; j += 4) {
        const float4 c = s_c[j];
        const float n0 = fmaxf(fmaxf(lo0 - c.x, c.x - hi0), 0.f), n1 = fmaxf(fmaxf(lo1 - c.y, c.y - hi1), 0.f),
                    n2 = fmaxf(fmaxf(lo2 - c.z, c.z - hi2), 0.f);
        if (n0 * n0 + n1 * n1 + n2 * n2 <= U) atomicOr(&s_mask[cl][j >> 5], 1u << (j & 31));
    }
    __syncthreads();
    // the survivors in ascending order (every lane of the cell reads the same list)
    uint8_t surv[16];
    int cnt = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        uint32_t m = s_mask[cl][w];
        while (m) {
            const int bit = __ffs((int)m) - 1;
            m &= m - 1u;
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (q == cnt) surv[q] = (uint8_t)(32 * w + bit);
            ++cnt;
        }
    }
    if (sub == 0) s_keep[cl] = cnt <= 16 ? ((1u << cnt) - 1u) : 0u;
    __syncthreads();
    if (cnt <= 16) {
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            if (a < cnt && (a & 3) == sub) {
                const float4 cj = s_c[surv[a]];
                bool drop = false;
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    if (b < cnt && b != a && !drop) {
                        const float4 ck = s_c[surv[b]];
                        // max over the box of |x-ck|^2 - |x-cj|^2 = 2 x.(cj - ck) + |ck|^2 - |cj|^2
                        const float d0 = cj.x - ck.x, d1 = cj.y - ck.y, d2 = cj.z - ck.z;
                        const float m = 2.f * ((d0 > 0.f ? hi0 : lo0) * d0 + (d1 > 0.f ? hi1 : lo1) * d1 + (d2 > 0.f ? hi2 : lo2) * d2) + (ck.w - cj.w);
                        drop = m < -1.0f;
                    }
                }
                if (drop) atomicAnd(&s_keep[cl], ~(1u << a));
            }
        }
    }
    __syncthreads();
    if (sub != 0) return;
    const uint32_t keep = s_keep[cl];
    const int n = __popc(keep);
    uint32_t w0, w1;
    const uint32_t dummy = (uint32_t)K;
    if (cnt > 16 || n > 7) {
        w0 = 2u | (dummy << 8) | (dummy << 16) | (dummy << 24);
        w1 = dummy * 0x01010101u;
    } else {
        uint32_t e[7];
        int k = 0;
#pragma unroll
        for (int a = 0; a < 16; ++a)
            if (a < cnt && ((keep >> a) & 1u)) {
#pragma unroll
                for (int q = 0; q < 7; ++q)
                    if (q == k) e[q] = surv[a];
                ++k;
            }
#pragma unroll
        for (int q = 0; q < 7; ++q)
            if (q >= n) e[q] = dummy;
        w0 = (uint32_t)n | (e[0] << 8) | (e[1] << 16) | (e[2] << 24);
        w1 = e[3] | (e[4] << 8) | (e[5] << 16) | (e[6] << 24);
    }
    cells[2 * cell] = w0;
    cells[2 * cell + 1] = w1;
}

// The same for 16-byte lists {n, e0 .. e14} (K <= 256): up to 32 survivors of the bound test go through the pairwise test
// (in LDS, loops instead of unrolled registers); unused positions name a centre that is NOT on the list and lies far
// from the cell -- the one with the largest distance to the box among those the bound test rejected, else one the
// pairwise test rejected (either way more than 1.0 behind some listed centre everywhere in the cell, so it neither wins
// nor looks like a near tie); a cell with more than 15 candidates holds n = 2 and the same entry twice, which ties.
__global__ __launch_bounds__(kBuildThreads) void kmeans_cells_build16_kernel(const double *__restrict__ centers, const int K,
                                                                             uint32_t *__restrict__ cells)
{
    __shared__ float4 s_c[kCellsMaxK];
    __shared__ uint32_t s_mask[kBuildCells][8];
    __shared__ uint32_t s_keep[kBuildCells];
    __shared__ uint8_t s_surv[kBuildCells][32];
    for (int i = threadIdx.x; i < K; i += kBuildThreads) {
        const float x = (float)centers[3 * i], y = (float)centers[3 * i + 1], z = (float)centers[3 * i + 2];
        s_c[i] = make_float4(x, y, z, x * x + y * y + z * z);
    }
    const int cl = threadIdx.x >> 2, sub = threadIdx.x & 3;
    if (sub == 0) {
#pragma unroll
        for (int w = 0; w < 8; ++w) s_mask[cl][w] = 0u;
    }
    __syncthreads();
    const int cell = blockIdx.x * kBuildCells + cl;
    const float lo0 = (float)(((cell >> 4) & 15) << 4), lo1 = (float)((cell & 15) << 4), lo2 = (float)(((cell >> 8) & 15) << 4);
    const float hi0 = lo0 + 15.f, hi1 = lo1 + 15.f, hi2 = lo2 + 15.f;
    float U = __int_as_float(0x7f800000);
    for (int j = sub; j < K; j += 4) {
        const float4 c = s_c[j];
        const float f0 = fmaxf(fabsf(c.x - lo0), fabsf(c.x - hi0)), f1 = fmaxf(fabsf(c.y - lo1), fabsf(c.y - hi1)),
                    f2 = fmaxf(fabsf(c.z - lo2), fabsf(c.z - hi2));
        U = fminf(U, f0 * f0 + f1 * f1 + f2 * f2);
    }
    U = fminf(U, __shfl_xor(U, 1));
    U = fminf(U, __shfl_xor(U, 2));
    U += 1.0f;
    float far_d = -1.f;  // the rejected centre farthest from the box
    int far_j = -1;
    for (int j = sub; j < K; j += 4) {
        const float4 c = s_c[j];
        const float n0 = fmaxf(fmaxf(lo0 - c.x, c.x - hi0), 0.f), n1 = fmaxf(fmaxf(lo1 - c.y, c.y - hi1), 0.f),
                    n2 = fmaxf(fmaxf(lo2 - c.z, c.z - hi2), 0.f);
        const float dmin = n0 * n0 + n1 * n1 + n2 * n2;
        if (dmin <= U) atomicOr(&s_mask[cl][j >> 5], 1u << (j & 31));
        else if (dmin > far_d) {
            far_d = dmin;
            far_j = j;
        }
    }
#pragma unroll
    for (int m = 1; m <= 2; m <<= 1) {
        const float od = __shfl_xor(far_d, m);
        const int oj = __shfl_xor(far_j, m);
        if (od > far_d || (od == far_d && oj > far_j)) {  // (any total order: the four lanes must agree)
            far_d = od;
            far_j = oj;
        }
    }
    __syncthreads();
    int cnt = 0;
    for (int w = 0; w < 8; ++w) {
        uint32_t m = s_mask[cl][w];
        while (m) {
            const int bit = __ffs((int)m) - 1;
            m &= m - 1u;
            if (cnt < 32 && sub == 0) s_surv[cl][cnt] = (uint8_t)(32 * w + bit);
            ++cnt;
        }
    }
    if (sub == 0) s_keep[cl] = cnt >= 32 ? 0xffffffffu : ((1u << cnt) - 1u);
    __syncthreads();
    if (cnt <= 32) {
        for (int a = sub; a < cnt; a += 4) {
            const float4 cj = s_c[s_surv[cl][a]];
            bool drop = false;
            for (int b = 0; b < cnt && !drop; ++b) {
                if (b == a) continue;
                const float4 ck = s_c[s_surv[cl][b]];
                const float d0 = cj.x - ck.x, d1 = cj.y - ck.y, d2 = cj.z - ck.z;
                const float m = 2.f * ((d0 > 0.f ? hi0 : lo0) * d0 + (d1 > 0.f ? hi1 : lo1) * d1 + (d2 > 0.f ? hi2 : lo2) * d2) + (ck.w - cj.w);
                drop = m < -1.0f;
            }
            if (drop) atomicAnd(&s_keep[cl], ~(1u << a));
        }
    }
    __syncthreads();
    if (sub != 0) return;
    const uint32_t keep = s_keep[cl];
    const int n = __popc(keep);
    uint32_t bytes[16];
    if (cnt > 32 || n > 15) {
#pragma unroll
        for (int q = 0; q < 16; ++q) bytes[q] = 0u;
        bytes[0] = 2u;
    } else {
        uint32_t filler = far_j >= 0 ? (uint32_t)far_j : 0u;
        if (far_j < 0)
            for (int a = 0; a < cnt; ++a)
                if (!((keep >> a) & 1u)) {
                    filler = s_surv[cl][a];
                    break;
                }
#pragma unroll
        for (int q = 1; q < 16; ++q) bytes[q] = filler;
        bytes[0] = (uint32_t)n;
        int k = 1;
        for (int a = 0; a < cnt; ++a)
            if ((keep >> a) & 1u) {
                const uint32_t v = s_surv[cl][a];
#pragma unroll
                for (int q = 1; q < 16; ++q)
                    if (q == k) bytes[q] = v;
                ++k;
            }
    }
#pragma unroll
    for (int w = 0; w < 4; ++w)
        cells[4 * cell + w] = bytes[4 * w] | (bytes[4 * w + 1] << 8) | (bytes[4 * w + 2] << 16) | (bytes[4 * w + 3] << 24);
}

// W = 2: 8-byte lists {n, e0..e6} padded with the dummy entry K (K <= 255), 512 threads, 6 waves per SIMD, wide totals per
// wave.  W = 4 (K > 64): 16-byte lists {n, e0..e14} padded with a real centre that is not on the list and far from the
// cell (K <= 256), 1024 threads, wide totals per workgroup (LDS atomics), the centre records first in LDS.
template <bool SQ, int W>
__global__ __launch_bounds__(W == 2 ? 512 : 1024, W == 2 ? 6 : 4) void kmeans_cells_kernel(const uint8_t *__restrict__ px, const int64_t n,
                                                                   const double *__restrict__ centers, const double *__restrict__ mean, const int K,
                                                                   const uint32_t *__restrict__ cells,
                                                                   unsigned long long *__restrict__ sums,
                                                                   unsigned long long *__restrict__ counts,
                                                                   unsigned long long *__restrict__ sumsq)
{
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int kCellsBlock = W == 2 ? 512 : 1024;
    constexpr int kCellsWaves = kCellsBlock / 64;
    constexpr int kAccSets = W == 2 ? kCellsWaves : 1;  // wide totals: per wave, or one set per workgroup
    constexpr uint32_t kC4Off = W == 2 ? kCellsGrid * 8u : 0u, kCellOff = W == 2 ? 0u : kC4Bytes16;
    uint32_t *s_cells = reinterpret_cast<uint32_t *>(smem + kCellOff);                  // 4096 x {n, e0..}
    float4 *s_c4 = reinterpret_cast<float4 *>(smem + kC4Off);                           // {-2c, |c|^2 + BIAS} (W = 2: + dummy)
    double *s_c = reinterpret_cast<double *>(smem + (W == 2 ? kC4Off + 16u * (K + 1) : kCellOff + kCellsGrid * 16u));  // 4 K float64
    unsigned long long *s_acc = reinterpret_cast<unsigned long long *>(s_c + 4 * K);    // [sets][K][2 or 3]
    constexpr int kW = SQ ? 3 : 2;
    unsigned long long *s_l1 = s_acc + (size_t)kAccSets * K * kW;                       // [waves][K] packed r18|g18|b18|n10
    for (int i = threadIdx.x; i < kCellsGrid * W; i += kCellsBlock) s_cells[i] = cells[i];
    for (int i = threadIdx.x; i < K + (W == 2 ? 1 : 0); i += kCellsBlock) {
        if (i < K) {
            const double c0 = centers[3 * i], c1 = centers[3 * i + 1], c2 = centers[3 * i + 2];
            stage_centre_f64(s_c + 4 * i, c0, c1, c2, mean);
            s_c4[i] = make_float4((float)(-2.0 * c0), (float)(-2.0 * c1), (float)(-2.0 * c2),
                                  (float)(c0 * c0 + c1 * c1 + c2 * c2 + (double)kScoreBias));
        } else {
            s_c4[i] = make_float4(0.f, 0.f, 0.f, kDummyScore);
        }
    }
    for (int i = threadIdx.x; i < (kAccSets * kW + kCellsWaves) * K; i += kCellsBlock) s_acc[i] = 0;
    __syncthreads();
    unsigned long long *acc = s_acc + (W == 2 ? (size_t)(threadIdx.x >> 6) * K * kW : (size_t)0);
    unsigned long long *l1 = s_l1 + (size_t)(threadIdx.x >> 6) * K;
    const int lane = threadIdx.x & 63;

    const int64_t n_groups = (n + 3) / 4;
    const bool aligned = ((uintptr_t)px & 3) == 0;
    uint32_t since_flush = 0, since_l1 = 0;
    // a wave's packed first-level totals into the wide ones (W = 2: both private to the wave, plain read-modify-write)
    auto flush_l1 = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int i = lane; i < K; i += 64) {
            const unsigned long long v = l1[i];
            if (v) {
                l1[i] = 0;
                const unsigned long long rg = (v & 0x3ffffull) | (((v >> 18) & 0x3ffffull) << 28);
                const unsigned long long bn = ((v >> 36) & 0x3ffffull) | ((v >> 54) << 28);
                if (W == 2) {
                    acc[i * kW] += rg;
                    acc[i * kW + 1] += bn;
                } else {
                    atomicAdd(&acc[i * kW], rg);
                    atomicAdd(&acc[i * kW + 1], bn);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };
    auto flush = [&]() {
        flush_l1();
        since_l1 = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < K; i += kCellsBlock) {
            unsigned long long rg = 0, bn = 0, sq = 0;
#pragma unroll
            for (int w = 0; w < kAccSets; ++w) {
                unsigned long long *a = s_acc + ((size_t)w * K + i) * kW;
                rg += a[0];
                bn += a[1];
                a[0] = a[1] = 0;
                if (SQ) {
                    sq += a[2];
                    a[2] = 0;
                }
            }
            // (W = 2: a wave's fields hold < 2^16 pixels, summed over the 8 waves r and g stay below 2^27; W = 4: the
            // workgroup's one set sees 2^16 pixels between flushes: below 2^24)
            if (bn >> 28) {
                atomicAdd(&sums[3 * i], rg & 0xfffffffull);
                atomicAdd(&sums[3 * i + 1], rg >> 28);
                atomicAdd(&sums[3 * i + 2], bn & 0xfffffffull);
                atomicAdd(&counts[i], bn >> 28);
                if (SQ) atomicAdd(&sumsq[i], sq);
            }
        }
        __syncthreads();
    };
    // the candidate reads below address s_c4 by its LDS offset (the kernel has no static LDS)
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)s_c4 != kC4Off) __builtin_trap();
    const unsigned char *cell_bytes = reinterpret_cast<const unsigned char *>(s_cells);
    // the next round's twelve bytes are in flight while this round runs (full, aligned groups; the others load in place)
    auto fetch = [&](const int64_t gi) -> uint3 {
        if (aligned && gi * 4 + 4 <= n) return reinterpret_cast<const uint3 *>(px)[gi];
        return make_uint3(0u, 0u, 0u);
    };
    uint3 w_next = fetch((int64_t)blockIdx.x * kCellsBlock + threadIdx.x);
    for (int64_t g0 = (int64_t)blockIdx.x * kCellsBlock; g0 < n_groups; g0 += (int64_t)gridDim.x * kCellsBlock) {
        const int64_t gi = g0 + threadIdx.x;
        const int64_t p0 = gi * 4;
        const int cnt = gi < n_groups ? (int)min<int64_t>(4, n - p0) : 0;
        uint32_t v[4] = {0u, 0u, 0u, 0u};
        const uint3 w = w_next;
        w_next = fetch(gi + (int64_t)gridDim.x * kCellsBlock);
        if (aligned && cnt == 4) {
            v[0] = w.x & 0xffffffu;
            v[1] = __builtin_amdgcn_perm(w.y, w.x, 0x0c050403u);
            v[2] = __builtin_amdgcn_perm(w.z, w.y, 0x0c040302u);
            v[3] = w.z >> 8;
        } else {
            for (int q = 0; q < cnt; ++q) {
                const uint8_t *b = px + (p0 + q) * 3;
                v[q] = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16);
            }
        }
        float fr[4], fg[4], fb[4];
        int k0[4], k1[4];
        uint32_t L[4][W];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            fr[q] = (float)(v[q] & 255u);
            fg[q] = (float)((v[q] >> 8) & 255u);
            fb[q] = (float)(v[q] >> 16);
            // byte offset of the cell's list, (g' | r' << 4 | b' << 8) * 8: the high nibbles r' << 4 | g' << 12 | b' << 20
            // times 2^20 + 2^8 (mod 2^32) = r' << 12 | g' << 20 | r' << 24 | b' << 28, bits 16..19 clear
            // (W = 4: times 16, one bit further up)
            const uint32_t off = (uint32_t)__umul24(v[q] & 0xf0f0f0u, 0x100100u) >> (W == 2 ? 17 : 16);  // (HIP's __umul24 returns int)
            if (W == 2) {
                const uint2 t = *reinterpret_cast<const uint2 *>(cell_bytes + off);
                L[q][0] = t.x;
                L[q][1] = t.y;
            } else {
                const uint4 t = *reinterpret_cast<const uint4 *>(cell_bytes + off);
                L[q][0] = t.x;
                L[q][1] = t.y;
                L[q][W - 2] = t.z;
                L[q][W - 1] = t.w;
            }
            k0[q] = k1[q] = 0x7fffffff;
        }
        const int nmax = (int)max(max(L[0][0] & 255u, L[1][0] & 255u), max(L[2][0] & 255u, L[3][0] & 255u));
        uint32_t four = 4u;
        asm volatile("" : "+v"(four));  // (SDWA takes no inline constant)
#pragma unroll
        for (int i = 0; i < 4 * W - 1; ++i) {
            if (__ballot(i < nmax) == 0ull) break;  // wave-uniform
            // (byte (i + 1) of each list) << 4 in one instruction, the four records in flight together, then per pixel three
            // v_fmac_f32 (kept apart: packed pairs would cost moves), the key with the list position as tag, v_med3 + v_min
            uint32_t eo[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t word = L[q][(i + 1) >> 2];
                if (((i + 1) & 3) == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(eo[q]) : "v"(four), "v"(word));
                else if (((i + 1) & 3) == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(eo[q]) : "v"(four), "v"(word));
                else if (((i + 1) & 3) == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(eo[q]) : "v"(four), "v"(word));
                else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(eo[q]) : "v"(four), "v"(word));
            }
            floatx4 c0, c1, c2, c3;
            asm volatile(
                "ds_read_b128 %0, %4 offset:%8\n\t"
                "ds_read_b128 %1, %5 offset:%8\n\t"
                "ds_read_b128 %2, %6 offset:%8\n\t"
                "ds_read_b128 %3, %7 offset:%8\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3)
                : "v"(eo[0]), "v"(eo[1]), "v"(eo[2]), "v"(eo[3]), "n"(kC4Off)
                : "memory");
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const floatx4 c = q == 0 ? c0 : q == 1 ? c1 : q == 2 ? c2 : c3;
                float t = c[3];
                asm("v_fmac_f32 %0, %1, %2" : "+v"(t) : "v"(fr[q]), "v"(c[0]));
                asm("v_fmac_f32 %0, %1, %2" : "+v"(t) : "v"(fg[q]), "v"(c[1]));
                asm("v_fmac_f32 %0, %1, %2" : "+v"(t) : "v"(fb[q]), "v"(c[2]));
                int key;
                asm("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(key) : "v"(t), "n"(i));
                asm("v_med3_i32 %0, %1, %0, %2" : "+v"(k1[q]) : "v"(k0[q]), "v"(key));
                asm("v_min_i32 %0, %0, %1" : "+v"(k0[q]) : "v"(key));
            }
        }
        int lab[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            lab[q] = 0;
            if (q < cnt) {
                const uint32_t r = v[q] & 255u, g = (v[q] >> 8) & 255u, b = v[q] >> 16;
                // the winner's list position is the key's tag: byte (tag + 1) of the list
                const uint32_t pos = (uint32_t)(k0[q] & 255) + 1u;
                if (W == 2) lab[q] = (int)__builtin_amdgcn_perm(L[q][1], L[q][0], 0x0c0c0c00u + pos);
                else lab[q] = (int)__builtin_amdgcn_perm(pos & 8u ? L[q][W - 1] : L[q][1], pos & 8u ? L[q][W - 2] : L[q][0], 0x0c0c0c00u + (pos & 7u));
                // scores within 0.15 of their exact value, keys 256 apart per ulp of 0.0625 (see kmeans_step_kernel):
                // a gap of more than 6 ulp settles the order among the listed centres, and an unlisted one is farther
                // than some listed one on the whole cell.  A list of one leaves k1 at the key of a dummy or of a far
                // unlisted centre (more than 1.0 behind a listed one everywhere in the cell); the two equal entries of an
                // overfull cell tie with each other.
                if (k1[q] - k0[q] <= (6 << 8) + 255) lab[q] = label_f64(s_c, K, mean, r, g, b, lab[q]);
            }
        }
        // Totals.  Images are coherent: most of a wave's 256 consecutive pixels carry one or two labels, and 256 LDS
        // atomics on one address serialise.  So: rounds that take the label of the first pixel not yet accounted for,
        // sum its pixels over the wave (packed r | g << 16 and b; v_add_u32_dpp reductions) and let one lane add the
        // totals; a round that found fewer than 32 pixels is the last one, and what is left goes through one packed
        // 64-bit LDS atomic per pixel (r, g, b in 18 bits each, the count in 10: flushed every 768 pixels).
        uint32_t rem = (1u << cnt) - 1u;
        // (not worth a round when fewer than 24 lanes share lane 0's first label: uniform random pixels, many clusters)
        const int first_label = __builtin_amdgcn_readfirstlane(lab[0]);
        const int n_rounds = __popcll(__ballot(lab[0] == first_label)) >= 24 ? 8 : 0;
        for (int round = 0; round < n_rounds; ++round) {
            const unsigned long long has = __ballot(rem != 0u);
            if (has == 0ull) break;
            const int src = __ffsll((long long)has) - 1;
            const int mine = (rem & 1u) ? lab[0] : (rem & 2u) ? lab[1] : (rem & 4u) ? lab[2] : lab[3];
            const int l = __builtin_amdgcn_readlane(mine, src);
            uint32_t rg = 0, bb = 0, sq = 0;
            int found = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool sel = ((rem >> q) & 1u) && lab[q] == l;
                const uint32_t r = v[q] & 255u, g = (v[q] >> 8) & 255u, b = v[q] >> 16;
                rg += sel ? (r | (g << 16)) : 0u;
                bb += sel ? b : 0u;
                if (SQ) sq += sel ? (r * r + g * g + b * b) : 0u;
                rem &= sel ? ~(1u << q) : ~0u;
                found += __popcll(__ballot(sel));
            }
            rg = wave_sum_to_lane63(rg);
            bb = wave_sum_to_lane63(bb);
            if (SQ) sq = wave_sum_to_lane63(sq);
            if (lane == 63) {
                atomicAdd(&acc[l * kW], (unsigned long long)(rg & 0xffffu) | ((unsigned long long)(rg >> 16) << 28));
                atomicAdd(&acc[l * kW + 1], (unsigned long long)bb | ((unsigned long long)found << 28));
                if (SQ) atomicAdd(&acc[l * kW + 2], (unsigned long long)sq);
            }
            if (found < 32) break;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if ((rem >> q) & 1u) {
                const uint32_t r = v[q] & 255u, g = (v[q] >> 8) & 255u, b = v[q] >> 16;
                atomicAdd(&l1[lab[q]], (unsigned long long)(r | (g << 18)) | ((unsigned long long)((b << 4) | (1u << 22)) << 32));
                if (SQ) atomicAdd(&acc[lab[q] * kW + 2], (unsigned long long)(r * r + g * g + b * b));
            }
        }
        if (++since_l1 == 3) {  // wave-uniform: 768 pixels at most in the packed fields
            flush_l1();
            since_l1 = 0;
        }
        since_flush += kCellsBlock;
        if (since_flush >= (uint32_t)kGroupsPerFlush) {  // block-uniform
            flush();
            since_flush = 0;
        }
    }
    flush();
}

// One workgroup: the centre update of one Lloyd iteration (lloyd_update_block, kmeans_label.hip.h).
__global__ __launch_bounds__(256) void kmeans_update_kernel(const long long *__restrict__ totals, double *__restrict__ centers,
                                                            long long *__restrict__ prev, double *__restrict__ status,
                                                            const int K, const double tol, const int max_iter)
{
    __shared__ double s_red[256];
    lloyd_update_block<false>(totals, centers, prev, status, K, tol, max_iter, s_red);
}

// ---------------------------------------------------------------------------------------------------------------
// kmeans_pp_kernel: sklearn's _kmeans_plusplus (greedy k-means++ with 2 + int(ln K) local trials) on the seeding sample,
// one workgroup.  The sample points are uint8 triples and every centre is one of them, so all squared distances, their
// prefix sums and the candidates' potentials are integers below 2^53: the float64 arithmetic of the reference
// (euclidean_distances, stable_cumsum, the dot with the sample weights) is exact and independent of summation order,
// and only `uniform * potential` rounds -- once, in float64, here as there.  The uniforms do not depend on the data:
// the host draws them from numpy's RandomState in the order sklearn would.
// Per centre: block prefix sum of the closest distances, the trials' picks (first index whose prefix sum >= value:
// np.searchsorted, clipped to n - 1), the potentials of the candidates, the first smallest one (np.argmin).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kPpThreads = 1024;
constexpr int kPpMaxN = 16384;   // points held in LDS (the reference's sample is 10 000)
constexpr int kPpPer = kPpMaxN / kPpThreads;
constexpr int kPpMaxTrials = 8;  // 2 + int(ln 1024) = 8

__global__ __launch_bounds__(kPpThreads) void kmeans_pp_kernel(const uint8_t *__restrict__ sample, const int n, const int K,
                                                                const int first, const double *__restrict__ uniforms,
                                                                const int n_trials, int *__restrict__ out_ids,
                                                                double *__restrict__ out_centers)
{
    __shared__ uint32_t s_pt[kPpMaxN];
    __shared__ unsigned long long s_scan[kPpThreads / 64];  // the waves' totals
    __shared__ unsigned long long s_red[kPpMaxTrials][kPpThreads / 64];
    __shared__ int s_pick[kPpMaxTrials];
    __shared__ int s_best;
    const int t = threadIdx.x;
    for (int i = t; i < n; i += kPpThreads) s_pt[i] = (uint32_t)sample[3 * i] | ((uint32_t)sample[3 * i + 1] << 8) | ((uint32_t)sample[3 * i + 2] << 16);
    __syncthreads();
    auto dist2 = [](const uint32_t a, const uint32_t b) -> uint32_t {
        const int d0 = (int)(a & 255u) - (int)(b & 255u), d1 = (int)((a >> 8) & 255u) - (int)((b >> 8) & 255u),
                  d2 = (int)(a >> 16) - (int)(b >> 16);
        return (uint32_t)(d0 * d0 + d1 * d1 + d2 * d2);
    };
    // thread t owns the points [lo, hi): contiguous, so that a prefix sum over threads is the prefix sum over points
    const int per = (n + kPpThreads - 1) / kPpThreads;
    const int lo = min(n, t * per), hi = min(n, lo + per);
    uint32_t closest[kPpPer];
    const uint32_t c0 = s_pt[first];
#pragma unroll
    for (int e = 0; e < kPpPer; ++e) closest[e] = lo + e < hi ? dist2(s_pt[lo + e], c0) : 0u;
    if (t == 0) {
        out_ids[0] = first;
        out_centers[0] = (double)(c0 & 255u);
        out_centers[1] = (double)((c0 >> 8) & 255u);
        out_centers[2] = (double)(c0 >> 16);
    }
    for (int c = 1; c < K; ++c) {
        // inclusive prefix sums over the threads' own totals
        unsigned long long mine = 0;
#pragma unroll
        for (int e = 0; e < kPpPer; ++e) mine += closest[e];
        // (inside a wave by shuffles, across the 16 waves through LDS: two barriers instead of the twenty of a Hillis-Steele scan
        // over all 1024 threads -- this kernel is one workgroup working through K dependent centres, every barrier is on its path)
        unsigned long long wincl = mine;
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long o = __shfl_up(wincl, off);
            if ((t & 63) >= off) wincl += o;
        }
        if ((t & 63) == 63) s_scan[t >> 6] = wincl;
        if (t < kPpMaxTrials) s_pick[t] = n - 1;  // np.clip(picks, None, n - 1): a value above the total
        __syncthreads();
        unsigned long long before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kPpThreads / 64; ++w) {
            const unsigned long long v = s_scan[w];
            before += w < (t >> 6) ? v : 0ull;
            total += v;
        }
        const unsigned long long incl = before + wincl, excl = incl - mine;
        const double pot = (double)total;
        // the trials' picks: the first point whose inclusive prefix sum is >= value lies in exactly one thread's range
        for (int q = 0; q < n_trials; ++q) {
            const double rv = __dmul_rn(uniforms[(size_t)(c - 1) * n_trials + q], pot);
            if (lo < hi && !((double)excl >= rv && t > 0) && (double)incl >= rv) {
                unsigned long long run = excl;
#pragma unroll
                for (int e = 0; e < kPpPer; ++e) {
                    if (lo + e < hi) {
                        run += closest[e];
                        if ((double)run >= rv) {
                            atomicMin(&s_pick[q], lo + e);
                            break;
                        }
                    }
                }
            }
        }
        __syncthreads();
        // potentials of the candidates: sum over all points of min(closest, distance to the candidate)
        // (this loop is most of the kernel -- own points x trials, on ONE compute unit: the squared distance as |x|^2 + |c|^2 - 2 x.c
        // with v_dot4_u32_u8 is 3 instructions instead of 12, and a thread's partial sum of at most kPpPer distances fits 32
        // bits; the values are the same integers.  The points are re-read from LDS: keeping them and their norms in registers
        // spilled at 1024 threads)
        uint32_t cand[kPpMaxTrials], cnrm[kPpMaxTrials], part[kPpMaxTrials];
#pragma unroll
        for (int q = 0; q < kPpMaxTrials; ++q) {
            cand[q] = q < n_trials ? s_pt[s_pick[q]] : 0u;
            cnrm[q] = __builtin_amdgcn_udot4(cand[q], cand[q], 0u, false);
            part[q] = 0u;
        }
#pragma unroll
        for (int e = 0; e < kPpPer; ++e) {
            if (lo + e < hi) {
                const uint32_t x = s_pt[lo + e];
                const uint32_t xn = __builtin_amdgcn_udot4(x, x, 0u, false);
#pragma unroll
                for (int q = 0; q < kPpMaxTrials; ++q)
                    if (q < n_trials) part[q] += min(closest[e], xn + cnrm[q] - 2u * __builtin_amdgcn_udot4(x, cand[q], 0u, false));
            }
        }
#pragma unroll
        for (int q = 0; q < kPpMaxTrials; ++q) {
            if (q < n_trials) {
                unsigned long long v = part[q];
                for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
                if ((t & 63) == 0) s_red[q][t >> 6] = v;
            }
        }
        __syncthreads();
        if (t == 0) {
            int best = 0;
            unsigned long long best_pot = ~0ull;
            for (int q = 0; q < n_trials; ++q) {
                unsigned long long v = 0;
                for (int w = 0; w < kPpThreads / 64; ++w) v += s_red[q][w];
                if (v < best_pot) {  // np.argmin: the first smallest
                    best_pot = v;
                    best = q;
                }
            }
            s_best = s_pick[best];
            out_ids[c] = s_pick[best];
            const uint32_t b = s_pt[s_pick[best]];
            out_centers[3 * c] = (double)(b & 255u);
            out_centers[3 * c + 1] = (double)((b >> 8) & 255u);
            out_centers[3 * c + 2] = (double)(b >> 16);
        }
        __syncthreads();
        const uint32_t nb = s_pt[s_best];
        const uint32_t nbn = __builtin_amdgcn_udot4(nb, nb, 0u, false);
#pragma unroll
        for (int e = 0; e < kPpPer; ++e)
            if (lo + e < hi) {
                const uint32_t x = s_pt[lo + e];
                closest[e] = min(closest[e], __builtin_amdgcn_udot4(x, x, 0u, false) + nbn - 2u * __builtin_amdgcn_udot4(x, nb, 0u, false));
            }
        __syncthreads();
    }
}

}  // namespace

// 64 KB of cell lists per (device, stream) that has run a pass, kept until the library is unloaded.  Launches on one
// stream are ordered, so a pass never sees another pass's lists as long as a call's two launches (lists, pass) are
// enqueued together: the entry's mutex is held across them (two host threads may share a stream).
struct CellsScratch {
    uint32_t *ptr = nullptr;
    std::mutex launch_mu;
};
static CellsScratch *cells_scratch(const int dev, hipStream_t s)
{
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, CellsScratch *> cache;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find({dev, s});
    if (it != cache.end()) return it->second;
    void *p = nullptr;
    if (hipMalloc(&p, (size_t)16 * kCellsGrid) != hipSuccess) {
        set_error("dp_kmeans_step_u8: hipMalloc of the cell lists failed");
        return nullptr;
    }
    CellsScratch *e = new CellsScratch;
    e->ptr = static_cast<uint32_t *>(p);
    cache[{dev, s}] = e;
    return e;
}

int launch_kmeans_step(const uint8_t *px, int64_t n, const double *centers, const double *mean, int K, int64_t *sums,
                       int64_t *counts, int64_t *sumsq, hipStream_t s)
{
    if (counts == sums + 3 * (size_t)K && (sumsq == nullptr || sumsq == counts + K)) {
        // one planar totals buffer (the device-side Lloyd loop, dp_kmeans_update): one memset instead of three launches
        DP_HIP(hipMemsetAsync(sums, 0, sizeof(int64_t) * (size_t)K * (sumsq ? 5 : 4), s));
    } else {
        DP_HIP(hipMemsetAsync(sums, 0, sizeof(int64_t) * 3 * (size_t)K, s));
        DP_HIP(hipMemsetAsync(counts, 0, sizeof(int64_t) * (size_t)K, s));
        if (sumsq) DP_HIP(hipMemsetAsync(sumsq, 0, sizeof(int64_t) * (size_t)K, s));
    }
    if (n == 0) return DP_OK;
    const int64_t groups = (n + 3) / 4;
    int cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const int64_t want = (groups + kBlock - 1) / kBlock;
    const unsigned blocks = (unsigned)std::min<int64_t>(want, (int64_t)cus * 8);  // persistent: 8 workgroups of 4 waves per CU
    ProfMark *pm = prof_begin(s);
    const size_t kw = sumsq ? 3 : 2;
    const size_t smem = sizeof(float4) * K + sizeof(double) * 4 * K + sizeof(unsigned long long) * kWavesPerBlock * K * kw;
    if (smem > 64 * 1024) {
        set_error("dp_kmeans_step_u8: too many clusters for the LDS accumulators");
        return DP_EUNSUPPORTED;
    }
    const bool want_mfma = exp_env("DP_KMEANS_MFMA") != nullptr;  // measured slower at every K: opt-in only
    if (want_mfma && K <= kMfmaMaxK) {
        // scores on the matrix cores: a wave takes 128 pixels per round, 4 workgroups of 4 waves per CU
        constexpr int kRoundPx = kWavesPerBlock * kTiles * 32;
        const int KP = ((K + 31) / 32) * 32;
        const size_t msmem = sizeof(float4) * KP + sizeof(double) * 4 * KP + sizeof(unsigned long long) * kWavesPerBlock * KP * kw;
        const int64_t rounds = (n + kRoundPx - 1) / kRoundPx;
        const unsigned mblocks = (unsigned)std::min<int64_t>(rounds, (int64_t)cus * 4);
        if (sumsq)
            hipLaunchKernelGGL(kmeans_mfma_kernel<true>, dim3(mblocks), dim3(kBlock), msmem, s, px, n, centers, mean, K,
                               reinterpret_cast<unsigned long long *>(sums), reinterpret_cast<unsigned long long *>(counts),
                               reinterpret_cast<unsigned long long *>(sumsq));
        else
            hipLaunchKernelGGL(kmeans_mfma_kernel<false>, dim3(mblocks), dim3(kBlock), msmem, s, px, n, centers, mean, K,
                               reinterpret_cast<unsigned long long *>(sums), reinterpret_cast<unsigned long long *>(counts), nullptr);
        prof_end(pm, s);
        DP_HIP(hipGetLastError());
        return DP_OK;
    }
    {
        // big images: per-cell candidate lists, rebuilt from the current centres in front of every pass
        // (DP_KMEANS_CELLS=0 keeps the full scan, =1 takes the lists at any size: tests)
        const char *e = exp_env("DP_KMEANS_CELLS");
        const bool force = e && e[0] == '1', off = e && e[0] == '0';
        const bool wide = K > kCellsMaxK8;  // 16-byte lists, 1024 threads, one set of wide totals per workgroup
        const int cblock = wide ? 1024 : 512, cwaves = cblock / 64;
        const size_t csmem = wide ? (size_t)kC4Bytes16 + 16 * kCellsGrid + sizeof(double) * 4 * K + sizeof(unsigned long long) * K * (kw + cwaves)
                                  : (size_t)8 * kCellsGrid + sizeof(float4) * (K + 1) + sizeof(double) * 4 * K +
                                        sizeof(unsigned long long) * cwaves * K * (kw + 1);
        if (!off && !want_mfma && K <= kCellsMaxK && (force || n >= kCellsMinPixels) && csmem <= 150 * 1024) {
            CellsScratch *scratch = cells_scratch(dev, s);
            if (!scratch) return DP_EHIP;
            std::lock_guard<std::mutex> launch_lock(scratch->launch_mu);
            uint32_t *cells = scratch->ptr;
            if (wide) hipLaunchKernelGGL(kmeans_cells_build16_kernel, dim3(kCellsGrid / kBuildCells), dim3(kBuildThreads), 0, s, centers, K, cells);
            else hipLaunchKernelGGL(kmeans_cells_build_kernel, dim3(kCellsGrid / kBuildCells), dim3(kBuildThreads), 0, s, centers, K, cells);
            const int64_t cwant = (groups + cblock - 1) / cblock;
            const int per_cu = wide ? 1 : (int)std::max<size_t>(1, std::min<size_t>(3, (160 * 1024) / (csmem + 1024)));  // 3 x 8 waves: 6 per SIMD
            const unsigned cblocks = (unsigned)std::min<int64_t>(cwant, (int64_t)cus * per_cu);
#define DP_KMC(SQF, WF)                                                                                                   \
    do {                                                                                                                 \
        auto kern = kmeans_cells_kernel<SQF, WF>;                                                                        \
        DP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)csmem)); \
        hipLaunchKernelGGL(kern, dim3(cblocks), dim3(cblock), csmem, s, px, n, centers, mean, K, cells,                        \
                           reinterpret_cast<unsigned long long *>(sums), reinterpret_cast<unsigned long long *>(counts), \
                           reinterpret_cast<unsigned long long *>(sumsq));                                              \
    } while (0)
            if (sumsq) {
                if (wide) DP_KMC(true, 4); else DP_KMC(true, 2);
            } else {
                if (wide) DP_KMC(false, 4); else DP_KMC(false, 2);
            }
#undef DP_KMC
            prof_end(pm, s);
            DP_HIP(hipGetLastError());
            return DP_OK;
        }
    }
    const bool keys = K <= 256 && !exp_env("DP_KMEANS_NO_KEYS");
#define DP_KM(SQF, KF)                                                                                                    \
    hipLaunchKernelGGL((kmeans_step_kernel<SQF, KF>), dim3(blocks), dim3(kBlock), smem, s, px, n, centers, mean, K,            \
                       reinterpret_cast<unsigned long long *>(sums), reinterpret_cast<unsigned long long *>(counts),   \
                       reinterpret_cast<unsigned long long *>(sumsq))
    if (sumsq) {
        if (keys) DP_KM(true, true); else DP_KM(true, false);
    } else {
        if (keys) DP_KM(false, true); else DP_KM(false, false);
    }
#undef DP_KM
    prof_end(pm, s);
    DP_HIP(hipGetLastError());
    return DP_OK;
}

int launch_kmeans_update(const int64_t *totals, double *centers, int64_t *prev, double *status, int K, double tol, int max_iter,
                         hipStream_t s)
{
    hipLaunchKernelGGL(kmeans_update_kernel, dim3(1), dim3(256), 0, s, reinterpret_cast<const long long *>(totals), centers,
                       reinterpret_cast<long long *>(prev), status, K, tol, max_iter);
    DP_HIP(hipGetLastError());
    return DP_OK;
}

}  // namespace dp

namespace dp {
int launch_kmeans_pp(const uint8_t *sample, int n, int K, int first, const double *uniforms, int n_trials, int *out_ids,
                     double *out_centers, hipStream_t s)
{
    if (n > kPpMaxN || n_trials > kPpMaxTrials) {
        set_error("dp_kmeans_plusplus_u8: sample larger than 16384 points (or more than 8 local trials)");
        return DP_EUNSUPPORTED;
    }
    hipLaunchKernelGGL(kmeans_pp_kernel, dim3(1), dim3(kPpThreads), 0, s, sample, n, K, first, uniforms, n_trials, out_ids,
                       out_centers);
    DP_HIP(hipGetLastError());
    return DP_OK;
}
}  // namespace dp
