// Ordered / nearest-palette kernels for gfx950.
//
// Replaces NoDitherStrategy.dither (dithering_lib.py:337-341), MatrixDitherStrategy.dither
// (:355-378) and the IGN strategy (:551-568) for packed uint8 RGB frames.
//
// Every kernel gives each lane 4 consecutive pixels = 12 bytes = one dwordx3 load and store, fully
// coalesced across the wave (768 B per wave each way), finds the two (three) nearest palette entries
// exactly and evaluates the reference's float64 decision `s0/(s0+s1) <= t` exactly (integer palettes:
// d0 * 2^sh <= m * (d0+d1) for t = m/2^sh, provably the same decision except when both sides are equal).
// Which kernel runs (launch_ordered):
//   ordered_lean_kernel<MODE>        integer palettes with a cell table (accel.hip): the fast path -- LDS
//                                    candidate blocks, branch-free main loop, rare pixels (split cells,
//                                    distance ties, exact equality, row-straddling groups) deferred to a
//                                    wave-private queue and resolved densely with tie codes / exceptions
//   ordered_lean_float_kernel<MODE>  float (use_gamma) palettes with a cell table: float32 ranking with a
//                                    certainty gap, float64 recomputation of the winners
//   ordered_cell_kernel<MODE>        the previous generation of the fast path (inline rare paths); kept for
//                                    buffers that are not dword-aligned
//   ordered_int_kernel / ordered_f64_kernel   brute force over the whole palette (no accelerator)
// Pixels whose result depends on scipy's visiting order in a way no table expresses, and near ties of the
// float path, only set a flag bit; fixup_kernel compacts the flagged pixels per workgroup into LDS and
// resolves them through the scipy-order KD-tree emulation (tree_query) and the literal float64 chain.
#include <type_traits>

#include "dp_internal.h"
#include "tree_query.hip.h"

namespace dp {

namespace {

constexpr int kBlock = 256;
constexpr int kListCap = 16384;           // LDS pixel list of the fix-up pass (64 KiB) = one full batch
constexpr int kLdsTreeK = 256;            // palettes up to this size (and kLdsTreeNodes nodes) get their tree staged in LDS
constexpr int kLdsTreeNodes = 512;
constexpr int kDrainAt = kListCap - kBlock * 64;

struct Geo {
    uint32_t n_px;   // pixels in this launch (< 2^31)
    uint32_t hw, w, h;
    double inv_hw, inv_w;
    int y0, x0;      // global coordinates of pixel (0,0)
    int ty0, tx0;    // y0 % th_h, x0 % th_w
    int aligned;     // in/out are 4-byte aligned
    int neg2;        // -(2 << kLocalBits), kept in a register on purpose (see cand8)
    uint32_t adv_y, adv_x;  // (tile stride in pixels) mod hw, split into rows and columns (persistent kernel)
    uint32_t *dirty;        // workspace word: number of waves that flagged a pixel (zeroed per launch)
};

__device__ __forceinline__ int med3i(const int a, const int b, const int c)
{
    return max(min(a, b), min(max(a, b), c));
}

// p -> (frame-local y, x) without integer division
__device__ __forceinline__ void locate(const Geo &g, const uint32_t p, uint32_t &y, uint32_t &x)
{
    uint32_t f = (uint32_t)((double)p * g.inv_hw);
    int32_t q = (int32_t)(p - f * g.hw);
    if (q < 0) q += (int32_t)g.hw;
    else if ((uint32_t)q >= g.hw) q -= (int32_t)g.hw;
    uint32_t yy = (uint32_t)((double)q * g.inv_w);
    int32_t xx = q - (int32_t)(yy * g.w);
    if (xx < 0) { xx += (int32_t)g.w; --yy; }
    else if ((uint32_t)xx >= g.w) { xx -= (int32_t)g.w; ++yy; }
    y = yy;
    x = (uint32_t)xx;
}

struct Cursor {  // walks 4 consecutive pixels in raster order, tracking the threshold-tile position
    uint32_t y, x;
    int ty, tx;
};

__device__ __forceinline__ void cursor_init(const Geo &g, const ThrDev &thr, const uint32_t p, Cursor &c,
                                            const bool need_tile)
{
    locate(g, p, c.y, c.x);
    c.ty = c.tx = 0;
    if (need_tile) {
        c.ty = (int)(((uint32_t)g.y0 + c.y) % (uint32_t)thr.th_h);
        c.tx = (int)(((uint32_t)g.x0 + c.x) % (uint32_t)thr.th_w);
    }
}

__device__ __forceinline__ void cursor_next(const Geo &g, const ThrDev &thr, Cursor &c)
{
    ++c.x;
    if (++c.tx == thr.th_w) c.tx = 0;
    if (c.x == g.w) {
        c.x = 0;
        c.tx = g.tx0;
        ++c.y;
        if (++c.ty == thr.th_h) c.ty = 0;
        if (c.y == g.h) {
            c.y = 0;
            c.ty = g.ty0;
        }
    }
}

__device__ __forceinline__ void load4(const uint8_t *__restrict__ in, const Geo &g, const uint32_t gidx,
                                      uint32_t px[4])
{
    const uint32_t p0 = gidx * 4u;
    if (g.aligned && p0 + 4u <= g.n_px) {
        const uint32_t *in32 = reinterpret_cast<const uint32_t *>(in) + (size_t)gidx * 3;
        const uint32_t w0 = in32[0], w1 = in32[1], w2 = in32[2];
        px[0] = w0 & 0xffffffu;
        px[1] = (w0 >> 24) | ((w1 & 0xffffu) << 8);
        px[2] = (w1 >> 16) | ((w2 & 0xffu) << 16);
        px[3] = w2 >> 8;
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            px[q] = 0;
            if (p0 + q < g.n_px) {
                const uint8_t *b = in + (size_t)(p0 + q) * 3;
                px[q] = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16);
            }
        }
    }
}

__device__ __forceinline__ void store4(uint8_t *__restrict__ out, const Geo &g, const uint32_t gidx,
                                       const uint32_t c[4])
{
    const uint32_t p0 = gidx * 4u;
    if (g.aligned && p0 + 4u <= g.n_px) {
        uint32_t *o32 = reinterpret_cast<uint32_t *>(out) + (size_t)gidx * 3;
        o32[0] = c[0] | (c[1] << 24);
        o32[1] = (c[1] >> 8) | (c[2] << 16);
        o32[2] = (c[2] >> 16) | (c[3] << 8);
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (p0 + q < g.n_px) {
                uint8_t *b = out + (size_t)(p0 + q) * 3;
                b[0] = (uint8_t)c[q];
                b[1] = (uint8_t)(c[q] >> 8);
                b[2] = (uint8_t)(c[q] >> 16);
            }
    }
}

__device__ __forceinline__ void store_flags(unsigned long long *__restrict__ flags, uint32_t *__restrict__ dirty,
                                            const uint32_t gidx, const bool slow[4])
{
    const uint32_t tile = gidx >> 6;
    const int lane = threadIdx.x & 63;
    unsigned long long any = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned long long b = __ballot(slow[q]);
        any |= b;
        if (lane == q) flags[(size_t)tile * 4 + q] = b;
    }
    // rare: queue this wave tile for the fix-up pass (dirty[0] = count, dirty[1..] = tile indices; when the
    // count exceeds the queue the fix-up pass scans the whole bitmap instead)
    if (any != 0ull && lane == 0) {
        const uint32_t slot = atomicAdd(dirty, 1u);
        if (slot < (uint32_t)kQueueTiles) dirty[1 + slot] = tile;
    }
}

// MODE: 0 nearest only; 1 matrix in integer form (uint32 compare); 2 matrix f32; 3 IGN
template <int MODE>
__global__ __launch_bounds__(kBlock) void ordered_int_kernel(const uint8_t *__restrict__ in,
                                                             uint8_t *__restrict__ out,
                                                             unsigned long long *__restrict__ flags, const Geo g,
                                                             const PalDev pal, const ThrDev thr, const float sx,
                                                             const float sy, const float sc)
{
    __shared__ uint32_t s_out[DP_MAX_COLORS];
    __shared__ uint32_t s_thr[MODE == 1 ? 256 : 1];
    for (int i = threadIdx.x; i < pal.K; i += kBlock) s_out[i] = pal.out_rgb[i];
    if (MODE == 1)
        for (int i = threadIdx.x; i < thr.th_h * thr.th_w; i += kBlock) s_thr[i] = thr.m[i];
    __syncthreads();

    const uint32_t gidx = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t p0 = gidx * 4u;
    uint32_t px[4];
    load4(in, g, gidx, px);

    constexpr int kBig = 0x7fffffff;
    int m0[4], m1[4], m2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) m0[q] = m1[q] = m2[q] = kBig;

    const int K = pal.K;
#pragma unroll 4
    for (int j = 0; j < K; ++j) {
        const uint32_t pj = pal.p4[j];  // wave-uniform: scalar loads
        const int nk = pal.nkey[j];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int dot = (int)__builtin_amdgcn_udot4(px[q], pj, 0u, false);
            const int key = nk - (dot << (kIdxBits + 1));  // ((|p|^2 - 2 x.p) << kIdxBits) | j
            const int n2 = med3i(m1[q], m2[q], key);
            const int n1 = med3i(m0[q], m1[q], key);
            m0[q] = min(m0[q], key);
            m1[q] = n1;
            m2[q] = n2;
        }
    }

    Cursor cur;
    cursor_init(g, thr, p0 < g.n_px ? p0 : 0u, cur, MODE == 1 || MODE == 2);
    uint32_t col[4];
    bool slow[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int xx = (int)__builtin_amdgcn_udot4(px[q], px[q], 0u, false);
        const uint32_t d0 = (uint32_t)((m0[q] >> kIdxBits) + xx);
        const uint32_t d1 = (uint32_t)((m1[q] >> kIdxBits) + xx);
        const uint32_t d2 = (uint32_t)((m2[q] >> kIdxBits) + xx);
        const int i0 = m0[q] & ((1 << kIdxBits) - 1);
        const int i1 = m1[q] & ((1 << kIdxBits) - 1);
        bool nearest = true, s;
        if (MODE == 0) {
            s = (d0 == d1);
        } else {
            const uint32_t S = d0 + d1;
            bool eq;
            if (MODE == 1) {
                const uint32_t mt = s_thr[cur.ty * thr.th_w + cur.tx];
                const uint32_t lhs = d0 << thr.sh;
                const uint32_t rhs = __umul24(mt, S);
                nearest = lhs <= rhs;
                eq = lhs == rhs;
            } else {
                float t;
                if (MODE == 2)
                    t = thr.f32[cur.ty * thr.th_w + cur.tx];
                else
                    t = ign_threshold(g.x0 + (int)cur.x, g.y0 + (int)cur.y, sx, sy, sc);
                // exact: a 24-bit significand times a 19-bit integer fits a double
                const double lhs = (double)d0, rhs = __dmul_rn((double)t, (double)S);
                nearest = lhs <= rhs;
                eq = lhs == rhs;
            }
            s = (d0 == d1) | eq | ((d1 == d2) & !nearest);
        }
        slow[q] = s & (p0 + q < g.n_px);
        col[q] = s_out[nearest ? i0 : i1];
        cursor_next(g, thr, cur);
    }
    store4(out, g, gidx, col);
    store_flags(flags, g.dirty, gidx, slow);
}


// ---------------------------------------------------------------------------------------------
// Fast path: integer palettes with a search accelerator (accel.hip).  One persistent 1024-lane
// workgroup per CU keeps the cell table in LDS (4096 blocks of 8 packed colours at a 32-byte stride,
// plus the blocks of split cells) and walks 4096-pixel tiles.  Per lane: 4 consecutive pixels; the
// eight ds_read_b128 of their candidate blocks are issued back to back, then everything is
// straight-line code: 7 VALU ops per candidate (two v_dot4, v_lshl_add, v_mad_i32_i24, min/med3/med3)
// and the exact integer decision.  Distance ties take their outcome from the 2-bit tie code of the
// colour (a rare L2 read); exact equality in the decision replays the reference's float64 chain
// inline; only tie code 3 and overflowing sub-cells are flagged for the fix-up pass.
// ---------------------------------------------------------------------------------------------
constexpr int kCellBlock = 1024;

// Keys of the 8 candidates of one block and the three smallest of them.
//   key = ((|p|^2 - 2 x.p) << 8) | (4*idx)      (4*idx = byte offset of the candidate inside its block)
// Hand scheduled: hipcc neither fuses the shift/add/multiply into v_lshl_add + v_mad_i32_i24 nor keeps
// med3 for the running top-3, and it has to pad DOT results with s_nop; here all eight v_dot4 pairs are
// issued first (a DOT result must not be read for 3 issue slots), then 2 ops per key, then the three
// smallest: sort the first three (min3/med3/max3), insert the other five (3 ops each).  `neg2` = -512 must sit in an SGPR (v_mad_i32_i24 takes no literal).
__device__ __forceinline__ void cand8(const uint32_t x, const uint4 ca, const uint4 cb, const int neg2, int &m0,
                                      int &m1, int &m2)
{
    int n0, n1, n2, n3, n4, n5, n6, n7, p0, p1, p2, p3, p4, p5, p6, p7;
    asm volatile(
        "v_dot4_u32_u8 %[n0], %[c0], %[c0], 0\n\t"
        "v_dot4_u32_u8 %[p0], %[x], %[c0], 0\n\t"
        "v_dot4_u32_u8 %[n1], %[c1], %[c1], 0\n\t"
        "v_dot4_u32_u8 %[p1], %[x], %[c1], 0\n\t"
        "v_dot4_u32_u8 %[n2], %[c2], %[c2], 0\n\t"
        "v_dot4_u32_u8 %[p2], %[x], %[c2], 0\n\t"
        "v_dot4_u32_u8 %[n3], %[c3], %[c3], 0\n\t"
        "v_dot4_u32_u8 %[p3], %[x], %[c3], 0\n\t"
        "v_dot4_u32_u8 %[n4], %[c4], %[c4], 0\n\t"
        "v_dot4_u32_u8 %[p4], %[x], %[c4], 0\n\t"
        "v_dot4_u32_u8 %[n5], %[c5], %[c5], 0\n\t"
        "v_dot4_u32_u8 %[p5], %[x], %[c5], 0\n\t"
        "v_dot4_u32_u8 %[n6], %[c6], %[c6], 0\n\t"
        "v_dot4_u32_u8 %[p6], %[x], %[c6], 0\n\t"
        "v_dot4_u32_u8 %[n7], %[c7], %[c7], 0\n\t"
        "v_dot4_u32_u8 %[p7], %[x], %[c7], 0\n\t"
        "v_lshl_add_u32 %[n0], %[n0], 8, 0\n\t"
        "v_lshl_add_u32 %[n1], %[n1], 8, 4\n\t"
        "v_lshl_add_u32 %[n2], %[n2], 8, 8\n\t"
        "v_lshl_add_u32 %[n3], %[n3], 8, 12\n\t"
        "v_lshl_add_u32 %[n4], %[n4], 8, 16\n\t"
        "v_lshl_add_u32 %[n5], %[n5], 8, 20\n\t"
        "v_lshl_add_u32 %[n6], %[n6], 8, 24\n\t"
        "v_lshl_add_u32 %[n7], %[n7], 8, 28\n\t"
        "v_mad_i32_i24 %[n0], %[p0], %[ng], %[n0]\n\t"
        "v_mad_i32_i24 %[n1], %[p1], %[ng], %[n1]\n\t"
        "v_mad_i32_i24 %[n2], %[p2], %[ng], %[n2]\n\t"
        "v_mad_i32_i24 %[n3], %[p3], %[ng], %[n3]\n\t"
        "v_mad_i32_i24 %[n4], %[p4], %[ng], %[n4]\n\t"
        "v_mad_i32_i24 %[n5], %[p5], %[ng], %[n5]\n\t"
        "v_mad_i32_i24 %[n6], %[p6], %[ng], %[n6]\n\t"
        "v_mad_i32_i24 %[n7], %[p7], %[ng], %[n7]\n\t"
        // top-3 insertion network (m0 <= m1 <= m2)
        "v_min3_i32 %[m0], %[n0], %[n1], %[n2]\n\t"
        "v_med3_i32 %[m1], %[n0], %[n1], %[n2]\n\t"
        "v_max3_i32 %[m2], %[n0], %[n1], %[n2]\n\t"
        "v_med3_i32 %[m2], %[m1], %[m2], %[n3]\n\t"
        "v_med3_i32 %[m1], %[m0], %[m1], %[n3]\n\t"
        "v_min_i32 %[m0], %[m0], %[n3]\n\t"
        "v_med3_i32 %[m2], %[m1], %[m2], %[n4]\n\t"
        "v_med3_i32 %[m1], %[m0], %[m1], %[n4]\n\t"
        "v_min_i32 %[m0], %[m0], %[n4]\n\t"
        "v_med3_i32 %[m2], %[m1], %[m2], %[n5]\n\t"
        "v_med3_i32 %[m1], %[m0], %[m1], %[n5]\n\t"
        "v_min_i32 %[m0], %[m0], %[n5]\n\t"
        "v_med3_i32 %[m2], %[m1], %[m2], %[n6]\n\t"
        "v_med3_i32 %[m1], %[m0], %[m1], %[n6]\n\t"
        "v_min_i32 %[m0], %[m0], %[n6]\n\t"
        "v_med3_i32 %[m2], %[m1], %[m2], %[n7]\n\t"
        "v_med3_i32 %[m1], %[m0], %[m1], %[n7]\n\t"
        "v_min_i32 %[m0], %[m0], %[n7]\n\t"
        : [n0] "=&v"(n0), [n1] "=&v"(n1), [n2] "=&v"(n2), [n3] "=&v"(n3), [n4] "=&v"(n4), [n5] "=&v"(n5),
          [n6] "=&v"(n6), [n7] "=&v"(n7), [p0] "=&v"(p0), [p1] "=&v"(p1), [p2] "=&v"(p2), [p3] "=&v"(p3),
          [p4] "=&v"(p4), [p5] "=&v"(p5), [p6] "=&v"(p6), [p7] "=&v"(p7), [m0] "=&v"(m0), [m1] "=&v"(m1),
          [m2] "=&v"(m2)
        : [x] "v"(x), [c0] "v"(ca.x), [c1] "v"(ca.y), [c2] "v"(ca.z), [c3] "v"(ca.w), [c4] "v"(cb.x),
          [c5] "v"(cb.y), [c6] "v"(cb.z), [c7] "v"(cb.w), [ng] "s"(neg2));
}

// The same keys, the TWO smallest only (pixels that take their nearest entry whatever the second one is): 44 operations
__device__ __forceinline__ void cand8n(const uint32_t x, const uint4 ca, const uint4 cb, const int neg2, int &m0, int &m1)
{
    int n0, n1, n2, n3, n4, n5, n6, n7, p0, p1, p2, p3, p4, p5, p6, p7;
    asm volatile(
        "v_dot4_u32_u8 %[n0], %[c0], %[c0], 0\n\t"
        "v_dot4_u32_u8 %[p0], %[x], %[c0], 0\n\t"
        "v_dot4_u32_u8 %[n1], %[c1], %[c1], 0\n\t"
        "v_dot4_u32_u8 %[p1], %[x], %[c1], 0\n\t"
        "v_dot4_u32_u8 %[n2], %[c2], %[c2], 0\n\t"
        "v_dot4_u32_u8 %[p2], %[x], %[c2], 0\n\t"
        "v_dot4_u32_u8 %[n3], %[c3], %[c3], 0\n\t"
        "v_dot4_u32_u8 %[p3], %[x], %[c3], 0\n\t"
        "v_dot4_u32_u8 %[n4], %[c4], %[c4], 0\n\t"
        "v_dot4_u32_u8 %[p4], %[x], %[c4], 0\n\t"
        "v_dot4_u32_u8 %[n5], %[c5], %[c5], 0\n\t"
        "v_dot4_u32_u8 %[p5], %[x], %[c5], 0\n\t"
        "v_dot4_u32_u8 %[n6], %[c6], %[c6], 0\n\t"
        "v_dot4_u32_u8 %[p6], %[x], %[c6], 0\n\t"
        "v_dot4_u32_u8 %[n7], %[c7], %[c7], 0\n\t"
        "v_dot4_u32_u8 %[p7], %[x], %[c7], 0\n\t"
        "v_lshl_add_u32 %[n0], %[n0], 8, 0\n\t"
        "v_lshl_add_u32 %[n1], %[n1], 8, 4\n\t"
        "v_lshl_add_u32 %[n2], %[n2], 8, 8\n\t"
        "v_lshl_add_u32 %[n3], %[n3], 8, 12\n\t"
        "v_lshl_add_u32 %[n4], %[n4], 8, 16\n\t"
        "v_lshl_add_u32 %[n5], %[n5], 8, 20\n\t"
        "v_lshl_add_u32 %[n6], %[n6], 8, 24\n\t"
        "v_lshl_add_u32 %[n7], %[n7], 8, 28\n\t"
        "v_mad_i32_i24 %[n0], %[p0], %[ng], %[n0]\n\t"
        "v_mad_i32_i24 %[n1], %[p1], %[ng], %[n1]\n\t"
        "v_mad_i32_i24 %[n2], %[p2], %[ng], %[n2]\n\t"
        "v_mad_i32_i24 %[n3], %[p3], %[ng], %[n3]\n\t"
        "v_mad_i32_i24 %[n4], %[p4], %[ng], %[n4]\n\t"
        "v_mad_i32_i24 %[n5], %[p5], %[ng], %[n5]\n\t"
        "v_mad_i32_i24 %[n6], %[p6], %[ng], %[n6]\n\t"
        "v_mad_i32_i24 %[n7], %[p7], %[ng], %[n7]\n\t"
        "v_min3_i32 %[m0], %[n0], %[n1], %[n2]\n\t"
        "v_med3_i32 %[m1], %[n0], %[n1], %[n2]\n\t"
        "v_med3_i32 %[m1], %[m0], %[m1], %[n3]\n\t"
        "v_min_i32 %[m0], %[m0], %[n3]\n\t"
        "v_med3_i32 %[m1], %[m0], %[m1], %[n4]\n\t"
        "v_min_i32 %[m0], %[m0], %[n4]\n\t"
        "v_med3_i32 %[m1], %[m0], %[m1], %[n5]\n\t"
        "v_min_i32 %[m0], %[m0], %[n5]\n\t"
        "v_med3_i32 %[m1], %[m0], %[m1], %[n6]\n\t"
        "v_min_i32 %[m0], %[m0], %[n6]\n\t"
        "v_med3_i32 %[m1], %[m0], %[m1], %[n7]\n\t"
        "v_min_i32 %[m0], %[m0], %[n7]\n\t"
        : [n0] "=&v"(n0), [n1] "=&v"(n1), [n2] "=&v"(n2), [n3] "=&v"(n3), [n4] "=&v"(n4), [n5] "=&v"(n5),
          [n6] "=&v"(n6), [n7] "=&v"(n7), [p0] "=&v"(p0), [p1] "=&v"(p1), [p2] "=&v"(p2), [p3] "=&v"(p3),
          [p4] "=&v"(p4), [p5] "=&v"(p5), [p6] "=&v"(p6), [p7] "=&v"(p7), [m0] "=&v"(m0), [m1] "=&v"(m1)
        : [x] "v"(x), [c0] "v"(ca.x), [c1] "v"(ca.y), [c2] "v"(ca.z), [c3] "v"(ca.w), [c4] "v"(cb.x),
          [c5] "v"(cb.y), [c6] "v"(cb.z), [c7] "v"(cb.w), [ng] "s"(neg2));
}

// Colours whose tie outcome no code expresses: binary search of the palette's exception list.
// Returns false when the colour is not listed (or the list overflowed at build time).
__device__ __forceinline__ bool find_exception(const PalDev &pal, const uint32_t x, uint32_t &pair, uint32_t &single)
{
    int lo = 0, hi = pal.n_exc - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const uint4 e = pal.exc[mid];
        if (e.x == x) {
            pair = e.y;
            single = e.z;
            return true;
        }
        if (e.x < x) lo = mid + 1;
        else hi = mid - 1;
    }
    return false;
}

// byte offset of the block of x's cell: slot = r' | b'<<4 | g'<<8 (dp_internal.h cell_slot), times 32
__device__ __forceinline__ uint32_t cell_offset(const uint32_t x)
{
    const uint32_t t = x & 0xf0f0f0u;
    const uint32_t y = t | (t << 12);  // bits 16..27 = r', b', g'
    return (y >> 11) & 0x1ffe0u;
}

// the same for the table of 4-entry blocks (16 bytes per cell)
__device__ __forceinline__ uint32_t cell_offset4(const uint32_t x)
{
    const uint32_t t = x & 0xf0f0f0u;
    const uint32_t y = t | (t << 12);
    return (y >> 12) & 0xfff0u;
}

// Keys of the 4 candidates of a small block and the three smallest (see cand8)
__device__ __forceinline__ void cand4(const uint32_t x, const uint4 ca, const int neg2, int &m0, int &m1, int &m2)
{
    int n0, n1, n2, n3, p0, p1, p2, p3;
    asm volatile(
        "v_dot4_u32_u8 %[n0], %[c0], %[c0], 0\n\t"
        "v_dot4_u32_u8 %[p0], %[x], %[c0], 0\n\t"
        "v_dot4_u32_u8 %[n1], %[c1], %[c1], 0\n\t"
        "v_dot4_u32_u8 %[p1], %[x], %[c1], 0\n\t"
        "v_dot4_u32_u8 %[n2], %[c2], %[c2], 0\n\t"
        "v_dot4_u32_u8 %[p2], %[x], %[c2], 0\n\t"
        "v_dot4_u32_u8 %[n3], %[c3], %[c3], 0\n\t"
        "v_dot4_u32_u8 %[p3], %[x], %[c3], 0\n\t"
        "v_lshl_add_u32 %[n0], %[n0], 8, 0\n\t"
        "v_lshl_add_u32 %[n1], %[n1], 8, 4\n\t"
        "v_lshl_add_u32 %[n2], %[n2], 8, 8\n\t"
        "v_lshl_add_u32 %[n3], %[n3], 8, 12\n\t"
        "v_mad_i32_i24 %[n0], %[p0], %[ng], %[n0]\n\t"
        "v_mad_i32_i24 %[n1], %[p1], %[ng], %[n1]\n\t"
        "v_mad_i32_i24 %[n2], %[p2], %[ng], %[n2]\n\t"
        "v_mad_i32_i24 %[n3], %[p3], %[ng], %[n3]\n\t"
        "v_min3_i32 %[m0], %[n0], %[n1], %[n2]\n\t"
        "v_med3_i32 %[m1], %[n0], %[n1], %[n2]\n\t"
        "v_max3_i32 %[m2], %[n0], %[n1], %[n2]\n\t"
        "v_med3_i32 %[m2], %[m1], %[m2], %[n3]\n\t"
        "v_med3_i32 %[m1], %[m0], %[m1], %[n3]\n\t"
        "v_min_i32 %[m0], %[m0], %[n3]\n\t"
        : [n0] "=&v"(n0), [n1] "=&v"(n1), [n2] "=&v"(n2), [n3] "=&v"(n3), [p0] "=&v"(p0), [p1] "=&v"(p1), [p2] "=&v"(p2),
          [p3] "=&v"(p3), [m0] "=&v"(m0), [m1] "=&v"(m1), [m2] "=&v"(m2)
        : [x] "v"(x), [c0] "v"(ca.x), [c1] "v"(ca.y), [c2] "v"(ca.z), [c3] "v"(ca.w), [ng] "s"(neg2));
}

template <int MODE>
__global__ __launch_bounds__(kCellBlock) void ordered_cell_kernel(const uint8_t *__restrict__ in,
                                                                  uint8_t *__restrict__ out,
                                                                  unsigned long long *__restrict__ flags,
                                                                  const Geo g, const PalDev pal, const ThrDev thr,
                                                                  const float sx, const float sy, const float sc,
                                                                  const uint32_t n_tiles)
{
    extern __shared__ __align__(16) uint32_t smem[];
    uint32_t *s_tab = smem;
    uint32_t *s_thr = smem + pal.tab_words;
    for (int i = threadIdx.x * 4; i < pal.tab_words; i += kCellBlock * 4)
        *reinterpret_cast<uint4 *>(&s_tab[i]) = *reinterpret_cast<const uint4 *>(&pal.cell_tab[i]);
    if (MODE == 1)
        for (int i = threadIdx.x; i < thr.th_h * thr.th_w; i += kCellBlock) s_thr[i] = thr.m[i];
    __syncthreads();
    const uint8_t *s_bytes = reinterpret_cast<const uint8_t *>(s_tab);

    const float thr_scale = 1.0f / (float)(1u << thr.sh);
    // threshold-tile addressing without divisions when both dimensions are powers of two (every Bayer table)
    const bool pow2 = (MODE == 1) && ((thr.th_w & (thr.th_w - 1)) == 0) && ((thr.th_h & (thr.th_h - 1)) == 0);

    uint32_t px[4];
    uint32_t tile = blockIdx.x;
    uint32_t fy = 0, fx = 0;  // frame-local coordinates of this lane's first pixel, advanced incrementally
    if (tile < n_tiles) {
        const uint32_t gidx0 = tile * kCellBlock + threadIdx.x;
        load4(in, g, gidx0, px);
        locate(g, gidx0 * 4u < g.n_px ? gidx0 * 4u : 0u, fy, fx);
    }
    for (; tile < n_tiles; tile += gridDim.x) {
        const uint32_t gidx = tile * kCellBlock + threadIdx.x;
        const uint32_t p0 = gidx * 4u;
        const uint32_t xq[4] = {px[0], px[1], px[2], px[3]};
        const uint32_t next = tile + gridDim.x;
        if (next < n_tiles) load4(in, g, next * kCellBlock + threadIdx.x, px);  // prefetch the next tile

        // candidate blocks of the four pixels: all LDS reads in flight together
        uint32_t blk[4];  // byte offset of the block in LDS
        uint4 ca[4], cb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t x = xq[q];
            blk[q] = cell_offset(x);
            ca[q] = *reinterpret_cast<const uint4 *>(s_bytes + blk[q]);
            cb[q] = *reinterpret_cast<const uint4 *>(s_bytes + blk[q] + 16);
        }
        bool slow[4] = {false, false, false, false};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // rare: the cell is split octree-fashion; descend by one colour bit per level
            for (int bit = 3; (ca[q].x >> 31) != 0; --bit) {
                if ((ca[q].x & 0x40000000u) || bit < 0) {
                    slow[q] = true;  // a single colour with more than 8 candidates: fix-up pass
                    break;
                }
                const uint32_t x = xq[q];
                const uint32_t sub = (((x >> bit) & 1u) << 2) | (((x >> (8 + bit)) & 1u) << 1) | ((x >> (16 + bit)) & 1u);
                blk[q] = (4096u * 8u + ((ca[q].x & 0xffffffu) * 8u + sub) * 8u) * 4u;
                ca[q] = *reinterpret_cast<const uint4 *>(s_bytes + blk[q]);
                cb[q] = *reinterpret_cast<const uint4 *>(s_bytes + blk[q] + 16);
            }
        }

        // threshold positions of the four pixels
        Cursor cur;
        cur.y = fy;
        cur.x = fx;
        cur.ty = cur.tx = 0;
        const bool same_row = pow2 && (fx + 3u < g.w);
        uint32_t trow = 0, tcol = 0;
        if (MODE == 1) {
            if (same_row) {
                trow = (((uint32_t)g.y0 + fy) & (uint32_t)(thr.th_h - 1)) * (uint32_t)thr.th_w;
                tcol = (uint32_t)g.x0 + fx;
            } else {
                cur.ty = (int)(((uint32_t)g.y0 + fy) % (uint32_t)thr.th_h);
                cur.tx = (int)(((uint32_t)g.x0 + fx) % (uint32_t)thr.th_w);
            }
        } else if (MODE == 2) {
            cur.ty = (int)(((uint32_t)g.y0 + fy) % (uint32_t)thr.th_h);
            cur.tx = (int)(((uint32_t)g.x0 + fx) % (uint32_t)thr.th_w);
        }

        uint32_t col[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t x = xq[q];
            int m0, m1, m2;
            cand8(x, ca[q], cb[q], g.neg2, m0, m1, m2);
            const int xx = (int)__builtin_amdgcn_udot4(x, x, 0u, false);
            const int a0 = m0 >> kLocalBits, a1 = m1 >> kLocalBits, a2 = m2 >> kLocalBits;
            const uint32_t d0 = (uint32_t)(a0 + xx), d1 = (uint32_t)(a1 + xx);
            uint32_t a = *reinterpret_cast<const uint32_t *>(s_bytes + blk[q] + (m0 & 0xfc));  // reported nearest
            uint32_t b = *reinterpret_cast<const uint32_t *>(s_bytes + blk[q] + (m1 & 0xfc));  // reported second
            bool s = slow[q];
            bool nearest = true;
            if (MODE == 0) {
                if (a0 == a1) {
                    const uint32_t code = (pal.code1[x >> 4] >> ((x & 15u) * 2)) & 3u;
                    if (code == 1) a = b;
                    else if (code == 2) a = *reinterpret_cast<const uint32_t *>(s_bytes + blk[q] + (m2 & 0xfc));
                    else if (code == 3) {
                        uint32_t pair, single;
                        if (find_exception(pal, x, pair, single)) a = pal.out_rgb[single];
                        else s = true;
                    }
                }
            } else {
                if (a0 == a1 || a1 == a2) {
                    const uint32_t code = (pal.code2[x >> 3] >> ((x & 7u) * 4)) & 15u;
                    if (code != 0u) {
                        const uint32_t c0 = a, c1 = b;
                        const uint32_t c2 = *reinterpret_cast<const uint32_t *>(s_bytes + blk[q] + (m2 & 0xfc));
                        if (code == 1) { a = c1; b = c0; }
                        else if (code == 2) { b = c2; }
                        else if (code == 3) { a = c2; b = c0; }
                        else if (code == 4) { a = c1; b = c2; }
                        else if (code == 5) { a = c2; b = c1; }
                        else {
                            uint32_t pair, single;
                            if (find_exception(pal, x, pair, single)) {
                                a = pal.out_rgb[pair & 0xffffu];
                                b = pal.out_rgb[pair >> 16];
                            } else {
                                s = true;
                            }
                        }
                    }
                }
                const uint32_t S = d0 + d1;
                float t;
                bool eq;
                if (MODE == 1) {
                    const uint32_t mt = same_row ? s_thr[trow + ((tcol + q) & (uint32_t)(thr.th_w - 1))]
                                                 : s_thr[cur.ty * thr.th_w + cur.tx];
                    const uint32_t lhs = d0 << thr.sh;
                    const uint32_t rhs = __umul24(mt, S);
                    nearest = lhs <= rhs;
                    eq = lhs == rhs;
                    t = __fmul_rn((float)mt, thr_scale);
                } else {
                    if (MODE == 2)
                        t = thr.f32[cur.ty * thr.th_w + cur.tx];
                    else
                        t = ign_threshold(g.x0 + (int)cur.x, g.y0 + (int)cur.y, sx, sy, sc);
                    const double lhs = (double)d0, rhs = __dmul_rn((double)t, (double)S);
                    nearest = lhs <= rhs;
                    eq = lhs == rhs;
                }
                if (eq) nearest = ordered_use_nearest((double)d0, (double)d1, t);  // rare: the literal float64 chain
            }
            slow[q] = s & (p0 + q < g.n_px);
            col[q] = nearest ? a : b;
            if (!(MODE == 1 && same_row) && MODE != 0) cursor_next(g, thr, cur);
        }
        store4(out, g, gidx, col);
        store_flags(flags, g.dirty, gidx, slow);

        // advance this lane's coordinates to its group in the next tile (no division)
        fx += g.adv_x;
        fy += g.adv_y;
        if (fx >= g.w) {
            fx -= g.w;
            ++fy;
        }
        if (fy >= g.h) fy -= g.h;
    }
}

// ---------------------------------------------------------------------------------------------
// Lean variant of the fast path: dword-aligned buffers, and for matrices power-of-two tables.  Same
// cell table, same candidate network, same decision, but the main loop is branch-free: byte shuffles
// through v_perm, the four thresholds of a lane from one padded table row, one colour read per pixel
// (of the chosen candidate only).  Everything that is rare per pixel but not per wave -- split cells
// (1.5 % of the pixels of the headline case, but ~100 % of its waves), distance ties (0.3 % / 18 %),
// exact equality in the decision, the one group per image row that straddles its end when the width is
// not a multiple of 4, a partial last group -- is deferred: the pixel index goes into a wave-private
// LDS queue, and whenever 64 have gathered the wave resolves them densely, one pixel per lane, with
// the complete code (lean_pixel_full) and overwrites their bytes.  Split cells need no test of their
// own: a marker block holds the marker word and seven zero words, i.e. seven equal colours, so its
// keys always contain a tie among the three smallest.
// MODE: 0 nearest only; 1 matrix in integer form (table in LDS); 2 matrix float32 (table read from
// global memory, it stays in L1); 3 IGN.
// ---------------------------------------------------------------------------------------------
constexpr int kLeanLdsWords = 160 * 1024 / 4;
constexpr int kLeanQueue = 128;                                         // entries per wave
constexpr int kLeanQueueWords = (kCellBlock / 64) * kLeanQueue;         // 8 KB at the top of LDS
constexpr int kLeanTabBytes = (kLeanLdsWords - kLeanQueueWords) * 4;    // table + thresholds must fit below
constexpr int kLeanHalfLdsWords = 80 * 1024 / 4;                        // HALF instances: two workgroups per CU
constexpr int kLeanHalfQueue = 112;                                     // entries per wave: drained from 48 up (47 + 64 at most)
constexpr int kLeanHalfDrain = 48;
constexpr int kLeanHalfTabBytes = (kLeanHalfLdsWords - (kCellBlock / 64) * kLeanHalfQueue) * 4;
constexpr int kWarpLutBytes = 768;                                      // tables over warped cells: the three maps ...
constexpr int kWarpLutAt = kLeanTabBytes - kWarpLutBytes;               // ... sit right below the queue

// Cell coordinates of a colour: the colour itself, or (tables over warped cells, accel.hip) its three bytes mapped
// through the per-channel tables staged in LDS.
template <bool WARP>
__device__ __forceinline__ uint32_t cell_coord(const uint32_t x, const uint8_t *s_bytes)
{
    if (!WARP) return x;
    const uint8_t *lut = s_bytes + kWarpLutAt;
    return (uint32_t)lut[x & 255u] | ((uint32_t)lut[256u + ((x >> 8) & 255u)] << 8) | ((uint32_t)lut[512u + (x >> 16)] << 16);
}

// x mod d for d > 0 with inv = 1.0/d (exact: the float64 quotient estimate is off by at most one)
__device__ __forceinline__ uint32_t umod_inv(const uint32_t x, const uint32_t d, const double inv)
{
    const uint32_t q = (uint32_t)((double)x * inv);
    int32_t r = (int32_t)(x - q * d);
    if (r < 0) r += (int32_t)d;
    else if ((uint32_t)r >= d) r -= (int32_t)d;
    return (uint32_t)r;
}

// row / column of a pixel position in the threshold table
__device__ __forceinline__ void thr_pos(const ThrDev &thr, const uint32_t gy, const uint32_t gx, uint32_t &row, uint32_t &col)
{
    if (thr.pow2) {
        row = gy & (uint32_t)(thr.th_h - 1);
        col = gx & (uint32_t)(thr.th_w - 1);
    } else {
        row = umod_inv(gy, (uint32_t)thr.th_h, thr.inv_h);
        col = umod_inv(gx, (uint32_t)thr.th_w, thr.inv_w);
    }
}

struct LeanThr {  // what the decision needs besides the distances
    uint32_t mt;  // MODE 1
    float t;      // MODE 2, 3
};

// nearest?  `eq` reports exact equality of the two sides (the float64 replay then decides)
template <int MODE>
__device__ __forceinline__ bool lean_decide(const uint32_t d0, const uint32_t S, const LeanThr &th, const int sh, bool &eq)
{
    if (MODE == 1) {
        const uint32_t lhs = d0 << sh;
        const uint32_t rhs = __umul24(th.mt, S);
        eq = lhs == rhs;
        return lhs <= rhs;
    }
    // exact: a 24-bit significand times a 19-bit integer fits a double
    const double lhs = (double)d0, rhs = __dmul_rn((double)th.t, (double)S);
    eq = lhs == rhs;
    return lhs <= rhs;
}

// Complete resolution of one pixel given its colour and threshold (split cells, tie codes, the float64 replay):
// returns the output colour; `slow`: left to the fix-up pass; `hard`: needed more than the plain block of its cell.
// The block that decides a pixel: its cell's, or -- split cells -- the leaf reached by one colour bit per level.
struct Leaf {
    uint32_t blk;        // byte offset of the block in the table
    bool in_lds;         // the block is in the staged part of the table (else: read it from global memory)
    uint4 ca;            // first 16 bytes of the block
    bool stuck;          // ended on a "single colour with too many candidates" marker
    bool split;          // the cell is split
};

// Reads from the table with the address space spelled out: the staged part through LDS instructions, deep split nodes
// through global loads (a pointer that may be either turns every access into a flat load, and two plain branches get
// merged into exactly that).
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) u32x4_t lds_uint4_t;
typedef const __attribute__((address_space(3))) uint32_t lds_uint32_t;
typedef const __attribute__((address_space(1))) u32x4_t glb_uint4_t;
typedef const __attribute__((address_space(1))) uint32_t glb_uint32_t;

__device__ __forceinline__ uint4 leaf_read4(const Leaf &lf, const PalDev &pal, const uint8_t *s_bytes, const uint32_t off)
{
    u32x4_t v;
    if (lf.in_lds) v = *(lds_uint4_t *)(s_bytes + off);
    else v = *(glb_uint4_t *)(reinterpret_cast<const uint8_t *>(pal.cell_tab) + off);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint32_t leaf_read1(const Leaf &lf, const PalDev &pal, const uint8_t *s_bytes, const uint32_t off)
{
    if (lf.in_lds) return *(lds_uint32_t *)(s_bytes + off);
    return *(glb_uint32_t *)(reinterpret_cast<const uint8_t *>(pal.cell_tab) + off);
}

template <int BW>
__device__ __forceinline__ void leaf_begin(const uint32_t x, const uint8_t *s_bytes, Leaf &lf)
{
    lf.blk = BW == 8 ? cell_offset(x) : cell_offset4(x);
    lf.in_lds = true;
    lf.ca = *reinterpret_cast<const uint4 *>(s_bytes + lf.blk);
    lf.stuck = false;
    lf.split = (lf.ca.x >> 31) != 0;
}

// one level down (call while leaf_pending); `bit`: 3 for the children of a 16^3 cell, then 2, 1, 0
template <int BW>
__device__ __forceinline__ void leaf_step(const uint32_t x, const int bit, const PalDev &pal, const uint8_t *s_bytes, Leaf &lf)
{
    if ((lf.ca.x & 0x40000000u) || bit < 0) {
        lf.stuck = true;
        return;
    }
    const uint32_t sub = (((x >> bit) & 1u) << 2) | (((x >> (8 + bit)) & 1u) << 1) | ((x >> (16 + bit)) & 1u);
    lf.blk = (4096u * BW + ((lf.ca.x & 0xffffffu) * 8u + sub) * BW) * 4u;
    // blocks beyond the staged part of the table (deep split nodes of clustered palettes) are read from global memory
    // (two explicit paths: a pointer that may be either would turn every access into a flat load)
    lf.in_lds = lf.blk < (uint32_t)pal.tab_words * 4u;
    lf.ca = leaf_read4(lf, pal, s_bytes, lf.blk);
}

__device__ __forceinline__ bool leaf_pending(const Leaf &lf) { return (lf.ca.x >> 31) != 0 && !lf.stuck; }

template <int BW>
__device__ __forceinline__ void leaf_find(const uint32_t x, const PalDev &pal, const uint8_t *s_bytes, Leaf &lf)
{
    leaf_begin<BW>(x, s_bytes, lf);
    for (int bit = 3; leaf_pending(lf); --bit) leaf_step<BW>(x, bit, pal, s_bytes, lf);
}

template <int MODE, int BW>  // BW: entries per block of the table in LDS (8, or 4: pal.cell_tab4)
__device__ __forceinline__ uint32_t resolve_pixel(const uint32_t x, const LeanThr &th, const Leaf &lf, const Geo &g,
                                                  const PalDev &pal, const ThrDev &thr, const uint8_t *s_bytes,
                                                  bool &slow_out, bool &hard)
{
    const uint32_t blk = lf.blk;
    const uint4 ca = lf.ca;
    // a single colour with more candidates than a block holds: fix-up pass, or (small palettes) a scan
    bool slow = lf.stuck && BW != 4;
    const bool scan = lf.stuck && BW == 4;
    hard = lf.split;
    // the three nearest: distances (minus |x|^2) a0 <= a1 <= a2, ordered by (distance, palette index); their colours
    // are fetched on demand: col(k) for the k-th nearest
    int a0, a1, a2;
    int m0, m1, m2;
    if (BW == 4 && scan) {
        constexpr int kBig = 0x7fffffff;
        m0 = m1 = m2 = kBig;
        for (int j = 0; j < pal.K; ++j) {
            const int dot = (int)__builtin_amdgcn_udot4(x, pal.p4[j], 0u, false);
            const int key = pal.nkey[j] - (dot << (kIdxBits + 1));  // ((|p|^2 - 2 x.p) << kIdxBits) | j
            const int n2 = med3i(m1, m2, key);
            const int n1 = med3i(m0, m1, key);
            m0 = min(m0, key);
            m1 = n1;
            m2 = n2;
        }
        a0 = m0 >> kIdxBits;
        a1 = m1 >> kIdxBits;
        a2 = pal.K > 2 ? (m2 >> kIdxBits) : kBig;
        if (pal.K < 3) m2 = m1;
    } else {
        if (BW == 8) {
            const uint4 cb = leaf_read4(lf, pal, s_bytes, blk + 16);
            cand8(x, ca, cb, g.neg2, m0, m1, m2);
        } else {
            cand4(x, ca, g.neg2, m0, m1, m2);
        }
        a0 = m0 >> kLocalBits;
        a1 = m1 >> kLocalBits;
        a2 = m2 >> kLocalBits;
    }
    auto col = [&](const int key) -> uint32_t {
        if (BW == 4 && scan) return pal.out_rgb[key & ((1 << kIdxBits) - 1)];
        return leaf_read1(lf, pal, s_bytes, blk | ((uint32_t)key & 0xfcu));
    };
    uint32_t c;
    if (MODE == 0) {
        int sel = m0;
        bool direct = false;
        c = 0;
        if (a0 == a1) {
            hard = true;
            const uint32_t code = (pal.code1[x >> 4] >> ((x & 15u) * 2)) & 3u;
            if (code == 1) sel = m1;
            else if (code == 2) sel = m2;
            else if (code == 3) {
                uint32_t pair, single;
                if (find_exception(pal, x, pair, single)) {
                    c = pal.out_rgb[single];
                    direct = true;
                } else {
                    slow = true;
                }
            }
        }
        if (!direct) c = col(sel);
    } else {
        const int xx = (int)__builtin_amdgcn_udot4(x, x, 0u, false);
        const uint32_t d0 = (uint32_t)(a0 + xx), d1 = (uint32_t)(a1 + xx);
        bool eq;
        bool nearest = lean_decide<MODE>(d0, d0 + d1, th, thr.sh, eq);
        int sa = m0, sb = m1;  // keys of the reported nearest / second
        bool direct = false;
        uint32_t exc_pair = 0;
        if (a0 == a1 || a1 == a2) {
            hard = true;
            const uint32_t code = (pal.code2[x >> 3] >> ((x & 7u) * 4)) & 15u;
            if (code == 1) { sa = m1; sb = m0; }
            else if (code == 2) { sb = m2; }
            else if (code == 3) { sa = m2; sb = m0; }
            else if (code == 4) { sa = m1; sb = m2; }
            else if (code == 5) { sa = m2; sb = m1; }
            else if (code != 0u) {
                uint32_t single;
                if (find_exception(pal, x, exc_pair, single)) direct = true;
                else slow = true;
            }
        }
        if (eq)  // the literal float64 chain decides
            nearest = ordered_use_nearest_call((double)d0, (double)d1,
                                               MODE == 1 ? __fmul_rn((float)th.mt, 1.0f / (float)(1u << thr.sh)) : th.t);
        if (direct) c = pal.out_rgb[nearest ? (exc_pair & 0xffffu) : (exc_pair >> 16)];
        else c = col(nearest ? sa : sb);
    }
    slow_out = slow;
    return c;
}

// fix-up bookkeeping of one pixel that stays unresolved
__device__ __forceinline__ void flag_slow_pixel(const uint32_t p, unsigned long long *__restrict__ flags, const Geo &g)
{
    const uint32_t wtile = p >> 8;  // 64 lanes x 4 pixels
    atomicOr(&flags[(size_t)wtile * 4 + (p & 3u)], 1ull << ((p >> 2) & 63u));
    // queue the wave tile for the fix-up pass (a tile may appear more than once: re-resolving is idempotent)
    const uint32_t slot = atomicAdd(g.dirty, 1u);
    if (slot < (uint32_t)kQueueTiles) g.dirty[1 + slot] = wtile;
}

// The deferred path: one queue entry per lane -- (group of four pixels) << 4 | the group's deferred pixels -- whose
// lowest pixel is resolved from its bytes in memory to its bytes in memory.  Returns the entry without that pixel
// (0 when the group is done; the caller queues the rest again: two deferred pixels in one group are rare).
template <int MODE, int BW, bool WARP>
__device__ __forceinline__ uint32_t lean_pixel_full(const uint32_t entry, const uint8_t *__restrict__ in,
                                                    uint8_t *__restrict__ out, unsigned long long *__restrict__ flags,
                                                    const Geo &g, const PalDev &pal, const ThrDev &thr,
                                                    const uint32_t *s_words, const float sx, const float sy, const float sc)
{
    const uint32_t mask = entry & 15u;
    const uint32_t p = (entry >> 4) * 4u + (uint32_t)__builtin_ctz(mask | 16u);
    const uint8_t *s_bytes = reinterpret_cast<const uint8_t *>(s_words);
    const uint8_t *b = in + (size_t)p * 3;
    const uint32_t x = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16);
    LeanThr th;
    th.mt = 0;
    th.t = 0.0f;
    if (MODE != 0) {
        uint32_t fy, fx;
        locate(g, p, fy, fx);
        if (MODE == 3) {
            th.t = ign_threshold(g.x0 + (int)fx, g.y0 + (int)fy, sx, sy, sc);
        } else {
            uint32_t row, col;
            thr_pos(thr, (uint32_t)g.y0 + fy, (uint32_t)g.x0 + fx, row, col);
            if (MODE == 1) th.mt = s_words[pal.tab_words + row * thr.tw_pad + col];
            else th.t = thr.fpad[row * thr.tw_pad + col];
        }
    }
    bool slow, hard;
    Leaf lf;
    leaf_find<BW>(cell_coord<WARP>(x, s_bytes), pal, s_bytes, lf);
    const uint32_t c = resolve_pixel<MODE, BW>(x, th, lf, g, pal, thr, s_bytes, slow, hard);
    uint8_t *o = out + (size_t)p * 3;
    o[0] = (uint8_t)c;
    o[1] = (uint8_t)(c >> 8);
    o[2] = (uint8_t)(c >> 16);
    if (slow) flag_slow_pixel(p, flags, g);
    const uint32_t rest = mask & (mask - 1u);
    return rest ? ((entry & ~15u) | rest) : 0u;
}

// ADAPT: compiled with the deep mode (see below); the launcher picks it for palettes crowded into few cells, whose
// pixels tend to sit in split cells -- the plain instantiation keeps its scalar registers for the lean loop
// HALF (4-entry blocks on plain cells, not adaptive): the workgroup declares 80 KB of LDS instead of all 160 -- the
// 64 KB table, thresholds and queue fit -- so that two workgroups share a CU (8 waves per SIMD; these instances need fewer
// than 64 VGPRs) and a wave's LDS waits are covered twice as often.  The launch bounds must SAY 8 waves per SIMD: without the
// hint the compiler used 106 scalar registers, and at 112 allocated per wave a SIMD's 800 hold only seven waves -- the second
// workgroup never fitted (rounds 2-3 ran this instance at one workgroup per CU without knowing; with 78 SGPRs C5's launch
// goes from 0.477 to 0.441 ms, same process, tools/bench_scripts/ab_libs.py bayer4:16:100).
template <int MODE, int BW, bool ADAPT, bool WARP, bool HALF = false>
__global__ __launch_bounds__(kCellBlock, HALF ? 8 : 4) void ordered_lean_kernel(const uint8_t *__restrict__ in,
                                                                  uint8_t *__restrict__ out,
                                                                  unsigned long long *__restrict__ flags,
                                                                  const Geo g, const PalDev pal, const ThrDev thr,
                                                                  const float sx, const float sy, const float sc,
                                                                  const uint32_t n_tiles)
{
    static_assert(!HALF || (BW == 4 && !ADAPT && !WARP), "HALF: 4-entry blocks on plain cells only");
    constexpr int kLdsWords = HALF ? kLeanHalfLdsWords : kLeanLdsWords;
    __shared__ __align__(16) uint32_t smem[kLdsWords];  // static: LDS addresses need no base register
    for (int i = threadIdx.x * 4; i < pal.tab_words; i += kCellBlock * 4) {
        uint4 v = *reinterpret_cast<const uint4 *>(&pal.cell_tab[i]);
        // The LDS copy of a marker block (marker word + equal entries) carries the marker TWICE: should the marker's key
        // be the smallest of the block, it ties with its copy, so that the top-2 network of the nearest-only slots
        // sees a split cell as a tie just as the top-3 network does.  (Blocks start at multiples of BW words.)
        if ((i % BW) == 0 && (v.x >> 31)) v.y = v.x;
        *reinterpret_cast<uint4 *>(&smem[i]) = v;
    }
    if (MODE == 1) {
        const int n = thr.th_h * thr.tw_pad;
        for (int i = threadIdx.x; i < n; i += kCellBlock) smem[pal.tab_words + i] = thr.mpad[i];
    }
    if (WARP && threadIdx.x < kWarpLutBytes / 4)
        smem[kWarpLutAt / 4 + threadIdx.x] = reinterpret_cast<const uint32_t *>(pal.warp_lut)[threadIdx.x];
    __syncthreads();
    const uint8_t *s_bytes = reinterpret_cast<const uint8_t *>(smem);
    const uint32_t lane = threadIdx.x & 63u;
    constexpr int kQueue = HALF ? kLeanHalfQueue : kLeanQueue;
    constexpr uint32_t kDrainAt = HALF ? kLeanHalfDrain : 64;  // (HALF: partly filled drains, so that the queue can be shorter)
    uint32_t *s_queue = smem + (kLdsWords - (kCellBlock / 64) * kQueue) + (threadIdx.x >> 6) * kQueue;
    uint32_t qcount = 0;  // wave-uniform
    bool deep = false;    // wave-uniform: resolve every pixel completely in place (see below)
    const uint3 *in3 = reinterpret_cast<const uint3 *>(in);
    uint3 *out3 = reinterpret_cast<uint3 *>(out);
    const uint32_t n_full = g.n_px >> 2;  // groups of four whole pixels; a partial last group goes through the queue

    uint32_t tile = blockIdx.x;
    uint32_t fy = 0, fx = 0;
    uint3 wn = make_uint3(0u, 0u, 0u);
    if (tile < n_tiles) {
        const uint32_t gidx0 = tile * kCellBlock + threadIdx.x;
        if (gidx0 < n_full) wn = in3[gidx0];
        if (gidx0 * 4u < g.n_px) locate(g, gidx0 * 4u, fy, fx);
    }
    // Which pixel slots of a wave tile can only take their nearest entry (thr.cls, host.cpp: every lane's threshold of that
    // slot is >= 1/2, and then f = s0/(s0+s1) <= 1/2 <= t whatever the second entry is -- every rounding of the
    // reference's float64 chain is monotone, dithering_lib.py:361-376): a property of the position of the wave's first
    // pixel in the threshold table, valid when the wave's 256 pixels lie in one image row.  With a Bayer matrix that is
    // every other slot.  Such slots run a top-2 network and no decision arithmetic (47 instead of 64 vector instructions).
    const bool use_cls = (MODE == 1 || MODE == 2) && thr.has_cls != 0 && !ADAPT;
    // (the table word is fetched one tile ahead and only shifted / masked when the tile starts, so that the scalar load
    // has a whole tile to arrive)
    uint32_t cls_word = 0, cls_shift = 0;  // wave-uniform
    auto tile_class = [&](const uint32_t ty, const uint32_t tx, const uint32_t gi) {
        cls_word = 0;
        cls_shift = 0;
        if (MODE == 0 || !use_cls) return;
        // (every lane looks at lane 0's position: uniform address, scalar load from the kernel arguments)
        const uint32_t y0l = (uint32_t)__builtin_amdgcn_readfirstlane((int)ty), x0l = (uint32_t)__builtin_amdgcn_readfirstlane((int)tx);
        const uint32_t g0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)gi);
        if (x0l + 256u > g.w || g0 + 64u > n_full) return;
        const uint32_t row = ((uint32_t)g.y0 + y0l) & (uint32_t)(thr.th_h - 1), col = ((uint32_t)g.x0 + x0l) & (uint32_t)(thr.th_w - 1);
        const uint32_t idx = row * (uint32_t)thr.th_w + col;  // < 256
        cls_word = thr.cls_nib[idx >> 3];
        cls_shift = (idx & 7u) * 4u;
    };
    if (tile < n_tiles) tile_class(fy, fx, tile * kCellBlock + threadIdx.x);
    for (; tile < n_tiles; tile += gridDim.x) {
        const uint32_t gidx = tile * kCellBlock + threadIdx.x;
        const uint3 wc = wn;
        const uint32_t cls = MODE == 0 ? 15u : ((cls_word >> cls_shift) & 15u);  // wave-uniform
        const uint32_t cy = fy, cx = fx;  // this tile's position; fy / fx move on to the next tile's
        {
            const uint32_t next = tile + gridDim.x;
            const uint32_t gn = next * kCellBlock + threadIdx.x;
            if (next < n_tiles && gn < n_full) wn = in3[gn];  // prefetch the next tile
            fx += g.adv_x;
            fy += g.adv_y;
            if (fx >= g.w) {
                fx -= g.w;
                ++fy;
            }
            if (fy >= g.h) fy -= g.h;
            cls_word = 0;
            if (next < n_tiles) tile_class(fy, fx, gn);
        }
        // all clear; lean_pixel_full ORs in the bits of the pixels it leaves to the fix-up pass later
        if (lane < 4u) flags[(size_t)(gidx >> 6) * 4 + lane] = 0ull;
        bool rare[4] = {false, false, false, false};
        uint32_t n_hard = 0;  // deep mode: pixels of this wave tile that needed more than their cell's block
        if (ADAPT && deep && gidx < n_full) {
            // Most recent pixels of this wave sat in split cells or on ties (palettes extracted from the image itself put
            // their colours exactly where the pixels are): resolve the four pixels completely right here instead of
            // computing a throw-away result and queueing them.
            uint32_t xq[4];
            xq[0] = wc.x & 0xffffffu;
            xq[1] = __builtin_amdgcn_perm(wc.y, wc.x, 0x0c050403u);
            xq[2] = __builtin_amdgcn_perm(wc.z, wc.y, 0x0c040302u);
            xq[3] = wc.z >> 8;
            LeanThr th[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                th[q].mt = 0;
                th[q].t = 0.0f;
            }
            if (MODE == 1 || MODE == 2) {
                uint32_t row, col;
                thr_pos(thr, (uint32_t)g.y0 + cy, (uint32_t)g.x0 + cx, row, col);
                const uint32_t at = row * (uint32_t)thr.tw_pad + col;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (MODE == 1) th[q].mt = smem[pal.tab_words + at + q];
                    else th[q].t = thr.fpad[at + q];
                }
            } else if (MODE == 3) {
#pragma unroll
                for (int q = 0; q < 4; ++q) th[q].t = ign_threshold(g.x0 + (int)cx + q, g.y0 + (int)cy, sx, sy, sc);
            }
            if ((MODE != 0) && (cx + 3u >= g.w)) {
#pragma unroll
                for (int q = 0; q < 4; ++q) rare[q] = true;  // the group straddles a row end: through the queue
            } else {
                uint32_t col[4];
                uint32_t hard_bits = 0;
                // the four descents advance level by level together, so that their reads are in flight at the same time
                Leaf lf[4];
                uint32_t xc[4];  // cell coordinates
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    xc[q] = cell_coord<WARP>(xq[q], s_bytes);
                    leaf_begin<BW>(xc[q], s_bytes, lf[q]);
                }
                for (int bit = 3; (int)leaf_pending(lf[0]) | (int)leaf_pending(lf[1]) | (int)leaf_pending(lf[2]) | (int)leaf_pending(lf[3]); --bit) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (leaf_pending(lf[q])) leaf_step<BW>(xc[q], bit, pal, s_bytes, lf[q]);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    bool slow, hard;
                    col[q] = resolve_pixel<MODE, BW>(xq[q], th[q], lf[q], g, pal, thr, s_bytes, slow, hard);
                    if (slow) flag_slow_pixel(gidx * 4u + (uint32_t)q, flags, g);
                    hard_bits += hard ? 1u : 0u;
                }
                uint3 wo;
                wo.x = __builtin_amdgcn_perm(col[1], col[0], 0x04020100u);
                wo.y = __builtin_amdgcn_perm(col[2], col[1], 0x05040201u);
                wo.z = __builtin_amdgcn_perm(col[3], col[2], 0x06050402u);
                out3[gidx] = wo;
                n_hard = hard_bits;
            }
        } else if (gidx < n_full) {
            uint32_t xq[4];
            xq[0] = wc.x & 0xffffffu;
            xq[1] = __builtin_amdgcn_perm(wc.y, wc.x, 0x0c050403u);
            xq[2] = __builtin_amdgcn_perm(wc.z, wc.y, 0x0c040302u);
            xq[3] = wc.z >> 8;
            uint32_t blk[4];
            uint4 ca[4], cb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t xc = cell_coord<WARP>(xq[q], s_bytes);
                blk[q] = BW == 8 ? cell_offset(xc) : cell_offset4(xc);
                ca[q] = *reinterpret_cast<const uint4 *>(s_bytes + blk[q]);
                if (BW == 8) cb[q] = *reinterpret_cast<const uint4 *>(s_bytes + blk[q] + 16);
            }
            // one straight-line body per class pattern (bit q: slot q takes its nearest entry whatever the second one is)
            auto body = [&](auto cls_c) {
                constexpr uint32_t CLS = (uint32_t)decltype(cls_c)::value;
                LeanThr th[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    th[q].mt = 0;
                    th[q].t = 0.0f;
                }
                if (CLS != 15u) {
                    if (MODE == 1 || MODE == 2) {
                        uint32_t row, col;
                        thr_pos(thr, (uint32_t)g.y0 + cy, (uint32_t)g.x0 + cx, row, col);
                        const uint32_t at = row * (uint32_t)thr.tw_pad + col;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            if ((CLS >> q) & 1u) continue;
                            if (MODE == 1) th[q].mt = smem[pal.tab_words + at + q];
                            else th[q].t = thr.fpad[at + q];
                        }
                    } else if (MODE == 3) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) th[q].t = ign_threshold(g.x0 + (int)cx + q, g.y0 + (int)cy, sx, sy, sc);
                    }
                }
                // the group runs over the end of its row: all four through the queue (their positions differ)
                const bool straddle = (MODE != 0) && (cx + 3u >= g.w);
                uint32_t col[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t x = xq[q];
                    int sel;
                    if ((CLS >> q) & 1u) {
                        int m0, m1;
                        if (BW == 8) {
                            cand8n(x, ca[q], cb[q], g.neg2, m0, m1);
                        } else {
                            int m2;
                            cand4(x, ca[q], g.neg2, m0, m1, m2);  // (a split cell's marker block: three equal entries behind the marker)
                            m1 = ((uint32_t)(m1 ^ m2) < (1u << kLocalBits)) ? m0 : m1;
                        }
                        // a tie for the nearest entry; also: any split cell (the marker word is staged twice, see above)
                        rare[q] = ((uint32_t)(m0 ^ m1) < (1u << kLocalBits)) | straddle;
                        sel = m0;
                    } else {
                        int m0, m1, m2;
                        if (BW == 8) cand8(x, ca[q], cb[q], g.neg2, m0, m1, m2);
                        else cand4(x, ca[q], g.neg2, m0, m1, m2);
                        const int a0 = m0 >> kLocalBits, a1 = m1 >> kLocalBits;
                        const bool tie = (a0 == a1) | ((uint32_t)(m1 ^ m2) < (1u << kLocalBits));  // also: any split cell
                        if (MODE == 0) {
                            sel = m0;
                            rare[q] = tie;
                        } else {
                            const int xx = (int)__builtin_amdgcn_udot4(x, x, 0u, false);
                            const uint32_t d0 = (uint32_t)(a0 + xx);
                            const uint32_t S = d0 + (uint32_t)a1 + (uint32_t)xx;
                            bool eq;
                            const bool nearest = lean_decide<MODE>(d0, S, th[q], thr.sh, eq);
                            rare[q] = tie | eq | straddle;
                            sel = nearest ? m0 : m1;
                        }
                    }
                    col[q] = *reinterpret_cast<const uint32_t *>(s_bytes + (blk[q] | ((uint32_t)sel & 0xfcu)));
                }
                uint3 wo;
                wo.x = __builtin_amdgcn_perm(col[1], col[0], 0x04020100u);
                wo.y = __builtin_amdgcn_perm(col[2], col[1], 0x05040201u);
                wo.z = __builtin_amdgcn_perm(col[3], col[2], 0x06050402u);
                out3[gidx] = wo;
            };
            if (MODE == 0) body(std::integral_constant<int, 15>{});
            else if (MODE == 3 || ADAPT) body(std::integral_constant<int, 0>{});
            else if (cls == 5u) body(std::integral_constant<int, 5>{});
            else if (cls == 10u) body(std::integral_constant<int, 10>{});
            else if (cls == 15u) body(std::integral_constant<int, 15>{});
            else body(std::integral_constant<int, 0>{});
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) rare[q] = gidx * 4u + (uint32_t)q < g.n_px;  // the partial last group
        }
        uint32_t rare_px = 0;  // wave-uniform
        if (ADAPT) {
#pragma unroll
            for (int q = 0; q < 4; ++q) rare_px += (uint32_t)__popcll(__ballot(rare[q]));
        }
        // one queue entry per group with deferred pixels: group << 4 | which of its four
        const uint32_t rmask = (rare[0] ? 1u : 0u) | (rare[1] ? 2u : 0u) | (rare[2] ? 4u : 0u) | (rare[3] ? 8u : 0u);
        const unsigned long long rb = __ballot(rmask != 0u);
        if (rb != 0ull) {  // wave-uniform
            if (rmask != 0u)
                s_queue[__builtin_amdgcn_mbcnt_hi((uint32_t)(rb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)rb, qcount))] = (gidx << 4) | rmask;
            qcount += (uint32_t)__popcll(rb);
            while (qcount >= kDrainAt) {
                const uint32_t nq = HALF ? (qcount < 64u ? qcount : 64u) : 64u;
                qcount -= nq;
                __threadfence_block();  // the group stores that are about to be overwritten
                uint32_t rest = 0u;
                if (!HALF || lane < nq) rest = lean_pixel_full<MODE, BW, WARP>(s_queue[qcount + lane], in, out, flags, g, pal, thr, smem, sx, sy, sc);
                const unsigned long long mb = __ballot(rest != 0u);  // groups with a further deferred pixel go back
                if (mb != 0ull) {
                    if (rest != 0u)
                        s_queue[__builtin_amdgcn_mbcnt_hi((uint32_t)(mb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mb, qcount))] = rest;
                    qcount += (uint32_t)__popcll(mb);
                }
            }
        }
        // adapt: more than a third of the last 256 pixels rare -> deep mode; fewer than an eighth hard -> back
        if (!ADAPT) {
        } else if (deep) {
            unsigned long long m1 = __ballot(n_hard & 1u), m2 = __ballot(n_hard & 2u), m4 = __ballot(n_hard & 4u);
            const uint32_t hard_px = (uint32_t)__popcll(m1) + 2u * (uint32_t)__popcll(m2) + 4u * (uint32_t)__popcll(m4);
            if (hard_px < 32u) deep = false;
        } else if (rare_px > 96u) {
            deep = true;
        }
    }
    while (qcount != 0u) {
        const uint32_t n = qcount < 64u ? qcount : 64u;
        qcount -= n;
        __threadfence_block();
        uint32_t rest = 0u;
        if (lane < n) rest = lean_pixel_full<MODE, BW, WARP>(s_queue[qcount + lane], in, out, flags, g, pal, thr, smem, sx, sy, sc);
        const unsigned long long mb = __ballot(rest != 0u);
        if (mb != 0ull) {
            if (rest != 0u)
                s_queue[__builtin_amdgcn_mbcnt_hi((uint32_t)(mb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mb, qcount))] = rest;
            qcount += (uint32_t)__popcll(mb);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// ordered_compact_kernel: crowded integer palettes (extracted from the image itself: median cut, k-means).
//
// Such a palette puts its colours where the pixels are, so most PIXELS sit in cells whose candidate set is larger
// than a block: the octree below the 16^3 cells is the main path, not the exception.  With 4-byte entries the octree
// of a 256-colour median-cut palette (550 nodes x 256 B) does not fit LDS next to the 4096 cell blocks, the lean
// kernel reads its deeper nodes from global memory and spends 61 % of its wave cycles waiting (round 2: 1.61 ms for
// 24 4K frames).  Here the SAME table (same cells -- plain or warped --, same node numbering, same block positions:
// host_logic.h compact_table) is held with ONE BYTE per entry, the index of the palette entry: 8 bytes per block, 64
// per node, the whole octree in 35 KB.  The palette itself sits in LDS as 256 records {colour, |p|^2 << 8 | index}
// (2 KB), so a candidate's key is TWO instructions -- v_dot4_u32_u8 (x.p) and v_mad_i32_i24 (-512 x.p + record) --
// plus one v_lshlrev_b32_sdwa for the record's address, against four in the lean kernel; the key's low byte is the
// palette index, which orders equal distances exactly as the block position does (blocks are in index order).
// Everything fits less than half of LDS: two workgroups share a CU (8 waves per SIMD) and nothing on the per-pixel
// path leaves the CU except the tie codes.  Every pixel is resolved in place (no deferral queue): cell block, descent
// while the block is a split marker, eight keys, top three, exact decision, tie codes / exception list / float64
// replay as resolve_pixel does.  A tie between the second and third candidate only matters when the decision takes
// the second one, so the tie-code words are requested for fewer pixels, and those of a lane's four pixels together.
// Compact block: two words of four index bytes; second word 0xffffffff = marker, first word then
// 0x80000000 | byte offset of the split node's eight blocks, or 0xC0000000 (a single colour with more than eight
// candidates: fix-up pass).
// ---------------------------------------------------------------------------------------------
constexpr int kCompactHalfWords = 80 * 1024 / 4;
constexpr uint32_t kCompactRecBytes = 256 * 8;                           // LDS: records at 0 ...
constexpr uint32_t kCompactLutAt = kCompactRecBytes;                     // ... the three warp maps ...
constexpr uint32_t kCompactTabAt = kCompactLutAt + kWarpLutBytes;        // ... the table, then integer thresholds

// keys of eight palette records and the three smallest (see cand8); r_k = {colour, |p|^2 << 8 | index}
__device__ __forceinline__ void cand8r(const uint32_t x, const uint2 r0, const uint2 r1, const uint2 r2, const uint2 r3,
                                       const uint2 r4, const uint2 r5, const uint2 r6, const uint2 r7, const int neg2, int &m0,
                                       int &m1, int &m2)
{
    int n0, n1, n2, n3, n4, n5, n6, n7;
    asm volatile(
        "v_dot4_u32_u8 %[n0], %[x], %[c0], 0\n\t"
        "v_dot4_u32_u8 %[n1], %[x], %[c1], 0\n\t"
        "v_dot4_u32_u8 %[n2], %[x], %[c2], 0\n\t"
        "v_dot4_u32_u8 %[n3], %[x], %[c3], 0\n\t"
        "v_dot4_u32_u8 %[n4], %[x], %[c4], 0\n\t"
        "v_dot4_u32_u8 %[n5], %[x], %[c5], 0\n\t"
        "v_dot4_u32_u8 %[n6], %[x], %[c6], 0\n\t"
        "v_dot4_u32_u8 %[n7], %[x], %[c7], 0\n\t"
        "v_mad_i32_i24 %[n0], %[n0], %[ng], %[k0]\n\t"
        "v_mad_i32_i24 %[n1], %[n1], %[ng], %[k1]\n\t"
        "v_mad_i32_i24 %[n2], %[n2], %[ng], %[k2]\n\t"
        "v_mad_i32_i24 %[n3], %[n3], %[ng], %[k3]\n\t"
        "v_mad_i32_i24 %[n4], %[n4], %[ng], %[k4]\n\t"
        "v_mad_i32_i24 %[n5], %[n5], %[ng], %[k5]\n\t"
        "v_mad_i32_i24 %[n6], %[n6], %[ng], %[k6]\n\t"
        "v_mad_i32_i24 %[n7], %[n7], %[ng], %[k7]\n\t"
        "v_min3_i32 %[m0], %[n0], %[n1], %[n2]\n\t"
        "v_med3_i32 %[m1], %[n0], %[n1], %[n2]\n\t"
        "v_max3_i32 %[m2], %[n0], %[n1], %[n2]\n\t"
        "v_med3_i32 %[m2], %[m1], %[m2], %[n3]\n\t"
        "v_med3_i32 %[m1], %[m0], %[m1], %[n3]\n\t"
        "v_min_i32 %[m0], %[m0], %[n3]\n\t"
        "v_med3_i32 %[m2], %[m1], %[m2], %[n4]\n\t"
        "v_med3_i32 %[m1], %[m0], %[m1], %[n4]\n\t"
        "v_min_i32 %[m0], %[m0], %[n4]\n\t"
        "v_med3_i32 %[m2], %[m1], %[m2], %[n5]\n\t"
        "v_med3_i32 %[m1], %[m0], %[m1], %[n5]\n\t"
        "v_min_i32 %[m0], %[m0], %[n5]\n\t"
        "v_med3_i32 %[m2], %[m1], %[m2], %[n6]\n\t"
        "v_med3_i32 %[m1], %[m0], %[m1], %[n6]\n\t"
        "v_min_i32 %[m0], %[m0], %[n6]\n\t"
        "v_med3_i32 %[m2], %[m1], %[m2], %[n7]\n\t"
        "v_med3_i32 %[m1], %[m0], %[m1], %[n7]\n\t"
        "v_min_i32 %[m0], %[m0], %[n7]\n\t"
        : [n0] "=&v"(n0), [n1] "=&v"(n1), [n2] "=&v"(n2), [n3] "=&v"(n3), [n4] "=&v"(n4), [n5] "=&v"(n5),
          [n6] "=&v"(n6), [n7] "=&v"(n7), [m0] "=&v"(m0), [m1] "=&v"(m1), [m2] "=&v"(m2)
        : [x] "v"(x), [c0] "v"(r0.x), [c1] "v"(r1.x), [c2] "v"(r2.x), [c3] "v"(r3.x), [c4] "v"(r4.x), [c5] "v"(r5.x),
          [c6] "v"(r6.x), [c7] "v"(r7.x), [k0] "v"(r0.y), [k1] "v"(r1.y), [k2] "v"(r2.y), [k3] "v"(r3.y), [k4] "v"(r4.y),
          [k5] "v"(r5.y), [k6] "v"(r6.y), [k7] "v"(r7.y), [ng] "s"(neg2));
}

// byte k of `word` times 8 (the LDS address of palette record number byte k): one SDWA shift; `three` holds 3
template <int BYTE>
__device__ __forceinline__ uint32_t rec_addr(const uint32_t word, const uint32_t three)
{
    uint32_t a;
    if (BYTE == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(a) : "v"(three), "v"(word));
    else if (BYTE == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(a) : "v"(three), "v"(word));
    else if (BYTE == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(a) : "v"(three), "v"(word));
    else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(a) : "v"(three), "v"(word));
    return a;
}

// Which of the three nearest candidates (0, 1, 2 by (distance, index)) is written: two bits per (tie code, decision).
// k=2 codes (accel.hip): 0 (c0,c1)  1 (c1,c0)  2 (c0,c2)  3 (c2,c0)  4 (c1,c2)  5 (c2,c1); field 2*code + (nearest ? 0 : 1)
constexpr uint32_t kPickTable = (0u << 0) | (1u << 2) | (1u << 4) | (0u << 6) | (0u << 8) | (2u << 10) | (2u << 12) | (0u << 14) |
                                (1u << 16) | (2u << 18) | (2u << 20) | (1u << 22);

template <int MODE, bool WARP, bool HALF>
__global__ __launch_bounds__(kCellBlock, HALF ? 8 : 4) void ordered_compact_kernel(const uint8_t *__restrict__ in,
                                                                     uint8_t *__restrict__ out,
                                                                     unsigned long long *__restrict__ flags,
                                                                     const Geo g, const PalDev pal, const ThrDev thr,
                                                                     const float sx, const float sy, const float sc,
                                                                     const uint32_t n_tiles)
{
    constexpr int kLdsWords = HALF ? kCompactHalfWords : kLeanLdsWords;
    __shared__ __align__(16) uint32_t smem[kLdsWords];
    const uint32_t thr_at = kCompactTabAt / 4 + (uint32_t)pal.comp_words;  // (words)
    for (int i = threadIdx.x; i < pal.K; i += kCellBlock) {
        const uint32_t c = pal.p4[i];
        smem[2 * i] = c;
        smem[2 * i + 1] = (__builtin_amdgcn_udot4(c, c, 0u, false) << kLocalBits) | (uint32_t)i;
    }
    if (WARP && threadIdx.x < kWarpLutBytes / 4) smem[kCompactLutAt / 4 + threadIdx.x] = reinterpret_cast<const uint32_t *>(pal.warp_lut)[threadIdx.x];
    for (int i = threadIdx.x; i < pal.comp_words; i += kCellBlock) smem[kCompactTabAt / 4 + i] = pal.comp_tab[i];
    if (MODE == 1) {
        const int n = thr.th_h * thr.tw_pad;
        for (int i = threadIdx.x; i < n; i += kCellBlock) smem[thr_at + i] = thr.mpad[i];
    }
    __syncthreads();
    const uint8_t *s_bytes = reinterpret_cast<const uint8_t *>(smem);
    const uint8_t *s_lut = s_bytes + kCompactLutAt;
    const uint8_t *s_tab = s_bytes + kCompactTabAt;
    const uint32_t lane = threadIdx.x & 63u;
    const uint3 *in3 = reinterpret_cast<const uint3 *>(in);
    uint3 *out3 = reinterpret_cast<uint3 *>(out);
    const uint32_t n_full = g.n_px >> 2;
    uint32_t three;
    asm volatile("v_mov_b32 %0, 3" : "=v"(three));  // (kept in a register: SDWA takes no inline constant here)

    // the block that decides a colour: its cell's, or the leaf reached by one colour bit per level
    auto find_block = [&](const uint32_t x, uint32_t &lo, uint32_t &hi, bool &stuck) {
        uint32_t xc = x;
        if (WARP) xc = (uint32_t)s_lut[x & 255u] | ((uint32_t)s_lut[256u + ((x >> 8) & 255u)] << 8) | ((uint32_t)s_lut[512u + (x >> 16)] << 16);
        const uint32_t t = xc & 0xf0f0f0u, y = t | (t << 12);
        uint2 b = *reinterpret_cast<const uint2 *>(s_tab + ((y >> 13) & 0x7ff8u));  // slot (r' | b'<<4 | g'<<8) x 8 bytes
        // (one exit condition: a loop with a break inside costs ~20 scalar instructions of mask bookkeeping per round)
        for (int bit = 3; bit >= 0 && b.y == 0xffffffffu && !(b.x & 0x40000000u); --bit) {
            // child number r<<2 | g<<1 | b from bit `bit` of the three coordinates, times 8 bytes: the three bits land on
            // bits 18, 17, 16 of the product (no carries: all partial products are distinct powers of two)
            const uint32_t sub8 = (__umul24((xc >> bit) & 0x010101u, 0x40201u) >> 13) & 0x38u;
            b = *reinterpret_cast<const uint2 *>(s_tab + ((b.x & 0xffffffu) | sub8));
        }
        stuck = b.y == 0xffffffffu;  // still a marker: a single colour with more than 8 candidates (or a table deeper than its colours)
        lo = b.x;
        hi = b.y;
    };
    auto keys_of = [&](const uint32_t x, const uint32_t lo, const uint32_t hi, int &m0, int &m1, int &m2) {
        const uint2 r0 = *reinterpret_cast<const uint2 *>(s_bytes + rec_addr<0>(lo, three));
        const uint2 r1 = *reinterpret_cast<const uint2 *>(s_bytes + rec_addr<1>(lo, three));
        const uint2 r2 = *reinterpret_cast<const uint2 *>(s_bytes + rec_addr<2>(lo, three));
        const uint2 r3 = *reinterpret_cast<const uint2 *>(s_bytes + rec_addr<3>(lo, three));
        const uint2 r4 = *reinterpret_cast<const uint2 *>(s_bytes + rec_addr<0>(hi, three));
        const uint2 r5 = *reinterpret_cast<const uint2 *>(s_bytes + rec_addr<1>(hi, three));
        const uint2 r6 = *reinterpret_cast<const uint2 *>(s_bytes + rec_addr<2>(hi, three));
        const uint2 r7 = *reinterpret_cast<const uint2 *>(s_bytes + rec_addr<3>(hi, three));
        cand8r(x, r0, r1, r2, r3, r4, r5, r6, r7, g.neg2, m0, m1, m2);
    };
    // decision from the three smallest keys: `nearest`, and whether the outcome depends on the order in which scipy reports
    // equidistant entries (then the colour's tie code is needed)
    auto decide = [&](const uint32_t x, const LeanThr &th, const int m0, const int m1, const int m2, bool &nearest) -> bool {
        const int a0 = m0 >> kLocalBits, a1 = m1 >> kLocalBits, a2 = m2 >> kLocalBits;
        if (MODE == 0) {
            nearest = true;
            return a0 == a1;
        }
        const int xx = (int)__builtin_amdgcn_udot4(x, x, 0u, false);
        const uint32_t d0 = (uint32_t)(a0 + xx), d1 = (uint32_t)(a1 + xx);
        bool eq;
        nearest = lean_decide<MODE>(d0, d0 + d1, th, thr.sh, eq);
        if (eq)  // the literal float64 chain decides
            nearest = ordered_use_nearest_call((double)d0, (double)d1,
                                               MODE == 1 ? __fmul_rn((float)th.mt, 1.0f / (float)(1u << thr.sh)) : th.t);
        // second and third equidistant: only the SECOND entry's identity is open, and it is written only when !nearest
        return (a0 == a1) | ((a1 == a2) & !nearest);
    };
    // the colour written for tie code `code` (0 when no tie matters); codes no pair expresses: exception list or fix-up pass
    auto pick = [&](const uint32_t x, const int m0, const int m1, const int m2, const bool nearest, const uint32_t code, bool &slow) -> uint32_t {
        uint32_t ic;
        if (MODE == 0) ic = code;
        else ic = (kPickTable >> (2u * (2u * code + (nearest ? 0u : 1u)))) & 3u;
        int sel = ic == 0u ? m0 : (ic == 1u ? m1 : m2);
        uint32_t c = *reinterpret_cast<const uint32_t *>(s_bytes + rec_addr<0>((uint32_t)sel, three));
        if (code > (MODE == 0 ? 2u : 5u)) {
            uint32_t pair, single;
            if (find_exception(pal, x, pair, single)) c = pal.out_rgb[MODE == 0 ? single : (nearest ? (pair & 0xffffu) : (pair >> 16))];
            else slow = true;
        }
        return c;
    };
    auto code_word = [&](const uint32_t xa) -> uint32_t { return MODE == 0 ? pal.code1[xa >> 4] : pal.code2[xa >> 3]; };
    auto code_in = [&](const uint32_t w, const uint32_t x) -> uint32_t {
        return MODE == 0 ? ((w >> ((x & 15u) * 2)) & 3u) : ((w >> ((x & 7u) * 4)) & 15u);
    };
    // a pixel on its own (row-straddling groups, the partial last group): position and threshold from the pixel index
    auto single_pixel = [&](const uint32_t p) -> bool {
        const uint8_t *b = in + (size_t)p * 3;
        const uint32_t x = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16);
        LeanThr th;
        th.mt = 0;
        th.t = 0.0f;
        if (MODE != 0) {
            uint32_t py, px;
            locate(g, p, py, px);
            if (MODE == 3) {
                th.t = ign_threshold(g.x0 + (int)px, g.y0 + (int)py, sx, sy, sc);
            } else {
                uint32_t row, col;
                thr_pos(thr, (uint32_t)g.y0 + py, (uint32_t)g.x0 + px, row, col);
                if (MODE == 1) th.mt = smem[thr_at + row * thr.tw_pad + col];
                else th.t = thr.fpad[row * thr.tw_pad + col];
            }
        }
        uint32_t lo, hi;
        bool stuck;
        find_block(x, lo, hi, stuck);
        bool slow = stuck;
        uint32_t c = 0;
        if (!stuck) {
            int m0, m1, m2;
            keys_of(x, lo, hi, m0, m1, m2);
            bool nearest;
            const bool need = decide(x, th, m0, m1, m2, nearest);
            const uint32_t code = need ? code_in(code_word(x), x) : 0u;
            c = pick(x, m0, m1, m2, nearest, code, slow);
        }
        uint8_t *o = out + (size_t)p * 3;
        o[0] = (uint8_t)c;
        o[1] = (uint8_t)(c >> 8);
        o[2] = (uint8_t)(c >> 16);
        return slow;
    };

    uint32_t tile = blockIdx.x;
    uint32_t fy = 0, fx = 0;
    uint3 wn = make_uint3(0u, 0u, 0u);
    if (tile < n_tiles) {
        const uint32_t gidx0 = tile * kCellBlock + threadIdx.x;
        if (gidx0 < n_full) wn = in3[gidx0];
        if (gidx0 * 4u < g.n_px) locate(g, gidx0 * 4u, fy, fx);
    }
    for (; tile < n_tiles; tile += gridDim.x) {
        const uint32_t gidx = tile * kCellBlock + threadIdx.x;
        const uint3 wc = wn;
        const uint32_t cy = fy, cx = fx;
        {
            const uint32_t next = tile + gridDim.x;
            const uint32_t gn = next * kCellBlock + threadIdx.x;
            if (next < n_tiles && gn < n_full) wn = in3[gn];  // prefetch the next tile
            fx += g.adv_x;
            fy += g.adv_y;
            if (fx >= g.w) {
                fx -= g.w;
                ++fy;
            }
            if (fy >= g.h) fy -= g.h;
        }
        bool slowq[4] = {false, false, false, false};  // pixels left to the fix-up pass (single colours with more than 8 candidates, codes no pair expresses)
        if (gidx < n_full && !((MODE != 0) && (cx + 3u >= g.w))) {
            uint32_t xq[4];
            xq[0] = wc.x & 0xffffffu;
            xq[1] = __builtin_amdgcn_perm(wc.y, wc.x, 0x0c050403u);
            xq[2] = __builtin_amdgcn_perm(wc.z, wc.y, 0x0c040302u);
            xq[3] = wc.z >> 8;
            LeanThr th[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                th[q].mt = 0;
                th[q].t = 0.0f;
            }
            if (MODE == 1 || MODE == 2) {
                uint32_t row, col;
                thr_pos(thr, (uint32_t)g.y0 + cy, (uint32_t)g.x0 + cx, row, col);
                const uint32_t at = row * (uint32_t)thr.tw_pad + col;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (MODE == 1) th[q].mt = smem[thr_at + at + q];
                    else th[q].t = thr.fpad[at + q];
                }
            } else if (MODE == 3) {
#pragma unroll
                for (int q = 0; q < 4; ++q) th[q].t = ign_threshold(g.x0 + (int)cx + q, g.y0 + (int)cy, sx, sy, sc);
            }
            uint32_t lo[4], hi[4];
            bool stuck[4], nearest[4], need[4];
            int m0[4], m1[4], m2[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) find_block(xq[q], lo[q], hi[q], stuck[q]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // (a marker block's bytes name records too: the keys of a stuck pixel are computed and not used)
                keys_of(xq[q], lo[q], hi[q], m0[q], m1[q], m2[q]);
                need[q] = decide(xq[q], th[q], m0[q], m1[q], m2[q], nearest[q]) & !stuck[q];
            }
            // the tie codes of the four pixels: requested together (one round trip), only when some lane needs one at all
            uint32_t code[4] = {0u, 0u, 0u, 0u};
            if (__ballot(need[0] | need[1] | need[2] | need[3]) != 0ull) {
                uint32_t w[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) w[q] = code_word(need[q] ? xq[q] : 0u);  // (lanes that need none read word 0: one line)
#pragma unroll
                for (int q = 0; q < 4; ++q) code[q] = need[q] ? code_in(w[q], xq[q]) : 0u;
            }
            uint32_t col[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bool slow = stuck[q];
                col[q] = pick(xq[q], m0[q], m1[q], m2[q], nearest[q], code[q], slow);
                slowq[q] = slow;
            }
            uint3 wo;
            wo.x = __builtin_amdgcn_perm(col[1], col[0], 0x04020100u);
            wo.y = __builtin_amdgcn_perm(col[2], col[1], 0x05040201u);
            wo.z = __builtin_amdgcn_perm(col[3], col[2], 0x06050402u);
            out3[gidx] = wo;
        } else {
            // a group that runs over the end of its row (its pixels' threshold positions differ) or the partial last group
#pragma unroll 1
            for (uint32_t q = 0; q < 4u; ++q) {
                if (gidx * 4u + q < g.n_px) {
                    const bool sl = single_pixel(gidx * 4u + q);
                    slowq[0] |= sl & (q == 0u);  // (no dynamic register indexing)
                    slowq[1] |= sl & (q == 1u);
                    slowq[2] |= sl & (q == 2u);
                    slowq[3] |= sl & (q == 3u);
                }
            }
        }
        // the wave tile's flag words: plain stores of the four ballots (usually zero), the tile queued for the fix-up pass if not
        if (__ballot(slowq[0] | slowq[1] | slowq[2] | slowq[3]) != 0ull)
            store_flags(flags, g.dirty, gidx, slowq);
        else if (lane < 4u)
            flags[(size_t)(gidx >> 6) * 4 + lane] = 0ull;
    }
}

// ---------------------------------------------------------------------------------------------
// ordered_fast_kernel: the lean kernel's successor for uncrowded integer palettes (plain cells).
//
// What bounds these kernels is the number of vector instructions per pixel: profiles/microbench/valu_rate*.txt
// -- every integer operation used here (v_dot4, v_mad_i32_i24, v_med3, v_perm, compares ...) issues once per ~4.3
// cycles per SIMD whatever the occupancy; only f32 add/mul/fma and v_mov reach ~2.4.  So the work is cut, not
// rescheduled:
//  * a pixel whose threshold is >= 1/2 always takes its NEAREST entry: f = s0/(s0+s1) <= 1/2 because s0 <= s1 and
//    every rounding of the reference's float64 chain is monotone (dithering_lib.py:361-376).  Such a pixel needs the
//    nearest entry and the knowledge that it is unique, nothing else.  Which entries can be nearest somewhere in a
//    cell, N(cell), is known at build time (accel.hip): the kernel stages every cell block into LDS with the members
//    of N first (at most kNearSlots8 = 6 of 8; the builder splits the ~1 % of cells with more) and runs those pixels on
//    six candidates with a top-2 network and no decision arithmetic: 35 instead of 64 vector instructions.
//  * whether a pixel slot q (pixel q of every lane's group of four) is of that kind is a property of the wave: with a
//    Bayer matrix (any size) the lanes of a wave see at most two thresholds per slot, both on the same side of 1/2,
//    so exactly half of the slots qualify; the test is one compare per slot and wave tile, the branch is scalar.
//    Blue noise and IGN mix both kinds within a slot and keep the general path (same code as the lean kernel).
//  * the table in global memory stays in palette-index order (the tie codes are defined on it); only the LDS copy is
//    permuted, so the deferred path reads blocks (and split nodes, which are no longer staged) from global memory.
// MODE as in ordered_lean_kernel.  BW = 8 or 4 (tables of small palettes: every slot is read either way).
// ---------------------------------------------------------------------------------------------
[[maybe_unused]] constexpr int kNearSlots8 = 6;

// Keys of the first six entries of a block and the two smallest (see cand8)
__device__ __forceinline__ void cand6n(const uint32_t x, const uint4 ca, const uint32_t c4, const uint32_t c5, const int neg2,
                                       int &m0, int &m1)
{
    int n0, n1, n2, n3, n4, n5, p0, p1, p2, p3, p4, p5;
    asm volatile(
        "v_dot4_u32_u8 %[n0], %[c0], %[c0], 0\n\t"
        "v_dot4_u32_u8 %[p0], %[x], %[c0], 0\n\t"
        "v_dot4_u32_u8 %[n1], %[c1], %[c1], 0\n\t"
        "v_dot4_u32_u8 %[p1], %[x], %[c1], 0\n\t"
        "v_dot4_u32_u8 %[n2], %[c2], %[c2], 0\n\t"
        "v_dot4_u32_u8 %[p2], %[x], %[c2], 0\n\t"
        "v_dot4_u32_u8 %[n3], %[c3], %[c3], 0\n\t"
        "v_dot4_u32_u8 %[p3], %[x], %[c3], 0\n\t"
        "v_dot4_u32_u8 %[n4], %[c4], %[c4], 0\n\t"
        "v_dot4_u32_u8 %[p4], %[x], %[c4], 0\n\t"
        "v_dot4_u32_u8 %[n5], %[c5], %[c5], 0\n\t"
        "v_dot4_u32_u8 %[p5], %[x], %[c5], 0\n\t"
        "v_lshl_add_u32 %[n0], %[n0], 8, 0\n\t"
        "v_lshl_add_u32 %[n1], %[n1], 8, 4\n\t"
        "v_lshl_add_u32 %[n2], %[n2], 8, 8\n\t"
        "v_lshl_add_u32 %[n3], %[n3], 8, 12\n\t"
        "v_lshl_add_u32 %[n4], %[n4], 8, 16\n\t"
        "v_lshl_add_u32 %[n5], %[n5], 8, 20\n\t"
        "v_mad_i32_i24 %[n0], %[p0], %[ng], %[n0]\n\t"
        "v_mad_i32_i24 %[n1], %[p1], %[ng], %[n1]\n\t"
        "v_mad_i32_i24 %[n2], %[p2], %[ng], %[n2]\n\t"
        "v_mad_i32_i24 %[n3], %[p3], %[ng], %[n3]\n\t"
        "v_mad_i32_i24 %[n4], %[p4], %[ng], %[n4]\n\t"
        "v_mad_i32_i24 %[n5], %[p5], %[ng], %[n5]\n\t"
        // the two smallest (m0 <= m1)
        "v_min3_i32 %[m0], %[n0], %[n1], %[n2]\n\t"
        "v_med3_i32 %[m1], %[n0], %[n1], %[n2]\n\t"
        "v_med3_i32 %[m1], %[m0], %[m1], %[n3]\n\t"
        "v_min_i32 %[m0], %[m0], %[n3]\n\t"
        "v_med3_i32 %[m1], %[m0], %[m1], %[n4]\n\t"
        "v_min_i32 %[m0], %[m0], %[n4]\n\t"
        "v_med3_i32 %[m1], %[m0], %[m1], %[n5]\n\t"
        "v_min_i32 %[m0], %[m0], %[n5]\n\t"
        : [n0] "=&v"(n0), [n1] "=&v"(n1), [n2] "=&v"(n2), [n3] "=&v"(n3), [n4] "=&v"(n4), [n5] "=&v"(n5), [p0] "=&v"(p0),
          [p1] "=&v"(p1), [p2] "=&v"(p2), [p3] "=&v"(p3), [p4] "=&v"(p4), [p5] "=&v"(p5), [m0] "=&v"(m0), [m1] "=&v"(m1)
        : [x] "v"(x), [c0] "v"(ca.x), [c1] "v"(ca.y), [c2] "v"(ca.z), [c3] "v"(ca.w), [c4] "v"(c4), [c5] "v"(c5),
          [ng] "s"(neg2));
}

// The same for the four entries of a small block
__device__ __forceinline__ void cand4n(const uint32_t x, const uint4 ca, const int neg2, int &m0, int &m1)
{
    int n0, n1, n2, n3, p0, p1, p2, p3;
    asm volatile(
        "v_dot4_u32_u8 %[n0], %[c0], %[c0], 0\n\t"
        "v_dot4_u32_u8 %[p0], %[x], %[c0], 0\n\t"
        "v_dot4_u32_u8 %[n1], %[c1], %[c1], 0\n\t"
        "v_dot4_u32_u8 %[p1], %[x], %[c1], 0\n\t"
        "v_dot4_u32_u8 %[n2], %[c2], %[c2], 0\n\t"
        "v_dot4_u32_u8 %[p2], %[x], %[c2], 0\n\t"
        "v_dot4_u32_u8 %[n3], %[c3], %[c3], 0\n\t"
        "v_dot4_u32_u8 %[p3], %[x], %[c3], 0\n\t"
        "v_lshl_add_u32 %[n0], %[n0], 8, 0\n\t"
        "v_lshl_add_u32 %[n1], %[n1], 8, 4\n\t"
        "v_lshl_add_u32 %[n2], %[n2], 8, 8\n\t"
        "v_lshl_add_u32 %[n3], %[n3], 8, 12\n\t"
        "v_mad_i32_i24 %[n0], %[p0], %[ng], %[n0]\n\t"
        "v_mad_i32_i24 %[n1], %[p1], %[ng], %[n1]\n\t"
        "v_mad_i32_i24 %[n2], %[p2], %[ng], %[n2]\n\t"
        "v_mad_i32_i24 %[n3], %[p3], %[ng], %[n3]\n\t"
        "v_min3_i32 %[m0], %[n0], %[n1], %[n2]\n\t"
        "v_med3_i32 %[m1], %[n0], %[n1], %[n2]\n\t"
        "v_med3_i32 %[m1], %[m0], %[m1], %[n3]\n\t"
        "v_min_i32 %[m0], %[m0], %[n3]\n\t"
        : [n0] "=&v"(n0), [n1] "=&v"(n1), [n2] "=&v"(n2), [n3] "=&v"(n3), [p0] "=&v"(p0), [p1] "=&v"(p1), [p2] "=&v"(p2),
          [p3] "=&v"(p3), [m0] "=&v"(m0), [m1] "=&v"(m1)
        : [x] "v"(x), [c0] "v"(ca.x), [c1] "v"(ca.y), [c2] "v"(ca.z), [c3] "v"(ca.w), [ng] "s"(neg2));
}

// the leaf of a colour, every block read from the (index-ordered) table in global memory
template <int BW>
__device__ __forceinline__ void leaf_find_global(const uint32_t x, const PalDev &pal, Leaf &lf)
{
    lf.blk = BW == 8 ? cell_offset(x) : cell_offset4(x);
    lf.in_lds = false;
    lf.stuck = false;
    lf.ca = leaf_read4(lf, pal, nullptr, lf.blk);
    lf.split = (lf.ca.x >> 31) != 0;
    for (int bit = 3; leaf_pending(lf); --bit) {
        if ((lf.ca.x & 0x40000000u) || bit < 0) {
            lf.stuck = true;
            break;
        }
        const uint32_t sub = (((x >> bit) & 1u) << 2) | (((x >> (8 + bit)) & 1u) << 1) | ((x >> (16 + bit)) & 1u);
        lf.blk = (4096u * BW + ((lf.ca.x & 0xffffffu) * 8u + sub) * BW) * 4u;
        lf.ca = leaf_read4(lf, pal, nullptr, lf.blk);
    }
}

// threshold of pixel p (MODE 1: from the integer table in LDS; MODE 2: the padded float table; MODE 3: computed)
template <int MODE>
__device__ __forceinline__ void pixel_threshold(const uint32_t p, const Geo &g, const ThrDev &thr, const uint32_t *s_thr,
                                                const float sx, const float sy, const float sc, LeanThr &th)
{
    th.mt = 0;
    th.t = 0.0f;
    if (MODE == 0) return;
    uint32_t fy, fx;
    locate(g, p, fy, fx);
    if (MODE == 3) {
        th.t = ign_threshold(g.x0 + (int)fx, g.y0 + (int)fy, sx, sy, sc);
    } else {
        uint32_t row, col;
        thr_pos(thr, (uint32_t)g.y0 + fy, (uint32_t)g.x0 + fx, row, col);
        if (MODE == 1) th.mt = s_thr[row * thr.tw_pad + col];
        else th.t = thr.fpad[row * thr.tw_pad + col];
    }
}

// Queue B of the fast kernel: one pixel p resolved completely from global memory (index-ordered table with its octree,
// tie codes, exception list, the float64 replay) -- what the few pixels need that the LDS copy cannot answer.
template <int MODE, int BW>
__device__ __forceinline__ void fast_pixel_global(const uint32_t p, const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                  unsigned long long *__restrict__ flags, const Geo &g, const PalDev &pal,
                                                  const ThrDev &thr, const uint32_t *s_thr, const float sx, const float sy,
                                                  const float sc)
{
    const uint8_t *b = in + (size_t)p * 3;
    const uint32_t x = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16);
    LeanThr th;
    pixel_threshold<MODE>(p, g, thr, s_thr, sx, sy, sc, th);
    bool slow, hard;
    Leaf lf;
    leaf_find_global<BW>(x, pal, lf);
    const uint32_t c = resolve_pixel<MODE, BW>(x, th, lf, g, pal, thr, nullptr, slow, hard);
    uint8_t *o = out + (size_t)p * 3;
    o[0] = (uint8_t)c;
    o[1] = (uint8_t)(c >> 8);
    o[2] = (uint8_t)(c >> 16);
    if (slow) flag_slow_pixel(p, flags, g);
}

// Queue A of the fast kernel: a pixel of a SPLIT cell on the cell's flat list in LDS (kWideList entries in index order,
// n_valid of them real).  Returns false -- nothing written -- when the answer needs more than that: a tie among the three
// nearest (tie codes) or exact equality in the decision (float64 replay); the caller passes the pixel on to queue B.
template <int MODE>
__device__ __forceinline__ bool resolve_wide(const uint32_t x, const LeanThr &th, const int sh, const uint32_t *s_list,
                                             const int n_valid, uint32_t &c)
{
    constexpr int kBig = 0x7fffffff;
    int m0 = kBig, m1 = kBig, m2 = kBig;
    uint32_t cj[kWideList];
#pragma unroll
    for (int j4 = 0; j4 < kWideList; j4 += 4) {
        const uint4 v = *reinterpret_cast<const uint4 *>(s_list + j4);
        cj[j4] = v.x;
        cj[j4 + 1] = v.y;
        cj[j4 + 2] = v.z;
        cj[j4 + 3] = v.w;
    }
#pragma unroll
    for (int j = 0; j < kWideList; ++j) {
        const int n = (int)__builtin_amdgcn_udot4(cj[j], cj[j], 0u, false);
        const int p = (int)__builtin_amdgcn_udot4(x, cj[j], 0u, false);
        int key = (n - 2 * p) * (1 << kLocalBits) + 4 * j;
        if (j >= n_valid) key = kBig;
        const int n2 = med3i(m1, m2, key);
        const int n1 = med3i(m0, m1, key);
        m0 = min(m0, key);
        m1 = n1;
        m2 = n2;
    }
    const int a0 = m0 >> kLocalBits, a1 = m1 >> kLocalBits, a2 = m2 >> kLocalBits;
    int sel = m0;
    if (MODE == 0) {
        if (a0 == a1) return false;
    } else {
        if (a0 == a1 || a1 == a2) return false;
        const int xx = (int)__builtin_amdgcn_udot4(x, x, 0u, false);
        const uint32_t d0 = (uint32_t)(a0 + xx);
        const uint32_t S = d0 + (uint32_t)a1 + (uint32_t)xx;
        bool eq;
        const bool nearest = lean_decide<MODE == 0 ? 1 : MODE>(d0, S, th, sh, eq);
        if (eq) return false;
        sel = nearest ? m0 : m1;
    }
    c = s_list[((uint32_t)sel & 0xfcu) >> 2];
    return true;
}

constexpr int kFastQueue = 128;  // entries per wave and queue (A and B)

template <int MODE, int BW, int DBG = 0>  // DBG (measurements only, wrong pixels): 1 = nothing deferred is resolved
__global__ __launch_bounds__(kCellBlock) void ordered_fast_kernel(const uint8_t *__restrict__ in,
                                                                  uint8_t *__restrict__ out,
                                                                  unsigned long long *__restrict__ flags,
                                                                  const Geo g, const PalDev pal, const ThrDev thr,
                                                                  const float sx, const float sy, const float sc,
                                                                  const uint32_t n_tiles)
{
    // LDS: cell blocks (nearest set first) | flat lists of the split cells | integer thresholds | queue A | queue B
    __shared__ __align__(16) uint32_t smem[kLeanLdsWords];
    constexpr int kTop = 4096 * BW;
    constexpr int kFieldBits = BW == 8 ? 3 : 2;
    constexpr int kQueueAt = kLeanLdsWords - 2 * (kCellBlock / 64) * kFastQueue;
    const int n_wide = BW == 8 ? pal.n_wide : pal.n_wide4;
    const int thr_at = kTop + n_wide * kWideList;
    {
        const uint32_t *perm = BW == 8 ? pal.cell_perm : pal.cell_perm4;
        for (int cell = threadIdx.x; cell < 4096; cell += kCellBlock) {
            const uint32_t pw = perm[cell];
            const bool split = (pw >> 24) == 0xffu;
            uint32_t w[BW];
#pragma unroll
            for (int k = 0; k < BW; ++k)
                w[k] = split ? 0u : pal.cell_tab[cell * BW + (int)((pw >> (kFieldBits * k)) & (uint32_t)(BW - 1))];
            // a split cell: equal entries (their keys tie: deferred), the last one carrying the number of its flat list
            if (split) w[BW - 1] = 0x80000000u | (pw & 0xffffffu);
            *reinterpret_cast<uint4 *>(&smem[cell * BW]) = make_uint4(w[0], w[1], w[2], w[3]);
            if (BW == 8) *reinterpret_cast<uint4 *>(&smem[cell * BW + 4]) = make_uint4(w[4 % BW], w[5 % BW], w[6 % BW], w[7 % BW]);
        }
        const uint32_t *wide = BW == 8 ? pal.cell_wide : pal.cell_wide4;
        for (int i = threadIdx.x; i < n_wide * kWideList; i += kCellBlock) smem[kTop + i] = wide[i];
    }
    if (MODE == 1) {
        const int n = thr.th_h * thr.tw_pad;
        for (int i = threadIdx.x; i < n; i += kCellBlock) smem[thr_at + i] = thr.mpad[i];
    }
    __syncthreads();
    const uint8_t *s_bytes = reinterpret_cast<const uint8_t *>(smem);
    const uint32_t *s_thr = smem + thr_at;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t *s_qa = smem + kQueueAt + (threadIdx.x >> 6) * 2 * kFastQueue;
    uint32_t *s_qb = s_qa + kFastQueue;
    uint32_t qa = 0, qb = 0;      // wave-uniform fill of the queues
    uint32_t pend_e = 0;          // queue-A entry (group << 4 | deferred pixels) whose bytes this lane has in flight (0: none)
    uint3 pend_w = make_uint3(0u, 0u, 0u);  // ... the group's twelve bytes
    bool pending = false;         // wave-uniform
    const int n_valid = pal.K < kWideList ? pal.K : kWideList;
    const uint3 *in3 = reinterpret_cast<const uint3 *>(in);
    uint3 *out3 = reinterpret_cast<uint3 *>(out);
    const uint32_t n_full = g.n_px >> 2;  // groups of four whole pixels; a partial last group goes through queue B

    // queue B: pixels that need global memory (ties, exact equality, pixels of unsplit cells in straddling groups, ...)
    auto drain_b = [&](const uint32_t n) {
        qb -= n;
        if (lane < n) fast_pixel_global<MODE, BW>(s_qb[qb + lane], in, out, flags, g, pal, thr, s_thr, sx, sy, sc);
    };
    auto push_b = [&](const bool want, const uint32_t p) {
        const unsigned long long mb = __ballot(want);
        if (mb != 0ull) {
            if (want) s_qb[__builtin_amdgcn_mbcnt_hi((uint32_t)(mb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mb, qb))] = p;
            qb += (uint32_t)__popcll(mb);
            if (qb >= 64u) {
                __threadfence_block();  // the stores of the main loop that the drain overwrites
                drain_b(64u);
            }
        }
    };
    // queue A, second half: the bytes asked for one tile ago have arrived.  Pixels of split cells are resolved on the
    // cell's flat list in LDS; whatever needs more (a tie, exact equality, an unsplit cell) goes on to queue B.
    auto finish_a = [&]() {
        pending = false;
        uint32_t mask = pend_e & 15u;
        const uint32_t group = pend_e >> 4;
        uint32_t xq[4];
        xq[0] = pend_w.x & 0xffffffu;
        xq[1] = __builtin_amdgcn_perm(pend_w.y, pend_w.x, 0x0c050403u);
        xq[2] = __builtin_amdgcn_perm(pend_w.z, pend_w.y, 0x0c040302u);
        xq[3] = pend_w.z >> 8;
        while (__ballot(mask != 0u) != 0ull) {  // wave-uniform: nearly always one round (one deferred pixel per group)
            const bool have = mask != 0u;
            const uint32_t k = (uint32_t)__builtin_ctz(mask | 16u) & 3u;
            const uint32_t p = group * 4u + k;
            bool to_b = have;
            if (have) {
                const uint32_t x = k == 0u ? xq[0] : (k == 1u ? xq[1] : (k == 2u ? xq[2] : xq[3]));
                const uint32_t blk = BW == 8 ? cell_offset(x) : cell_offset4(x);
                const uint32_t mk = *reinterpret_cast<const uint32_t *>(s_bytes + blk + (BW - 1) * 4);
                if (mk >> 31) {
                    LeanThr th;
                    pixel_threshold<MODE>(p, g, thr, s_thr, sx, sy, sc, th);
                    uint32_t c;
                    if (resolve_wide<MODE>(x, th, thr.sh, smem + kTop + (mk & 0xffffffu) * kWideList, n_valid, c)) {
                        uint8_t *o = out + (size_t)p * 3;
                        o[0] = (uint8_t)c;
                        o[1] = (uint8_t)(c >> 8);
                        o[2] = (uint8_t)(c >> 16);
                        to_b = false;
                    }
                }
            }
            push_b(to_b, p);
            mask &= mask - 1u;
        }
        pend_e = 0u;
    };
    // queue A, first half: take up to 64 entries and ask for the bytes of their groups
    auto start_a = [&]() {
        const uint32_t n = qa < 64u ? qa : 64u;
        qa -= n;
        __threadfence_block();  // the group stores that are about to be overwritten
        pend_e = 0u;
        if (lane < n) {
            pend_e = s_qa[qa + lane];
            pend_w = in3[pend_e >> 4];
        }
        pending = true;
    };

    uint32_t tile = blockIdx.x;
    uint32_t fy = 0, fx = 0;
    uint3 wn = make_uint3(0u, 0u, 0u);
    if (tile < n_tiles) {
        const uint32_t gidx0 = tile * kCellBlock + threadIdx.x;
        if (gidx0 < n_full) wn = in3[gidx0];
        if (gidx0 * 4u < g.n_px) locate(g, gidx0 * 4u, fy, fx);
    }
    // Which pixel slots of this wave tile can only take their nearest entry (thr.cls, host.cpp): a property of the position
    // of the wave's first pixel in the threshold table, valid when the 256 pixels of the wave lie in one image row.
    const bool use_cls = (MODE == 1 || MODE == 2) && thr.has_cls != 0 && !(DBG & 2);
    auto tile_class = [&](const uint32_t ty, const uint32_t tx, const uint32_t gi) -> uint32_t {
        if (MODE == 0) return 15u;
        if (!use_cls) return 0u;
        // (every lane looks at lane 0's position: uniform address, scalar load)
        const uint32_t y0l = (uint32_t)__builtin_amdgcn_readfirstlane((int)ty), x0l = (uint32_t)__builtin_amdgcn_readfirstlane((int)tx);
        const uint32_t g0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)gi);
        if (x0l + 256u > g.w || g0 + 64u > n_full) return 0u;
        const uint32_t row = ((uint32_t)g.y0 + y0l) & (uint32_t)(thr.th_h - 1), col = ((uint32_t)g.x0 + x0l) & (uint32_t)(thr.th_w - 1);
        const uint32_t idx = row * (uint32_t)thr.th_w + col;  // < 256
        return (thr.cls_nib[idx >> 3] >> ((idx & 7u) * 4u)) & 15u;
    };
    uint32_t cls_next = tile < n_tiles ? tile_class(fy, fx, tile * kCellBlock + threadIdx.x) : 0u;
    for (; tile < n_tiles; tile += gridDim.x) {
        const uint32_t gidx = tile * kCellBlock + threadIdx.x;
        const uint3 wc = wn;
        const uint32_t cls = cls_next;  // wave-uniform
        const uint32_t cy = fy, cx = fx;
        {
            const uint32_t next = tile + gridDim.x;
            const uint32_t gn = next * kCellBlock + threadIdx.x;
            if (next < n_tiles && gn < n_full) wn = in3[gn];  // prefetch the next tile
            // ... and its position and class
            fx += g.adv_x;
            fy += g.adv_y;
            if (fx >= g.w) {
                fx -= g.w;
                ++fy;
            }
            if (fy >= g.h) fy -= g.h;
            cls_next = next < n_tiles ? tile_class(fy, fx, gn) : 0u;
        }
        // all clear; the deferred paths OR in the bits of the pixels they leave to the fix-up pass later
        if (lane < 4u) flags[(size_t)(gidx >> 6) * 4 + lane] = 0ull;
        bool rare[4] = {false, false, false, false};
        if (gidx < n_full) {
            uint32_t xq[4];
            xq[0] = wc.x & 0xffffffu;
            xq[1] = __builtin_amdgcn_perm(wc.y, wc.x, 0x0c050403u);
            xq[2] = __builtin_amdgcn_perm(wc.z, wc.y, 0x0c040302u);
            xq[3] = wc.z >> 8;
            uint32_t blk[4];
            uint4 ca[4], cb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                blk[q] = BW == 8 ? cell_offset(xq[q]) : cell_offset4(xq[q]);
                ca[q] = *reinterpret_cast<const uint4 *>(s_bytes + blk[q]);
                if (BW == 8) {
                    if (cls & (1u << q)) {  // scalar branch: the nearest-only path reads six entries
                        const uint2 v = *reinterpret_cast<const uint2 *>(s_bytes + blk[q] + 16);
                        cb[q] = make_uint4(v.x, v.y, 0u, 0u);
                    } else {
                        cb[q] = *reinterpret_cast<const uint4 *>(s_bytes + blk[q] + 16);
                    }
                }
            }
            LeanThr th[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                th[q].mt = 0;
                th[q].t = 0.0f;
            }
            if (cls != 15u) {
                if (MODE == 1 || MODE == 2) {
                    uint32_t row, col;
                    thr_pos(thr, (uint32_t)g.y0 + cy, (uint32_t)g.x0 + cx, row, col);
                    const uint32_t at = row * (uint32_t)thr.tw_pad + col;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (MODE == 1) th[q].mt = s_thr[at + q];
                        else th[q].t = thr.fpad[at + q];
                    }
                } else if (MODE == 3) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) th[q].t = ign_threshold(g.x0 + (int)cx + q, g.y0 + (int)cy, sx, sy, sc);
                }
            }
            // the group runs over the end of its row: all four through the queue (their positions differ)
            const bool straddle = (MODE != 0) && (cx + 3u >= g.w);
            uint32_t col[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t x = xq[q];
                int sel;
                if (cls & (1u << q)) {
                    int m0, m1;
                    if (BW == 8) {
                        cand6n(x, ca[q], cb[q].x, cb[q].y, g.neg2, m0, m1);
                    } else {
                        int m2;
                        cand4(x, ca[q], g.neg2, m0, m1, m2);  // (all four entries: a split cell's marker must meet two equal keys)
                        m1 = ((uint32_t)(m1 ^ m2) < (1u << kLocalBits)) ? m0 : m1;
                    }
                    rare[q] = ((uint32_t)(m0 ^ m1) < (1u << kLocalBits)) | straddle;  // also: any split cell
                    sel = m0;
                } else {
                    int m0, m1, m2;
                    if (BW == 8) cand8(x, ca[q], cb[q], g.neg2, m0, m1, m2);
                    else cand4(x, ca[q], g.neg2, m0, m1, m2);
                    const int a0 = m0 >> kLocalBits, a1 = m1 >> kLocalBits;
                    const bool tie = (a0 == a1) | ((uint32_t)(m1 ^ m2) < (1u << kLocalBits));  // also: any split cell
                    const int xx = (int)__builtin_amdgcn_udot4(x, x, 0u, false);
                    const uint32_t d0 = (uint32_t)(a0 + xx);
                    const uint32_t S = d0 + (uint32_t)a1 + (uint32_t)xx;
                    bool eq;
                    const bool nearest = lean_decide<MODE == 0 ? 1 : MODE>(d0, S, th[q], thr.sh, eq);
                    rare[q] = tie | eq | straddle;
                    sel = nearest ? m0 : m1;
                }
                col[q] = *reinterpret_cast<const uint32_t *>(s_bytes + (blk[q] | ((uint32_t)sel & 0xfcu)));
            }
            uint3 wo;
            wo.x = __builtin_amdgcn_perm(col[1], col[0], 0x04020100u);
            wo.y = __builtin_amdgcn_perm(col[2], col[1], 0x05040201u);
            wo.z = __builtin_amdgcn_perm(col[3], col[2], 0x06050402u);
            out3[gidx] = wo;
        }
        // one queue entry per group with deferred pixels: group << 4 | which of its four
        const uint32_t rmask = (rare[0] ? 1u : 0u) | (rare[1] ? 2u : 0u) | (rare[2] ? 4u : 0u) | (rare[3] ? 8u : 0u);
        const unsigned long long rb = __ballot(rmask != 0u);
        if (rb != 0ull) {  // wave-uniform
            if (rmask != 0u)
                s_qa[__builtin_amdgcn_mbcnt_hi((uint32_t)(rb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)rb, qa))] = (gidx << 4) | rmask;
            qa += (uint32_t)__popcll(rb);
        }
        // the partial last group (its twelve bytes may not all exist): pixel by pixel through queue B
        if ((tile + 1u) * kCellBlock > n_full && !(DBG & 1)) {  // scalar: the last tile of the launch only
#pragma unroll
            for (int q = 0; q < 4; ++q) push_b(gidx == n_full && gidx * 4u + (uint32_t)q < g.n_px, gidx * 4u + (uint32_t)q);
        }
        if (DBG & 1) {
            qa = 0u;
        } else {
            if (pending) finish_a();
            if (qa >= 64u) start_a();
        }
    }
    if (!(DBG & 1)) {
        while (pending || qa != 0u) {
            if (pending) finish_a();
            if (qa != 0u) start_a();
        }
        while (qb != 0u) {
            __threadfence_block();
            drain_b(qb < 64u ? qb : 64u);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Float (gamma) palettes with a cell table (accel.hip, build_accel_float): blocks of 8 byte offsets into
// the candidate table {x, y, z, out_rgb} held in LDS next to the cell table and lut_in.  Candidates are
// ranked in float32 (key = distance bits with the block position in the low 3 bits); the ranking is
// accepted only when neighbouring keys differ by more than 128 ulp -- the lists hold every entry within
// 2^-14 of the second distance, the float32 evaluation is off by < 1.5e-6 -- and then the two winners'
// distances are recomputed in float64 exactly as scipy does and the reference's float64 chain decides.
// Anything else (near ties, a single colour with more than 8 candidates, groups straddling a row end)
// is flagged for the fix-up pass.  MODE: 0 nearest only, 2 matrix (float32 thresholds), 3 IGN.
// ---------------------------------------------------------------------------------------------
constexpr int kFloatKeyGap = 128;

struct alignas(16) Cand3 {  // the coordinates of a candidate record: one ds_read_b96
    float x, y, z;
};

__device__ __forceinline__ int med3_i32(const int a, const int b, const int c)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <int MODE>
__global__ __launch_bounds__(kCellBlock) void ordered_lean_float_kernel(const uint8_t *__restrict__ in,
                                                                        uint8_t *__restrict__ out,
                                                                        unsigned long long *__restrict__ flags,
                                                                        const Geo g, const PalDev pal, const ThrDev thr,
                                                                        const float sx, const float sy, const float sc,
                                                                        const uint32_t n_tiles)
{
    __shared__ __align__(16) uint32_t smem[kLeanLdsWords];
    const uint32_t cand_base = (uint32_t)pal.ftab_words * 4u;  // bytes
    // table words are byte offsets into the candidate table: make them absolute LDS addresses while copying
    // (markers have bit 31 set and stay as they are)
    for (int i = threadIdx.x * 4; i < pal.ftab_words; i += kCellBlock * 4) {
        uint4 v = *reinterpret_cast<const uint4 *>(&pal.ftab[i]);
        v.x += (v.x >> 31) ? 0u : cand_base;
        v.y += (v.y >> 31) ? 0u : cand_base;
        v.z += (v.z >> 31) ? 0u : cand_base;
        v.w += (v.w >> 31) ? 0u : cand_base;
        *reinterpret_cast<uint4 *>(&smem[i]) = v;
    }
    const uint32_t lut_base = cand_base + (uint32_t)pal.K * 16u;
    for (int i = threadIdx.x; i < pal.K; i += kCellBlock)
        *reinterpret_cast<float4 *>(&smem[pal.ftab_words + 4 * i]) = pal.fcand[i];
    if (threadIdx.x < 256)
        reinterpret_cast<uint8_t *>(smem)[lut_base + threadIdx.x] = pal.lut_in ? pal.lut_in[threadIdx.x] : (uint8_t)threadIdx.x;
    __syncthreads();
    const uint8_t *s_bytes = reinterpret_cast<const uint8_t *>(smem);
    const uint8_t *s_lut = s_bytes + lut_base;
    const uint32_t lane = threadIdx.x & 63u;
    const uint3 *in3 = reinterpret_cast<const uint3 *>(in);
    uint3 *out3 = reinterpret_cast<uint3 *>(out);
    const uint32_t n_full = g.n_px >> 2;

    uint32_t tile = blockIdx.x;
    uint32_t fy = 0, fx = 0;
    uint3 wn = make_uint3(0u, 0u, 0u);
    if (tile < n_tiles) {
        const uint32_t gidx0 = tile * kCellBlock + threadIdx.x;
        if (gidx0 < n_full) wn = in3[gidx0];
        if (gidx0 * 4u < g.n_px) locate(g, gidx0 * 4u, fy, fx);
    }
    for (; tile < n_tiles; tile += gridDim.x) {
        const uint32_t gidx = tile * kCellBlock + threadIdx.x;
        const uint3 wc = wn;
        {
            const uint32_t next = tile + gridDim.x;
            const uint32_t gn = next * kCellBlock + threadIdx.x;
            if (next < n_tiles && gn < n_full) wn = in3[gn];  // prefetch the next tile
        }
        bool slow[4] = {false, false, false, false};
        if (gidx < n_full) {
            uint32_t xs[4];
            xs[0] = wc.x & 0xffffffu;
            xs[1] = __builtin_amdgcn_perm(wc.y, wc.x, 0x0c050403u);
            xs[2] = __builtin_amdgcn_perm(wc.z, wc.y, 0x0c040302u);
            xs[3] = wc.z >> 8;
            float tq[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (MODE == 2) {
                uint32_t row, col;
                thr_pos(thr, (uint32_t)g.y0 + fy, (uint32_t)g.x0 + fx, row, col);
                const uint32_t at = row * (uint32_t)thr.tw_pad + col;
#pragma unroll
                for (int q = 0; q < 4; ++q) tq[q] = thr.fpad[at + q];
            } else if (MODE == 3) {
#pragma unroll
                for (int q = 0; q < 4; ++q) tq[q] = ign_threshold(g.x0 + (int)fx + q, g.y0 + (int)fy, sx, sy, sc);
            }
            const bool straddle = (MODE != 0) && (fx + 3u >= g.w);
            uint32_t col[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t r = s_lut[xs[q] & 255u], gg = s_lut[(xs[q] >> 8) & 255u], b = s_lut[xs[q] >> 16];
                const uint32_t x = r | (gg << 8) | (b << 16);
                uint32_t blk = cell_offset(x);
                uint4 ca = *reinterpret_cast<const uint4 *>(s_bytes + blk);
                bool s = straddle;
                // split cells: descend by one colour bit per level; nodes beyond the staged part of the table (clustered
                // palettes) are read from global memory, where the offsets are still relative to the candidate table
                bool from_global = false;
                for (int bit = 3; (ca.x >> 31) != 0; --bit) {
                    if ((ca.x & 0x40000000u) || bit < 0) {
                        s = true;  // a single colour with more than 8 candidates: fix-up pass
                        ca = make_uint4(cand_base, cand_base, cand_base, cand_base);  // any valid entry
                        from_global = false;
                        blk = 0u;
                        break;
                    }
                    const uint32_t sub = (((x >> bit) & 1u) << 2) | (((x >> (8 + bit)) & 1u) << 1) | ((x >> (16 + bit)) & 1u);
                    blk = (4096u * 8u + ((ca.x & 0xffffffu) * 8u + sub) * 8u) * 4u;
                    from_global = blk >= cand_base;
                    ca = from_global ? *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(pal.ftab) + blk)
                                     : *reinterpret_cast<const uint4 *>(s_bytes + blk);
                }
                uint4 cb;
                if (from_global) {
                    cb = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(pal.ftab) + blk + 16);
                    ca.x += cand_base;  // a leaf block: eight offsets
                    ca.y += cand_base;
                    ca.z += cand_base;
                    ca.w += cand_base;
                    cb.x += cand_base;
                    cb.y += cand_base;
                    cb.z += cand_base;
                    cb.w += cand_base;
                } else {
                    cb = *reinterpret_cast<const uint4 *>(s_bytes + blk + 16);
                    if (s && (ca.x == cand_base)) cb = ca;  // the block of a slow marker holds no offsets
                }
                const uint32_t off[8] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
                const float fr = (float)r, fg = (float)gg, fb = (float)b;
                int key[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    // the whole 16-byte record: ds_read_b128 takes 4 LDS cycles per wave, ds_read_b96 takes 8
                    // (MI355X_MICROARCH.md, LDS); measured: no difference, the kernel is bound by its 228 vector instructions
                    const float4 cc = *reinterpret_cast<const float4 *>(s_bytes + off[c]);
                    const float dx = cc.x - fr, dy = cc.y - fg, dz = cc.z - fb;
                    const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                    key[c] = (int)((__float_as_uint(d) & ~7u) | (uint32_t)c);
                }
                int m0 = min(min(key[0], key[1]), key[2]);
                int m1 = med3_i32(key[0], key[1], key[2]);
                int m2 = max(max(key[0], key[1]), key[2]);
#pragma unroll
                for (int c = 3; c < 8; ++c) {
                    m2 = med3_i32(m1, m2, key[c]);
                    m1 = med3_i32(m0, m1, key[c]);
                    m0 = min(m0, key[c]);
                }
                s |= (m1 - m0) <= kFloatKeyGap;
                uint32_t o0 = off[0], o1 = off[0];
#pragma unroll
                for (int c = 1; c < 8; ++c) {
                    o0 = ((uint32_t)m0 & 7u) == (uint32_t)c ? off[c] : o0;
                    o1 = ((uint32_t)m1 & 7u) == (uint32_t)c ? off[c] : o1;
                }
                const uint32_t a0 = s ? cand_base : o0;
                uint32_t cpick = *reinterpret_cast<const uint32_t *>(s_bytes + a0 + 12);  // the winner's output colour
                if (MODE != 0) {
                    s |= (m2 - m1) <= kFloatKeyGap;
                    const uint32_t a1 = s ? cand_base : o1;
                    const uint32_t cnext = *reinterpret_cast<const uint32_t *>(s_bytes + a1 + 12);
                    // The decision s0/(s0+s1) <= t in float32 first: the keys hold the two distances to within 8 ulp on
                    // top of the < 1.5e-6 of their float32 evaluation, so the quotient (<= 0.5) is off by < 3e-6 absolute
                    // while the reference's float64 chain is off by < 1e-15; a gap of more than 2e-5 to the threshold
                    // settles it.  Only pixels closer than that (about one in 10^4) replay the float64 chain.
                    const float s0f = __uint_as_float((uint32_t)m0 & ~7u), s1f = __uint_as_float((uint32_t)m1 & ~7u);
                    const float gap = __fdividef(s0f, s0f + s1f) - tq[q];
                    bool use_nearest = gap <= 0.0f;
                    if (fabsf(gap) <= 2e-5f && !s) {
                        const float4 c0 = *reinterpret_cast<const float4 *>(s_bytes + a0);
                        const float4 c1 = *reinterpret_cast<const float4 *>(s_bytes + a1);
                        const double p0[3] = {(double)c0.x, (double)c0.y, (double)c0.z};
                        const double p1[3] = {(double)c1.x, (double)c1.y, (double)c1.z};
                        const double d0 = sq_dist3(p0, (double)r, (double)gg, (double)b);
                        const double d1 = sq_dist3(p1, (double)r, (double)gg, (double)b);
                        use_nearest = ordered_use_nearest(d0, d1, tq[q]);
                    }
                    if (!use_nearest) cpick = cnext;
                }
                slow[q] = s;
                col[q] = cpick;
            }
            uint3 wo;
            wo.x = __builtin_amdgcn_perm(col[1], col[0], 0x04020100u);
            wo.y = __builtin_amdgcn_perm(col[2], col[1], 0x05040201u);
            wo.z = __builtin_amdgcn_perm(col[3], col[2], 0x06050402u);
            out3[gidx] = wo;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) slow[q] = gidx * 4u + (uint32_t)q < g.n_px;  // the partial last group
        }
        if (__ballot(slow[0] | slow[1] | slow[2] | slow[3]) != 0ull)
            store_flags(flags, g.dirty, gidx, slow);
        else if (lane < 4u)
            flags[(size_t)(gidx >> 6) * 4 + lane] = 0ull;
        fx += g.adv_x;
        fy += g.adv_y;
        if (fx >= g.w) {
            fx -= g.w;
            ++fy;
        }
        if (fy >= g.h) fy -= g.h;
    }
}

// ---------------------------------------------------------------------------------------------
// ordered_compact_float_kernel: float (gamma) palettes on the one-byte-per-entry table (round 3).
// The float table of ordered_lean_float_kernel with every byte offset replaced by the record's index (accel.hip,
// build_accel_float -> compact_table): 8 bytes per block, the 256 records {x, y, z, out_rgb} at LDS address 0 (4 KB), lut_in
// behind them -- for an uncrowded 256-colour palette 38 KB in all.  Per pixel: three lut
// reads, one ds_read_b64 (block), descent while it is a marker, eight record addresses by v_lshlrev_b32_sdwa and eight
// ds_read_b128, the float32 keys (distance bits | position) and top-3 network of the lean float kernel with its
// margins; the winners' records are found again through the block's index bytes (v_perm_b32 by position) instead of a
// 28-instruction select chain over the eight offsets, and the float64 replay of the decision sits behind a wave-uniform
// branch instead of being if-converted into every pixel's path; and a near tie of the float32 keys is no longer a case for
// the fix-up pass: the same rare branch ranks the block's eight candidates in float64 (compact_float_rare), only exact
// float64 ties remain flagged -- the fix-up pass of the 24-frame batch drops from 0.12 ms to 0.006.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kCfRecBytes = 256 * 16;
constexpr uint32_t kCfLutAt = kCfRecBytes;
constexpr uint32_t kCfTabAt = kCfLutAt + 256;

// The rare pixels of ordered_compact_float_kernel, out of line so that their float64 arithmetic does not take registers
// from the main loop (a call about once per 50 tiles).  what = 1: the float64 chain decides between the two winners
// (win = i0 | i1 << 8); what = 2: neighbouring float32 keys were too close to rank -- the block's eight candidates again in
// float64 exactly as scipy evaluates them (sq_dist3): distinct float64 distances order uniquely; an exact float64 tie for
// first place, or for second when the second entry matters, depends on scipy's traversal order and goes to the fix-up pass
// (bit 31 of the result).  rec: LDS address of the candidate records.  x: the pixel after lut_in.
__device__ __noinline__ uint32_t compact_float_rare(const uint32_t what, const uint32_t x, const uint32_t blo, const uint32_t bhi,
                                                    const uint32_t win, const float t, const int mode, const uint32_t rec)
{
    typedef const __attribute__((address_space(3))) float lds_f32_t;
    auto record = [&](const uint32_t j, double p[3]) -> uint32_t {
        lds_f32_t *r = (lds_f32_t *)(uintptr_t)(rec + (j << 4));
        p[0] = (double)r[0];
        p[1] = (double)r[1];
        p[2] = (double)r[2];
        return __float_as_uint(r[3]);
    };
    const double xr = (double)(x & 255u), xg = (double)((x >> 8) & 255u), xb = (double)(x >> 16);
    uint32_t j0 = win & 255u, j1 = win >> 8;
    bool tie = false;
    if (what == 2u) {
        double b0 = __longlong_as_double(0x7ff0000000000000LL), b1 = b0, b2 = b0;
#pragma unroll 1
        for (int c = 0; c < 8; ++c) {
            const uint32_t jc = ((c < 4 ? blo : bhi) >> (8 * (c & 3))) & 255u;
            double pc[3];
            (void)record(jc, pc);
            const double d = sq_dist3(pc, xr, xg, xb);
            if (d < b0) {
                b2 = b1;
                b1 = b0;
                j1 = j0;
                b0 = d;
                j0 = jc;
            } else if (d < b1) {
                b2 = b1;
                b1 = d;
                j1 = jc;
            } else if (d < b2) {
                b2 = d;
            }
        }
        tie = (b0 == b1) || (mode != 0 && b1 == b2);
    }
    double p0[3], p1[3];
    const uint32_t o0 = record(j0, p0), o1 = record(j1, p1);
    bool use_nearest = true;
    if (mode != 0) use_nearest = ordered_use_nearest(sq_dist3(p0, xr, xg, xb), sq_dist3(p1, xr, xg, xb), t);
    return (use_nearest ? o0 : o1) | (tie ? 0x80000000u : 0u);
}

template <int MODE>
__global__ __launch_bounds__(kCellBlock) void ordered_compact_float_kernel(const uint8_t *__restrict__ in,
                                                                                         uint8_t *__restrict__ out,
                                                                                         unsigned long long *__restrict__ flags,
                                                                                         const Geo g, const PalDev pal, const ThrDev thr,
                                                                                         const float sx, const float sy, const float sc,
                                                                                         const uint32_t n_tiles)
{
    // One workgroup per CU.  Two of 1024 lanes (64 registers per lane) spill in the main loop and measured 10-50 % slower; three
    // of 512 lanes (80 registers, 6 waves per SIMD, no spills) measured the same as one of 1024: the kernel is bound by its
    // instruction and LDS throughput, not by latency.
    __shared__ __align__(16) uint32_t smem[kLeanLdsWords];
    for (int i = threadIdx.x; i < pal.K; i += kCellBlock) *reinterpret_cast<float4 *>(&smem[4 * i]) = pal.fcand[i];
    if (threadIdx.x < 256)
        reinterpret_cast<uint8_t *>(smem)[kCfLutAt + threadIdx.x] = pal.lut_in ? pal.lut_in[threadIdx.x] : (uint8_t)threadIdx.x;
    for (int i = threadIdx.x; i < pal.comp_words; i += kCellBlock) smem[kCfTabAt / 4 + i] = pal.comp_tab[i];
    // MODE 2: the padded float32 threshold table behind the cell table when it fits (every Bayer / blue-noise size does)
    const uint32_t thr_at = kCfTabAt / 4 + (uint32_t)pal.comp_words;
    const bool thr_lds = MODE == 2 && (size_t)thr_at + (size_t)thr.th_h * thr.tw_pad <= (size_t)kLeanLdsWords;
    if (thr_lds) {
        const int n = thr.th_h * thr.tw_pad;
        for (int i = threadIdx.x; i < n; i += kCellBlock) smem[thr_at + i] = __float_as_uint(thr.fpad[i]);
    }
    __syncthreads();
    const uint8_t *s_bytes = reinterpret_cast<const uint8_t *>(smem);
    const uint8_t *s_lut = s_bytes + kCfLutAt;
    const uint8_t *s_tab = s_bytes + kCfTabAt;
    const uint32_t lane = threadIdx.x & 63u;
    const uint3 *in3 = reinterpret_cast<const uint3 *>(in);
    uint3 *out3 = reinterpret_cast<uint3 *>(out);
    const uint32_t n_full = g.n_px >> 2;
    uint32_t four;
    asm volatile("v_mov_b32 %0, 4" : "=v"(four));

    uint32_t tile = blockIdx.x;
    uint32_t fy = 0, fx = 0;
    uint3 wn = make_uint3(0u, 0u, 0u);
    if (tile < n_tiles) {
        const uint32_t gidx0 = tile * kCellBlock + threadIdx.x;
        if (gidx0 < n_full) wn = in3[gidx0];
        if (gidx0 * 4u < g.n_px) locate(g, gidx0 * 4u, fy, fx);
    }
    for (; tile < n_tiles; tile += gridDim.x) {
        const uint32_t gidx = tile * kCellBlock + threadIdx.x;
        const uint3 wc = wn;
        {
            const uint32_t next = tile + gridDim.x;
            const uint32_t gn = next * kCellBlock + threadIdx.x;
            if (next < n_tiles && gn < n_full) wn = in3[gn];  // prefetch the next tile
        }
        bool slow[4] = {false, false, false, false};
        if (gidx < n_full) {
            uint32_t xs[4];
            xs[0] = wc.x & 0xffffffu;
            xs[1] = __builtin_amdgcn_perm(wc.y, wc.x, 0x0c050403u);
            xs[2] = __builtin_amdgcn_perm(wc.z, wc.y, 0x0c040302u);
            xs[3] = wc.z >> 8;
            float tq[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (MODE == 2) {
                uint32_t row, col;
                thr_pos(thr, (uint32_t)g.y0 + fy, (uint32_t)g.x0 + fx, row, col);
                const uint32_t at = row * (uint32_t)thr.tw_pad + col;
#pragma unroll
                for (int q = 0; q < 4; ++q) tq[q] = thr_lds ? __uint_as_float(smem[thr_at + at + q]) : thr.fpad[at + q];
            } else if (MODE == 3) {
#pragma unroll
                for (int q = 0; q < 4; ++q) tq[q] = ign_threshold(g.x0 + (int)fx + q, g.y0 + (int)fy, sx, sy, sc);
            }
            const bool straddle = (MODE != 0) && (fx + 3u >= g.w);
            uint32_t col[4], blo[4], bhi[4], win[4];  // win: index bytes of the two winners (i0 | i1 << 8)
            // what is left for the rare branch below: 1 = the float64 chain decides between the two winners,
            // 2 = neighbouring float32 keys too close to rank: all eight candidates again in float64
            uint32_t rare[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t r = s_lut[xs[q] & 255u], gg = s_lut[(xs[q] >> 8) & 255u], b = s_lut[xs[q] >> 16];
                const uint32_t x = r | (gg << 8) | (b << 16);
                const uint32_t t = x & 0xf0f0f0u, y = t | (t << 12);
                uint2 blk = *reinterpret_cast<const uint2 *>(s_tab + ((y >> 13) & 0x7ff8u));
                bool s = straddle;
                for (int bit = 3; bit >= 0 && blk.y == 0xffffffffu && !(blk.x & 0x40000000u); --bit) {  // (one exit condition: see ordered_compact_kernel)
                    const uint32_t sub8 = (__umul24((x >> bit) & 0x010101u, 0x40201u) >> 13) & 0x38u;
                    blk = *reinterpret_cast<const uint2 *>(s_tab + ((blk.x & 0xffffffu) | sub8));
                }
                if (blk.y == 0xffffffffu) {  // still a marker: a single colour with more than 8 candidates -- fix-up pass
                    s = true;
                    blk = make_uint2(0u, 0u);  // any valid entries
                }
                blo[q] = blk.x;
                bhi[q] = blk.y;
                const float4 c0 = *reinterpret_cast<const float4 *>(s_bytes + rec_addr<0>(blk.x, four));
                const float4 c1 = *reinterpret_cast<const float4 *>(s_bytes + rec_addr<1>(blk.x, four));
                const float4 c2 = *reinterpret_cast<const float4 *>(s_bytes + rec_addr<2>(blk.x, four));
                const float4 c3 = *reinterpret_cast<const float4 *>(s_bytes + rec_addr<3>(blk.x, four));
                const float4 c4 = *reinterpret_cast<const float4 *>(s_bytes + rec_addr<0>(blk.y, four));
                const float4 c5 = *reinterpret_cast<const float4 *>(s_bytes + rec_addr<1>(blk.y, four));
                const float4 c6 = *reinterpret_cast<const float4 *>(s_bytes + rec_addr<2>(blk.y, four));
                const float4 c7 = *reinterpret_cast<const float4 *>(s_bytes + rec_addr<3>(blk.y, four));
                const float fr = (float)r, fg = (float)gg, fb = (float)b;
                int key[8];
                auto keyof = [&](const float4 cc, const uint32_t c) -> int {
                    // (naming the fourth word keeps the read a ds_read_b128: 4 LDS cycles per wave in groups of 16 lanes over 64
                    // banks, where the ds_read_b96 the compiler would narrow it to takes 8 in groups of 8 over 32)
                    asm volatile("" ::"v"(cc.w));
                    const float dx = cc.x - fr, dy = cc.y - fg, dz = cc.z - fb;
                    const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                    return (int)((__float_as_uint(d) & ~7u) | c);
                };
                key[0] = keyof(c0, 0u);
                key[1] = keyof(c1, 1u);
                key[2] = keyof(c2, 2u);
                key[3] = keyof(c3, 3u);
                key[4] = keyof(c4, 4u);
                key[5] = keyof(c5, 5u);
                key[6] = keyof(c6, 6u);
                key[7] = keyof(c7, 7u);
                int m0 = min(min(key[0], key[1]), key[2]);
                int m1 = med3_i32(key[0], key[1], key[2]);
                int m2 = max(max(key[0], key[1]), key[2]);
#pragma unroll
                for (int c = 3; c < 8; ++c) {
                    m2 = med3_i32(m1, m2, key[c]);
                    m1 = med3_i32(m0, m1, key[c]);
                    m0 = min(m0, key[c]);
                }
                bool close = (m1 - m0) <= kFloatKeyGap;
                if (MODE != 0) close |= (m2 - m1) <= kFloatKeyGap;
                // the winners' records: index byte number (key & 7) of the block
                const uint32_t i0 = __builtin_amdgcn_perm(blk.y, blk.x, 0x0c0c0c00u | ((uint32_t)m0 & 7u));
                const uint32_t i1 = __builtin_amdgcn_perm(blk.y, blk.x, 0x0c0c0c00u | ((uint32_t)m1 & 7u));
                win[q] = i0 | (i1 << 8);
                uint32_t cpick = *reinterpret_cast<const uint32_t *>(s_bytes + (i0 << 4) + 12);  // the winner's output colour
                uint32_t what = (close && !s) ? 2u : 0u;
                if (MODE != 0) {
                    const uint32_t cnext = *reinterpret_cast<const uint32_t *>(s_bytes + (i1 << 4) + 12);
                    // the decision s0/(s0+s1) <= t in float32 first (see ordered_lean_float_kernel): a gap of more than 2e-5
                    // to the threshold settles it; closer pixels (about one in 10^4) replay the float64 chain below
                    const float s0f = __uint_as_float((uint32_t)m0 & ~7u), s1f = __uint_as_float((uint32_t)m1 & ~7u);
                    // s0/(s0+s1) - t without the division: (s0 - t S) / S, compared with a margin of 2e-5 S
                    const float S = s0f + s1f;
                    const float gap = fmaf(-tq[q], S, s0f);
                    if (what == 0u && !s && fabsf(gap) <= 2e-5f * S) what = 1u;
                    if (!(gap <= 0.0f)) cpick = cnext;
                }
                rare[q] = what;
                slow[q] = s;
                col[q] = cpick;
            }
            if (__ballot((rare[0] | rare[1] | rare[2] | rare[3]) != 0u) != 0ull) {  // wave-uniform: about one tile in 50
                const uint32_t rec = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)smem;
#pragma unroll 1
                for (int q = 0; q < 4; ++q) {
                    // (dynamic indexing of the per-pixel registers would spill them: select by compare)
                    const uint32_t what = q == 0 ? rare[0] : q == 1 ? rare[1] : q == 2 ? rare[2] : rare[3];
                    if (what != 0u) {
                        const uint32_t xv = q == 0 ? xs[0] : q == 1 ? xs[1] : q == 2 ? xs[2] : xs[3];
                        const uint32_t x = (uint32_t)s_lut[xv & 255u] | ((uint32_t)s_lut[(xv >> 8) & 255u] << 8) | ((uint32_t)s_lut[xv >> 16] << 16);
                        const uint32_t res = compact_float_rare(what, x, q == 0 ? blo[0] : q == 1 ? blo[1] : q == 2 ? blo[2] : blo[3],
                                                                q == 0 ? bhi[0] : q == 1 ? bhi[1] : q == 2 ? bhi[2] : bhi[3],
                                                                q == 0 ? win[0] : q == 1 ? win[1] : q == 2 ? win[2] : win[3],
                                                                q == 0 ? tq[0] : q == 1 ? tq[1] : q == 2 ? tq[2] : tq[3], MODE, rec);
                        const uint32_t c = res & 0xffffffu;
                        const bool tie = (res >> 31) != 0u;
                        if (q == 0) { col[0] = c; slow[0] |= tie; }
                        else if (q == 1) { col[1] = c; slow[1] |= tie; }
                        else if (q == 2) { col[2] = c; slow[2] |= tie; }
                        else { col[3] = c; slow[3] |= tie; }
                    }
                }
            }
            uint3 wo;
            wo.x = __builtin_amdgcn_perm(col[1], col[0], 0x04020100u);
            wo.y = __builtin_amdgcn_perm(col[2], col[1], 0x05040201u);
            wo.z = __builtin_amdgcn_perm(col[3], col[2], 0x06050402u);
            out3[gidx] = wo;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) slow[q] = gidx * 4u + (uint32_t)q < g.n_px;  // the partial last group
        }
        if (__ballot(slow[0] | slow[1] | slow[2] | slow[3]) != 0ull)
            store_flags(flags, g.dirty, gidx, slow);
        else if (lane < 4u)
            flags[(size_t)(gidx >> 6) * 4 + lane] = 0ull;
        fx += g.adv_x;
        fy += g.adv_y;
        if (fx >= g.w) {
            fx -= g.w;
            ++fy;
        }
        if (fy >= g.h) fy -= g.h;
    }
}

// General palettes (non-integer: gamma on) -- float64 brute force with the reference's arithmetic.
template <int MODE>
__global__ __launch_bounds__(kBlock) void ordered_f64_kernel(const uint8_t *__restrict__ in,
                                                             uint8_t *__restrict__ out,
                                                             unsigned long long *__restrict__ flags, const Geo g,
                                                             const PalDev pal, const ThrDev thr, const float sx,
                                                             const float sy, const float sc)
{
    __shared__ uint32_t s_out[DP_MAX_COLORS];
    __shared__ uint8_t s_lut[256];
    for (int i = threadIdx.x; i < pal.K; i += kBlock) s_out[i] = pal.out_rgb[i];
    for (int i = threadIdx.x; i < 256; i += kBlock) s_lut[i] = pal.lut_in ? pal.lut_in[i] : (uint8_t)i;
    __syncthreads();

    const uint32_t gidx = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t p0 = gidx * 4u;
    uint32_t px[4];
    load4(in, g, gidx, px);

    const double inf = __longlong_as_double(0x7ff0000000000000LL);
    Cursor cur;
    cursor_init(g, thr, p0 < g.n_px ? p0 : 0u, cur, MODE == 1 || MODE == 2);
    uint32_t col[4];
    bool slow[4];
    const int K = pal.K;
    for (int q = 0; q < 4; ++q) {
        const double x0 = (double)s_lut[px[q] & 255u], x1 = (double)s_lut[(px[q] >> 8) & 255u],
                     x2 = (double)s_lut[(px[q] >> 16) & 255u];
        double b0 = inf, b1 = inf, b2 = inf;
        int i0 = 0, i1 = 0;
        for (int j = 0; j < K; ++j) {
            const double d = sq_dist3(pal.pts + 3 * j, x0, x1, x2);
            if (d < b2) {
                if (d < b1) {
                    b2 = b1;
                    if (d < b0) {
                        b1 = b0;
                        i1 = i0;
                        b0 = d;
                        i0 = j;
                    } else {
                        b1 = d;
                        i1 = j;
                    }
                } else {
                    b2 = d;
                }
            }
        }
        bool nearest = true, s;
        if (MODE == 0) {
            s = (b0 == b1);
        } else {
            float t;
            if (MODE == 3)
                t = ign_threshold(g.x0 + (int)cur.x, g.y0 + (int)cur.y, sx, sy, sc);
            else
                t = thr.f32[cur.ty * thr.th_w + cur.tx];
            nearest = ordered_use_nearest(b0, b1, t);
            s = (b0 == b1) | ((b1 == b2) & !nearest & (b1 != inf));
        }
        slow[q] = s & (p0 + q < g.n_px);
        col[q] = s_out[nearest ? i0 : i1];
        cursor_next(g, thr, cur);
    }
    store4(out, g, gidx, col);
    store_flags(flags, g.dirty, gidx, slow);
}

// Pass 2: resolve flagged pixels in scipy's order.
template <int MODE, int CAP>  // MODE: 0 nearest, 2 matrix (f32 thresholds), 3 IGN; CAP: traversal queue entries
__global__ __launch_bounds__(kBlock) void fixup_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                       const unsigned long long *__restrict__ flags,
                                                       const uint32_t n_words, const Geo g, const PalDev pal_in,
                                                       const ThrDev thr, const float sx, const float sy,
                                                       const float sc)
{
    __shared__ uint32_t s_list[kListCap];
    __shared__ uint32_t s_count;
    const uint32_t n_dirty = *g.dirty;
    if (n_dirty == 0u) return;  // nothing was flagged in pass 1
    // The traversal is a chain of dependent reads (node -> child -> leaf -> point): from global memory one
    // query costs tens of microseconds, which is the whole pass when only a few hundred pixels are flagged.
    // Stage the tree in LDS and point a private copy of the palette descriptor at it.
    __shared__ double s_pts[kLdsTreeK * 3];
    __shared__ double s_split[kLdsTreeNodes];
    __shared__ int32_t s_indices[kLdsTreeK];
    __shared__ int32_t s_nodes[5][kLdsTreeNodes];  // split_dim, start, end, less, greater
    __shared__ uint32_t s_outrgb[kLdsTreeK];
    PalDev pal = pal_in;
    if (pal_in.K <= kLdsTreeK && pal_in.n_nodes <= kLdsTreeNodes) {
        for (int i = threadIdx.x; i < pal_in.K * 3; i += kBlock) s_pts[i] = pal_in.pts[i];
        for (int i = threadIdx.x; i < pal_in.K; i += kBlock) {
            s_indices[i] = pal_in.indices[i];
            s_outrgb[i] = pal_in.out_rgb[i];
        }
        for (int i = threadIdx.x; i < pal_in.n_nodes; i += kBlock) {
            s_split[i] = pal_in.split[i];
            s_nodes[0][i] = pal_in.split_dim[i];
            s_nodes[1][i] = pal_in.start[i];
            s_nodes[2][i] = pal_in.end[i];
            s_nodes[3][i] = pal_in.less[i];
            s_nodes[4][i] = pal_in.greater[i];
        }
        pal.pts = s_pts;
        pal.split = s_split;
        pal.indices = s_indices;
        pal.split_dim = s_nodes[0];
        pal.start = s_nodes[1];
        pal.end = s_nodes[2];
        pal.less = s_nodes[3];
        pal.greater = s_nodes[4];
        pal.out_rgb = s_outrgb;
    }
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();

    auto drain = [&]() {
        const uint32_t n = s_count;
        for (uint32_t i = threadIdx.x; i < n; i += kBlock) {
            const uint32_t p = s_list[i];
            const uint8_t *b = in + (size_t)p * 3;
            uint32_t c0 = b[0], c1 = b[1], c2 = b[2];
            if (pal.lut_in) {
                c0 = pal.lut_in[c0];
                c1 = pal.lut_in[c1];
                c2 = pal.lut_in[c2];
            }
            double d2[2];
            int ii[2];
            int pick;
            if (MODE == 0) {
                tree_query<1, CAP>(pal, (double)c0, (double)c1, (double)c2, d2, ii);
                pick = ii[0];
            } else {
                tree_query<2, CAP>(pal, (double)c0, (double)c1, (double)c2, d2, ii);
                uint32_t y, x;
                locate(g, p, y, x);
                float t;
                if (MODE == 3)
                    t = ign_threshold(g.x0 + (int)x, g.y0 + (int)y, sx, sy, sc);
                else
                    t = thr.f32[(((uint32_t)g.y0 + y) % (uint32_t)thr.th_h) * thr.th_w +
                                (((uint32_t)g.x0 + x) % (uint32_t)thr.th_w)];
                pick = ordered_use_nearest(d2[0], d2[1], t) ? ii[0] : ii[1];
                if (pick >= pal.K) pick = ii[0];
            }
            const uint32_t c = pal.out_rgb[pick];
            uint8_t *o = out + (size_t)p * 3;
            o[0] = (uint8_t)c;
            o[1] = (uint8_t)(c >> 8);
            o[2] = (uint8_t)(c >> 16);
        }
        __syncthreads();
        if (threadIdx.x == 0) s_count = 0;
        __syncthreads();
    };

    if (n_dirty <= (uint32_t)kQueueTiles) {
        // few flagged wave tiles: visit exactly those (4 flag words each), 64 tiles per iteration
        for (uint32_t base = blockIdx.x * 64u; base < n_dirty; base += gridDim.x * 64u) {
            const uint32_t e = base + (threadIdx.x >> 2), q = threadIdx.x & 3u;
            unsigned long long wd = 0;
            uint32_t tile = 0;
            if (e < n_dirty) {
                tile = g.dirty[1 + e];
                wd = flags[(size_t)tile * 4 + q];
            }
            if (wd) {
                const int n = __popcll(wd);
                uint32_t pos = atomicAdd(&s_count, (uint32_t)n);
                while (wd) {
                    const int lane = __ffsll((long long)wd) - 1;
                    wd &= wd - 1;
                    s_list[pos++] = ((tile * 64u + (uint32_t)lane) << 2) + q;
                }
            }
            __syncthreads();
            if (s_count > (uint32_t)kDrainAt) drain();
        }
        drain();
        return;
    }
    // stream the bitmap 8 x 256 words per iteration; blocks without any flag (almost all of them once the
    // tie codes are in use) cost one coalesced 8-byte load per lane and one barrier
    constexpr int kWordsPerIter = 8;
    for (uint32_t base = blockIdx.x * kBlock * kWordsPerIter; base < n_words; base += gridDim.x * kBlock * kWordsPerIter) {
        unsigned long long word[kWordsPerIter];
        unsigned long long any = 0;
#pragma unroll
        for (int k = 0; k < kWordsPerIter; ++k) {
            const uint32_t wi = base + k * kBlock + threadIdx.x;
            word[k] = (wi < n_words) ? flags[wi] : 0ull;
            any |= word[k];
        }
        if (!__syncthreads_or(any != 0ull)) continue;  // wave- and block-uniform
#pragma unroll
        for (int k = 0; k < kWordsPerIter; ++k) {
            const uint32_t wi = base + k * kBlock + threadIdx.x;
            unsigned long long wd = word[k];
            if (wd) {
                const int n = __popcll(wd);
                uint32_t pos = atomicAdd(&s_count, (uint32_t)n);
                const uint32_t tile = wi >> 2, q = wi & 3u;
                while (wd) {
                    const int lane = __ffsll((long long)wd) - 1;
                    wd &= wd - 1;
                    s_list[pos++] = ((tile * 64u + (uint32_t)lane) << 2) + q;
                }
            }
            __syncthreads();
            if (s_count > (uint32_t)kDrainAt) drain();  // uniform: every thread reads the same s_count
        }
    }
    drain();
}

__global__ void ign_field_kernel(float *__restrict__ out, const int h, const int w, const int y0, const int x0,
                                 const float sx, const float sy, const float sc)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x < w && y < h) out[(size_t)y * w + x] = ign_threshold(x0 + x, y0 + y, sx, sy, sc);
}

// Pillow's NEAREST resize (ImagingScaleAffine): the source index of output column x is int(xo_x) with
// xo_0 = 0.5*scale and xo_{x+1} = xo_x + scale ACCUMULATED in double -- not (x+0.5)*scale, which differs
// whenever the product is an exact integer.  Thread 0 builds the column table, thread 1 the row table.
__global__ void resize_tables_kernel(int *__restrict__ xtab, int *__restrict__ ytab, const int w, const int ow,
                                     const int h, const int oh)
{
    const int which = threadIdx.x;
    if (which > 1) return;
    const int n_in = which ? h : w, n_out = which ? oh : ow;
    int *tab = which ? ytab : xtab;
    const double a = (double)n_in / (double)n_out;
    double xo = __dmul_rn(a, 0.5);
    for (int x = 0; x < n_out; ++x) {
        int v = (int)xo;
        tab[x] = v < n_in ? v : n_in - 1;
        xo = __dadd_rn(xo, a);
    }
}

__global__ void resize_nearest_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const int h,
                                      const int w, const int oh, const int ow, const int *__restrict__ xtab,
                                      const int *__restrict__ ytab)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const size_t f = blockIdx.z;
    if (x >= ow || y >= oh) return;
    const uint8_t *s = in + (f * h * w + (size_t)ytab[y] * w + xtab[x]) * 3;
    uint8_t *d = out + (f * oh * ow + (size_t)y * ow + x) * 3;
    d[0] = s[0];
    d[1] = s[1];
    d[2] = s[2];
}

int num_cus()
{
    static thread_local int cached_dev = -1, cached = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev != cached_dev) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached = n;
        cached_dev = dev;
    }
    return cached;
}

static inline bool env_set(const char *name) { return exp_env(name) != nullptr; }

template <int MODE>
int launch_cell(uint32_t grid, size_t lds, hipStream_t s, const uint8_t *in, uint8_t *out, unsigned long long *flags,
                const Geo &g, const PalDev &pal, const ThrDev &thr, float sx, float sy, float sc, uint32_t n_tiles)
{
    auto kern = ordered_cell_kernel<MODE>;
    // more than 64 KB of dynamic LDS has to be granted per function (once per device is enough,
    // repeating it is cheap)
    DP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kCellBlock), lds, s, in, out, flags, g, pal, thr, sx, sy, sc, n_tiles);
    return DP_OK;
}

template <int MODE>
void launch_pass1(bool integer, dim3 grid, hipStream_t s, const uint8_t *in, uint8_t *out, unsigned long long *flags,
                  const Geo &g, const PalDev &pal, const ThrDev &thr, float sx, float sy, float sc)
{
    if (integer)
        hipLaunchKernelGGL(ordered_int_kernel<MODE>, grid, dim3(kBlock), 0, s, in, out, flags, g, pal, thr, sx, sy, sc);
    else
        hipLaunchKernelGGL(ordered_f64_kernel<MODE>, grid, dim3(kBlock), 0, s, in, out, flags, g, pal, thr, sx, sy, sc);
}

}  // namespace

int launch_ordered(const uint8_t *in, uint8_t *out, int64_t n_frames, int h, int w, int y0, int x0,
                   const PalDev &pal, int mode, const ThrDev *thr_in, float ign_scale, int ign_seed, void *ws,
                   size_t ws_bytes, hipStream_t s)
{
    (void)ws_bytes;
    ThrDev thr;
    thr.th_h = thr.th_w = 1;
    thr.f32 = nullptr;
    thr.m = nullptr;
    thr.sh = 0;
    thr.fpad = nullptr;
    thr.mpad = nullptr;
    for (uint32_t &wd : thr.cls_nib) wd = 0;
    thr.has_cls = 0;
    thr.tw_pad = 0;
    thr.pow2 = 1;
    thr.inv_h = thr.inv_w = 1.0;
    if (mode == DP_MODE_MATRIX) thr = *thr_in;
    // a single colour: every pixel maps to it, and the k=2 query of the reference has no second entry
    if (pal.K == 1) mode = DP_MODE_NEAREST;

    const float sx = (float)((double)ign_seed * 0.37), sy = (float)((double)ign_seed * 0.73);
    const int64_t hw = (int64_t)h * w;
    // frames per launch so that pixel indices stay below 2^30
    // (2^30: the lean kernels' queue entries keep the index of a group of four pixels in 28 bits)
    const int64_t max_frames = std::max<int64_t>(1, ((int64_t)1 << 30) / hw - 1);
    unsigned long long *flags = reinterpret_cast<unsigned long long *>(ws);
    // the dirty word lives in the slack behind the largest possible bitmap of this call
    const size_t dirty_off = ((size_t)((std::min(max_frames, n_frames) * hw + 255) / 256) * 32 + 768) & ~size_t(7);

    for (int64_t f0 = 0; f0 < n_frames; f0 += max_frames) {
        const int64_t nf = std::min(max_frames, n_frames - f0);
        const uint8_t *in_c = in + (size_t)f0 * hw * 3;
        uint8_t *out_c = out + (size_t)f0 * hw * 3;
        Geo g;
        g.n_px = (uint32_t)(nf * hw);
        g.hw = (uint32_t)hw;
        g.w = (uint32_t)w;
        g.h = (uint32_t)h;
        g.inv_hw = 1.0 / (double)hw;
        g.inv_w = 1.0 / (double)w;
        g.y0 = y0;
        g.x0 = x0;
        g.ty0 = y0 % thr.th_h;
        g.tx0 = x0 % thr.th_w;
        g.aligned = (((uintptr_t)in_c | (uintptr_t)out_c) & 3) == 0;
        g.neg2 = -(2 << kLocalBits);
        g.adv_y = g.adv_x = 0;
        g.dirty = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(ws) + dirty_off);
        DP_HIP(hipMemsetAsync(g.dirty, 0, sizeof(uint32_t), s));  // the count; queue entries need no reset
        const uint32_t groups = (g.n_px + 3) / 4;
        const uint32_t blocks = (groups + kBlock - 1) / kBlock;
        uint32_t n_words = blocks * (kBlock / 64) * 4;
        unsigned long long *fl = flags;  // chunks run back to back on one stream: the bitmap is reused
        const bool integer = pal.is_integer != 0;
        int fix_mode;
        ProfMark *pm = prof_begin(s);
        const bool int_thr_ok = thr.m != nullptr && thr.th_h * thr.th_w <= 256;
        // the table the lean kernels would stage: 4-entry blocks when the accelerator built them, else 8-entry blocks
        // ... or, crowded palettes, the table over warped cells (with its maps)
        const bool warp = pal.warp_tab != nullptr;
        const bool small = warp ? pal.warp_bw == 4 : pal.cell_tab4 != nullptr;
        const size_t lean_tab_bytes = warp ? 4 * (size_t)pal.warp_words + kWarpLutBytes : 4 * (size_t)(small ? pal.tab4_words : pal.tab_words);
        const bool geo_ok = integer && (warp || small || pal.cell_tab != nullptr) && g.aligned && y0 >= 0 && x0 >= 0 && g.n_px <= (1u << 30);
        const bool lean_geo = geo_ok && lean_tab_bytes <= (size_t)kLeanTabBytes;
        const bool int_lean = thr.mpad != nullptr && lean_tab_bytes + (size_t)thr.th_h * thr.tw_pad * 4 <= (size_t)kLeanTabBytes;
        const bool lean_ok = lean_geo && (mode == DP_MODE_NEAREST || mode == DP_MODE_IGN ||
                                          (mode == DP_MODE_MATRIX && (int_lean || thr.fpad != nullptr)));
        const bool whole_tab = pal.cell_tab != nullptr && pal.tab_total == pal.tab_words;  // the general kernel stages all of it
        if (integer && (lean_ok || whole_tab)) {
            // fast path: LDS cell table + tie codes, persistent 1024-lane workgroups over 4096-pixel tiles
            const uint32_t n_tiles = (groups + kCellBlock - 1) / kCellBlock;
            n_words = n_tiles * (kCellBlock / 64) * 4;
            const size_t lds = sizeof(uint32_t) * ((size_t)pal.tab_words + 256);  // launch_cell (plain table) only
            // 4-entry blocks on plain cells that leave half of LDS free: two workgroups per CU (DP_LEAN_NO_HALF=1: one)
            const bool half = lean_ok && small && !warp && !exp_env("DP_LEAN_NO_HALF") &&
                              lean_tab_bytes + (mode == DP_MODE_MATRIX && int_lean ? (size_t)thr.th_h * thr.tw_pad * 4 : 0) <= (size_t)kLeanHalfTabBytes;
            const uint32_t cgrid = std::min<uint32_t>(n_tiles, (uint32_t)num_cus() * (half ? 2u : 1u));
            {
                const uint64_t adv = ((uint64_t)cgrid * kCellBlock * 4u) % (uint64_t)hw;
                g.adv_y = (uint32_t)(adv / (uint64_t)w);
                g.adv_x = (uint32_t)(adv % (uint64_t)w);
            }
            int rc;
            PalDev pal4 = pal;  // the table the lean kernel stages, in the fields it reads
            if (warp) {
                pal4.cell_tab = pal.warp_tab;
                pal4.tab_words = pal.warp_words;
                pal4.tab_total = pal.warp_total;
            } else if (small) {
                pal4.cell_tab = pal.cell_tab4;
                pal4.tab_words = pal.tab4_words;
                pal4.tab_total = pal.tab4_words;
            }
            // crowded palettes (many split cells, or a table larger than LDS): the instantiation that adapts per wave
            const bool adapt = warp ? pal.warp_adapt != 0 : (!small && pal.adapt != 0);
#define DP_LEAN_K(M, BW, AD, WP)                                                                                         \
    hipLaunchKernelGGL((ordered_lean_kernel<M, BW, AD, WP>), dim3(cgrid), dim3(kCellBlock), 0, s, in_c, out_c, fl, g, pal4, thr, \
                       sx, sy, ign_scale, n_tiles)
#define DP_LEAN(M)                                                                                                        \
    do {                                                                                                                 \
        if (warp && small) DP_LEAN_K(M, 4, false, true);                                                                 \
        else if (warp && adapt) DP_LEAN_K(M, 8, true, true);                                                             \
        else if (warp) DP_LEAN_K(M, 8, false, true);                                                                     \
        else if (small && half)                                                                                          \
            hipLaunchKernelGGL((ordered_lean_kernel<M, 4, false, false, true>), dim3(cgrid), dim3(kCellBlock), 0, s, in_c, out_c, fl, g, \
                               pal4, thr, sx, sy, ign_scale, n_tiles);                                                   \
        else if (small) DP_LEAN_K(M, 4, false, false);                                                                   \
        else if (adapt) DP_LEAN_K(M, 8, true, false);                                                                    \
        else DP_LEAN_K(M, 8, false, false);                                                                              \
    } while (0)
            // uncrowded palettes on plain cells: the fast kernel (nearest set staged first; see ordered_fast_kernel)
            const uint32_t *perm = small ? pal.cell_perm4 : pal.cell_perm;
            const size_t fast_tab_bytes = 4096u * (small ? 4u : 8u) * 4u;
            // (the fast kernel stages the 4096 cell blocks, the flat lists of the split cells, integer thresholds, two queues)
            const size_t fast_fixed = fast_tab_bytes + (size_t)(small ? pal.n_wide4 : pal.n_wide) * kWideList * 4 +
                                      2 * (kCellBlock / 64) * kFastQueue * 4;
            const bool int_fast = thr.mpad != nullptr && fast_fixed + (size_t)thr.th_h * thr.tw_pad * 4 <= sizeof(uint32_t) * kLeanLdsWords;
            // Measured on MI355X (tools/bench_scripts/fast_vs_lean.py, 24 4K frames, 256 colours): nearest-only mode
            // 0.45 ms against 0.50 ms of the lean kernel; with a matrix the per-slot class branches cost more than the
            // shorter candidate network saves (0.66 against 0.55 ms), so the matrix / IGN modes stay on the lean kernel
            // unless DP_FAST_ALL is set (experiments).
            const bool fast_mode = mode == DP_MODE_NEAREST || exp_env("DP_FAST_ALL") != nullptr;
            const bool fast_ok = fast_mode && geo_ok && !warp && !adapt && perm != nullptr && fast_fixed <= sizeof(uint32_t) * kLeanLdsWords &&
                                 (mode == DP_MODE_NEAREST || mode == DP_MODE_IGN ||
                                  (mode == DP_MODE_MATRIX && (int_fast || thr.fpad != nullptr)));
            const int dbg = exp_env("DP_FAST_DBG") ? atoi(exp_env("DP_FAST_DBG")) : 0;  // (measurement switch)
#define DP_FAST(M)                                                                                                        \
    do {                                                                                                                 \
        if (dbg == 1 && !small) hipLaunchKernelGGL((ordered_fast_kernel<M, 8, 1>), dim3(cgrid), dim3(kCellBlock), 0, s, in_c, out_c, fl, g, pal4, thr, sx, sy, ign_scale, n_tiles); \
        else if (dbg == 2 && !small) hipLaunchKernelGGL((ordered_fast_kernel<M, 8, 2>), dim3(cgrid), dim3(kCellBlock), 0, s, in_c, out_c, fl, g, pal4, thr, sx, sy, ign_scale, n_tiles); \
        else if (dbg == 3 && !small) hipLaunchKernelGGL((ordered_fast_kernel<M, 8, 3>), dim3(cgrid), dim3(kCellBlock), 0, s, in_c, out_c, fl, g, pal4, thr, sx, sy, ign_scale, n_tiles); \
        else if (small) hipLaunchKernelGGL((ordered_fast_kernel<M, 4>), dim3(cgrid), dim3(kCellBlock), 0, s, in_c, out_c, fl, g, pal4, thr, sx, sy, ign_scale, n_tiles); \
        else hipLaunchKernelGGL((ordered_fast_kernel<M, 8>), dim3(cgrid), dim3(kCellBlock), 0, s, in_c, out_c, fl, g, pal4, thr, sx, sy, ign_scale, n_tiles); \
    } while (0)
            // crowded palettes: the compact kernel (the whole octree in LDS, every pixel resolved in place), two workgroups
            // per CU when table + colours + maps + thresholds fit 80 KB
            const size_t comp_base = pal.comp_tab ? (size_t)kCompactTabAt + 4 * (size_t)pal.comp_words : 0;
            // (integer thresholds go to LDS behind the table; a table too large for that -- blue noise -- is read as float32 from L1)
            const bool int_comp = mode == DP_MODE_MATRIX && thr.mpad != nullptr &&
                                  comp_base + (size_t)thr.th_h * thr.tw_pad * 4 <= sizeof(uint32_t) * kLeanLdsWords;
            const size_t comp_bytes = comp_base + (int_comp ? (size_t)thr.th_h * thr.tw_pad * 4 : 0);
            const bool comp_ok = pal.comp_tab != nullptr && geo_ok && (adapt || env_set("DP_FORCE_COMPACT")) && comp_bytes <= sizeof(uint32_t) * kLeanLdsWords &&
                                 (mode == DP_MODE_NEAREST || mode == DP_MODE_IGN || (mode == DP_MODE_MATRIX && (int_comp || thr.fpad != nullptr))) &&
                                 !env_set("DP_NO_COMPACT_KERNEL");
            const bool comp_half = comp_ok && comp_bytes <= sizeof(uint32_t) * kCompactHalfWords && !env_set("DP_COMPACT_NO_HALF");
            const uint32_t pgrid = std::min<uint32_t>(n_tiles, (uint32_t)num_cus() * (comp_half ? 2u : 1u));
            if (comp_ok) {
                const uint64_t adv = ((uint64_t)pgrid * kCellBlock * 4u) % (uint64_t)hw;
                g.adv_y = (uint32_t)(adv / (uint64_t)w);
                g.adv_x = (uint32_t)(adv % (uint64_t)w);
            }
#define DP_COMP_K(M, WP, HF)                                                                                             \
    hipLaunchKernelGGL((ordered_compact_kernel<M, WP, HF>), dim3(pgrid), dim3(kCellBlock), 0, s, in_c, out_c, fl, g, pal, thr, sx, sy, \
                       ign_scale, n_tiles)
#define DP_COMP(M)                                                                                                        \
    do {                                                                                                                 \
        if (pal.comp_warp && comp_half) DP_COMP_K(M, true, true);                                                        \
        else if (pal.comp_warp) DP_COMP_K(M, true, false);                                                               \
        else if (comp_half) DP_COMP_K(M, false, true);                                                                   \
        else DP_COMP_K(M, false, false);                                                                                 \
    } while (0)
            if (comp_ok && mode == DP_MODE_NEAREST) {
                DP_COMP(0);
                rc = DP_OK;
                fix_mode = 0;
            } else if (comp_ok && mode == DP_MODE_IGN) {
                DP_COMP(3);
                rc = DP_OK;
                fix_mode = 3;
            } else if (comp_ok && mode == DP_MODE_MATRIX && int_comp) {
                DP_COMP(1);
                rc = DP_OK;
                fix_mode = 2;
            } else if (comp_ok && mode == DP_MODE_MATRIX) {
                DP_COMP(2);
                rc = DP_OK;
                fix_mode = 2;
#undef DP_COMP
#undef DP_COMP_K
            } else if (fast_ok && mode == DP_MODE_NEAREST) {
                DP_FAST(0);
                rc = DP_OK;
                fix_mode = 0;
            } else if (fast_ok && mode == DP_MODE_IGN) {
                DP_FAST(3);
                rc = DP_OK;
                fix_mode = 3;
            } else if (fast_ok && mode == DP_MODE_MATRIX && int_fast) {
                DP_FAST(1);
                rc = DP_OK;
                fix_mode = 2;
            } else if (fast_ok && mode == DP_MODE_MATRIX && thr.fpad != nullptr) {
                DP_FAST(2);
                rc = DP_OK;
                fix_mode = 2;
#undef DP_FAST
            } else if (lean_geo && mode == DP_MODE_NEAREST) {
                DP_LEAN(0);
                rc = DP_OK;
                fix_mode = 0;
            } else if (lean_geo && mode == DP_MODE_IGN) {
                DP_LEAN(3);
                rc = DP_OK;
                fix_mode = 3;
            } else if (lean_geo && mode == DP_MODE_MATRIX && int_lean) {
                DP_LEAN(1);
                rc = DP_OK;
                fix_mode = 2;
            } else if (lean_geo && mode == DP_MODE_MATRIX && thr.fpad != nullptr) {
                DP_LEAN(2);
                rc = DP_OK;
                fix_mode = 2;
#undef DP_LEAN
#undef DP_LEAN_K
            } else if (mode == DP_MODE_NEAREST) {
                rc = launch_cell<0>(cgrid, lds, s, in_c, out_c, fl, g, pal, thr, sx, sy, ign_scale, n_tiles);
                fix_mode = 0;
            } else if (mode == DP_MODE_IGN) {
                rc = launch_cell<3>(cgrid, lds, s, in_c, out_c, fl, g, pal, thr, sx, sy, ign_scale, n_tiles);
                fix_mode = 3;
            } else if (int_thr_ok) {
                rc = launch_cell<1>(cgrid, lds, s, in_c, out_c, fl, g, pal, thr, sx, sy, ign_scale, n_tiles);
                fix_mode = 2;
            } else {
                rc = launch_cell<2>(cgrid, lds, s, in_c, out_c, fl, g, pal, thr, sx, sy, ign_scale, n_tiles);
                fix_mode = 2;
            }
            if (rc != DP_OK) return rc;
        } else if (!integer && pal.ftab != nullptr && g.aligned && y0 >= 0 && x0 >= 0 &&
                   (size_t)pal.ftab_words * 4 + (size_t)pal.K * 16 + 256 <= sizeof(uint32_t) * kLeanLdsWords &&  // staged part
                   (mode != DP_MODE_MATRIX || thr.fpad != nullptr)) {
            // float (gamma) palettes with a cell table
            // the one-byte-per-entry table (K <= 256): records + lut + table in LDS
            const size_t cf_bytes = pal.comp_tab ? (size_t)kCfTabAt + 4 * (size_t)pal.comp_words : 0;
            const bool cf_ok = pal.comp_tab != nullptr && cf_bytes <= sizeof(uint32_t) * kLeanLdsWords && !env_set("DP_NO_COMPACT_KERNEL");
            const uint32_t n_tiles = (groups + kCellBlock - 1) / kCellBlock;
            n_words = n_tiles * (kCellBlock / 64) * 4;
            const uint32_t cgrid = std::min<uint32_t>(n_tiles, (uint32_t)num_cus());
            const uint64_t adv = ((uint64_t)cgrid * kCellBlock * 4u) % (uint64_t)hw;
            g.adv_y = (uint32_t)(adv / (uint64_t)w);
            g.adv_x = (uint32_t)(adv % (uint64_t)w);
#define DP_LEANF(M)                                                                                                      \
    do {                                                                                                                 \
        if (cf_ok) hipLaunchKernelGGL((ordered_compact_float_kernel<M>), dim3(cgrid), dim3(kCellBlock), 0, s, in_c, out_c, fl, g, pal, thr, sx, sy, ign_scale, n_tiles); \
        else hipLaunchKernelGGL(ordered_lean_float_kernel<M>, dim3(cgrid), dim3(kCellBlock), 0, s, in_c, out_c, fl, g, pal, thr, sx, sy, ign_scale, n_tiles); \
    } while (0)
            if (mode == DP_MODE_NEAREST) {
                DP_LEANF(0);
                fix_mode = 0;
            } else if (mode == DP_MODE_IGN) {
                DP_LEANF(3);
                fix_mode = 3;
            } else {
                DP_LEANF(2);
                fix_mode = 2;
            }
#undef DP_LEANF
        } else if (mode == DP_MODE_NEAREST) {
            launch_pass1<0>(integer, dim3(blocks), s, in_c, out_c, fl, g, pal, thr, sx, sy, ign_scale);
            fix_mode = 0;
        } else if (mode == DP_MODE_IGN) {
            launch_pass1<3>(integer, dim3(blocks), s, in_c, out_c, fl, g, pal, thr, sx, sy, ign_scale);
            fix_mode = 3;
        } else {
            if (integer && int_thr_ok)
                launch_pass1<1>(integer, dim3(blocks), s, in_c, out_c, fl, g, pal, thr, sx, sy, ign_scale);
            else
                launch_pass1<2>(integer, dim3(blocks), s, in_c, out_c, fl, g, pal, thr, sx, sy, ign_scale);
            fix_mode = 2;
        }
        DP_HIP(hipGetLastError());
        prof_mid(pm, s);
        // one resident workgroup per CU (the 96 KB LDS list admits no more): a persistent grid avoids queueing
        const uint32_t fgrid = std::min<uint32_t>((n_words + kBlock * 8 - 1) / (kBlock * 8), (uint32_t)num_cus());
#define DP_FIX(M, C) hipLaunchKernelGGL((fixup_kernel<M, C>), dim3(fgrid), dim3(kBlock), 0, s, in_c, out_c, fl, n_words, g, pal, thr, sx, sy, ign_scale)
        const bool big_q = pal.n_inner > kQueueSmall;
        if (fix_mode == 0) {
            if (big_q) DP_FIX(0, kQueueLarge); else DP_FIX(0, kQueueSmall);
        } else if (fix_mode == 3) {
            if (big_q) DP_FIX(3, kQueueLarge); else DP_FIX(3, kQueueSmall);
        } else {
            if (big_q) DP_FIX(2, kQueueLarge); else DP_FIX(2, kQueueSmall);
        }
#undef DP_FIX
        prof_end(pm, s);
        DP_HIP(hipGetLastError());
    }
    return DP_OK;
}

int launch_ign_thresholds(float *out, int h, int w, int y0, int x0, float scale, int seed, hipStream_t s)
{
    const float sx = (float)((double)seed * 0.37), sy = (float)((double)seed * 0.73);
    if (h > 65535) {
        set_error("dp_ign_thresholds: h > 65535 not supported");
        return DP_EUNSUPPORTED;
    }
    hipLaunchKernelGGL(ign_field_kernel, dim3((w + 255) / 256, h), dim3(256), 0, s, out, h, w, y0, x0, sx, sy, scale);
    DP_HIP(hipGetLastError());
    return DP_OK;
}

int launch_resize_nearest(const uint8_t *in, uint8_t *out, int64_t n_frames, int h, int w, int oh, int ow,
                          hipStream_t s)
{
    if (oh > 65535 || n_frames > 65535) {
        set_error("dp_resize_nearest_u8: oh or n_frames > 65535 not supported");
        return DP_EUNSUPPORTED;
    }
    int *tabs = nullptr;
    DP_HIP(hipMallocAsync((void **)&tabs, sizeof(int) * ((size_t)ow + oh), s));  // stream-ordered scratch
    hipLaunchKernelGGL(resize_tables_kernel, dim3(1), dim3(64), 0, s, tabs, tabs + ow, w, ow, h, oh);
    hipLaunchKernelGGL(resize_nearest_kernel, dim3((ow + 255) / 256, oh, (unsigned)n_frames), dim3(256), 0, s, in, out,
                       h, w, oh, ow, tabs, tabs + ow);
    hipError_t e = hipGetLastError();
    (void)hipFreeAsync(tabs, s);
    if (e != hipSuccess) return hip_fail(e, "resize launch");
    return DP_OK;
}

}  // namespace dp
