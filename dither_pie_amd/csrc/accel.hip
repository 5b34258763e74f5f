// Per-palette search accelerator for integer palettes (built once in dp_palette_create).
//
// The nearest / second-nearest palette entry of a colour x can only be an entry of
//     T(x) = { j : d_j(x) <= d_(2)(x) }        (everything at least as close as the 2nd nearest),
// and which of several equidistant entries scipy reports depends only on the order in which its
// KD-tree traversal meets the members of T(x).  Both facts are properties of the 2^24 colours, not
// of an image, so they are tabulated per palette:
//
//   cell lists   for each of the 16x16x16 cells of the RGB cube, the exact union of T(x) over the
//                cell's 4096 colours (~5 entries for a 256-colour palette), sorted by palette index
//                and padded to exactly 8 with further (harmless) palette entries: one 32-byte block
//                per cell at a fixed stride, so the dither kernel needs no descriptor and no loop.
//                The few cells with more than 8 members (~1.5 %) hold a marker instead and are split
//                octree-fashion into eight half-size sub-cells (8^3, then 4^3, 2^3, single colours),
//                each with its own 8-entry block; only a single colour with more than 8 equidistant
//                candidates is marked "slow" (its pixels go to the generic fix-up pass).
//                The whole table (128 KB + ~64 words per split) lives in LDS in the dither kernel.
//   tie codes    per colour and per query kind: for colours whose three smallest distances contain a
//                tie, the outcome of scipy's traversal (tree_query) expressed relative to the
//                candidates sorted by (distance, index):
//                  k=2 (4 bits): 0 -> (c0,c1)  1 -> (c1,c0)  2 -> (c0,c2)  3 -> (c2,c0)  4 -> (c1,c2)
//                       5 -> (c2,c1)  15 -> none of these (four-way ties; the dither kernel flags the
//                       pixel for the generic fix-up pass)
//                  k=1 (2 bits): 0 -> c0  1 -> c1  2 -> c2  3 -> other
//                "equal" is judged on the output colour, so duplicate palette entries never need the
//                fix-up pass.  8 MB + 4 MB, touched only by tied pixels (~0.3 %).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "dp_internal.h"
#include "tree_query.hip.h"

namespace dp {
namespace {

__device__ __forceinline__ int med3i(const int a, const int b, const int c)
{
    return max(min(a, b), min(max(a, b), c));
}

constexpr int kExcCap = 1 << 16;  // colours with an outcome outside the codes (four-way ties); normally a few dozen

template <int MW, int CAP>  // MW: mask words (32 palette entries each); CAP: traversal queue of the tie queries
__global__ __launch_bounds__(256) void accel_scan_kernel(const PalDev pal, uint32_t *__restrict__ masks,
                                                         uint32_t *__restrict__ nmasks, uint32_t *__restrict__ code1,
                                                         uint32_t *__restrict__ code2, uint4 *__restrict__ exc,
                                                         uint32_t *__restrict__ exc_count)
{
    __shared__ uint32_t s_mask[9][MW];  // [0] the whole cell, [1+s] its 8x8x8 sub-cell s
    __shared__ uint32_t s_nmask[MW];    // N(cell): the entries that are NEAREST (ties included) to some colour of the cell
    for (int i = threadIdx.x; i < 9 * MW; i += 256) (&s_mask[0][0])[i] = 0;
    if (threadIdx.x < MW) s_nmask[threadIdx.x] = 0;
    __syncthreads();
    const int cell = blockIdx.x;
    const int rc = cell >> 8, gc = (cell >> 4) & 15, bc = cell & 15;
    const int K = pal.K;
    constexpr int kBig = 0x7fffffff;
    constexpr int IM = (1 << kIdxBits) - 1;

    for (int t = 0; t < 16; ++t) {
        const int id = threadIdx.x + 256 * t;
        const uint32_t r = rc * 16 + (id >> 8), g = gc * 16 + ((id >> 4) & 15), b = bc * 16 + (id & 15);
        const uint32_t x4 = r | (g << 8) | (b << 16);
        int m0 = kBig, m1 = kBig, m2 = kBig, m3 = kBig;
        for (int j = 0; j < K; ++j) {
            const int dot = (int)__builtin_amdgcn_udot4(x4, pal.p4[j], 0u, false);
            const int key = pal.nkey[j] - (dot << (kIdxBits + 1));
            const int n3 = med3i(m2, m3, key);
            const int n2 = med3i(m1, m2, key);
            const int n1 = med3i(m0, m1, key);
            m0 = min(m0, key);
            m1 = n1;
            m2 = n2;
            m3 = n3;
        }
        const int d0 = m0 >> kIdxBits, d1 = m1 >> kIdxBits, d2 = m2 >> kIdxBits, d3 = m3 >> kIdxBits;
        const int c0 = m0 & IM, c1 = m1 & IM, c2 = m2 & IM;
        const int sub = 1 + (((((id >> 8) >> 3) & 1) << 2) | (((((id >> 4) & 15) >> 3) & 1) << 1) | (((id & 15) >> 3) & 1));
        auto mark = [&](int j) {
            atomicOr(&s_mask[0][j >> 5], 1u << (j & 31));
            atomicOr(&s_mask[sub][j >> 5], 1u << (j & 31));
        };
        mark(c0);
        if (K > 1) mark(c1);
        if (K > 2 && d2 == d1) mark(c2);
        if (K > 3 && d3 == d1) {  // four or more at the second distance: take every one of them
            for (int j = 0; j < K; ++j) {
                const int dot = (int)__builtin_amdgcn_udot4(x4, pal.p4[j], 0u, false);
                const int dj = (pal.nkey[j] - (dot << (kIdxBits + 1))) >> kIdxBits;
                if (dj <= d1) mark(j);
            }
        }
        // the nearest set: everything at the smallest distance
        atomicOr(&s_nmask[c0 >> 5], 1u << (c0 & 31));
        if (K > 1 && d1 == d0) atomicOr(&s_nmask[c1 >> 5], 1u << (c1 & 31));
        if (K > 2 && d2 == d0) atomicOr(&s_nmask[c2 >> 5], 1u << (c2 & 31));
        if (K > 3 && d3 == d0) {
            for (int j = 0; j < K; ++j) {
                const int dot = (int)__builtin_amdgcn_udot4(x4, pal.p4[j], 0u, false);
                const int dj = (pal.nkey[j] - (dot << (kIdxBits + 1))) >> kIdxBits;
                if (dj <= d0) atomicOr(&s_nmask[j >> 5], 1u << (j & 31));
            }
        }
        if (K < 2) continue;
        const bool tie01 = d0 == d1, tie12 = (K > 2) && d1 == d2;
        if (!(tie01 || tie12)) continue;
        const uint32_t o0 = pal.out_rgb[c0], o1 = pal.out_rgb[c1], o2 = K > 2 ? pal.out_rgb[c2] : 0xffffffffu;
        double dd[2];
        int ii[2], i1 = 0;
        bool other = false;
        {
            tree_query<2, CAP>(pal, (double)r, (double)g, (double)b, dd, ii);
            const uint32_t on = pal.out_rgb[ii[0]], os = pal.out_rgb[ii[1]];
            uint32_t code = 15;
            if (on == o0 && os == o1) code = 0;
            else if (on == o1 && os == o0) code = 1;
            else if (on == o0 && os == o2) code = 2;
            else if (on == o2 && os == o0) code = 3;
            else if (on == o1 && os == o2) code = 4;
            else if (on == o2 && os == o1) code = 5;
            if (code) atomicOr(&code2[x4 >> 3], code << ((x4 & 7u) * 4));
            other = code == 15;
            i1 = ii[0];  // the k=1 answer unless the two nearest tie
        }
        const uint32_t pair = (uint32_t)ii[0] | ((uint32_t)ii[1] << 16);
        if (tie01) {
            tree_query<1, CAP>(pal, (double)r, (double)g, (double)b, dd, ii);
            const uint32_t on = pal.out_rgb[ii[0]];
            uint32_t code = 3;
            if (on == o0) code = 0;
            else if (on == o1) code = 1;
            else if (on == o2) code = 2;
            if (code) atomicOr(&code1[x4 >> 4], code << ((x4 & 15u) * 2));
            other |= code == 3;
            i1 = ii[0];
        }
        if (other) {  // spell the outcome out
            const uint32_t slot = atomicAdd(exc_count, 1u);
            if (slot < (uint32_t)kExcCap) exc[slot] = make_uint4(x4, pair, (uint32_t)i1, 0u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 9 * MW; i += 256) masks[(size_t)cell * 9 * MW + i] = (&s_mask[0][0])[i];
    if (threadIdx.x < MW) nmasks[(size_t)cell * MW + threadIdx.x] = s_nmask[threadIdx.x];
}


// T(x)-union masks for arbitrary axis-aligned boxes of colours (used below the 8x8x8 level; Box: host_logic.h)
template <int MW>
__global__ __launch_bounds__(64) void accel_box_kernel(const PalDev pal, const Box *__restrict__ boxes,
                                                       uint32_t *__restrict__ masks)
{
    __shared__ uint32_t s_mask[MW];
    if (threadIdx.x < MW) s_mask[threadIdx.x] = 0;
    __syncthreads();
    const Box bx = boxes[blockIdx.x];
    const int K = pal.K;
    constexpr int kBig = 0x7fffffff;
    constexpr int IM = (1 << kIdxBits) - 1;
    const int n = bx.size * bx.size * bx.size;
    for (int id = threadIdx.x; id < n; id += 64) {
        const uint32_t r = bx.r0 + id / (bx.size * bx.size), g = bx.g0 + (id / bx.size) % bx.size, b = bx.b0 + id % bx.size;
        const uint32_t x4 = r | (g << 8) | (b << 16);
        int m0 = kBig, m1 = kBig, m2 = kBig, m3 = kBig;
        for (int j = 0; j < K; ++j) {
            const int dot = (int)__builtin_amdgcn_udot4(x4, pal.p4[j], 0u, false);
            const int key = pal.nkey[j] - (dot << (kIdxBits + 1));
            const int n3 = med3i(m2, m3, key);
            const int n2 = med3i(m1, m2, key);
            const int n1 = med3i(m0, m1, key);
            m0 = min(m0, key);
            m1 = n1;
            m2 = n2;
            m3 = n3;
        }
        const int d1 = m1 >> kIdxBits, d2 = m2 >> kIdxBits, d3 = m3 >> kIdxBits;
        atomicOr(&s_mask[(m0 & IM) >> 5], 1u << (m0 & 31));
        if (K > 1) atomicOr(&s_mask[(m1 & IM) >> 5], 1u << (m1 & 31));
        if (K > 2 && d2 == d1) atomicOr(&s_mask[(m2 & IM) >> 5], 1u << (m2 & 31));
        if (K > 3 && d3 == d1)
            for (int j = 0; j < K; ++j) {
                const int dot = (int)__builtin_amdgcn_udot4(x4, pal.p4[j], 0u, false);
                const int dj = (pal.nkey[j] - (dot << (kIdxBits + 1))) >> kIdxBits;
                if (dj <= d1) atomicOr(&s_mask[j >> 5], 1u << (j & 31));
            }
    }
    __syncthreads();
    if (threadIdx.x < MW) masks[(size_t)blockIdx.x * MW + threadIdx.x] = s_mask[threadIdx.x];
}


// ---- warped cells ------------------------------------------------------------------------------------
// A palette extracted from an image crowds its colours into a small part of the cube -- exactly where the image's
// pixels are -- and the uniform 16^3 cells there hold far more than 8 candidates.  For such palettes the table is
// built a second time over WARPED coordinates: per channel a monotone map u = lut[v] (0..255 -> 0..255) that gives
// each sixteenth of the palette's coordinates (by rank) its own cell, with the 16 sub-positions of a cell spread
// evenly over the cell's values.  Cells, sub-cells and the octree below them are defined on (lut_r[r], lut_g[g],
// lut_b[b]) exactly as the plain table defines them on (r, g, b); a warped position may stand for several colours
// (wide cells in empty regions) or for none.  The dither kernel looks the three bytes up in LDS.
struct Range {
    int r0, r1, g0, g1, b0, b1;  // the colours r0 <= r < r1, ...
};

template <int MW>
__global__ __launch_bounds__(256) void accel_range_kernel(const PalDev pal, const Range *__restrict__ ranges,
                                                          uint32_t *__restrict__ masks)
{
    __shared__ uint32_t s_mask[MW];
    if (threadIdx.x < MW) s_mask[threadIdx.x] = 0;
    __syncthreads();
    const Range bx = ranges[blockIdx.x];
    const int K = pal.K;
    constexpr int kBig = 0x7fffffff;
    constexpr int IM = (1 << kIdxBits) - 1;
    const int ng = bx.g1 - bx.g0, nb = bx.b1 - bx.b0;
    const int n = (bx.r1 - bx.r0) * ng * nb;
    for (int id = threadIdx.x; id < n; id += 256) {
        const uint32_t r = bx.r0 + id / (ng * nb), g = bx.g0 + (id / nb) % ng, b = bx.b0 + id % nb;
        const uint32_t x4 = r | (g << 8) | (b << 16);
        int m0 = kBig, m1 = kBig, m2 = kBig, m3 = kBig;
        for (int j = 0; j < K; ++j) {
            const int dot = (int)__builtin_amdgcn_udot4(x4, pal.p4[j], 0u, false);
            const int key = pal.nkey[j] - (dot << (kIdxBits + 1));
            const int n3 = med3i(m2, m3, key);
            const int n2 = med3i(m1, m2, key);
            const int n1 = med3i(m0, m1, key);
            m0 = min(m0, key);
            m1 = n1;
            m2 = n2;
            m3 = n3;
        }
        const int d1 = m1 >> kIdxBits, d2 = m2 >> kIdxBits, d3 = m3 >> kIdxBits;
        atomicOr(&s_mask[(m0 & IM) >> 5], 1u << (m0 & 31));
        if (K > 1) atomicOr(&s_mask[(m1 & IM) >> 5], 1u << (m1 & 31));
        if (K > 2 && d2 == d1) atomicOr(&s_mask[(m2 & IM) >> 5], 1u << (m2 & 31));
        if (K > 3 && d3 == d1)
            for (int j = 0; j < K; ++j) {
                const int dot = (int)__builtin_amdgcn_udot4(x4, pal.p4[j], 0u, false);
                const int dj = (pal.nkey[j] - (dot << (kIdxBits + 1))) >> kIdxBits;
                if (dj <= d1) atomicOr(&s_mask[j >> 5], 1u << (j & 31));
            }
    }
    __syncthreads();
    if (threadIdx.x < MW) masks[(size_t)blockIdx.x * MW + threadIdx.x] = s_mask[threadIdx.x];
}

// T(x) of every colour, OR-ed into the masks of its WARPED cell and 8^3 sub-cell (masks zeroed by the caller; same
// layout as accel_scan_kernel's: [cell = r'<<8 | g'<<4 | b'][9][MW]).  One block per plain cell of colours.
template <int MW>
__global__ __launch_bounds__(256) void accel_scan_warp_kernel(const PalDev pal, const uint8_t *__restrict__ lut,
                                                              uint32_t *__restrict__ masks)
{
    const int cell = blockIdx.x;
    const int rc = cell >> 8, gc = (cell >> 4) & 15, bc = cell & 15;
    const int K = pal.K;
    constexpr int kBig = 0x7fffffff;
    constexpr int IM = (1 << kIdxBits) - 1;
    for (int t = 0; t < 16; ++t) {
        const int id = threadIdx.x + 256 * t;
        const uint32_t r = rc * 16 + (id >> 8), g = gc * 16 + ((id >> 4) & 15), b = bc * 16 + (id & 15);
        const uint32_t x4 = r | (g << 8) | (b << 16);
        int m0 = kBig, m1 = kBig, m2 = kBig, m3 = kBig;
        for (int j = 0; j < K; ++j) {
            const int dot = (int)__builtin_amdgcn_udot4(x4, pal.p4[j], 0u, false);
            const int key = pal.nkey[j] - (dot << (kIdxBits + 1));
            const int n3 = med3i(m2, m3, key);
            const int n2 = med3i(m1, m2, key);
            const int n1 = med3i(m0, m1, key);
            m0 = min(m0, key);
            m1 = n1;
            m2 = n2;
            m3 = n3;
        }
        const int d1 = m1 >> kIdxBits, d2 = m2 >> kIdxBits, d3 = m3 >> kIdxBits;
        const uint32_t ur = lut[r], ug = lut[256 + g], ub = lut[512 + b];
        const uint32_t wcell = ((ur >> 4) << 8) | ((ug >> 4) << 4) | (ub >> 4);
        const uint32_t sub = 1u + ((((ur >> 3) & 1u) << 2) | (((ug >> 3) & 1u) << 1) | ((ub >> 3) & 1u));
        uint32_t *whole = masks + (size_t)wcell * 9 * MW, *part = whole + (size_t)sub * MW;
        auto mark = [&](int j) {
            const uint32_t bit = 1u << (j & 31);
            if (!(whole[j >> 5] & bit)) atomicOr(&whole[j >> 5], bit);
            if (!(part[j >> 5] & bit)) atomicOr(&part[j >> 5], bit);
        };
        mark(m0 & IM);
        if (K > 1) mark(m1 & IM);
        if (K > 2 && d2 == d1) mark(m2 & IM);
        if (K > 3 && d3 == d1)
            for (int j = 0; j < K; ++j) {
                const int dot = (int)__builtin_amdgcn_udot4(x4, pal.p4[j], 0u, false);
                const int dj = (pal.nkey[j] - (dot << (kIdxBits + 1))) >> kIdxBits;
                if (dj <= d1) mark(j);
            }
    }
}


// ---- float (gamma) palettes ----------------------------------------------------------------------
// Same cell lists for palettes whose coordinates are not integers (use_gamma: float32 linearised colours,
// pixels looked up through lut_in first).  The dither kernel ranks a cell's candidates in float32 and
// accepts the ranking only when neighbouring keys differ by more than 128 ulp (>= 7.6e-6 relative; its own
// rounding error is < 1.5e-6), otherwise the pixel goes to the fix-up pass.  So a list must hold every entry
// within that margin of the second smallest distance: T'(x) = { j : d_j <= d_(2) * (1 + 2^-14) }, d in
// float64 exactly as scipy computes it.  Colours no pixel can take (values outside lut_in's image) are skipped.
constexpr double kFloatListMargin = 1.0 + 6.103515625e-5;  // 1 + 2^-14

__device__ __forceinline__ void float_members(const double *s_pts, const int K, const double r, const double g,
                                              const double b, uint32_t *mask_a, uint32_t *mask_b)
{
    const double inf = __longlong_as_double(0x7ff0000000000000LL);
    double b0 = inf, b1 = inf;
    for (int j = 0; j < K; ++j) {
        const double d = sq_dist3(s_pts + 3 * j, r, g, b);
        if (d < b1) {
            if (d < b0) {
                b1 = b0;
                b0 = d;
            } else {
                b1 = d;
            }
        }
    }
    const double bound = __dmul_rn(b1, kFloatListMargin);
    for (int j = 0; j < K; ++j) {
        const double d = sq_dist3(s_pts + 3 * j, r, g, b);
        if (d <= bound) {
            atomicOr(&mask_a[j >> 5], 1u << (j & 31));
            if (mask_b) atomicOr(&mask_b[j >> 5], 1u << (j & 31));
        }
    }
}

__global__ __launch_bounds__(256) void accel_scan_float_kernel(const PalDev pal, const uint8_t *__restrict__ reach,
                                                               uint32_t *__restrict__ masks)
{
    __shared__ uint32_t s_mask[9][8];
    __shared__ double s_pts[256 * 3];
    if (threadIdx.x < 72) (&s_mask[0][0])[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < pal.K * 3; i += 256) s_pts[i] = pal.pts[i];
    __syncthreads();
    const int cell = blockIdx.x;
    const int rc = cell >> 8, gc = (cell >> 4) & 15, bc = cell & 15;
    for (int t = 0; t < 16; ++t) {
        const int id = threadIdx.x + 256 * t;
        const int r = rc * 16 + (id >> 8), g = gc * 16 + ((id >> 4) & 15), b = bc * 16 + (id & 15);
        if (!(reach[r] && reach[g] && reach[b])) continue;
        const int sub = 1 + (((((id >> 8) >> 3) & 1) << 2) | (((((id >> 4) & 15) >> 3) & 1) << 1) | (((id & 15) >> 3) & 1));
        float_members(s_pts, pal.K, (double)r, (double)g, (double)b, s_mask[0], s_mask[sub]);
    }
    __syncthreads();
    if (threadIdx.x < 72) masks[cell * 72 + threadIdx.x] = (&s_mask[0][0])[threadIdx.x];
}

__global__ __launch_bounds__(64) void accel_box_float_kernel(const PalDev pal, const uint8_t *__restrict__ reach,
                                                             const Box *__restrict__ boxes, uint32_t *__restrict__ masks)
{
    __shared__ uint32_t s_mask[8];
    __shared__ double s_pts[256 * 3];
    if (threadIdx.x < 8) s_mask[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < pal.K * 3; i += 64) s_pts[i] = pal.pts[i];
    __syncthreads();
    const Box bx = boxes[blockIdx.x];
    const int n = bx.size * bx.size * bx.size;
    for (int id = threadIdx.x; id < n; id += 64) {
        const int r = bx.r0 + id / (bx.size * bx.size), g = bx.g0 + (id / bx.size) % bx.size, b = bx.b0 + id % bx.size;
        if (!(reach[r] && reach[g] && reach[b])) continue;
        float_members(s_pts, pal.K, (double)r, (double)g, (double)b, s_mask, nullptr);
    }
    __syncthreads();
    if (threadIdx.x < 8) masks[blockIdx.x * 8 + threadIdx.x] = s_mask[threadIdx.x];
}

}  // namespace

namespace {

// (assemble_table, crowded_nodes_first, make_warp, mass_points, entries_in_split_cells and their constants: host_logic.h)

// runs `launch(d_boxes, d_masks, n)` over a list of boxes and brings the masks back
template <class B, class Launch>
int run_box_kernel(const std::vector<B> &boxes, std::vector<uint32_t> &bm, Launch launch)
{
    B *d_boxes = nullptr;
    uint32_t *d_bm = nullptr;
    hipError_t e = hipMalloc((void **)&d_boxes, sizeof(B) * boxes.size());
    if (e == hipSuccess) e = hipMalloc((void **)&d_bm, sizeof(uint32_t) * bm.size());
    if (e == hipSuccess) e = hipMemcpy(d_boxes, boxes.data(), sizeof(B) * boxes.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        launch(d_boxes, d_bm, (unsigned)boxes.size());
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(bm.data(), d_bm, sizeof(uint32_t) * bm.size(), hipMemcpyDeviceToHost);
    if (d_boxes) (void)hipFree(d_boxes);
    if (d_bm) (void)hipFree(d_bm);
    if (e != hipSuccess) return hip_fail(e, "accelerator refinement");
    return DP_OK;
}

}  // namespace

// Builds the accelerator for an integer palette whose output colours equal its search colours.
// On success fills the accel fields of `dev` and returns the device allocation through *blob_out.
int build_accel(PalDev &dev, const std::vector<uint32_t> &p4_host, void **blob_out, size_t *blob_bytes)
{
    *blob_out = nullptr;
    *blob_bytes = 0;
    const int K = dev.K;
    constexpr size_t kCodeWords = (size_t)1 << 20;  // 2^24 colours x 2 bits (k=1); the k=2 table has twice as many
    uint32_t *d_masks = nullptr;
    uint8_t *blob = nullptr;
    const int mw = K <= 256 ? 8 : 32;               // mask words per (sub-)cell
    const bool big_q = dev.n_inner > kQueueSmall;   // traversal queue of the tie queries
    // layout: code1 | code2 | table[max] | exceptions | exception count | table of 4-entry blocks [cap] |
    //         table over warped cells [max] | the three warp maps [768 bytes] | staging orders of the 8- and the
    //         4-entry table [2 x 4096 words]
    //         | the flat lists of their split cells [2 x kWideCap x kWideList words]
    //         | the compact (one byte per entry) copy of the crowded table [kCompactMaxWords]
    const size_t bytes = sizeof(uint32_t) * (3 * kCodeWords + 2 * kTabMaxWords + kTabCapWords) + sizeof(uint4) * kExcCap + 16 + 768 +
                         sizeof(uint32_t) * (2 * kCells + 2 * kWideCap * kWideList + kCompactMaxWords);
    DP_HIP(hipMalloc((void **)&blob, bytes));
    struct DevFree {  // frees the scan masks on every way out
        void *p = nullptr;
        ~DevFree()
        {
            if (p) (void)hipFree(p);
        }
    } masks_guard;
    hipError_t e = hipMalloc((void **)&d_masks, sizeof(uint32_t) * kCells * 10 * mw);  // T masks [9] + N mask [1] per cell
    masks_guard.p = d_masks;
    uint32_t *d_nmasks = d_masks + (size_t)kCells * 9 * mw;
    if (e == hipSuccess) e = hipMemset(blob, 0, sizeof(uint32_t) * 3 * kCodeWords);
    if (e != hipSuccess) {
        (void)hipFree(blob);
        return hip_fail(e, "accelerator allocation");
    }
    uint32_t *code1 = reinterpret_cast<uint32_t *>(blob);
    uint32_t *code2 = code1 + kCodeWords;
    uint32_t *d_tab = code2 + 2 * kCodeWords;
    uint4 *d_exc = reinterpret_cast<uint4 *>(d_tab + kTabMaxWords);  // 16-byte aligned: every part is a multiple of 16
    uint32_t *d_exc_count = reinterpret_cast<uint32_t *>(d_exc + kExcCap);
    e = hipMemset(d_exc_count, 0, sizeof(uint32_t));
    if (e != hipSuccess) {
        (void)hipFree(blob);
        return hip_fail(e, "accelerator allocation");
    }
#define DP_SCAN(MW, C) hipLaunchKernelGGL((accel_scan_kernel<MW, C>), dim3(kCells), dim3(256), 0, 0, dev, d_masks, d_nmasks, code1, code2, d_exc, d_exc_count)
    if (mw == 8) {
        if (big_q) DP_SCAN(8, kQueueLarge); else DP_SCAN(8, kQueueSmall);
    } else {
        if (big_q) DP_SCAN(32, kQueueLarge); else DP_SCAN(32, kQueueSmall);
    }
#undef DP_SCAN
    std::vector<uint32_t> masks((size_t)kCells * 9 * mw), nmasks((size_t)kCells * mw);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(masks.data(), d_masks, sizeof(uint32_t) * masks.size(), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(nmasks.data(), d_nmasks, sizeof(uint32_t) * nmasks.size(), hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        (void)hipFree(blob);
        return hip_fail(e, "accelerator scan");
    }

    std::vector<uint32_t> tab;
    TableStats st;
    auto box_masks = [&](const std::vector<Box> &boxes, std::vector<uint32_t> &bm) {
        return run_box_kernel(boxes, bm, [&](const Box *db, uint32_t *dm, unsigned n) {
            if (mw == 8) hipLaunchKernelGGL(accel_box_kernel<8>, dim3(n), dim3(64), 0, 0, dev, db, dm);
            else hipLaunchKernelGGL(accel_box_kernel<32>, dim3(n), dim3(64), 0, 0, dev, db, dm);
        });
    };
    // small palettes: most cells hold at most 4 candidates; a table of 4-entry blocks halves the work of the dither
    // kernel as long as few cells overflow into splits (their pixels take the deferred path)
    std::vector<uint32_t> tab4, perm4, perm8, wide4, wide8;
    TableStats st4;
    bool use4 = false;
    if (K <= 64) {
        // (staging order with the nearest set first; four slots hold any nearest set that fits the block)
        const int rc4 = assemble_table(masks, mw, 4, kTabCapWords, K, p4_host, p4_host, box_masks, tab4, st4, nmasks.data(), 4, &perm4, &wide4);
        if (rc4 != DP_OK) {
            (void)hipFree(blob);
            return rc4;
        }
        // worth it while few pixels fall into split cells (they take the deferred path): up to 4 % of the cells.
        // Single colours with more than 4 equidistant entries (marked slow; uniform grids have them) are resolved by
        // the deferred path with a scan of the whole (small) palette.
        use4 = !st4.too_big && st4.n_split_cells <= kCells * 4 / 100;
        if (exp_env("DP_DEBUG_ACCEL"))
            fprintf(stderr, "accel K=%d: 4-entry table: %d split cells, %d split nodes, %d slow, %zu words, too_big=%d -> use=%d\n", K,
                    st4.n_split_cells, st4.n_split, st4.n_slow, tab4.size(), (int)st4.too_big, (int)use4);
    }
    // the table of 8-entry blocks (palettes of fewer than 8 colours cannot fill a block: 4-entry blocks only)
    const bool have8 = K >= 8;
    if (have8) {
        const int rc = assemble_table(masks, mw, 8, kTabMaxWords, K, p4_host, p4_host, box_masks, tab, st, nmasks.data(), kNearSlots, &perm8, &wide8);
        if (rc != DP_OK) {
            (void)hipFree(blob);
            return rc;
        }
        if (exp_env("DP_DEBUG_ACCEL"))
            fprintf(stderr, "accel K=%d: 8-entry table: %d split cells, %d split nodes, %d slow, %zu words (%s), longest list %d, too_big=%d\n",
                    K, st.n_split_cells, st.n_split, st.n_slow, tab.size(), tab.size() <= (size_t)kTabCapWords ? "all in LDS" : "deep nodes in global memory",
                    st.max_cnt, (int)st.too_big);
    }
    if ((!have8 || st.too_big) && !use4) {
        (void)hipFree(blob);
        return DP_OK;  // no table that fits LDS: the brute-force kernel stays in charge
    }
    // Where an image's pixels are when the palette was extracted from that image: at the palette entries and between
    // neighbouring entries.  The share of these points that sits in split cells tells how many pixels would leave the
    // main path of the dither kernel.
    const std::vector<uint32_t> mass = mass_points(p4_host);
    const int n_mass = (int)mass.size();
    const bool ok8 = have8 && !st.too_big;
    const bool u8_spilled = ok8 && tab.size() > (size_t)kTabCapWords;
    const int m4 = use4 ? entries_in_split_cells(tab4, 4, mass) : n_mass;
    const int m8 = ok8 ? entries_in_split_cells(tab, 8, mass) : n_mass;
    // the 4-entry table loses its advantage when pixels crowd its split cells (more than 5 % of the mass) and the 8-entry
    // table keeps them on the main path
    if (use4 && ok8 && m4 * 20 > n_mass && m8 * 2 <= m4) use4 = false;
    // experiments: DP_FORCE_TABLE = u4 | u8 | w4 | w8 picks the table regardless of the estimates (when it exists)
    const char *force_env = exp_env("DP_FORCE_TABLE");
    const std::string force = force_env ? force_env : "";
    if (force == "u4") use4 = K <= 64 && !st4.too_big;
    else if (!force.empty() && ok8) use4 = false;
    const int u_mass = use4 ? m4 : m8;
    const int u_cells = use4 ? st4.n_split_cells : st.n_split_cells;
    const bool u_spilled = !use4 && u8_spilled;
    const bool u_crowded = u_spilled || u_cells > kCells * 3 / 100 || u_mass * 4 >= n_mass;
    // worth a second table: the table in use is crowded, or more than 2 % of the mass sits in its split cells (a table over
    // warped cells costs the kernel three more LDS reads per pixel, so it has to win something)
    const bool try_warp = (u_crowded && !(use4 && m4 * 50 <= n_mass)) || u_mass * 50 > n_mass;
    // Crowded palettes (extracted from an image: the colours sit where the pixels are, in a few cells that hold far more
    // than a block): build the table a second time over warped cells and keep it when it leaves at most half as much
    // of the mass in split cells.
    std::vector<uint32_t> wtab;
    TableStats wst;
    WarpMaps wm;
    int wbw = 0, w_mass = 0;
    bool w_spilled = false;
    uint32_t *d_wtab = d_exc_count + 4 + kTabCapWords;
    uint8_t *d_lut = reinterpret_cast<uint8_t *>(d_wtab + kTabMaxWords);
    if (K >= 8 && (try_warp || force[0] == 'w') && force[0] != 'u' && !exp_env("DP_NO_WARP")) {
        make_warp(p4_host, wm);
        auto warped = [&](const uint32_t c) {
            return (uint32_t)wm.lut[0][c & 255] | ((uint32_t)wm.lut[1][(c >> 8) & 255] << 8) | ((uint32_t)wm.lut[2][(c >> 16) & 255] << 16);
        };
        std::vector<uint32_t> coordw(K), massw(mass.size());
        for (int j = 0; j < K; ++j) coordw[j] = warped(p4_host[j]);
        for (size_t j = 0; j < mass.size(); ++j) massw[j] = warped(mass[j]);
        e = hipMemcpy(d_lut, &wm.lut[0][0], 768, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemset(d_masks, 0, sizeof(uint32_t) * kCells * 9 * mw);
        if (e == hipSuccess) {
            if (mw == 8) hipLaunchKernelGGL(accel_scan_warp_kernel<8>, dim3(kCells), dim3(256), 0, 0, dev, d_lut, d_masks);
            else hipLaunchKernelGGL(accel_scan_warp_kernel<32>, dim3(kCells), dim3(256), 0, 0, dev, d_lut, d_masks);
            e = hipGetLastError();
        }
        std::vector<uint32_t> wmasks((size_t)kCells * 9 * mw);
        if (e == hipSuccess) e = hipMemcpy(wmasks.data(), d_masks, sizeof(uint32_t) * wmasks.size(), hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            (void)hipFree(blob);
            return hip_fail(e, "accelerator scan (warped cells)");
        }
        auto box_masks_w = [&](const std::vector<Box> &boxes, std::vector<uint32_t> &bm) {
            std::vector<Range> ranges(boxes.size());
            for (size_t q = 0; q < boxes.size(); ++q) {
                const Box &b = boxes[q];
                ranges[q] = Range{wm.lo[0][b.r0], wm.lo[0][b.r0 + b.size], wm.lo[1][b.g0], wm.lo[1][b.g0 + b.size],
                                  wm.lo[2][b.b0], wm.lo[2][b.b0 + b.size]};
            }
            return run_box_kernel(ranges, bm, [&](const Range *db, uint32_t *dm, unsigned n) {
                if (mw == 8) hipLaunchKernelGGL(accel_range_kernel<8>, dim3(n), dim3(256), 0, 0, dev, db, dm);
                else hipLaunchKernelGGL(accel_range_kernel<32>, dim3(n), dim3(256), 0, 0, dev, db, dm);
            });
        };
        // (the warp maps take 768 bytes of the kernels' LDS)
        int w4_mass = -1;
        if (K <= 64 && force != "w8") {
            TableStats s4;
            std::vector<uint32_t> t4;
            const int rc4 = assemble_table(wmasks, mw, 4, kTabCapWords - 192, K, coordw, p4_host, box_masks_w, t4, s4);
            if (rc4 != DP_OK) {
                (void)hipFree(blob);
                return rc4;
            }
            w4_mass = s4.too_big ? n_mass : entries_in_split_cells(t4, 4, massw);
            // 4-entry blocks only when practically nothing leaves the main path (else 8-entry blocks are the safer choice)
            if (!s4.too_big && ((s4.n_split_cells <= kCells / 100 && s4.n_slow == 0 && w4_mass * 50 <= n_mass) || force == "w4")) {
                wtab.swap(t4);
                wst = s4;
                wbw = 4;
                w_mass = w4_mass;
            }
        }
        if (wbw == 0) {
            const int rc8 = assemble_table(wmasks, mw, 8, kTabMaxWords, K, coordw, p4_host, box_masks_w, wtab, wst);
            if (rc8 != DP_OK) {
                (void)hipFree(blob);
                return rc8;
            }
            if (!wst.too_big) {
                wbw = 8;
                w_spilled = wtab.size() > (size_t)(kTabCapWords - 192);
                w_mass = entries_in_split_cells(wtab, 8, massw);
                if (w_spilled) crowded_nodes_first(wtab, wst, 8, coordw);
            }
        }
        if (exp_env("DP_DEBUG_ACCEL"))
            fprintf(stderr, "accel K=%d: mass in split cells: plain 4-entry %d, plain 8-entry %d of %d (in use: %d-entry, %d cells%s); warped 4-entry %d, "
                    "warped %d-entry table: %d, %d split cells, %d nodes, %d slow, %zu words%s\n",
                    K, m4, m8, n_mass, use4 ? 4 : 8, u_cells, u_spilled ? ", spilled" : "", w4_mass, wbw, w_mass, wst.n_split_cells, wst.n_split,
                    wst.n_slow, wtab.size(), w_spilled ? " (deep nodes in global memory)" : "");
        // no real gain (less than half of the mass brought back, or less than 2 % of it): stay with the plain cells
        if (wbw != 0 && (w_mass * 2 > u_mass || (u_mass - w_mass) * 50 < n_mass) && !(u_spilled && !w_spilled) && force.empty()) wbw = 0;
    } else if (exp_env("DP_DEBUG_ACCEL")) {
        fprintf(stderr, "accel K=%d: mass in split cells: plain 4-entry %d, plain 8-entry %d of %d (in use: %d-entry, %d cells)\n", K, m4, m8, n_mass,
                use4 ? 4 : 8, u_cells);
    }
    dev.warp_tab = nullptr;
    dev.warp_lut = nullptr;
    dev.warp_words = dev.warp_total = dev.warp_bw = dev.warp_adapt = 0;
    dev.adapt = (!use4 && u_crowded) ? 1 : 0;
    if (wbw != 0) {
        e = hipMemcpy(d_wtab, wtab.data(), sizeof(uint32_t) * wtab.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(blob);
            return hip_fail(e, "accelerator upload (warped cells)");
        }
        dev.warp_tab = d_wtab;
        dev.warp_lut = d_lut;
        dev.warp_bw = wbw;
        dev.warp_total = (int)wtab.size();
        dev.warp_words = w_spilled ? std::min((int)wtab.size(), kTabStageWords - 3 * 64) : (int)wtab.size();
        dev.warp_adapt = (wbw == 8 && (w_spilled || wst.n_split_cells > kCells * 3 / 100 || w_mass * 4 >= n_mass)) ? 1 : 0;
    }
    // The fast ordered kernel (ordered.hip: ordered_fast_kernel) runs the pixels that can only take their nearest entry on
    // the first kNearSlots entries of the cell's block, staged nearest set first (perm8 / perm4 from assemble_table); it is
    // used with the plain tables of uncrowded palettes only.
    dev.cell_perm = dev.cell_perm4 = nullptr;
    dev.near_slots = 0;
    const bool fast8 = have8 && !st.too_big && !st.wide_overflow && !use4 && wbw == 0 && !dev.adapt && tab.size() <= (size_t)kTabCapWords &&
                       wide8.size() <= (size_t)kWideCap * kWideList && !exp_env("DP_NO_FAST");
    if (exp_env("DP_DEBUG_ACCEL") && have8)
        fprintf(stderr, "accel K=%d: fast kernel on the 8-entry table: %s (%d cells with a nearest set above %d entries, %zu flat lists)\n", K,
                fast8 ? "yes" : "no", st.n_near_overflow, kNearSlots, wide8.size() / kWideList);
    if (!fast8) perm8.clear();
    uint32_t *d_perm8 = reinterpret_cast<uint32_t *>(d_lut + 768), *d_perm4 = d_perm8 + kCells;
    uint32_t *d_wide8 = d_perm4 + kCells, *d_wide4 = d_wide8 + kWideCap * kWideList;
    dev.cell_wide = dev.cell_wide4 = nullptr;
    dev.n_wide = dev.n_wide4 = 0;
    if (!perm8.empty()) {
        e = hipMemcpy(d_perm8, perm8.data(), sizeof(uint32_t) * kCells, hipMemcpyHostToDevice);
        if (e == hipSuccess && !wide8.empty()) e = hipMemcpy(d_wide8, wide8.data(), sizeof(uint32_t) * wide8.size(), hipMemcpyHostToDevice);
        dev.cell_wide = d_wide8;
        dev.n_wide = (int)(wide8.size() / kWideList);
        if (e != hipSuccess) {
            (void)hipFree(blob);
            return hip_fail(e, "accelerator upload (staging order)");
        }
        dev.cell_perm = d_perm8;
        dev.near_slots = kNearSlots;
    }
    if (use4 && wbw == 0 && !perm4.empty() && !st4.wide_overflow && wide4.size() <= (size_t)kWideCap * kWideList && !exp_env("DP_NO_FAST")) {
        e = hipMemcpy(d_perm4, perm4.data(), sizeof(uint32_t) * kCells, hipMemcpyHostToDevice);
        if (e == hipSuccess && !wide4.empty()) e = hipMemcpy(d_wide4, wide4.data(), sizeof(uint32_t) * wide4.size(), hipMemcpyHostToDevice);
        dev.cell_wide4 = d_wide4;
        dev.n_wide4 = (int)(wide4.size() / kWideList);
        if (e != hipSuccess) {
            (void)hipFree(blob);
            return hip_fail(e, "accelerator upload (staging order)");
        }
        dev.cell_perm4 = d_perm4;
    }
    dev.cell_tab = nullptr;
    dev.tab_words = 0;
    dev.tab_total = 0;
    if (have8 && !st.too_big && tab.size() > (size_t)kTabCapWords) crowded_nodes_first(tab, st, 8, p4_host);
    // Crowded palettes: the table the adaptive lean kernel would run on (8-entry blocks over warped or plain cells), one
    // byte per entry, for ordered_compact_kernel (everything in LDS, no deferral)
    dev.comp_tab = nullptr;
    dev.comp_words = dev.comp_warp = 0;
    {
        const bool from_warp = wbw == 8 && (dev.warp_adapt != 0 || exp_env("DP_FORCE_COMPACT"));
        const bool from_plain = wbw == 0 && (dev.adapt != 0 || exp_env("DP_FORCE_COMPACT")) && have8 && !st.too_big;
        const std::vector<uint32_t> &src = from_warp ? wtab : tab;
        if (K <= 256 && (from_warp || from_plain) && src.size() / 4 <= (size_t)kCompactMaxWords && !exp_env("DP_NO_COMPACT")) {
            const std::vector<uint32_t> ct = compact_table(src, p4_host);
            uint32_t *d_comp = d_wide4 + kWideCap * kWideList;
            e = hipMemcpy(d_comp, ct.data(), sizeof(uint32_t) * ct.size(), hipMemcpyHostToDevice);
            if (e != hipSuccess) {
                (void)hipFree(blob);
                return hip_fail(e, "accelerator upload (compact table)");
            }
            dev.comp_tab = d_comp;
            dev.comp_words = (int)ct.size();
            dev.comp_warp = from_warp ? 1 : 0;
            if (exp_env("DP_DEBUG_ACCEL"))
                fprintf(stderr, "accel K=%d: compact table over %s cells: %zu bytes\n", K, from_warp ? "warped" : "plain", ct.size() * 4);
        }
    }
    if (have8 && !st.too_big) {
        e = hipMemcpy(d_tab, tab.data(), sizeof(uint32_t) * tab.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(blob);
            return hip_fail(e, "accelerator upload");
        }
        dev.cell_tab = d_tab;
        dev.tab_total = (int)tab.size();
        dev.tab_words = tab.size() <= (size_t)kTabCapWords ? (int)tab.size() : std::min((int)tab.size(), kTabStageWords);
    } else {
        st = st4;  // statistics of the table in use
    }
    dev.cell_tab4 = nullptr;
    dev.tab4_words = 0;
    if (use4) {
        uint32_t *d_tab4 = d_exc_count + 4;
        e = hipMemcpy(d_tab4, tab4.data(), sizeof(uint32_t) * tab4.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            dev.cell_tab = nullptr;
            (void)hipFree(blob);
            return hip_fail(e, "accelerator upload");
        }
        dev.cell_tab4 = d_tab4;
        dev.tab4_words = (int)tab4.size();
    }
    dev.n_split = st.n_split;
    dev.n_slow_blocks = st.n_slow;
    dev.n_split_cells = st.n_split_cells;
    dev.max_cell = st.max_cnt;
    dev.code1 = code1;
    dev.code2 = code2;
    {
        // the exception list, sorted by colour for the kernels' binary search
        uint32_t n_exc = 0;
        e = hipMemcpy(&n_exc, d_exc_count, sizeof(uint32_t), hipMemcpyDeviceToHost);
        if (e == hipSuccess && n_exc > 0 && n_exc <= (uint32_t)kExcCap) {
            std::vector<uint4> ex(n_exc);
            e = hipMemcpy(ex.data(), d_exc, sizeof(uint4) * n_exc, hipMemcpyDeviceToHost);
            std::sort(ex.begin(), ex.end(), [](const uint4 &a, const uint4 &b) { return a.x < b.x; });
            if (e == hipSuccess) e = hipMemcpy(d_exc, ex.data(), sizeof(uint4) * n_exc, hipMemcpyHostToDevice);
        }
        if (e != hipSuccess) {
            dev.cell_tab = nullptr;
            (void)hipFree(blob);
            return hip_fail(e, "accelerator exception list");
        }
        dev.exc = d_exc;
        dev.n_exc = n_exc <= (uint32_t)kExcCap ? (int)n_exc : -1;
    }
    *blob_out = blob;
    *blob_bytes = bytes;
    return DP_OK;
}

// The same for a palette with float coordinates (use_gamma).  pal_f32: K x 3 as the tree sees them; lut_host:
// the 256-entry input table or nullptr.  Table words are byte offsets (16 * j) into dev.fcand.
int build_accel_float(PalDev &dev, const float *pal_f32, const uint8_t *lut_host, void **blob_out, size_t *blob_bytes)
{
    *blob_out = nullptr;
    *blob_bytes = 0;
    const int K = dev.K;
    std::vector<uint8_t> reach(256, lut_host ? 0 : 1);
    if (lut_host)
        for (int v = 0; v < 256; ++v) reach[lut_host[v]] = 1;
    std::vector<uint32_t> coord4(K), word(K);
    for (int j = 0; j < K; ++j) {
        uint32_t c[3];
        for (int k = 0; k < 3; ++k) {
            const float v = pal_f32[3 * j + k];
            c[k] = (uint32_t)std::min(255.0f, std::max(0.0f, std::nearbyint(v)));
        }
        coord4[j] = c[0] | (c[1] << 8) | (c[2] << 16);
        word[j] = (uint32_t)j * 16u;
    }
    // layout: table[max] | reach[256] | compact table [kCompactMaxWords]
    const size_t bytes = sizeof(uint32_t) * kTabMaxWords + 256 + sizeof(uint32_t) * kCompactMaxWords;
    uint8_t *blob = nullptr;
    uint32_t *d_masks = nullptr;
    DP_HIP(hipMalloc((void **)&blob, bytes));
    uint32_t *d_tab = reinterpret_cast<uint32_t *>(blob);
    uint8_t *d_reach = blob + sizeof(uint32_t) * kTabMaxWords;
    hipError_t e = hipMalloc((void **)&d_masks, sizeof(uint32_t) * kCells * 72);
    if (e == hipSuccess) e = hipMemcpy(d_reach, reach.data(), 256, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(blob);
        if (d_masks) (void)hipFree(d_masks);
        return hip_fail(e, "float accelerator allocation");
    }
    hipLaunchKernelGGL(accel_scan_float_kernel, dim3(kCells), dim3(256), 0, 0, dev, d_reach, d_masks);
    std::vector<uint32_t> masks((size_t)kCells * 72);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(masks.data(), d_masks, sizeof(uint32_t) * masks.size(), hipMemcpyDeviceToHost);
    (void)hipFree(d_masks);
    if (e != hipSuccess) {
        (void)hipFree(blob);
        return hip_fail(e, "float accelerator scan");
    }
    // a cell none of whose colours can occur has an empty mask: it gets eight padding entries
    std::vector<uint32_t> tab;
    TableStats st;
    const int rc = assemble_table(
        masks, 8, 8, kTabMaxWords, K, coord4, word,
        [&](const std::vector<Box> &boxes, std::vector<uint32_t> &bm) {
            return run_box_kernel(boxes, bm, [&](const Box *db, uint32_t *dm, unsigned n) {
                hipLaunchKernelGGL(accel_box_float_kernel, dim3(n), dim3(64), 0, 0, dev, d_reach, db, dm);
            });
        },
        tab, st);
    if (rc != DP_OK || st.too_big) {
        (void)hipFree(blob);
        return rc;  // too big even for global memory: the brute-force kernel stays in charge
    }
    if ((int)tab.size() > (160 * 1024 - K * 16 - 256) / 4) crowded_nodes_first(tab, st, 8, coord4);
    e = hipMemcpy(d_tab, tab.data(), sizeof(uint32_t) * tab.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(blob);
        return hip_fail(e, "float accelerator upload");
    }
    dev.ftab = d_tab;
    dev.ftab_total = (int)tab.size();
    // staged in LDS next to the candidate table (16 bytes per colour) and lut_in: everything that fits, else the 4096
    // cell blocks and the first split nodes
    const int stage_cap = (160 * 1024 - K * 16 - 256) / 4;
    dev.ftab_words = (int)tab.size() <= stage_cap ? (int)tab.size() : 4096 * 8 + ((stage_cap - 4096 * 8) / 64) * 64;
    dev.n_split = st.n_split;
    dev.n_slow_blocks = st.n_slow;
    dev.max_cell = st.max_cnt;
    // the same table with one byte per entry (the index of the candidate record) for ordered_compact_float_kernel
    dev.comp_tab = nullptr;
    dev.comp_words = dev.comp_warp = 0;
    if (K <= 256 && tab.size() / 4 <= (size_t)kCompactMaxWords && !exp_env("DP_NO_COMPACT")) {
        const std::vector<uint32_t> ct = compact_table(tab, word);  // (word[j] = 16 j: unique, so entry -> j)
        uint32_t *d_comp = reinterpret_cast<uint32_t *>(d_reach + 256);
        e = hipMemcpy(d_comp, ct.data(), sizeof(uint32_t) * ct.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(blob);
            dev.ftab = nullptr;
            return hip_fail(e, "float accelerator upload (compact table)");
        }
        dev.comp_tab = d_comp;
        dev.comp_words = (int)ct.size();
    }
    *blob_out = blob;
    *blob_bytes = bytes;
    return DP_OK;
}

}  // namespace dp
