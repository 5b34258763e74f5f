// Nearest palette entry of an arbitrary float32 point, as scipy's KD-tree query (k=1) reports it: float32 prefilter,
// float64 validation, tree replay for exact ties; optionally restricted to the candidate list of the point's cell
// (ediff.hip: build_ed_cells).  Shared by the error-diffusion kernels (ediff.hip) and the variable-weight diffusers
// (vardiff.hip).
#pragma once
#include "dp_internal.h"
#include "tree_query.hip.h"

namespace dp {
namespace {

// Naming a candidate record's fourth word keeps its LDS read a ds_read_b128: four LDS cycles per wave (groups of 16 lanes
// over 64 banks), where the ds_read_b96 the compiler narrows a three-component use to takes eight (groups of 8 over 32
// banks; MI355X_MICROARCH.md, LDS).  Found on ordered_compact_float_kernel in round 3: 1.22 -> 0.94 ms from this alone.
__device__ __forceinline__ void keep_record_whole(const float4 &c) { asm volatile("" ::"v"(c.w)); }

// float32 squared distance to a candidate as an integer key, the low 3 bits replaced by `tag`
__device__ __forceinline__ int ed_key(const float4 c, const float o0, const float o1, const float o2, const uint32_t tag)
{
    keep_record_whole(c);
    const float a = c.x - o0, b = c.y - o1, cc = c.z - o2;
    const float d = __fmaf_rn(a, a, __fmaf_rn(b, b, cc * cc));
    return (int)((__float_as_uint(d) & ~7u) | tag);
}

__device__ __forceinline__ int ed_key16(const float4 c, const float o0, const float o1, const float o2, const uint32_t tag)
{
    keep_record_whole(c);
    const float a = c.x - o0, b = c.y - o1, cc = c.z - o2;
    const float d = __fmaf_rn(a, a, __fmaf_rn(b, b, cc * cc));
    return (int)((__float_as_uint(d) & ~15u) | tag);
}

// The same key from a candidate's EXPANDED record {-2x, -2y, -2z, |c|^2 + 2^18} (ediff.hip, palettes of up to 16 colours):
// |c - o|^2 - |o|^2 + 2^18 in three fused multiply-adds, four instructions per candidate with the tag instead of seven.  The
// term |o|^2 is common to all candidates and the bias keeps the value inside [2^16, 2^19) for points of the cube -- positive,
// so the bit patterns order like the values, and with an ulp of at most 2^-5: one rounding of the fourth word (float
// palettes; integer ones are exact), one per multiply-add and up to 8 ulp from the tag bits are below 0.32 in absolute
// terms per key (ed_expanded_margin).
__device__ __forceinline__ int ed_key_expanded(const float4 x, const float o0, const float o1, const float o2, const uint32_t tag)
{
    keep_record_whole(x);
    const float d = __fmaf_rn(x.x, o0, __fmaf_rn(x.y, o1, __fmaf_rn(x.z, o2, x.w)));
    return (int)((__float_as_uint(d) & ~7u) | tag);
}
constexpr float kEdExpandedBias = 262144.0f;
constexpr float kEdExpandedMargin = 0.75f;  // > 2 * 0.32: a second key this far above the first proves the float64 order

__device__ __forceinline__ int ed_med3(const int a, const int b, const int c)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <int CAP>
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
        tree_query<1, CAP>(pal, x0, x1, x2, d2, ii);
        i0 = ii[0];
    }
    return i0;
}

// An exact tie at an integer point of the cube, integer palette with tie codes (accel.hip): the float32 distances are
// exact there, and which of several entries at the smallest distance scipy's traversal reports is tabulated (k=1 code:
// 0 / 1 / 2 = the first / second / third of them in index order).  Flat image regions under the adaptive-variance gate
// are exactly this case (no error arrives, the query is the pixel itself), and lattice or image-derived palettes tie on
// 5-20 % of such pixels; the generic answer (float64 scan + traversal replay with its heaps in scratch memory) costs
// microseconds per occurrence.  Returns -1 when the shortcut does not apply or the code says "other".
__device__ __forceinline__ int integer_tie_choice(const PalDev &pal, const float4 *__restrict__ cand, const float o0,
                                                  const float o1, const float o2, const float b0, const int i0)
{
    if (!pal.is_integer || pal.code1 == nullptr) return -1;
    if (!(o0 >= 0.0f && o0 <= 255.0f && o1 >= 0.0f && o1 <= 255.0f && o2 >= 0.0f && o2 <= 255.0f)) return -1;
    const uint32_t r = (uint32_t)o0, g = (uint32_t)o1, b = (uint32_t)o2;
    if ((float)r != o0 || (float)g != o1 || (float)b != o2) return -1;
    const uint32_t x = r | (g << 8) | (b << 16);
    const uint32_t code = (pal.code1[x >> 4] >> ((x & 15u) * 2)) & 3u;
    if (code == 0u) return i0;  // the lowest index among the tied entries (candidates are visited in index order)
    if (code == 3u) return -1;
    uint32_t seen = 0;
    for (int j = 0; j < pal.K; ++j) {
        const float4 c = cand[j];
        const float a = c.x - o0, bb = c.y - o1, cc = c.z - o2;
        const float d = __fmaf_rn(a, a, __fmaf_rn(bb, bb, cc * cc));
        if (d == b0) {
            if (seen == code) return j;
            ++seen;
        }
    }
    return -1;
}

// float32 prefilter: |d_f32 - d| <= ~3.6e-7 d (one rounding per subtract, square and add), so a gap of
// 2e-6 (relative) between the two smallest float32 distances proves the float64 order.
// `cand`: the palette as {x, y, z, out_rgb bits} -- in LDS for the wavefront kernel (every lane reads the same
// entry: one broadcast ds_read_b128 per colour, issued four at a time), in global memory for the serial one.
template <int CAP>
__device__ __forceinline__ int nearest_color(const PalDev &pal, const float4 *__restrict__ cand, const float o0,
                                             const float o1, const float o2)
{
    float b0 = __int_as_float(0x7f800000), b1 = b0;
    int i0 = 0;
    const int K = pal.K;
    auto visit = [&](const float4 c, const int j) {
        keep_record_whole(c);
        const float a = c.x - o0, b = c.y - o1, cc = c.z - o2;
        const float d = __fmaf_rn(a, a, __fmaf_rn(b, b, cc * cc));  // a filter only: any rounding within the margin
        const bool lt0 = d < b0;
        b1 = lt0 ? b0 : (d < b1 ? d : b1);
        i0 = lt0 ? j : i0;
        b0 = lt0 ? d : b0;
    };
    int j = 0;
    for (; j + 4 <= K; j += 4) {
        const float4 c0 = cand[j], c1 = cand[j + 1], c2 = cand[j + 2], c3 = cand[j + 3];
        visit(c0, j);
        visit(c1, j + 1);
        visit(c2, j + 2);
        visit(c3, j + 3);
    }
    for (; j < K; ++j) visit(cand[j], j);
    if (b1 > b0 * 1.000002f) return i0;
    if (b0 == b1) {
        const int tied = integer_tie_choice(pal, cand, o0, o1, o2, b0, i0);
        if (tied >= 0) return tied;
    }
    return nearest_f64<CAP>(pal, o0, o1, o2);
}

// nearest_color restricted to the cell's list (same validation, same fallbacks).
// `coarse` (wavefront kernel, palettes of 9..16 colours; else nullptr): an LDS copy of the lists of the 16x16x16 cells,
// count and up to 7 indices as nibbles of one word (count 15: longer than that) -- 99.8 % of the cells of a 16-colour
// palette; a step then needs no load from global memory at all (the 8x8x8 lists live in L2: ~600 cycles of latency that
// every step of the dependency chain would pay).
// `lists16` (wavefront kernel on the few-frames schedule, palettes of 17..256 colours; else nullptr): an LDS copy of the
// The hierarchical table of at most four entries per leaf (host_logic.h: EdTables::h4; palettes of 17..256 colours): the wave pays
// for the longest candidate list among its 64 lanes, so instead of scanning a 16^3 cell's list in rounds of four, a lane whose cell
// has more than four possible nearest entries descends to the cell's 8-wide child and, if need be, to the 4-wide and 2-wide ones -- at
// most three more dependent reads -- and ranks ONE group of four.  Returns the palette index, or -1 when the table has no answer for this
// point (a 4-wide cell that is still longer, or a float32 near tie): the caller's lists decide then, as before.
__device__ __forceinline__ bool h4_marker(const uint32_t w) { return (w & 0xffu) >= ((w >> 8) & 0xffu); }

__device__ __forceinline__ int nearest_h4(const uint32_t *__restrict__ h4, const float4 *__restrict__ cand, const float o0, const float o1,
                                          const float o2)
{
    const uint32_t i0 = (uint32_t)o0, i1 = (uint32_t)o1, i2 = (uint32_t)o2;
    uint32_t w = h4[(i0 >> 4) | ((i1 >> 4) << 4) | ((i2 >> 4) << 8)];
#pragma unroll
    for (int bit = 3; bit >= 1; --bit) {   // the 8-wide, 4-wide, 2-wide child: as far as this lane's cell is cut
        if (!h4_marker(w)) break;
        if ((w >> 16) == 0xffffu) return -1;
        w = h4[4096u + (w >> 16) * 8u + (((i0 >> bit) & 1u) | (((i1 >> bit) & 1u) << 1) | (((i2 >> bit) & 1u) << 2))];
    }
    if (h4_marker(w)) return -1;
    const float4 c1 = cand[w & 255u], c2 = cand[(w >> 8) & 255u], c3 = cand[(w >> 16) & 255u], c4 = cand[w >> 24];
    keep_record_whole(c1);
    keep_record_whole(c2);
    keep_record_whole(c3);
    keep_record_whole(c4);
    const int k1 = ed_key(c1, o0, o1, o2, 0u), k2 = ed_key(c2, o0, o1, o2, 1u), k3 = ed_key(c3, o0, o1, o2, 2u),
              k4 = ed_key(c4, o0, o1, o2, 3u);
    int m0 = min(min(k1, k2), k3), m1 = ed_med3(k1, k2, k3);
    m1 = ed_med3(m0, m1, k4);
    m0 = min(m0, k4);
    // (the margin of the 16^3 key scan: 2e-6 of the float32 evaluation + 7 ulp of the tag bits)
    const float f0 = __int_as_float(m0 & ~7), f1 = __int_as_float(m1 & ~7);
    if (f1 > f0 * 1.000003f) return (int)((w >> (8u * ((uint32_t)m0 & 3u))) & 255u);
    return -1;
}

// lists of the 16x16x16 cells in the format of the 8x8x8 table (count byte 255: more than 15 entries, use that table).
// `expanded` (with `coarse`, EXPANDED instances): the candidates' expanded records for ed_key_expanded.
// entry `pos` (0-based) of a WIDE list block -- palettes of 257..1024 colours: ten bits per entry from bit 8 on (host_logic.h:
// ed_list_put; ediff.hip: ed_cells_kernel)
__device__ __forceinline__ int ed_wide_entry(const uint4 blk, const uint32_t pos)
{
    const uint32_t off = 8u + 10u * pos, wi = off >> 5, sh = off & 31u;
    const uint32_t lo = wi == 0u ? blk.x : (wi == 1u ? blk.y : (wi == 2u ? blk.z : blk.w));
    const uint32_t hi = wi == 0u ? blk.y : (wi == 1u ? blk.z : (wi == 2u ? blk.w : 0u));
    return (int)(__funnelshift_r(lo, hi, sh) & 1023u);
}

// WIDE_OK: the instance also reads the wide list blocks of palettes above 256 colours (only the large-queue instances do: the
// launchers send every such palette there)
template <int CAP, bool EXPANDED = false, bool WIDE_OK = false>
__device__ __forceinline__ int nearest_color_cells(const PalDev &pal, const float4 *__restrict__ cand,
                                                   const uint32_t *__restrict__ coarse, const float o0, const float o1,
                                                   const float o2, const uint4 *__restrict__ lists16 = nullptr,
                                                   const float4 *__restrict__ expanded = nullptr, const uint32_t *__restrict__ h4 = nullptr)
{
    float b0 = __int_as_float(0x7f800000), b1 = b0;
    int i0 = 0;
    auto visit = [&](const float4 c, const int j, const bool ok) {
        keep_record_whole(c);
        const float a = c.x - o0, b = c.y - o1, cc = c.z - o2;
        float d = __fmaf_rn(a, a, __fmaf_rn(b, b, cc * cc));
        d = ok ? d : __int_as_float(0x7f800000);
        const bool lt0 = d < b0;
        b1 = lt0 ? b0 : (d < b1 ? d : b1);
        i0 = lt0 ? j : i0;
        b0 = lt0 ? d : b0;
    };
    bool listed = false;
    if (coarse) {
        const uint32_t e = coarse[((uint32_t)o0 >> 4) | (((uint32_t)o1 >> 4) << 4) | (((uint32_t)o2 >> 4) << 8)];
        const int n = (int)(e & 15u);
        if (n <= 7) {
            const int j1 = (e >> 4) & 15, j2 = (e >> 8) & 15, j3 = (e >> 12) & 15, j4 = (e >> 16) & 15;
            if (EXPANDED) {
                const float4 x1 = expanded[j1], x2 = expanded[j2], x3 = expanded[j3], x4 = expanded[j4];
                int k1 = ed_key_expanded(x1, o0, o1, o2, 1u), k2 = ed_key_expanded(x2, o0, o1, o2, 2u),
                    k3 = ed_key_expanded(x3, o0, o1, o2, 3u), k4 = ed_key_expanded(x4, o0, o1, o2, 4u);
                int m0 = min(min(k1, k2), k3), m1 = ed_med3(k1, k2, k3);
                m1 = ed_med3(m0, m1, k4);
                m0 = min(m0, k4);
                if (n > 4) {
                    const float4 x5 = expanded[(e >> 20) & 15], x6 = expanded[(e >> 24) & 15], x7 = expanded[(e >> 28) & 15];
                    const int k5 = ed_key_expanded(x5, o0, o1, o2, 5u), k6 = ed_key_expanded(x6, o0, o1, o2, 6u),
                              k7 = ed_key_expanded(x7, o0, o1, o2, 7u);
                    m1 = ed_med3(m0, m1, k5);
                    m0 = min(m0, k5);
                    m1 = ed_med3(m0, m1, k6);
                    m0 = min(m0, k6);
                    m1 = ed_med3(m0, m1, k7);
                    m0 = min(m0, k7);
                }
                if (__int_as_float(m1 & ~7) - __int_as_float(m0 & ~7) > kEdExpandedMargin) return (int)((e >> (4 * (m0 & 7))) & 15u);
            }
            const float4 c1 = cand[j1], c2 = cand[j2], c3 = cand[j3], c4 = cand[j4];
            // First a scan without bookkeeping: key = float32 distance bits (non-negative floats order as integers) with
            // the list position in the low 3 bits, the two smallest keys from a min3/med3 network -- 8 instructions per
            // position instead of 16.  Unused positions hold an entry that is not on the list (ediff.hip, build_ed_cells).
            // The masked bits cost up to 7 ulp (8.4e-7 relative) on top of the 2e-6 margin of the float32 evaluation: a
            // second key more than 3e-6 above the first proves the float64 order; otherwise the scan below decides as before.
            if (!EXPANDED) {
                int k1 = ed_key(c1, o0, o1, o2, 1u), k2 = ed_key(c2, o0, o1, o2, 2u), k3 = ed_key(c3, o0, o1, o2, 3u),
                    k4 = ed_key(c4, o0, o1, o2, 4u);
                int m0 = min(min(k1, k2), k3), m1 = ed_med3(k1, k2, k3);
                m1 = ed_med3(m0, m1, k4);
                m0 = min(m0, k4);
                if (n > 4) {
                    const int j5 = (e >> 20) & 15, j6 = (e >> 24) & 15, j7 = (e >> 28) & 15;
                    const float4 c5 = cand[j5], c6 = cand[j6], c7 = cand[j7];
                    const int k5 = ed_key(c5, o0, o1, o2, 5u), k6 = ed_key(c6, o0, o1, o2, 6u), k7 = ed_key(c7, o0, o1, o2, 7u);
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
            visit(c1, j1, n >= 1);
            visit(c2, j2, n >= 2);
            visit(c3, j3, n >= 3);
            visit(c4, j4, n >= 4);
            if (n > 4) {
                const int j5 = (e >> 20) & 15, j6 = (e >> 24) & 15, j7 = (e >> 28) & 15;
                const float4 c5 = cand[j5], c6 = cand[j6], c7 = cand[j7];
                visit(c5, j5, true);
                visit(c6, j6, n >= 6);
                visit(c7, j7, n >= 7);
            }
            listed = true;
        }
    }
    if (!listed && h4) {
        const int j = nearest_h4(h4, cand, o0, o1, o2);
        if (j >= 0) return j;
    }
    if (!listed) {
        uint4 blk = make_uint4(255u, 0u, 0u, 0u);
        if (lists16) blk = lists16[((uint32_t)o0 >> 4) | (((uint32_t)o1 >> 4) << 4) | (((uint32_t)o2 >> 4) << 8)];
        int n = (int)(blk.x & 255u);
        if (n == 255) {  // (no LDS lists, or a cell with more than 15 entries)
            const uint32_t ci = ((uint32_t)o0 >> 3) | (((uint32_t)o1 >> 3) << 5) | (((uint32_t)o2 >> 3) << 10);
            blk = pal.ed_cells[ci];
            n = (int)(blk.x & 255u);
        }
        if (n == 254) {  // a crowded cell (clustered palettes): refined into 4^3, 2^3, 1^3 sub-cells
            const uint32_t i0 = (uint32_t)o0, i1 = (uint32_t)o1, i2 = (uint32_t)o2;
            for (int bit = 2; n == 254; --bit) {
                const uint32_t sub = ((i0 >> bit) & 1u) | (((i1 >> bit) & 1u) << 1) | (((i2 >> bit) & 1u) << 2);
                blk = pal.ed_nodes[(size_t)(blk.x >> 8) * 8 + sub];
                n = (int)(blk.x & 255u);
            }
        }
        if (n > 15) return nearest_color<CAP>(pal, cand, o0, o1, o2);
        if (WIDE_OK && pal.K > 256) {
            // 257..1024 colours: up to twelve ten-bit entries (padded to a multiple of four positions by the builder).  The same two
            // stages as below: the key scan in groups of four, then -- near ties -- the exact scan over the listed entries.
            int m0 = 0x7fffffff, m1 = 0x7fffffff;
            for (uint32_t g = 0; (int)g < n; g += 4u) {
                const int j1 = ed_wide_entry(blk, g), j2 = ed_wide_entry(blk, g + 1u), j3 = ed_wide_entry(blk, g + 2u), j4 = ed_wide_entry(blk, g + 3u);
                const float4 c1 = cand[j1], c2 = cand[j2], c3 = cand[j3], c4 = cand[j4];
                const int k1 = ed_key16(c1, o0, o1, o2, g), k2 = ed_key16(c2, o0, o1, o2, g + 1u), k3 = ed_key16(c3, o0, o1, o2, g + 2u),
                          k4 = ed_key16(c4, o0, o1, o2, g + 3u);
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
            if (n >= 1 && f1 > f0 * 1.000004f) return ed_wide_entry(blk, (uint32_t)m0 & 15u);
            for (int i = 0; i < n; ++i) {
                const int j = ed_wide_entry(blk, (uint32_t)i);
                visit(cand[j], j, true);
            }
            if (b1 > b0 * 1.000002f) return i0;
            if (b0 == b1) {
                const int tied = integer_tie_choice(pal, cand, o0, o1, o2, b0, i0);
                if (tied >= 0) return tied;
            }
            return nearest_f64<CAP>(pal, o0, o1, o2);
        }
        // Lists of up to 12 entries (padded to a multiple of 4 positions by the builder, ediff.hip: pack): the key scan first --
        // float32 distance bits with the position in the low 4 bits, the two smallest kept by v_med3 / v_min, groups of
        // four without per-position tests; a second key within 4e-6 (relative: 2e-6 of the evaluation + 15 ulp of the tag)
        // of the first leaves the decision to the exact scan below.
        if (n >= 1 && n <= 12 && pal.K > 16) {
            uint4 b = blk;
            b.x = __funnelshift_r(b.x, b.y, 8);  // drop the count byte
            b.y = __funnelshift_r(b.y, b.z, 8);
            b.z = __funnelshift_r(b.z, b.w, 8);
            int m0 = 0x7fffffff, m1 = 0x7fffffff;
            uint32_t tag = 0u;
            for (int left = n; left > 0; left -= 4, tag += 4u) {
                const int j1 = (int)(b.x & 255u), j2 = (int)((b.x >> 8) & 255u), j3 = (int)((b.x >> 16) & 255u), j4 = (int)(b.x >> 24);
                const float4 c1 = cand[j1], c2 = cand[j2], c3 = cand[j3], c4 = cand[j4];
                const int k1 = ed_key16(c1, o0, o1, o2, tag), k2 = ed_key16(c2, o0, o1, o2, tag + 1u),
                          k3 = ed_key16(c3, o0, o1, o2, tag + 2u), k4 = ed_key16(c4, o0, o1, o2, tag + 3u);
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
        // four entries per round, their reads in flight together (unused slots hold index 0: a valid, ignored read)
        blk.x = __funnelshift_r(blk.x, blk.y, 8);  // drop the count byte
        blk.y = __funnelshift_r(blk.y, blk.z, 8);
        blk.z = __funnelshift_r(blk.z, blk.w, 8);
        blk.w >>= 8;
        for (; n > 0; n -= 4) {
            const int j1 = (int)(blk.x & 255u), j2 = (int)((blk.x >> 8) & 255u), j3 = (int)((blk.x >> 16) & 255u), j4 = (int)(blk.x >> 24);
            const float4 c1 = cand[j1], c2 = cand[j2], c3 = cand[j3], c4 = cand[j4];
            visit(c1, j1, true);
            visit(c2, j2, n >= 2);
            visit(c3, j3, n >= 3);
            visit(c4, j4, n >= 4);
            blk.x = blk.y;
            blk.y = blk.z;
            blk.z = blk.w;
            blk.w = 0u;
        }
    }
    if (b1 > b0 * 1.000002f) return i0;
    if (b0 == b1) {
        const int tied = integer_tie_choice(pal, cand, o0, o1, o2, b0, i0);
        if (tied >= 0) return tied;
    }
    return nearest_f64<CAP>(pal, o0, o1, o2);
}

}  // namespace
}  // namespace dp
