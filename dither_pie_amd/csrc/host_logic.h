// Host-side logic of libditherpie_hip.so that never touches the GPU: the scipy-order KD-tree build, the assembly of
// the search accelerator's cell table from membership masks, and the candidate tables of the diffusion kernels (with
// their multi-threaded per-cell loops).  Pure C++17, no HIP headers: included by the .hip / .cpp translation units through
// dp_internal.h AND compiled on its own under AddressSanitizer + UBSan and under ThreadSanitizer
// (`make host_asan host_tsan` -> build/host_asan, build/host_tsan from host_sanitize.cpp; run by
// tests/test_host_sanitizers.py in the CPU tier).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <initializer_list>
#include <limits>
#include <numeric>
#include <thread>
#include <utility>
#include <exception>
#include <vector>

#include "../../include/ditherpie_hip.h"

namespace dp {

constexpr int kLeafSize = 10;     // scipy.spatial.KDTree default (dithering_lib.py:339)
constexpr int kWideList = 16;     // entries of the flat candidate list of a split cell (ordered_fast_kernel, accel.hip)

// KD-tree as scipy builds it; node 0 is the root, children follow in pre-order.
struct HostTree {
    int K = 0;
    std::vector<double> pts;  // K*3
    std::vector<int32_t> indices;
    std::vector<int32_t> split_dim, start, end, less, greater;
    std::vector<double> split;
    double mins[3], maxes[3];
};


// ---------------------------------------------------------------------------------------------
// scipy.spatial.KDTree(points) with its defaults (leafsize=10, compact_nodes, balanced_tree), the
// structure behind every palette search of the reference (dithering_lib.py:339, 358, 554, 655).
// The permutation std::nth_element leaves behind decides scipy's tie order, so the same standard
// algorithm (libstdc++ introselect) is used here, with scipy's coordinate-only comparator.
// ---------------------------------------------------------------------------------------------
namespace kd_detail {
struct Builder {
    HostTree &t;
    const double *P;

    double coord(int i, int d) const { return P[(size_t)i * 3 + d]; }

    // two-pointer partition of indices[s,e) by coord < split; returns the first index of the >= part
    int split_range(int s, int e, int d, double split)
    {
        int lo = s, hi = e - 1;
        while (lo <= hi) {
            if (coord(t.indices[lo], d) < split)
                ++lo;
            else if (coord(t.indices[hi], d) >= split)
                --hi;
            else
                std::swap(t.indices[lo++], t.indices[hi--]);
        }
        return lo;
    }

    int add_node(int s, int e)
    {
        int id = (int)t.split_dim.size();
        t.split_dim.push_back(-1);
        t.split.push_back(0.0);
        t.start.push_back(s);
        t.end.push_back(e);
        t.less.push_back(-1);
        t.greater.push_back(-1);
        return id;
    }

    int build(int s, int e)
    {
        const int id = add_node(s, e);
        if (e - s <= kLeafSize) return id;

        double lo[3], hi[3];
        for (int d = 0; d < 3; ++d) lo[d] = hi[d] = coord(t.indices[s], d);
        for (int j = s + 1; j < e; ++j)
            for (int d = 0; d < 3; ++d) {
                const double v = coord(t.indices[j], d);
                hi[d] = hi[d] > v ? hi[d] : v;
                lo[d] = lo[d] < v ? lo[d] : v;
            }
        int dim = 0;
        double extent = 0;
        for (int d = 0; d < 3; ++d)
            if (hi[d] - lo[d] > extent) {
                dim = d;
                extent = hi[d] - lo[d];
            }
        if (hi[dim] == lo[dim]) return id;  // all points coincide

        int32_t *first = t.indices.data() + s;
        const int half = (e - s) / 2;
        std::nth_element(first, first + half, first + (e - s),
                         [&](int32_t a, int32_t b) { return coord(a, dim) < coord(b, dim); });
        double split = coord(t.indices[s + half], dim);
        int cut = split_range(s, e, dim, split);
        if (cut == s) {
            // nothing lies strictly below the median value: cut just above the minimum instead
            double mn = coord(t.indices[s], dim);
            for (int j = s + 1; j < e; ++j) mn = std::min(mn, coord(t.indices[j], dim));
            split = std::nextafter(mn, std::numeric_limits<double>::infinity());
            cut = split_range(s, e, dim, split);
        }
        t.split_dim[id] = dim;
        t.split[id] = split;
        const int l = build(s, cut);
        const int g = build(cut, e);
        t.less[id] = l;
        t.greater[id] = g;
        return id;
    }
};
}  // namespace kd_detail

inline void build_tree(const double *pts, int K, HostTree &t)
{
    t = HostTree();
    t.K = K;
    t.pts.assign(pts, pts + (size_t)K * 3);
    t.indices.resize(K);
    std::iota(t.indices.begin(), t.indices.end(), 0);
    for (int d = 0; d < 3; ++d) t.mins[d] = t.maxes[d] = pts[d];
    for (int j = 1; j < K; ++j)
        for (int d = 0; d < 3; ++d) {
            t.mins[d] = std::min(t.mins[d], pts[(size_t)j * 3 + d]);
            t.maxes[d] = std::max(t.maxes[d], pts[(size_t)j * 3 + d]);
        }
    kd_detail::Builder b{t, t.pts.data()};
    b.build(0, K);
}


// Position of a 16x16x16 cell in the LDS table: the kernels form it as r' | b'<<4 | g'<<8 (x & 0xf0f0f0,
// OR-ed with itself shifted left by 12, bits 16..27), three operations fewer than r'<<8 | g'<<4 | b'.
inline int cell_slot(int rc, int gc, int bc) { return rc | (bc << 4) | (gc << 8); }


// an axis-aligned box of colours (the accelerator's octree below the 16^3 cells)
struct Box {
    int r0, g0, b0, size;
};


constexpr int kCells = 4096;
constexpr int kWideCap = 512;   // such lists per table at most
constexpr int kNearSlots = 6;  // entries of an 8-entry block that the nearest-only path of ordered_fast_kernel reads
constexpr int kTabCapWords = (160 * 1024 - 2048) / 4;  // LDS budget of the dither kernels
constexpr int kTabMaxWords = 1 << 20;                 // largest table built (4 MB); what exceeds LDS stays in global memory
// words the lean kernels stage when the table is larger than LDS: the 4096 cell blocks + the first split nodes
constexpr int kTabStageWords = 4096 * 8 + 88 * 64;

struct TableStats {
    int n_split = 0, n_slow = 0, max_cnt = 0, max_near = 0, n_near_overflow = 0;
    bool wide_overflow = false;
    std::vector<Box> node_box;  // the box each split node covers, by node index
    int n_split_cells = 0;  // 16^3 cells that are split (the pixels of these cells leave the main path of the kernels)
    bool too_big = false;
};

// Turns the per-cell membership masks into the LDS table: [4096 cells][8 words], then [n_split][8 sub-cells][8].
// coord4[j]: integer coordinates r | g<<8 | b<<16 of entry j (used to choose padding entries); word[j]: what a
// block stores for entry j.  box_masks(boxes, out) computes the membership masks (8 words each) of further boxes.
// nmasks (optional): the nearest sets N(cell) of the 4096 cells.  With them a cell whose N has more than `ns` members is
// split like an overflowing one, and perm[slot] receives, for every unsplit cell block, the order in which the fast
// ordered kernel stages the block's entries into LDS -- the members of N first (field k, 3 bits for bw = 8, 2 bits
// for bw = 4: which entry of the index-ordered block goes to LDS slot k; bits 28..31: |N|); 0xffffffff = split cell.
// The table itself stays in palette-index order (the tie codes are defined on that order).
// wide (with perm): for every SPLIT cell the whole list T(cell) in index order, padded to kWideList entries -- the fast
// kernel resolves the pixels of split cells on it (one block read instead of a descent through the octree);
// perm[slot] = 0xff000000 | list number.  A cell with a longer list sets st.wide_overflow (no fast kernel then).
template <class BoxMasks>
int assemble_table(const std::vector<uint32_t> &masks, const int mw, const int bw, const int cap_words, const int K,
                   const std::vector<uint32_t> &coord4, const std::vector<uint32_t> &word, BoxMasks box_masks,
                   std::vector<uint32_t> &tab, TableStats &st, const uint32_t *nmasks = nullptr, const int ns = 0,
                   std::vector<uint32_t> *perm = nullptr, std::vector<uint32_t> *wide = nullptr)
{
    // bw: entries per block (8, or 4 for small palettes); a split node is 8 child blocks
    tab.assign((size_t)kCells * bw, 0u);
    std::vector<int> list, extra;
    std::vector<uint64_t> dkey;
    // members of a mask, padded to bw with unused entries; false if more than bw
    auto make_block = [&](const uint32_t *mask, int cr, int cg, int cb, uint32_t *out8) {
        list.clear();
        for (int wi = 0; wi < mw && 32 * wi < K; ++wi)   // the members: the set bits of the mask below K
            for (uint32_t bits = mask[wi]; bits; bits &= bits - 1u) {
                const int j = 32 * wi + __builtin_ctz(bits);
                if (j < K) list.push_back(j);
            }
        st.max_cnt = std::max(st.max_cnt, (int)list.size());
        if (list.size() > (size_t)bw) return false;
        if (list.size() < (size_t)bw) {
            extra.clear();   // the lowest unused indices, as many as the padding needs
            for (int j = 0; j < K && extra.size() < (size_t)bw; ++j)
                if (!(mask[j >> 5] >> (j & 31) & 1u)) extra.push_back(j);
            // pad with unused entries: ANY real entry is harmless (it is not among the three nearest of any colour of the box, so its
            // key never decides anything), so the lowest unused indices do.  Rounds 2-5 took the unused entries nearest to the
            // centre -- a distance key for every unused entry of every cell and a partial sort: 3 of the 4.4 ms this assembly took
            // for 256 colours (tools/bench_scripts/accel_build_stages.py).
            const size_t need = (size_t)bw - list.size();
            (void)cr, (void)cg, (void)cb;
            for (size_t q = 0; q < need; ++q) list.push_back(extra[q]);
            std::sort(list.begin(), list.end());  // key ties must break towards the lower palette index
        }
        for (int i = 0; i < bw; ++i) out8[i] = word[list[i]];
        return true;
    };
    // pending splits below the 8^3 level: (block position in tab, box) resolved level by level
    struct Pending {
        size_t pos;
        Box box;
    };
    std::vector<Pending> pending;
    // turn the block at `pos` into a split node with 8 children of half size; children masks come
    // either from `child_masks` (8 x 8 words) or, if null, are requested for the next round
    auto split = [&](size_t pos, const Box &bx, const uint32_t *child_masks) {
        const size_t base = tab.size();
        if (base + 8 * (size_t)bw > (size_t)cap_words) {
            st.too_big = true;
            return;
        }
        tab.resize(base + 8 * (size_t)bw, 0u);
        tab[pos] = 0x80000000u | (uint32_t)((base - (size_t)kCells * bw) / (8 * (size_t)bw));
        ++st.n_split;
        st.node_box.push_back(bx);
        const int hs = bx.size / 2;
        for (int sidx = 0; sidx < 8; ++sidx) {
            Box c{bx.r0 + ((sidx >> 2) & 1) * hs, bx.g0 + ((sidx >> 1) & 1) * hs, bx.b0 + (sidx & 1) * hs, hs};
            const size_t cpos = base + (size_t)sidx * bw;
            if (child_masks) {
                uint32_t blk[8];
                if (make_block(child_masks + (size_t)mw * sidx, c.r0 + hs / 2, c.g0 + hs / 2, c.b0 + hs / 2, blk))
                    std::copy(blk, blk + bw, tab.begin() + cpos);
                else
                    pending.push_back({cpos, c});
            } else {
                pending.push_back({cpos, c});
            }
        }
    };
    if (perm) perm->assign(kCells, 0xffffffffu);
    for (int cell = 0; cell < kCells && !st.too_big; ++cell) {
        const uint32_t *m = &masks[(size_t)cell * 9 * mw];
        const int r0 = (cell >> 8) * 16, g0 = ((cell >> 4) & 15) * 16, b0 = (cell & 15) * 16;
        uint32_t blk[8];
        const size_t slot = (size_t)cell_slot(cell >> 8, (cell >> 4) & 15, cell & 15);
        bool fits = make_block(m, r0 + 8, g0 + 8, b0 + 8, blk);
        bool near_overflow = false;
        if (fits && nmasks) {
            // `list` holds the block's entries in index order: the nearest set first
            const uint32_t *nm = nmasks + (size_t)cell * mw;
            const int fb = bw == 8 ? 3 : 2;
            uint32_t pw = 0;
            int k = 0, n_near = 0;
            for (int pass = 0; pass < 2; ++pass)
                for (int i = 0; i < bw; ++i) {
                    const int j = list[i];
                    const bool near = (nm[j >> 5] >> (j & 31)) & 1u;
                    if (near == (pass == 0)) {
                        pw |= (uint32_t)i << (fb * k++);
                        n_near += near;
                    }
                }
            // a nearest set larger than the nearest-only path of the fast kernel reads: the cell keeps its block in the table,
            // but the fast kernel treats it as split (staging order = flat list below)
            near_overflow = n_near > ns;
            if (!near_overflow && perm) (*perm)[slot] = pw | ((uint32_t)n_near << 28);
            st.max_near = std::max(st.max_near, n_near);
            st.n_near_overflow += near_overflow;
        }
        if (fits) std::copy(blk, blk + bw, tab.begin() + slot * bw);
        if (!fits || near_overflow) {
            if (perm && wide) {
                // the cell's whole list, index order, padded with the unused entries nearest to the centre
                list.clear();
                extra.clear();
                for (int j = 0; j < K; ++j) ((m[j >> 5] >> (j & 31) & 1u) ? list : extra).push_back(j);
                if (K <= kWideList) {
                    // the whole palette (the kernel reads min(K, kWideList) entries)
                    (*perm)[slot] = 0xff000000u | (uint32_t)(wide->size() / kWideList);
                    for (int i = 0; i < kWideList; ++i) wide->push_back(i < K ? word[i] : 0u);
                } else if ((int)list.size() > kWideList) {
                    st.wide_overflow = true;
                } else {
                    const size_t need = (size_t)kWideList - list.size();
                    dkey.resize(extra.size());
                    for (size_t q = 0; q < extra.size(); ++q) {
                        const uint32_t c = coord4[extra[q]];
                        const int r = c & 255, g = (c >> 8) & 255, b = (c >> 16) & 255;
                        dkey[q] = ((uint64_t)((r - r0 - 8) * (r - r0 - 8) + (g - g0 - 8) * (g - g0 - 8) + (b - b0 - 8) * (b - b0 - 8)) << 16) | (uint64_t)extra[q];
                    }
                    std::partial_sort(dkey.begin(), dkey.begin() + need, dkey.end());
                    for (size_t q = 0; q < need; ++q) list.push_back((int)(dkey[q] & 0xffff));
                    std::sort(list.begin(), list.end());
                    (*perm)[slot] = 0xff000000u | (uint32_t)(wide->size() / kWideList);
                    for (int i = 0; i < kWideList; ++i) wide->push_back(word[list[i]]);
                }
            }
        }
        if (!fits) {
            split(slot * bw, Box{r0, g0, b0, 16}, m + mw);
            ++st.n_split_cells;
        }
    }
    // deeper levels: boxes that still hold more than 8 members are split again; a single colour that still
    // overflows is left to the fix-up pass
    while (!pending.empty() && !st.too_big) {
        std::vector<Pending> todo;
        todo.swap(pending);
        std::vector<Pending> kids;  // children whose masks we need this round
        for (const Pending &pd : todo) {
            if (pd.box.size == 1) {
                tab[pd.pos] = 0xC0000000u;
                ++st.n_slow;
                continue;
            }
            const size_t before = pending.size();
            split(pd.pos, pd.box, nullptr);
            if (st.too_big) break;
            for (size_t q = before; q < pending.size(); ++q) kids.push_back(pending[q]);
            pending.resize(before);
        }
        if (st.too_big || kids.empty()) break;
        std::vector<Box> boxes(kids.size());
        for (size_t q = 0; q < kids.size(); ++q) boxes[q] = kids[q].box;
        std::vector<uint32_t> bm(kids.size() * (size_t)mw);
        const int rc = box_masks(boxes, bm);
        if (rc != DP_OK) return rc;
        for (size_t q = 0; q < kids.size(); ++q) {
            const Box &c = kids[q].box;
            uint32_t blk[8];
            if (make_block(&bm[q * (size_t)mw], c.r0 + c.size / 2, c.g0 + c.size / 2, c.b0 + c.size / 2, blk))
                std::copy(blk, blk + bw, tab.begin() + kids[q].pos);
            else
                pending.push_back(kids[q]);
        }
    }
    return DP_OK;
}

// Tables larger than LDS: the kernels stage the cell blocks and the FIRST split nodes, the rest is read from global
// memory.  Put the nodes of the most crowded boxes first -- a palette extracted from an image crowds its colours where
// the image's pixels are, so those are the nodes the pixels visit.
inline void crowded_nodes_first(std::vector<uint32_t> &tab, const TableStats &st, const int bw, const std::vector<uint32_t> &coord4)
{
    const size_t n = st.node_box.size();
    if (n < 2) return;
    std::vector<int> score(n, 0);
    for (size_t k = 0; k < n; ++k) {
        const Box &b = st.node_box[k];
        for (const uint32_t c : coord4) {
            const int r = c & 255, g = (c >> 8) & 255, bl = (c >> 16) & 255;
            score[k] += (r >= b.r0 && r < b.r0 + b.size && g >= b.g0 && g < b.g0 + b.size && bl >= b.b0 && bl < b.b0 + b.size);
        }
    }
    std::vector<uint32_t> order(n), where(n);
    for (size_t k = 0; k < n; ++k) order[k] = (uint32_t)k;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return score[a] > score[b]; });
    for (size_t k = 0; k < n; ++k) where[order[k]] = (uint32_t)k;
    const size_t base = (size_t)kCells * bw, node_words = 8 * (size_t)bw;
    std::vector<uint32_t> moved(tab.size());
    std::copy(tab.begin(), tab.begin() + base, moved.begin());
    for (size_t k = 0; k < n; ++k)
        std::copy(tab.begin() + base + k * node_words, tab.begin() + base + (k + 1) * node_words,
                  moved.begin() + base + where[k] * node_words);
    for (uint32_t &w : moved)
        if ((w >> 30) == 2u) w = 0x80000000u | where[w & 0xffffffu];  // split markers: the child node's new index
    tab.swap(moved);
}

// The per-channel maps of a warped table (see accel_scan_warp_kernel).
struct WarpMaps {
    uint8_t lut[3][256];  // colour value -> warped coordinate
    int lo[3][257];       // lo[c][u]: the smallest value whose warped coordinate is >= u (256 when there is none)
};

// 16 cells per channel, cell i starting at the palette's coordinate of rank K*i/16 (boundaries kept strictly
// increasing); inside a cell of w values, value number k sits at sub-position k*16/w.
inline void make_warp(const std::vector<uint32_t> &coord4, WarpMaps &wm)
{
    const size_t K = coord4.size();
    for (int c = 0; c < 3; ++c) {
        std::vector<int> v(K);
        for (size_t j = 0; j < K; ++j) v[j] = (coord4[j] >> (8 * c)) & 255;
        std::sort(v.begin(), v.end());
        int a[17];
        a[0] = 0;
        a[16] = 256;
        for (int i = 1; i < 16; ++i) {
            int q = v[K * (size_t)i / 16];
            q = std::max(q, a[i - 1] + 1);
            q = std::min(q, 256 - (16 - i));
            a[i] = q;
        }
        for (int i = 0; i < 16; ++i) {
            const int w = a[i + 1] - a[i];
            for (int x = a[i]; x < a[i + 1]; ++x) wm.lut[c][x] = (uint8_t)(16 * i + ((x - a[i]) * 16) / w);
        }
        int x = 0;
        for (int u = 0; u <= 256; ++u) {
            while (x < 256 && (int)wm.lut[c][x] < u) ++x;
            wm.lo[c][u] = x;
        }
    }
}

// the palette entries and, for each entry, the points a quarter and half of the way to its four nearest other entries
inline std::vector<uint32_t> mass_points(const std::vector<uint32_t> &coord4)
{
    std::vector<uint32_t> out(coord4);
    const size_t K = coord4.size();
    constexpr int kNear = 4;
    for (size_t j = 0; j < K && K > (size_t)kNear; ++j) {
        const int r = coord4[j] & 255, g = (coord4[j] >> 8) & 255, b = (coord4[j] >> 16) & 255;
        uint64_t best[kNear];
        for (uint64_t &v : best) v = ~0ull;
        for (size_t k = 0; k < K; ++k) {
            if (k == j) continue;
            const int dr = (int)(coord4[k] & 255) - r, dg = (int)((coord4[k] >> 8) & 255) - g, db = (int)((coord4[k] >> 16) & 255) - b;
            uint64_t key = ((uint64_t)(dr * dr + dg * dg + db * db) << 32) | k;
            for (uint64_t &v : best)
                if (key < v) std::swap(key, v);
        }
        for (const uint64_t key : best) {
            const uint32_t c = coord4[key & 0xffffffffu];
            const int cr = c & 255, cg = (c >> 8) & 255, cb = (c >> 16) & 255;
            for (const int q : {1, 2})  // quarters of the way
                out.push_back((uint32_t)(r + (cr - r) * q / 4) | ((uint32_t)(g + (cg - g) * q / 4) << 8) | ((uint32_t)(b + (cb - b) * q / 4) << 16));
        }
    }
    return out;
}

// One byte per entry: the table `tab` of 8-entry blocks (4096 cell blocks, then 8 blocks per split node; entries = packed
// colours in palette-index order, a block whose first word has bit 31 set = marker) with every colour replaced by the
// index of its palette entry, two words per block.  Entries of equal colour (duplicated palette entries) get their indices
// in ascending order, as the index-ordered block has them.  Marker blocks become {marker, 0xffffffff} with the marker
// 0x80000000 | byte offset of the node's eight blocks in THIS table (split) or 0xC0000000 (too many candidates).
constexpr int kCompactMaxWords = 36 * 1024;  // 144 KB: what ordered_compact_kernel can hold next to records, maps, thresholds
inline std::vector<uint32_t> compact_table(const std::vector<uint32_t> &tab, const std::vector<uint32_t> &coord4)
{
    std::vector<std::pair<uint32_t, uint32_t>> by_colour(coord4.size());
    for (size_t j = 0; j < coord4.size(); ++j) by_colour[j] = {coord4[j], (uint32_t)j};
    std::sort(by_colour.begin(), by_colour.end());  // ascending index among equal colours
    std::vector<uint32_t> out(tab.size() / 4);
    for (size_t b = 0; b + 8 <= tab.size(); b += 8) {
        if (tab[b] >> 31) {
            const bool split = (tab[b] >> 30) == 2u;
            out[b / 4] = split ? (0x80000000u | (uint32_t)((4096u + (tab[b] & 0xffffffu) * 8u) * 8u)) : 0xC0000000u;
            out[b / 4 + 1] = 0xffffffffu;
            continue;
        }
        uint32_t w[2] = {0u, 0u};
        for (int i = 0; i < 8; ++i) {
            int nth = 0;  // how many earlier entries of the block have this colour
            for (int k = 0; k < i; ++k) nth += tab[b + k] == tab[b + i];
            auto it = std::lower_bound(by_colour.begin(), by_colour.end(), std::make_pair(tab[b + i], 0u));
            uint32_t idx = 0u;
            if (it != by_colour.end() && it->first == tab[b + i]) {
                if ((size_t)(it - by_colour.begin()) + nth < by_colour.size() && (it + nth)->first == tab[b + i]) it += nth;
                idx = it->second;
            }
            w[i >> 2] |= idx << (8 * (i & 3));
        }
        out[b / 4] = w[0];
        out[b / 4 + 1] = w[1];
    }
    return out;
}

// how many of the given points (coordinates as the table sees them) sit in split cells
inline int entries_in_split_cells(const std::vector<uint32_t> &tab, const int bw, const std::vector<uint32_t> &coord4)
{
    int n = 0;
    for (const uint32_t c : coord4) {
        const size_t slot = (size_t)cell_slot((c & 255) >> 4, ((c >> 8) & 255) >> 4, ((c >> 16) & 255) >> 4);
        n += (int)(tab[slot * bw] >> 31);
    }
    return n;
}


// ---------------------------------------------------------------------------------------------
// Candidate tables of the diffusion kernels (ediff.hip / vardiff.hip): everything that happens on the host between the
// download of ed_cells_kernel's geometric lists and the upload of the finished tables.
// ---------------------------------------------------------------------------------------------
struct U4 {  // layout of HIP's uint4
    uint32_t x, y, z, w;
};
inline U4 make_u4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return U4{x, y, z, w}; }

constexpr int kEdCells = 32 * 32 * 32;

// The per-cell loops below are independent: split them over a few host threads (the tables of a new palette are built
// at its first diffusion call, i.e. in front of a user's image).  Each thread writes only the cells of its own range.
template <class F>
inline void parallel_cells(const int n, F &&body, const int serial_below = 1024)
{
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt < 1 ? 1 : (nt > 8 ? 8 : nt);
    if (n < serial_below || nt == 1) {
        body(0, n);
        return;
    }
    std::vector<std::thread> th;
    const int chunk = (n + (int)nt - 1) / (int)nt;
    for (unsigned t = 1; t < nt; ++t) {
        const int lo = (int)t * chunk, hi = std::min(n, lo + chunk);
        if (lo < hi) th.emplace_back([&body, lo, hi]() { body(lo, hi); });
    }
    body(0, std::min(n, chunk));
    for (auto &t : th) t.join();
}

struct EdTables {
    std::vector<U4> nodes;         // refinement of overflowing 8^3 cells (8 entries per node); empty if none / given up
    bool give_up = false;          // more than 2^22 nodes: the overflowing cells keep count 255 (scan the palette)
    std::vector<U4> l16;           // K > 16: lists of the 16^3 cells in the 8^3 table's format
    std::vector<uint32_t> coarse;  // K <= 16: lists of the 16^3 cells, count | 7 index nibbles
    std::vector<uint32_t> ext;     // K <= 16: the same over the extended grid (outermost cells = half-spaces)
    std::vector<U4> ext16;         // 17..256 colours: l16 with the OUTERMOST cells standing for everything beyond them (unbounded boxes),
                                   // for the diffusers that do not clamp their values (vardiff.hip: nearest_ext16); count 255: too long
    std::vector<U4> ext_nodes;     // ... and the octree below the outermost cells whose list is too long (count 254 | node << 8: eight
                                   // half-size children, the outer ones unbounded again), at most kEdExtMaxNodes nodes
    // K > 16: the hierarchical nearest table of at most four entries per leaf (ed_nearest.hip.h: nearest_h4).  Words [0, 4096): the
    // 16^3 cells; then nodes of 8 words: the 8-wide children of a cell, the 4-wide children of a child, the 2-wide ones of those.  A LEAF word holds four
    // palette indices, byte 0 < byte 1 (listed entries, then an unlisted far entry as padding); a word with byte 0 >= byte 1 is a
    // MARKER: 0x00ff | node << 16 (node 0xffff: no answer here, use the lists).  Empty when it would not fit kEdH4MaxWords.
    std::vector<uint32_t> h4;
    size_t h4_wanted = 0;          // words the table would take (reported by the experiments build)
    double h4_none = 0.0;          // fraction of the palette's own colours at which the table has no answer
    double h4_depth = 0.0;         // mean number of descents of nearest_h4 at the palette's own colours (4 = no answer): where a
                                   // palette puts its colours is where its images put their pixels
};
constexpr size_t kEdH4LdsWords = 27648;   // 108 KB: all the LDS the few-frames diffusion kernel has left (256 random colours: 21 336 words, median cut 256 of smooth content: 25 120)
constexpr size_t kEdH4MaxWords = 65536;   // largest table built (256 KB, node numbers stay below 0xffff): beyond the LDS size it is read from L2
constexpr uint32_t kEdH4NoAnswer = 0xffff00ffu;

// cells [kEdCells]: in = the kernel's lists (count byte | up to 15 index bytes, count 255 = overflow), out = sharpened,
// padded, overflowing cells refined into `nodes`.  pts: K*3 float64 palette coordinates.
// A list block (8^3 cells, their octree nodes, 16^3 cells): word 0's low byte is the count (254: node pointer in the upper 24 bits,
// 255: too long -- scan the palette), then the entries' palette indices: one BYTE each (up to 15) for palettes of up to 256 colours,
// TEN BITS each from bit 8 on (up to 12: 8 + 120 = 128 bits) for 257..1024 colours ("wide"; round 5 -- until then error diffusion with
// more than 256 colours had no lists at all and ran the KD-tree query at every pixel step: 0.44 s per 4K frame at 1024 colours).
constexpr size_t kEdExtMaxNodes = 2048;   // refinement nodes of the extended 16^3 lists (8 entries each: 256 KB)
inline int ed_list_cap(const int K) { return K > 256 ? 12 : 15; }
inline int ed_list_get(const uint32_t (&w)[4], const int pos /* 0-based */, const bool wide)
{
    if (!wide) return (int)((w[(pos + 1) >> 2] >> (8 * ((pos + 1) & 3))) & 255u);
    const int off = 8 + 10 * pos, wi = off >> 5, sh = off & 31;
    uint32_t v = w[wi] >> sh;
    if (sh > 22 && wi < 3) v |= w[wi + 1] << (32 - sh);
    return (int)(v & 1023u);
}
inline void ed_list_put(uint32_t (&w)[4], const int pos /* 0-based */, const int j, const bool wide)
{
    if (!wide) {
        w[(pos + 1) >> 2] |= (uint32_t)j << (8 * ((pos + 1) & 3));
        return;
    }
    const int off = 8 + 10 * pos, wi = off >> 5, sh = off & 31;
    w[wi] |= (uint32_t)j << sh;
    if (sh > 22 && wi < 3) w[wi + 1] |= (uint32_t)j >> (32 - sh);
}

// what: 1 = the tables every diffusion kernel uses (the 8^3 lists in `host` refined in place, their octree, 16^3 lists, hierarchical
// table, nibble tables), 2 = ONLY the extended 16^3 lists of the unclamped diffusers (ext16 / ext_nodes; `host` is not touched) --
// built when such a diffuser first meets the palette, so that plain error diffusion does not pay for them (5-16 ms) -- 3 = both.
inline void ed_tables_refine(const double *pts, const int K, std::vector<U4> &host, EdTables &out, const int what = 1)
{
    const bool wide = K > 256;
    const int cap = ed_list_cap(K);
    std::vector<U4> &nodes = out.nodes;
    nodes.clear();
    auto box_list = [&](const std::vector<int> &from, const double lo[3], const double size, std::vector<int> &list) {
        double bound = std::numeric_limits<double>::infinity();
        for (int j : from) {
            double far2 = 0.0;
            for (int k = 0; k < 3; ++k) {
                const double c = pts[3 * j + k];
                const double m = std::max(std::fabs(c - lo[k]), std::fabs(c - (lo[k] + size)));
                far2 += m * m;
            }
            bound = std::min(bound, far2);
        }
        bound = bound * (1.0 + 1e-12) + 1e-9;
        list.clear();
        for (int j : from) {
            double near2 = 0.0;
            for (int k = 0; k < 3; ++k) {
                const double c = pts[3 * j + k];
                const double m = std::max(std::max(lo[k] - c, c - (lo[k] + size)), 0.0);
                near2 += m * m;
            }
            if (near2 <= bound) list.push_back(j);
        }
    };
    // A sharper (still conservative) list for a box: entry j is dropped if some other listed entry k is closer to EVERY
    // point of the box, i.e. the box lies strictly on k's side of the bisector of j and k:
    //   max over the box of |x - c_k|^2 - |x - c_j|^2 = max of 2 x.(c_j - c_k) + |c_k|^2 - |c_j|^2 < 0   (linear in x).
    // The true nearest entry of a point of the box is dominated by nobody, so it stays listed, with everything tied with it.
    auto prune_list = [&](const double lo[3], const double size, std::vector<int> &list) {
        std::vector<int> keep;
        for (int j : list) {
            bool dominated = false;
            for (int k : list) {
                if (k == j) continue;
                double mx = 0.0;
                for (int d = 0; d < 3; ++d) {
                    const double a = 2.0 * (pts[3 * j + d] - pts[3 * k + d]);
                    mx += std::max(a * lo[d], a * (lo[d] + size));
                    mx += pts[3 * k + d] * pts[3 * k + d] - pts[3 * j + d] * pts[3 * j + d];
                }
                if (mx < -1e-9 * (1.0 + std::fabs(mx))) {
                    dominated = true;
                    break;
                }
            }
            if (!dominated) keep.push_back(j);
        }
        list.swap(keep);
    };
    // the palette entry nearest to each corner of the cube: bit d of the index set = the corner's coordinate d is 255
    int corner_entry[8];
    for (int c = 0; c < 8; ++c) {
        double best = std::numeric_limits<double>::infinity();
        corner_entry[c] = 0;
        for (int j = 0; j < K; ++j) {
            double d2 = 0.0;
            for (int d = 0; d < 3; ++d) {
                const double m = pts[3 * j + d] - ((c >> d) & 1 ? 255.0 : 0.0);
                d2 += m * m;
            }
            if (d2 < best) {
                best = d2;
                corner_entry[c] = j;
            }
        }
    }
    // count byte + up to 15 index bytes.  Lists of up to 12 entries are padded to a multiple of 4 positions with an entry that is
    // NOT on the list (the one farthest from the box): the key scan of nearest_color_cells evaluates whole groups of four
    // without per-position tests; an unlisted entry is never the nearest of a point of the box, and should float32
    // rounding bring it within the margin of the nearest, the exact scan -- which honours the count -- decides.
    auto pack = [&](const std::vector<int> &list, const double *lo = nullptr, const double size = 0.0) {
        uint32_t w[4] = {(uint32_t)list.size(), 0u, 0u, 0u};
        for (size_t n = 1; n <= list.size(); ++n) ed_list_put(w, (int)n - 1, list[n - 1], wide);
        if (lo && !list.empty() && list.size() <= 12 && (int)list.size() < K) {
            // (any unlisted entry will do; a far one keeps it out of the margin: the entry nearest to the cube corner
            // opposite to the box, unless that one is listed -- then the farthest by scan)
            const int oct = (lo[0] + 0.5 * size < 128.0 ? 1 : 0) | (lo[1] + 0.5 * size < 128.0 ? 2 : 0) | (lo[2] + 0.5 * size < 128.0 ? 4 : 0);
            int filler = corner_entry[oct];
            if (std::find(list.begin(), list.end(), filler) != list.end()) {
                filler = -1;
                double far_d = -1.0;
                for (int j = 0; j < K; ++j) {
                    if (std::find(list.begin(), list.end(), j) != list.end()) continue;
                    double d2 = 0.0;
                    for (int d = 0; d < 3; ++d) {
                        const double m = pts[3 * j + d] - (lo[d] + 0.5 * size);
                        d2 += m * m;
                    }
                    if (d2 > far_d) {
                        far_d = d2;
                        filler = j;
                    }
                }
            }
            const size_t upto = (list.size() + 3) / 4 * 4;
            for (size_t n = list.size() + 1; n <= upto && filler >= 0; ++n) ed_list_put(w, (int)n - 1, filler, wide);
        }
        return make_u4(w[0], w[1], w[2], w[3]);
    };
    struct Work {
        size_t slot;      // index into `nodes` (or, with top == true, into `host`)
        bool top;
        double lo[3];
        double size;
        std::vector<int> from;
    };
    std::vector<Work> stack;
    std::vector<int> all(K);
    for (int j = 0; j < K; ++j) all[j] = j;
    // the kernel's lists (the geometric criterion), sharpened by the pairwise test
    if (what & 1) parallel_cells(kEdCells, [&](const int c0, const int c1) {
        std::vector<int> list;
        for (int cell = c0; cell < c1; ++cell) {
            const uint32_t w4[4] = {host[cell].x, host[cell].y, host[cell].z, host[cell].w};
            const int n = (int)(w4[0] & 255u);
            if (n < 1 || n > cap) continue;
            list.clear();
            for (int i = 1; i <= n; ++i) list.push_back(ed_list_get(w4, i - 1, wide));
            const double lo[3] = {(double)((cell & 31) * 8), (double)(((cell >> 5) & 31) * 8), (double)((cell >> 10) * 8)};
            prune_list(lo, 8.0, list);
            host[cell] = pack(list, lo, 8.0);  // (re-packed even when nothing was dropped: the padding)
        }
    });
    // The cells whose geometric list overflowed: their full pruned lists and, where even those are longer than 15, their octrees.
    // Each such cell is refined on its own (in parallel: a crowded palette has hundreds of them, and a list of a hundred entries
    // costs ten thousand pairwise tests) into a LOCAL node array whose node numbers are made global when the arrays are
    // joined below -- in cell order and with the same depth-first order inside a cell as the single stack of rounds 2-5 had,
    // so the tables come out byte for byte as before.
    std::vector<int> over;
    for (int cell = 0; (what & 1) && cell < kEdCells; ++cell)
        if ((host[cell].x & 255u) == 255u) over.push_back(cell);
    struct Local {
        U4 top;
        std::vector<U4> nodes;   // 8 per local node; child pointers hold LOCAL node numbers
    };
    std::vector<Local> locals(over.size());
    auto refine_cell = [&](const int cell, Local &lc) {
        std::vector<Work> st;
        Work w0;
        w0.slot = 0;
        w0.top = true;
        w0.lo[0] = (double)((cell & 31) * 8);
        w0.lo[1] = (double)(((cell >> 5) & 31) * 8);
        w0.lo[2] = (double)((cell >> 10) * 8);
        w0.size = 8.0;
        box_list(all, w0.lo, 8.0, w0.from);  // the cell's full list
        prune_list(w0.lo, 8.0, w0.from);
        st.push_back(std::move(w0));
        while (!st.empty()) {
            Work wk = std::move(st.back());
            st.pop_back();
            U4 entry;
            if ((int)wk.from.size() <= cap) {
                entry = pack(wk.from, wk.lo, wk.size);
            } else if (wk.size <= 1.0) {
                entry = make_u4(255u, 0u, 0u, 0u);  // a unit cube that still sees more than 15 entries: scan the palette
            } else {
                const size_t node = lc.nodes.size() / 8;
                lc.nodes.resize(lc.nodes.size() + 8, make_u4(255u, 0u, 0u, 0u));
                entry = make_u4(254u | ((uint32_t)node << 8), 0u, 0u, 0u);
                const double hs = wk.size * 0.5;
                for (int sub = 0; sub < 8; ++sub) {
                    Work ch;
                    ch.slot = node * 8 + (size_t)sub;
                    ch.top = false;
                    ch.lo[0] = wk.lo[0] + ((sub & 1) ? hs : 0.0);
                    ch.lo[1] = wk.lo[1] + ((sub & 2) ? hs : 0.0);
                    ch.lo[2] = wk.lo[2] + ((sub & 4) ? hs : 0.0);
                    ch.size = hs;
                    box_list(wk.from, ch.lo, hs, ch.from);
                    prune_list(ch.lo, hs, ch.from);
                    st.push_back(std::move(ch));
                }
            }
            if (wk.top) lc.top = entry;
            else lc.nodes[wk.slot] = entry;
        }
    };
    if (over.size() >= 16) {
        parallel_cells((int)over.size(), [&](const int i0, const int i1) {
            for (int i = i0; i < i1; ++i) refine_cell(over[(size_t)i], locals[(size_t)i]);
        }, 16);
    } else {
        for (size_t i = 0; i < over.size(); ++i) refine_cell(over[i], locals[i]);
    }
    bool give_up = false;
    (void)stack;
    // joined in the order the single stack visited them: it held the overflowing cells in ascending order and popped the LAST first
    for (size_t ii = over.size(); ii-- > 0 && !give_up;) {
        const Local &lc = locals[ii];
        const size_t base = nodes.size() / 8;
        if (base + lc.nodes.size() / 8 >= (1u << 22)) {
            give_up = true;
            break;
        }
        auto fix = [&](U4 e) {
            if ((e.x & 255u) == 254u) e.x = 254u | ((uint32_t)(base + (e.x >> 8)) << 8);
            return e;
        };
        host[(size_t)over[ii]] = fix(lc.top);
        for (const U4 &e : lc.nodes) nodes.push_back(fix(e));
    }
    out.give_up = give_up;
    if (give_up) {
        // abandoned: no cell may point into the node array any more
        for (int cell = 0; cell < kEdCells; ++cell)
            if ((host[cell].x & 255u) == 254u) host[cell] = make_u4(255u, 0u, 0u, 0u);
        nodes.clear();
    }
    out.l16.clear();
    out.coarse.clear();
    out.ext.clear();
    out.ext16.clear();
    out.ext_nodes.clear();
    out.h4.clear();
    if (K > 16) {
        // lists of the 16x16x16 cells in the format of the 8x8x8 table, for the LDS of the wavefront kernel's few-frames
        // variant (one wave per SIMD: the read of the 8x8x8 table from L2 is half of a step's latency there)
        std::vector<U4> &l16 = out.l16;
        l16.resize(4096);
        std::vector<std::vector<int>> top16(4096);   // the cells' pruned lists: the hierarchical table below starts from them
        parallel_cells(4096, [&](const int c0, const int c1) {
            for (int cell = c0; cell < c1; ++cell) {
                std::vector<int> &list = top16[cell];
                const double lo[3] = {(double)((cell & 15) * 16), (double)(((cell >> 4) & 15) * 16), (double)((cell >> 8) * 16)};
                box_list(all, lo, 16.0, list);
                prune_list(lo, 16.0, list);
                l16[cell] = (int)list.size() <= cap ? pack(list, lo, 16.0) : make_u4(255u, 0u, 0u, 0u);
            }
        });
        // The extended table for query points that are NOT clamped to the cube (perceptual / hybrid / adaptive-variance): a point is
        // looked up in the cell of its clamped coordinates, so an outermost cell stands for everything beyond it.  Over such an
        // unbounded box the geometric criterion lists everybody; the pairwise test alone decides -- entry j stays unless some k is
        // closer on the WHOLE box (in a direction in which the box is unbounded and |x - c_k|^2 - |x - c_j|^2 grows, k cannot be).
        // Candidates k are tried nearest to the cell first (most entries fall to one of the first few).  Until round 5 these
        // diffusers scanned the whole palette at every step above 16 colours: 59 ms per 1080p frame at 256 colours against 4.6 ms
        // for plain error diffusion (tools/bench_scripts/cliff_hunt.py).
        out.ext16.clear();
        if (what & 2) {   // (byte entries up to 256 colours, ten-bit entries above: pack())
            std::vector<U4> &ext16 = out.ext16;
            ext16 = l16;
            const double kInfE = std::numeric_limits<double>::infinity();
            std::vector<int> shell;
            for (int cell = 0; cell < 4096; ++cell) {
                const int c0 = cell & 15, c1 = (cell >> 4) & 15, c2 = cell >> 8;
                if (c0 == 0 || c0 == 15 || c1 == 0 || c1 == 15 || c2 == 0 || c2 == 15) shell.push_back(cell);
            }
            // the entries of `from` that no other entry of `from` dominates over the box [blo, bhi] (sides may be infinite); `lo`/`size`:
            // the box's bounded part (where a point's clamped coordinates fall), by which the candidates are tried nearest-first
            auto dom_list = [&](const std::vector<int> &from, const double blo[3], const double bhi[3], const double lo[3], const double size,
                                std::vector<std::pair<double, int>> &order, std::vector<int> &list) {
                order.clear();
                for (int j : from) {
                    double d2 = 0.0;
                    for (int d = 0; d < 3; ++d) {
                        const double c = pts[3 * j + d];
                        const double m = std::max(std::max(lo[d] - c, c - (lo[d] + size)), 0.0);
                        d2 += m * m;
                    }
                    order.emplace_back(d2, j);
                }
                std::sort(order.begin(), order.end());
                list.clear();
                for (int j : from) {
                    bool dominated = false;
                    for (size_t kk = 0; kk < order.size() && !dominated; ++kk) {
                        const int k = order[kk].second;
                        if (k == j) continue;
                        double mx = 0.0;
                        for (int d = 0; d < 3; ++d) {
                            const double a = 2.0 * (pts[3 * j + d] - pts[3 * k + d]);
                            if (a > 0.0) mx += bhi[d] == kInfE ? kInfE : a * bhi[d];
                            else if (a < 0.0) mx += blo[d] == -kInfE ? kInfE : a * blo[d];
                            mx += pts[3 * k + d] * pts[3 * k + d] - pts[3 * j + d] * pts[3 * j + d];
                        }
                        if (mx < -1e-9 * (1.0 + std::fabs(mx))) dominated = true;
                    }
                    if (!dominated) list.push_back(j);
                }
            };
            // A cell whose list is too long (a palette crowded at a face of the cube -- every palette under use_gamma, at the dark end)
            // is cut into its eight half-size children like the 8^3 cells are: an outer child stays unbounded on its outer side, a point
            // beyond the cube falls into it by its clamped coordinates.  Local node arrays, joined in cell order below.
            struct XItem {
                size_t slot;   // 0: the cell's own entry; else 1 + index into the local node array
                double blo[3], bhi[3], lo[3], size;
                std::vector<int> from;
            };
            std::vector<std::vector<U4>> xlocal(shell.size());   // per shell cell: [0] its entry, then its nodes' entries (8 per node)
            parallel_cells((int)shell.size(), [&](const int i0, const int i1) {
                std::vector<std::pair<double, int>> order;
                std::vector<int> list;
                std::vector<XItem> todo;
                for (int si = i0; si < i1; ++si) {
                    const int cell = shell[(size_t)si];
                    const int ci[3] = {cell & 15, (cell >> 4) & 15, cell >> 8};
                    std::vector<U4> &xl = xlocal[(size_t)si];
                    xl.assign(1, make_u4(255u, 0u, 0u, 0u));
                    XItem top;
                    top.slot = 0;
                    top.size = 16.0;
                    for (int d = 0; d < 3; ++d) {
                        top.lo[d] = (double)(ci[d] * 16);
                        top.blo[d] = ci[d] == 0 ? -kInfE : top.lo[d];
                        top.bhi[d] = ci[d] == 15 ? kInfE : top.lo[d] + 16.0;
                    }
                    top.from = all;
                    todo.clear();
                    todo.push_back(std::move(top));
                    while (!todo.empty()) {
                        XItem it = std::move(todo.back());
                        todo.pop_back();
                        dom_list(it.from, it.blo, it.bhi, it.lo, it.size, order, list);
                        if ((int)list.size() <= cap) {
                            xl[it.slot] = pack(list, it.lo, it.size);
                            continue;
                        }
                        // (stays 255, the whole-palette scan: unit boxes; more than 24 nodes in one cell; and cells far from every
                        // colour of the palette -- error diffusion keeps its values near the palette, and it is exactly the far
                        // cells of a crowded palette, to which all its colours look alike, that are expensive to refine: with them
                        // the tables of a 256-colour median-cut palette took 77 ms instead of 17)
                        if (it.size <= 1.0 || (xl.size() - 1) / 8 >= 24 || order.empty() || order[0].first > 40.0 * 40.0) continue;
                        const size_t node = (xl.size() - 1) / 8;
                        xl[it.slot] = make_u4(254u | ((uint32_t)node << 8), 0u, 0u, 0u);
                        xl.resize(xl.size() + 8, make_u4(255u, 0u, 0u, 0u));
                        const double hs = it.size * 0.5;
                        for (int sub = 0; sub < 8; ++sub) {
                            XItem ch;
                            ch.slot = 1 + 8 * node + (size_t)sub;
                            ch.size = hs;
                            for (int d = 0; d < 3; ++d) {
                                const int bit = (sub >> d) & 1;
                                ch.lo[d] = it.lo[d] + (bit ? hs : 0.0);
                                ch.blo[d] = (!bit && it.blo[d] == -kInfE) ? -kInfE : ch.lo[d];
                                ch.bhi[d] = (bit && it.bhi[d] == kInfE) ? kInfE : ch.lo[d] + hs;
                            }
                            ch.from = list;
                            todo.push_back(std::move(ch));
                        }
                    }
                }
            }, 64);
            std::vector<U4> &xn = out.ext_nodes;
            xn.clear();
            for (size_t si = 0; si < shell.size(); ++si) {
                const std::vector<U4> &xl = xlocal[si];
                const size_t base = xn.size() / 8;
                U4 topw = xl[0];
                if (xl.size() > 1 && base + (xl.size() - 1) / 8 > kEdExtMaxNodes) topw = make_u4(255u, 0u, 0u, 0u);   // no room: scan
                else
                    for (size_t i = 1; i < xl.size(); ++i) {
                        U4 e = xl[i];
                        if ((e.x & 255u) == 254u) e.x = 254u | ((uint32_t)(base + (e.x >> 8)) << 8);
                        xn.push_back(e);
                    }
                if ((topw.x & 255u) == 254u) topw.x = 254u | ((uint32_t)(base + (topw.x >> 8)) << 8);
                ext16[(size_t)shell[si]] = topw;
            }
        }
        out.h4_wanted = 0;
        out.h4_depth = 0.0;
        out.h4_none = 0.0;
        if (!wide && (what & 1)) {   // (its leaves hold index BYTES: palettes of up to 256 colours)
        // The hierarchical table: a wave of the diffusion kernel pays for the LONGEST list among its 64 lanes, and with 256 random
        // colours 12 % of the 16^3 cells, 0.8 % of the 8-wide and 0.03 % of the 4-wide cells have more than four possible nearest
        // entries -- so a cell with more than four is cut into its eight 8-wide children, such a child into its 4-wide children,
        // and every lane ends at a leaf of at most four (what is still longer at 4-wide keeps the lists above).
        auto leaf = [&](const std::vector<int> &list, const double lo[3], const double size) -> uint32_t {
            // filler: an unlisted entry far from the box, as pack() chooses it
            const int oct = (lo[0] + 0.5 * size < 128.0 ? 1 : 0) | (lo[1] + 0.5 * size < 128.0 ? 2 : 0) | (lo[2] + 0.5 * size < 128.0 ? 4 : 0);
            int filler = corner_entry[oct];
            if (std::find(list.begin(), list.end(), filler) != list.end()) {
                filler = -1;
                double far_d = -1.0;
                for (int j = 0; j < K; ++j) {
                    if (std::find(list.begin(), list.end(), j) != list.end()) continue;
                    double d2 = 0.0;
                    for (int d = 0; d < 3; ++d) {
                        const double m = pts[3 * j + d] - (lo[d] + 0.5 * size);
                        d2 += m * m;
                    }
                    if (d2 > far_d) {
                        far_d = d2;
                        filler = j;
                    }
                }
            }
            int b[4];
            for (int n = 0; n < 4; ++n) b[n] = n < (int)list.size() ? list[n] : filler;
            // byte 0 < byte 1 marks a leaf: two distinct indices always exist (a list of one is padded with its filler)
            if (b[0] > b[1]) std::swap(b[0], b[1]);
            return (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
        };
        std::vector<std::vector<uint32_t>> local(4096);   // per cell: [0] its word (marker: local node 0), then its nodes' words
        parallel_cells(4096, [&](const int c0, const int c1) {
            // (depth-first: a box with more than four entries becomes a node of its eight half-size children, down to 2-wide boxes)
            struct Item {
                size_t slot;
                double lo[3];
                double size;
                std::vector<int> from;
            };
            std::vector<Item> todo;
            std::vector<int> list;
            for (int cell = c0; cell < c1; ++cell) {
                std::vector<uint32_t> &lw = local[cell];
                lw.assign(1, kEdH4NoAnswer);
                todo.clear();
                Item top;
                top.slot = 0;
                top.lo[0] = (double)((cell & 15) * 16);
                top.lo[1] = (double)(((cell >> 4) & 15) * 16);
                top.lo[2] = (double)((cell >> 8) * 16);
                top.size = 16.0;
                todo.push_back(std::move(top));
                while (!todo.empty()) {
                    Item it = std::move(todo.back());
                    todo.pop_back();
                    if (it.size == 16.0) {
                        list = top16[cell];   // (box_list + prune_list of the whole palette: done above)
                    } else {
                        box_list(it.from, it.lo, it.size, list);
                        prune_list(it.lo, it.size, list);
                    }
                    if (list.size() <= 4) {
                        lw[it.slot] = leaf(list, it.lo, it.size);
                        continue;
                    }
                    if (it.size <= 2.0) continue;   // still longer in a 2-wide box: no answer here (the lists decide)
                    const uint32_t node = (uint32_t)((lw.size() - 1) / 8);   // local node number of the children
                    lw[it.slot] = 0x00ffu | (node << 16);
                    lw.resize(lw.size() + 8, kEdH4NoAnswer);
                    const double hs = it.size * 0.5;
                    for (int sub = 0; sub < 8; ++sub) {
                        Item ch;
                        ch.slot = 1 + 8 * (size_t)node + (size_t)sub;
                        ch.lo[0] = it.lo[0] + ((sub & 1) ? hs : 0.0);
                        ch.lo[1] = it.lo[1] + ((sub & 2) ? hs : 0.0);
                        ch.lo[2] = it.lo[2] + ((sub & 4) ? hs : 0.0);
                        ch.size = hs;
                        ch.from = list;
                        todo.push_back(std::move(ch));
                    }
                }
            }
        });
        std::vector<uint32_t> &h4 = out.h4;
        out.h4_wanted = 0;
        for (int cell = 0; cell < 4096; ++cell) out.h4_wanted += local[cell].size();
        h4.assign(4096, kEdH4NoAnswer);
        bool fits = true;
        for (int cell = 0; cell < 4096 && fits; ++cell) {
            const std::vector<uint32_t> &lw = local[cell];
            if (lw.size() == 1) {
                h4[cell] = lw[0];
                continue;
            }
            const size_t base = (h4.size() - 4096) / 8;   // global number of this cell's local node 0
            if (h4.size() + (lw.size() - 1) > kEdH4MaxWords || base + (lw.size() - 1) / 8 >= 0xffffu) {
                fits = false;
                break;
            }
            h4[cell] = 0x00ffu | ((uint32_t)base << 16);
            for (size_t i = 1; i < lw.size(); ++i) {
                uint32_t w = lw[i];
                const bool marker = (w & 0xffu) >= ((w >> 8) & 0xffu);
                if (marker && w != kEdH4NoAnswer) w = 0x00ffu | ((uint32_t)(base + (w >> 16)) << 16);
                h4.push_back(w);
            }
        }
        if (!fits) h4.clear();   // a clustered palette: the lists (and their octree refinement) serve it
        out.h4_depth = 0.0;
        out.h4_none = 0.0;
        if (!h4.empty()) {
            auto marker = [](const uint32_t w) { return (w & 0xffu) >= ((w >> 8) & 0xffu); };
            double total = 0.0, none = 0.0;
            for (int j = 0; j < K; ++j) {
                uint32_t x[3];
                for (int d = 0; d < 3; ++d) x[d] = (uint32_t)std::min(255.0, std::max(0.0, std::floor(pts[3 * j + d] + 0.5)));
                uint32_t w = h4[(x[0] >> 4) | ((x[1] >> 4) << 4) | ((x[2] >> 4) << 8)];
                int depth = 0;
                for (int bit = 3; bit >= 1 && marker(w); --bit) {
                    if ((w >> 16) == 0xffffu) break;
                    w = h4[4096u + (size_t)(w >> 16) * 8u + (((x[0] >> bit) & 1u) | (((x[1] >> bit) & 1u) << 1) | (((x[2] >> bit) & 1u) << 2))];
                    ++depth;
                }
                total += marker(w) ? 4.0 : (double)depth;
                none += marker(w) ? 1.0 : 0.0;
            }
            out.h4_depth = total / (double)K;
            out.h4_none = none / (double)K;
        }
        }
    }
    if (K <= 16 && (what & 1)) {
        // lists of the 16x16x16 cells for the wavefront kernel's LDS: count | up to 7 indices, one nibble each
        std::vector<uint32_t> &coarse = out.coarse;
        coarse.resize(4096);
        parallel_cells(4096, [&](const int c0, const int c1) {
            std::vector<int> list;
            for (int cell = c0; cell < c1; ++cell) {
                const double lo[3] = {(double)((cell & 15) * 16), (double)(((cell >> 4) & 15) * 16), (double)((cell >> 8) * 16)};
                box_list(all, lo, 16.0, list);
                prune_list(lo, 16.0, list);
                uint32_t word = 15u;
                if (list.size() <= 7) {
                    word = (uint32_t)list.size();
                    for (size_t n = 0; n < list.size(); ++n) word |= (uint32_t)list[n] << (4 * (n + 1));
                    // The unused positions name the entry farthest from the cell that is not on the list (K > 8 > list size:
                    // there is one).  The key scan of nearest_color_cells evaluates every position of a group of 4 (or 7)
                    // without a per-position validity test; an entry that is not listed can never be the nearest one, and
                    // if it comes within the margin of the nearest the exact scan (which honours the count) decides.
                    int filler = -1;
                    double far_d = -1.0;
                    for (int j = 0; j < K; ++j) {
                        if (std::find(list.begin(), list.end(), j) != list.end()) continue;
                        double near2 = 0.0;
                        for (int k = 0; k < 3; ++k) {
                            const double c = pts[3 * j + k];
                            const double m = std::max(std::max(lo[k] - c, c - (lo[k] + 16.0)), 0.0);
                            near2 += m * m;
                        }
                        if (near2 > far_d) {
                            far_d = near2;
                            filler = j;
                        }
                    }
                    for (size_t n = list.size(); n < 7 && filler >= 0; ++n) word |= (uint32_t)filler << (4 * (n + 1));
                }
                coarse[cell] = word;
            }
        });
        // The same table for query points that are NOT clamped to the cube (the perceptual / hybrid / adaptive-variance
        // diffusers of vardiff.hip): a point is looked up in the cell of its clamped coordinates, so the outermost cells
        // stand for everything beyond them -- their boxes are unbounded on that side.  With an unbounded box the geometric
        // criterion lists everybody; the pairwise test alone decides (K <= 16: 256 pairs per cell): over a box that is
        // unbounded in a direction in which |x - c_k|^2 - |x - c_j|^2 grows, k cannot dominate j.
        std::vector<uint32_t> &ext = out.ext;
        ext.resize(4096);
        const double kInf = std::numeric_limits<double>::infinity();
        parallel_cells(4096, [&](const int c0, const int c1) {
            std::vector<int> list;
            for (int cell = c0; cell < c1; ++cell) {
                const int ci[3] = {cell & 15, (cell >> 4) & 15, cell >> 8};
                double blo[3], bhi[3];
                for (int d = 0; d < 3; ++d) {
                    blo[d] = ci[d] == 0 ? -kInf : (double)(ci[d] * 16);
                    bhi[d] = ci[d] == 15 ? kInf : (double)(ci[d] * 16 + 16);
                }
                list.clear();
                for (int j = 0; j < K; ++j) {
                    bool dominated = false;
                    for (int k = 0; k < K && !dominated; ++k) {
                        if (k == j) continue;
                        double mx = 0.0;
                        for (int d = 0; d < 3; ++d) {
                            const double a = 2.0 * (pts[3 * j + d] - pts[3 * k + d]);
                            if (a > 0.0) mx += bhi[d] == kInf ? kInf : a * bhi[d];
                            else if (a < 0.0) mx += blo[d] == -kInf ? kInf : a * blo[d];
                            mx += pts[3 * k + d] * pts[3 * k + d] - pts[3 * j + d] * pts[3 * j + d];
                        }
                        if (mx < -1e-9 * (1.0 + std::fabs(mx))) dominated = true;
                    }
                    if (!dominated) list.push_back(j);
                }
                uint32_t word = 15u;
                if (list.size() <= 7) {
                    word = (uint32_t)list.size();
                    for (size_t n = 0; n < list.size(); ++n) word |= (uint32_t)list[n] << (4 * (n + 1));
                    int filler = -1;
                    double far_d = -1.0;
                    for (int j = 0; j < K; ++j) {  // unused positions: the unlisted entry farthest from the cell's inner corner
                        if (std::find(list.begin(), list.end(), j) != list.end()) continue;
                        double d2 = 0.0;
                        for (int d = 0; d < 3; ++d) {
                            const double m = pts[3 * j + d] - (double)(ci[d] * 16 + 8);
                            d2 += m * m;
                        }
                        if (d2 > far_d) {
                            far_d = d2;
                            filler = j;
                        }
                    }
                    for (size_t n = list.size(); n < 7 && filler >= 0; ++n) word |= (uint32_t)filler << (4 * (n + 1));
                }
                ext[cell] = word;
            }
        });
    }
}

// ---------------------------------------------------------------------------------------------
// Median cut as the reference runs it (ColorReducer.reduce_colors, dithering_lib.py:1813-1843):
//     median_cut(list(set(image.getdata())), depth)
// The cut sorts stably, so the ITERATION ORDER OF THE PYTHON SET is observable in the palette.  That order is a pure
// function of the insertion sequence: CPython's tuple hash (xxHash-style, Objects/tupleobject.c, 3.8+; small ints hash to
// themselves) and the open-addressing table of Objects/setobject.c (linear probes of 9 + perturbed jumps, growth to
// used*4 -- used*2 above 50 000 -- whenever fill*5 >= mask*3, re-insertion in slot order).  pyset_order() replays it and
// returns the elements in the order `list(set(...))` yields them, without creating a Python object per colour (1.2 M
// tuples of a 4K photograph: ~1 s of interpreter time).  The Python side checks the replay against the running
// interpreter's own set once per process and keeps building real sets if they ever disagree.
// ---------------------------------------------------------------------------------------------
inline uint64_t py_tuple3_hash(const uint64_t a, const uint64_t b, const uint64_t c)
{
    constexpr uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P5 = 2870177450012600261ULL;
    uint64_t acc = P5;
    for (const uint64_t lane : {a, b, c}) {
        acc += lane * P2;
        acc = (acc << 31) | (acc >> 33);
        acc *= P1;
    }
    acc += 3ULL ^ (P5 ^ 3527539ULL);
    return acc == ~0ULL ? 1546275796ULL : acc;
}

// rgb: n colours (3 bytes each) in insertion order, duplicates allowed.  order: for every element of the resulting set, in
// the set's iteration order, the index of its first occurrence in rgb.
// An 8-byte slot {low 32 bits of the hash, index + 1}: the home slot and the duplicate test need no more (tables have fewer than
// 2^32 slots; equal low words are settled by comparing the colours), and the high bits only enter the perturbed jumps after nine
// occupied neighbours, where the hash is simply recomputed from the colour.  Half the table (4 MB instead of 8 for 180 k
// colours) is half the cache misses.  (Prefetching home slots ahead was measured on the bench box's EPYC 9575F and costs time:
// 4.8 -> 6.1 ms, profiles/experiments/r04_pyset_prefetch_ab.txt.)
struct PySetSlot {
    uint32_t h;     // low word of the hash
    uint32_t idx1;  // index of the colour + 1 (0 = unused slot)
};

// Two slot arrays per thread, kept from call to call (a 180 k-colour set walks through 8 MB of tables; fresh allocations cost
// more in page faults than the insertions themselves).  A buffer is handed out zeroed over the slots the caller asks for:
// only what an earlier table left dirty is cleared.
struct PySetTables {
    std::vector<PySetSlot> buf[2];
    size_t dirty[2] = {0, 0};   // slots of each buffer that may be non-zero
    PySetSlot *fresh(const int which, const size_t size)
    {
        if (buf[which].size() < size) {
            buf[which].assign(size, PySetSlot{0, 0});
        } else if (dirty[which]) {
            std::memset(buf[which].data(), 0, std::min(dirty[which], buf[which].size()) * sizeof(PySetSlot));
        }
        dirty[which] = size;
        return buf[which].data();
    }
    void trim()   // (a one-off giant image must not pin hundreds of megabytes for the life of the thread)
    {
        for (int w = 0; w < 2; ++w)
            if (buf[w].size() > ((size_t)1 << 23)) {
                std::vector<PySetSlot>().swap(buf[w]);
                dirty[w] = 0;
            }
    }
};

inline void pyset_order(const uint8_t *rgb, const size_t n, std::vector<uint32_t> &order)
{
    constexpr size_t kLinearProbes = 9;
    constexpr int kPerturbShift = 5;
    order.clear();
    static thread_local PySetTables tables;
    size_t mask = 7, fill = 0;
    int cur = 0;
    PySetSlot *tab = tables.fresh(0, 8);
    auto hash_of = [&](const size_t k) { return py_tuple3_hash(rgb[3 * k], rgb[3 * k + 1], rgb[3 * k + 2]); };
    auto same = [&](const uint32_t slot_idx, const size_t k) {
        const uint8_t *p = rgb + 3 * (size_t)(slot_idx - 1), *q = rgb + 3 * k;
        return p[0] == q[0] && p[1] == q[1] && p[2] == q[2];
    };
    // insertion of an element that is known to be absent (growth steps)
    auto insert_clean = [&](PySetSlot *nt, const size_t m, const uint32_t hlo, const uint32_t idx1) {
        size_t i = (size_t)hlo & m;
        if (nt[i].idx1 == 0) {
            nt[i] = PySetSlot{hlo, idx1};
            return;
        }
        uint64_t perturb = 0;
        bool have_hash = false;
        for (;;) {
            if (i + kLinearProbes <= m)
                for (size_t j = 1; j <= kLinearProbes; ++j)
                    if (nt[i + j].idx1 == 0) {
                        nt[i + j] = PySetSlot{hlo, idx1};
                        return;
                    }
            if (!have_hash) {
                perturb = hash_of((size_t)idx1 - 1);
                have_hash = true;
            }
            perturb >>= kPerturbShift;
            i = (i * 5 + 1 + (size_t)perturb) & m;
            if (nt[i].idx1 == 0) {
                nt[i] = PySetSlot{hlo, idx1};
                return;
            }
        }
    };
    for (size_t k = 0; k < n; ++k) {
        const uint64_t h = hash_of(k);
        const uint32_t hlo = (uint32_t)h;
        uint64_t perturb = h;
        size_t i = (size_t)h & mask;
        bool added = false;
        for (bool done = false; !done;) {
            size_t probes = i + kLinearProbes <= mask ? kLinearProbes : 0;
            for (size_t j = i;; ++j) {
                if (tab[j].idx1 == 0) {
                    tab[j] = PySetSlot{hlo, (uint32_t)k + 1u};
                    added = done = true;
                    break;
                }
                if (tab[j].h == hlo && same(tab[j].idx1, k)) {
                    done = true;  // already in the set
                    break;
                }
                if (probes == 0) break;
                --probes;
            }
            if (!done) {
                perturb >>= kPerturbShift;
                i = (i * 5 + 1 + (size_t)perturb) & mask;
            }
        }
        if (added && ++fill * 5 >= mask * 3) {
            const size_t minused = fill > 50000 ? fill * 2 : fill * 4;  // (no deletions: used == fill)
            size_t newsize = 8;
            while (newsize <= minused) newsize <<= 1;
            PySetSlot *nt = tables.fresh(cur ^ 1, newsize);
            const size_t m = newsize - 1;
            for (size_t slot = 0; slot <= mask; ++slot)   // re-insertion in slot order (the old table is read sequentially)
                if (tab[slot].idx1 != 0) insert_clean(nt, m, tab[slot].h, tab[slot].idx1);
            tab = nt;
            cur ^= 1;
            mask = m;
        }
    }
    order.reserve(fill);
    for (size_t slot = 0; slot <= mask; ++slot)
        if (tab[slot].idx1 != 0) order.push_back(tab[slot].idx1 - 1u);
    tables.trim();
}

// median_cut(colors, depth) of the reference (dithering_lib.py:1822-1833) on n colours in list order: the first widest
// channel, a STABLE sort on it (a counting sort over the 256 values), split at n // 2, at depth 0 the per-channel mean
// int(sum / n) (true division in float64, truncated); an empty bucket yields the single entry (0, 0, 0) at any depth.
// The colours travel as packed words r | g << 8 | b << 16 between two buffers (cur -> other at every level: no copy back);
// the palette entries are appended to out (3 ints each).
// threads > 1: the two halves of a big bucket are cut concurrently (they work on disjoint parts of both buffers; the left
// half's entries come first in the palette, as in the recursion).
inline void median_cut_u32(uint32_t *cur, uint32_t *other, const size_t n, const int depth, std::vector<int32_t> &out, const int threads = 1)
{
    if (n == 0) {
        out.insert(out.end(), {0, 0, 0});
        return;
    }
    if (depth == 0) {
        uint64_t s0 = 0, s1 = 0, s2 = 0;
        for (size_t i = 0; i < n; ++i) {
            const uint32_t v = cur[i];
            s0 += v & 255u;
            s1 += (v >> 8) & 255u;
            s2 += v >> 16;
        }
        out.push_back((int32_t)((double)s0 / (double)n));
        out.push_back((int32_t)((double)s1 / (double)n));
        out.push_back((int32_t)((double)s2 / (double)n));
        return;
    }
    // per-channel minimum and maximum, all three at once on the packed word (byte-wise min / max: the compiler vectorises it)
    uint8_t lo[4] = {255, 255, 255, 255}, hi[4] = {0, 0, 0, 0};
    {
        const uint8_t *bytes = reinterpret_cast<const uint8_t *>(cur);
        uint8_t l0 = 255, l1 = 255, l2 = 255, h0 = 0, h1 = 0, h2 = 0;
        for (size_t i = 0; i < n; ++i) {
            const uint8_t a = bytes[4 * i], b = bytes[4 * i + 1], c = bytes[4 * i + 2];
            l0 = a < l0 ? a : l0;
            h0 = a > h0 ? a : h0;
            l1 = b < l1 ? b : l1;
            h1 = b > h1 ? b : h1;
            l2 = c < l2 ? c : l2;
            h2 = c > h2 ? c : h2;
        }
        lo[0] = l0; lo[1] = l1; lo[2] = l2;
        hi[0] = h0; hi[1] = h1; hi[2] = h2;
    }
    int ch = 0;
    for (int c = 1; c < 3; ++c)
        if (hi[c] - lo[c] > hi[ch] - lo[ch]) ch = c;  // the first of equal spans
    const int sh = 8 * ch;
    size_t start[257] = {0};
    for (size_t i = 0; i < n; ++i) ++start[((cur[i] >> sh) & 255u) + 1];
    for (int v = 0; v < 256; ++v) start[v + 1] += start[v];
    for (size_t i = 0; i < n; ++i) other[start[(cur[i] >> sh) & 255u]++] = cur[i];
    const size_t half = n / 2;
    if (threads > 1 && n >= 16384) {
        // (an exception -- bad_alloc while `right` or `out` grows -- must neither leave the lambda, which would call
        // std::terminate, nor destroy `t` while it is joinable, which would too: it is carried across the join and rethrown)
        std::vector<int32_t> right;
        std::exception_ptr failed;
        std::thread t([&] {
            try {
                median_cut_u32(other + half, cur + half, n - half, depth - 1, right, threads / 2);
            } catch (...) {
                failed = std::current_exception();
            }
        });
        struct Joiner {
            std::thread &t;
            ~Joiner() { if (t.joinable()) t.join(); }
        } joiner{t};
        median_cut_u32(other, cur, half, depth - 1, out, threads - threads / 2);
        t.join();
        if (failed) std::rethrow_exception(failed);
        out.insert(out.end(), right.begin(), right.end());
        return;
    }
    median_cut_u32(other, cur, half, depth - 1, out, 1);
    median_cut_u32(other + half, cur + half, n - half, depth - 1, out, 1);
}

}  // namespace dp
