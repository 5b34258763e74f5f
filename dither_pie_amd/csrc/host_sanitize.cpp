// CPU-sanitizer harness for the host logic of libditherpie_hip.so (host_logic.h): built WITHOUT HIP by
//   make host_asan   ->  build/host_asan   (g++ -fsanitize=address,undefined)
//   make host_tsan   ->  build/host_tsan   (g++ -fsanitize=thread: the multi-threaded per-cell loops)
// and run by tests/test_host_sanitizers.py in the CPU tier.  Test infrastructure, not part of the product library.
//
//   host_xxx kdtree   <pts.f64> <K>   the scipy-order KD-tree build; prints indices / nodes / splits (compared with
//                                     the golden cKDTree structures by the test)
//   host_xxx edtables <pts.f64> <K>   the diffusion candidate tables from host-made geometric lists (the criterion of
//                                     ed_cells_kernel), then checks on sampled points that the true nearest entry (and
//                                     everything tied with it) is on the list the kernels would search
//   host_xxx accel    <pts.f64> <K> <bw>   the accelerator's cell table from brute-force membership masks (what
//                                     accel_scan_kernel computes on the device), nearest sets, staging orders, wide
//                                     lists, node reordering, warp maps; checks on sampled colours that the block a
//                                     kernel would reach holds all of T(x)
//   host_xxx mediancut <rgb.u8> <n> <depth>   the replay of CPython's set order + the counting-sort median cut; prints the
//                                     palette and the first 2000 entries of the order (compared with the interpreter's own
//                                     set and with the Python cut by the test)
// Exit code 0 = all checks passed (and the sanitizer had nothing to say).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "host_logic.h"

using namespace dp;

static std::vector<double> read_pts(const char *path, int K)
{
    std::vector<double> p((size_t)K * 3);
    FILE *f = fopen(path, "rb");
    if (!f || fread(p.data(), sizeof(double), p.size(), f) != p.size()) {
        fprintf(stderr, "cannot read %d points from %s\n", K, path);
        exit(2);
    }
    fclose(f);
    return p;
}

static uint32_t lcg(uint32_t &s)
{
    s = s * 1664525u + 1013904223u;
    return s >> 8;
}

static int run_kdtree(const std::vector<double> &pts, int K)
{
    HostTree t;
    build_tree(pts.data(), K, t);
    printf("indices");
    for (int v : t.indices) printf(" %d", v);
    printf("\nnodes %zu\n", t.split_dim.size());
    for (size_t i = 0; i < t.split_dim.size(); ++i)
        printf("%d %d %d %d %d %.17g\n", t.split_dim[i], t.start[i], t.end[i], t.less[i], t.greater[i], t.split[i]);
    printf("box %.17g %.17g %.17g %.17g %.17g %.17g\n", t.mins[0], t.mins[1], t.mins[2], t.maxes[0], t.maxes[1], t.maxes[2]);
    return 0;
}

// ---- diffusion tables ----------------------------------------------------------------------------------------------
static void geometric_list(const std::vector<double> &pts, int K, const double lo[3], double size, std::vector<int> &list)
{
    double bound = std::numeric_limits<double>::infinity();
    for (int j = 0; j < K; ++j) {
        double far2 = 0.0;
        for (int k = 0; k < 3; ++k) {
            const double c = pts[3 * j + k], m = std::max(std::fabs(c - lo[k]), std::fabs(c - (lo[k] + size)));
            far2 += m * m;
        }
        bound = std::min(bound, far2);
    }
    bound = bound * (1.0 + 1e-6) + 1e-3;  // (the kernel works in float32 with a slack)
    list.clear();
    for (int j = 0; j < K; ++j) {
        double near2 = 0.0;
        for (int k = 0; k < 3; ++k) {
            const double c = pts[3 * j + k], m = std::max(std::max(lo[k] - c, c - (lo[k] + size)), 0.0);
            near2 += m * m;
        }
        if (near2 <= bound) list.push_back(j);
    }
}

static bool listed(const U4 &e, int n_max, int j, bool wide = false)
{
    const uint32_t w[4] = {e.x, e.y, e.z, e.w};
    const int n = (int)(w[0] & 255u);
    if (n > n_max) return true;  // overflow marker: the kernel scans the palette
    for (int i = 1; i <= n; ++i)
        if (ed_list_get(w, i - 1, wide) == j) return true;
    return false;
}

static int run_edtables(const std::vector<double> &pts, int K)
{
    if (K < 9 || K > 1024) {
        fprintf(stderr, "edtables: 9 <= K <= 1024\n");
        return 2;
    }
    const bool wide = K > 256;   // ten-bit list entries (host_logic.h: ed_list_put)
    const int cap = ed_list_cap(K);
    std::vector<U4> cells(kEdCells);
    std::vector<int> list;
    for (int cell = 0; cell < kEdCells; ++cell) {
        const double lo[3] = {(double)((cell & 31) * 8), (double)(((cell >> 5) & 31) * 8), (double)((cell >> 10) * 8)};
        geometric_list(pts, K, lo, 8.0, list);
        uint32_t w[4] = {255u, 0u, 0u, 0u};
        if ((int)list.size() <= cap) {
            w[0] = (uint32_t)list.size();
            for (size_t n = 1; n <= list.size(); ++n) ed_list_put(w, (int)n - 1, list[n - 1], wide);
        }
        cells[cell] = make_u4(w[0], w[1], w[2], w[3]);
    }
    EdTables tb;
    ed_tables_refine(pts.data(), K, cells, tb, 3);   // (the diffusion tables AND the extended lists of the unclamped diffusers)
    // sampled points: uniform integers, the palette entries themselves (rounded) and their neighbours
    uint32_t seed = 12345u + (uint32_t)K;
    long checked = 0, bad = 0, h4_answers = 0, h4_none = 0;
    auto check_point = [&](const int x[3]) {
        double best = std::numeric_limits<double>::infinity();
        for (int j = 0; j < K; ++j) {
            double d = 0;
            for (int k = 0; k < 3; ++k) d += (x[k] - pts[3 * j + k]) * (x[k] - pts[3 * j + k]);
            best = std::min(best, d);
        }
        // walk: 8^3 cell -> nodes
        U4 e = cells[(x[0] >> 3) | ((x[1] >> 3) << 5) | ((x[2] >> 3) << 10)];
        double lo[3] = {(double)(x[0] & ~7), (double)(x[1] & ~7), (double)(x[2] & ~7)}, size = 8.0;
        while ((e.x & 255u) == 254u) {
            const size_t node = e.x >> 8;
            size *= 0.5;
            int sub = 0;
            for (int k = 0; k < 3; ++k)
                if (x[k] >= lo[k] + size) {
                    sub |= 1 << k;
                    lo[k] += size;
                }
            if (node * 8 + sub >= tb.nodes.size()) {
                ++bad;
                return;
            }
            e = tb.nodes[node * 8 + sub];
        }
        const int c16 = (x[0] >> 4) | ((x[1] >> 4) << 4) | ((x[2] >> 4) << 8);
        for (int j = 0; j < K; ++j) {
            double d = 0;
            for (int k = 0; k < 3; ++k) d += (x[k] - pts[3 * j + k]) * (x[k] - pts[3 * j + k]);
            if (d != best) continue;
            ++checked;
            if (!listed(e, cap, j, wide)) ++bad;
            if (!tb.l16.empty() && !listed(tb.l16[c16], cap, j, wide)) ++bad;
            if (!tb.h4.empty()) {
                // the hierarchical <= 4-entry table as nearest_h4 (ed_nearest.hip.h) walks it: a leaf must hold every nearest entry
                auto marker = [](const uint32_t w) { return (w & 0xffu) >= ((w >> 8) & 0xffu); };
                uint32_t w = tb.h4[c16];
                bool answer = true;
                for (int bit = 3; bit >= 1 && marker(w); --bit) {
                    if ((w >> 16) == 0xffffu) {
                        answer = false;
                        break;
                    }
                    const size_t at = 4096u + (size_t)(w >> 16) * 8u + (size_t)(((x[0] >> bit) & 1) | (((x[1] >> bit) & 1) << 1) | (((x[2] >> bit) & 1) << 2));
                    if (at >= tb.h4.size()) {
                        ++bad;
                        answer = false;
                        break;
                    }
                    w = tb.h4[at];
                }
                if (answer && marker(w)) answer = false;   // (still a node pointer below 2-wide: no answer, the lists decide)
                if (answer) {
                    ++h4_answers;
                    bool on_leaf = false;
                    for (int b = 0; b < 4; ++b) on_leaf |= (int)((w >> (8 * b)) & 255u) == j;
                    if (!on_leaf) ++bad;
                } else {
                    ++h4_none;
                }
            }
            for (const std::vector<uint32_t> *tab : {&tb.coarse, &tb.ext}) {
                if (tab->empty()) continue;
                const uint32_t word = (*tab)[c16];
                const int n = (int)(word & 15u);
                bool ok = n > 7;
                for (int i = 0; i < n && !ok; ++i) ok = (int)((word >> (4 * (i + 1))) & 15u) == j;
                if (!ok) ++bad;
            }
        }
    };
    for (int i = 0; i < 40000; ++i) {
        const int x[3] = {(int)(lcg(seed) & 255u), (int)(lcg(seed) & 255u), (int)(lcg(seed) & 255u)};
        check_point(x);
    }
    for (int j = 0; j < K; ++j)
        for (int d = -1; d <= 1; ++d) {
            int x[3];
            for (int k = 0; k < 3; ++k) x[k] = std::min(255, std::max(0, (int)std::lround(pts[3 * j + k]) + d));
            check_point(x);
        }
    // the extended 16^3 lists (17..256 colours, unclamped diffusers): points in and around the cube, looked up by their clamped cell
    long ext_checked = 0, ext_long = 0;
    if (!tb.ext16.empty()) {
        for (int i = 0; i < 60000; ++i) {
            double x[3];
            for (int k = 0; k < 3; ++k) x[k] = (double)((int)(lcg(seed) % 640u) - 192) + (double)(lcg(seed) & 255u) / 256.0;   // [-192, 448)
            double best = std::numeric_limits<double>::infinity();
            for (int j = 0; j < K; ++j) {
                double d = 0;
                for (int k = 0; k < 3; ++k) d += (x[k] - pts[3 * j + k]) * (x[k] - pts[3 * j + k]);
                best = std::min(best, d);
            }
            int c[3];
            for (int k = 0; k < 3; ++k) c[k] = std::min(15, std::max(0, (int)x[k] >> 4));   // as nearest_ext16 (vardiff.hip)
            U4 e = tb.ext16[(size_t)(c[0] | (c[1] << 4) | (c[2] << 8))];
            int xi[3];
            for (int k = 0; k < 3; ++k) xi[k] = std::min(255, std::max(0, (int)x[k]));   // clamped integer coordinates choose the children
            for (int bit = 3; (e.x & 255u) == 254u; --bit) {
                const size_t at = (size_t)(e.x >> 8) * 8 + (size_t)(((xi[0] >> bit) & 1) | (((xi[1] >> bit) & 1) << 1) | (((xi[2] >> bit) & 1) << 2));
                if (bit < 0 || at >= tb.ext_nodes.size()) {
                    ++bad;
                    break;
                }
                e = tb.ext_nodes[at];
            }
            if ((e.x & 255u) == 254u) continue;
            if ((int)(e.x & 255u) > cap) {
                ++ext_long;
                continue;
            }
            for (int j = 0; j < K; ++j) {
                double d = 0;
                for (int k = 0; k < 3; ++k) d += (x[k] - pts[3 * j + k]) * (x[k] - pts[3 * j + k]);
                if (d != best) continue;
                ++ext_checked;
                if (!listed(e, cap, j, wide)) ++bad;
            }
        }
    }
    printf("ext16=%zu nodes=%zu (checked %ld, in cells with long lists %ld) ", tb.ext16.size(), tb.ext_nodes.size() / 8, ext_checked, ext_long);
    printf("edtables K=%d nodes=%zu give_up=%d l16=%zu coarse=%zu ext=%zu h4=%zu (answers %ld, none %ld) checked=%ld bad=%ld\n", K, tb.nodes.size(),
           (int)tb.give_up, tb.l16.size(), tb.coarse.size(), tb.ext.size(), tb.h4.size(), h4_answers, h4_none, checked, bad);
    if (K > 16 && !tb.h4.empty() && h4_answers < 100 * std::max(h4_none, 1L)) return 1;   // the table must answer nearly always
    return bad ? 1 : 0;
}

// ---- accelerator table -----------------------------------------------------------------------------------------------
static int run_accel(const std::vector<double> &pts, int K, int bw)
{
    if (K < 2 || K > 64 || (bw != 4 && bw != 8)) {
        fprintf(stderr, "accel: 2 <= K <= 64 (the masks are brute-forced over 2^24 colours), bw 4 or 8\n");
        return 2;
    }
    const int mw = (K + 31) / 32;
    std::vector<int> pr(K), pg(K), pb(K);
    std::vector<uint32_t> coord4(K);
    for (int j = 0; j < K; ++j) {
        pr[j] = (int)pts[3 * j], pg[j] = (int)pts[3 * j + 1], pb[j] = (int)pts[3 * j + 2];
        coord4[j] = (uint32_t)pr[j] | ((uint32_t)pg[j] << 8) | ((uint32_t)pb[j] << 16);
    }
    // T(x) = everything at least as close as the second nearest; N(x) = the nearest set
    auto sets_of = [&](int r, int g, int b, uint32_t *tm, uint32_t *nm) {
        int d[64], d0 = 1 << 30, d1 = 1 << 30;
        for (int j = 0; j < K; ++j) {
            d[j] = (r - pr[j]) * (r - pr[j]) + (g - pg[j]) * (g - pg[j]) + (b - pb[j]) * (b - pb[j]);
            if (d[j] < d0) {
                d1 = d0;
                d0 = d[j];
            } else if (d[j] < d1) {
                d1 = d[j];
            }
        }
        for (int j = 0; j < K; ++j) {
            if (d[j] <= d1) tm[j >> 5] |= 1u << (j & 31);
            if (nm && d[j] == d0) nm[j >> 5] |= 1u << (j & 31);
        }
    };
    constexpr int kNCells = 4096;
    std::vector<uint32_t> masks((size_t)kNCells * 9 * mw, 0u), nmasks((size_t)kNCells * mw, 0u);
    for (int r = 0; r < 256; ++r)
        for (int g = 0; g < 256; ++g)
            for (int b = 0; b < 256; ++b) {
                const int cell = ((r >> 4) << 8) | ((g >> 4) << 4) | (b >> 4);
                const int sidx = (((r >> 3) & 1) << 2) | (((g >> 3) & 1) << 1) | ((b >> 3) & 1);
                uint32_t tm[2] = {0, 0};
                sets_of(r, g, b, tm, &nmasks[(size_t)cell * mw]);
                for (int w = 0; w < mw; ++w) {
                    masks[((size_t)cell * 9) * mw + w] |= tm[w];
                    masks[((size_t)cell * 9 + 1 + sidx) * mw + w] |= tm[w];
                }
            }
    auto box_masks = [&](const std::vector<Box> &boxes, std::vector<uint32_t> &bm) {
        std::fill(bm.begin(), bm.end(), 0u);
        for (size_t q = 0; q < boxes.size(); ++q)
            for (int r = boxes[q].r0; r < boxes[q].r0 + boxes[q].size; ++r)
                for (int g = boxes[q].g0; g < boxes[q].g0 + boxes[q].size; ++g)
                    for (int b = boxes[q].b0; b < boxes[q].b0 + boxes[q].size; ++b) sets_of(r, g, b, &bm[q * (size_t)mw], nullptr);
        return (int)DP_OK;
    };
    std::vector<uint32_t> tab, perm, wide;
    TableStats st;
    int rc = assemble_table(masks, mw, bw, kTabMaxWords, K, coord4, coord4, box_masks, tab, st, nmasks.data(), bw == 8 ? kNearSlots : 4,
                            &perm, &wide);
    if (rc != DP_OK || st.too_big) {
        fprintf(stderr, "assemble_table failed (rc %d, too_big %d)\n", rc, (int)st.too_big);
        return 1;
    }
    crowded_nodes_first(tab, st, bw, coord4);
    WarpMaps wm;
    make_warp(coord4, wm);
    const std::vector<uint32_t> mp = mass_points(coord4);
    const int in_split = entries_in_split_cells(tab, bw, mp);
    // every sampled colour must find all of T(x) in the block the kernel reaches
    uint32_t seed = 777u + (uint32_t)K;
    long bad = 0, checked = 0;
    for (int i = 0; i < 60000 + 27 * K; ++i) {
        int r, g, b;
        if (i < 60000) {
            r = lcg(seed) & 255, g = lcg(seed) & 255, b = lcg(seed) & 255;
        } else {
            const int q = i - 60000, j = q / 27, o = q % 27;
            r = std::min(255, std::max(0, pr[j] + o % 3 - 1));
            g = std::min(255, std::max(0, pg[j] + (o / 3) % 3 - 1));
            b = std::min(255, std::max(0, pb[j] + o / 9 - 1));
        }
        uint32_t tm[2] = {0, 0};
        sets_of(r, g, b, tm, nullptr);
        size_t pos = (size_t)cell_slot(r >> 4, g >> 4, b >> 4) * bw;
        int size = 16, r0 = r & ~15, g0 = g & ~15, b0 = b & ~15;
        bool slow = false;
        while (tab[pos] >> 31) {
            if ((tab[pos] >> 30) == 3u) {
                slow = true;  // a single colour with more candidates than a block: the fix-up pass resolves it
                break;
            }
            const size_t node = tab[pos] & 0xffffffu;
            size /= 2;
            const int sidx = ((r >= r0 + size) << 2) | ((g >= g0 + size) << 1) | (b >= b0 + size);
            r0 += (r >= r0 + size) * size, g0 += (g >= g0 + size) * size, b0 += (b >= b0 + size) * size;
            pos = (size_t)kCells * bw + (node * 8 + sidx) * bw;
            if (pos + bw > tab.size()) {
                ++bad;
                slow = true;
                break;
            }
        }
        if (slow) continue;
        for (int j = 0; j < K; ++j)
            if (tm[j >> 5] >> (j & 31) & 1u) {
                ++checked;
                bool found = false;
                for (int e = 0; e < bw; ++e) found |= tab[pos + e] == coord4[j];
                bad += !found;
            }
    }
    // the fast kernel's staging orders: a permutation of the block's positions, nearest set first
    for (int slot = 0; slot < kCells; ++slot) {
        const uint32_t pw = perm[slot];
        if (pw == 0xffffffffu) continue;
        if ((pw >> 24) == 0xffu) {
            if (((size_t)(pw & 0xffffffu) + 1) * kWideList > wide.size()) ++bad;
            continue;
        }
        const int fb = bw == 8 ? 3 : 2;
        uint32_t seen = 0;
        for (int k = 0; k < bw; ++k) seen |= 1u << ((pw >> (fb * k)) & ((1u << fb) - 1));
        if (seen != (1u << bw) - 1u) ++bad;
    }
    // the one-byte-per-entry copy (ordered_compact_kernel): same blocks, same positions, markers as byte offsets of the node
    int compact_bad = 0;
    if (bw == 8) {
        const std::vector<uint32_t> ct = compact_table(tab, coord4);
        compact_bad += ct.size() * 4 != tab.size();
        for (size_t b = 0; b + 8 <= tab.size(); b += 8) {
            const uint32_t lo = ct[b / 4], hi = ct[b / 4 + 1];
            if (tab[b] >> 31) {
                compact_bad += hi != 0xffffffffu;
                if ((tab[b] >> 30) == 2u) {
                    const uint32_t off = lo & 0xffffffu;  // bytes from the start of the compact table
                    compact_bad += (lo >> 30) != 2u || off != (4096u + (tab[b] & 0xffffffu) * 8u) * 8u || off + 64u > ct.size() * 4u;
                } else {
                    compact_bad += lo != 0xC0000000u;
                }
                continue;
            }
            uint32_t prev = 0;
            for (int i = 0; i < 8; ++i) {
                const uint32_t idx = ((i < 4 ? lo : hi) >> (8 * (i & 3))) & 255u;
                compact_bad += idx >= (uint32_t)K || coord4[idx] != tab[b + i];
                compact_bad += i > 0 && idx <= prev && coord4[idx] != coord4[prev];  // index order (equal colours: ascending too)
                prev = idx;
            }
        }
    }
    int warp_bad = 0;
    for (int c = 0; c < 3; ++c)
        for (int x = 1; x < 256; ++x) warp_bad += wm.lut[c][x] < wm.lut[c][x - 1];
    printf("accel K=%d bw=%d words=%zu split=%d split_cells=%d slow=%d max_cnt=%d wide=%zu mass_in_split=%d/%zu checked=%ld bad=%ld warp_bad=%d compact_bad=%d\n",
           K, bw, tab.size(), st.n_split, st.n_split_cells, st.n_slow, st.max_cnt, wide.size() / kWideList, in_split, mp.size(),
           checked, bad, warp_bad, compact_bad);
    return (bad || warp_bad || compact_bad) ? 1 : 0;
}

// ---- median cut (set-order replay + counting-sort cut) ----------------------------------------------------------------
static int run_mediancut(const char *path, long n, int depth)
{
    std::vector<uint8_t> rgb((size_t)n * 3 + 1);
    FILE *f = fopen(path, "rb");
    if (!f || fread(rgb.data(), 3, (size_t)n, f) != (size_t)n) {
        fprintf(stderr, "cannot read %ld colours from %s\n", n, path);
        return 2;
    }
    fclose(f);
    std::vector<uint32_t> order;
    pyset_order(rgb.data(), (size_t)n, order);
    std::vector<uint32_t> colours(order.size() + 1), scratch(order.size() + 1);
    for (size_t i = 0; i < order.size(); ++i) {
        const uint8_t *c = rgb.data() + 3 * (size_t)order[i];
        colours[i] = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16);
    }
    std::vector<int32_t> out;
    median_cut_u32(colours.data(), scratch.data(), order.size(), depth, out, 8);
    printf("distinct %zu\npalette", order.size());
    for (int32_t v : out) printf(" %d", v);
    printf("\norder");
    for (size_t i = 0; i < order.size() && i < 2000; ++i) printf(" %u", order[i]);
    printf("\n");
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 4) {
        fprintf(stderr, "usage: %s kdtree|edtables|accel <pts.f64> <K> [bw]  |  mediancut <rgb.u8> <n> <depth>\n", argv[0]);
        return 2;
    }
    if (std::string(argv[1]) == "mediancut") return run_mediancut(argv[2], atol(argv[3]), argc > 4 ? atoi(argv[4]) : 4);
    const int K = atoi(argv[3]);
    if (K < 1 || K > 1024) return 2;
    const std::vector<double> pts = read_pts(argv[2], K);
    const std::string cmd = argv[1];
    if (cmd == "kdtree") return run_kdtree(pts, K);
    if (cmd == "edtables") return run_edtables(pts, K);
    if (cmd == "accel") return run_accel(pts, K, argc > 4 ? atoi(argv[4]) : 8);
    return 2;
}
