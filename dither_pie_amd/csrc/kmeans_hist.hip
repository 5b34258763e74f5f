// Lloyd passes over the COLOUR HISTOGRAM instead of the pixels (ColorReducer.generate_kmeans_palette,
// dithering_lib.py:1845-1857 -> sklearn KMeans; the fit of SURVEY 8(e)'s k-means row).
//
// A pixel's label is a function of its colour, and the totals a pass produces are sums of count x colour -- so the pixels
// are read ONCE (3 B/pixel) into count[colour] over all 2^24 colours, and every Lloyd pass then runs over that table:
//   hist_count / plan / scatter / parts kernels   pixels -> table by PARTITION (below): the pixels are bucketed by their 16^3
//                       cell (2 bytes per pixel), every bucket becomes its cell's 16 KB table slice through an LDS histogram.
//                       0.4 ms for 33 M pixels where one global atomic per pixel takes 1.2 ms (and 376 ms on a flat frame).
//                       The per-cell pixel counts fall out of the first step: passes skip empty cells without reading them.
//   hist_pass_kernel    a persistent grid of independent waves over the OCCUPIED cells (a cell is cut over 2 or 4 waves when few
//                       are occupied).  The table is CELL-MAJOR (index = cell << 12 | r_lo << 8 | g_lo << 4 | b_lo), so a wave
//                       reads its cell's 16 KB contiguously, 4 KB at a time; it builds the cell's candidate list itself (the
//                       centres that can be nearest somewhere in the cell: bound test + pairwise bisector test, as
//                       kmeans_cells_build_kernel -- no separate list launch, no 4096-list table), and
//                         * a cell with ONE candidate (most cells at 32 centres) needs no distance at all: its totals are
//                           sum(count), sum(count * r), ... by separable sums;
//                         * otherwise every colour of the cell is scored against the list: float32 scores with the biased
//                           key trick of kmeans_step_kernel, ONE v_fma_f32 per (colour, candidate) because r is uniform
//                           over a wave's register, g over a lane, and only b varies inside a lane's four counts; near
//                           ties go to the same float64 decision as everywhere (label_f64 semantics, incl. sklearn's
//                           rule for equidistant colours), over the list only: every centre that can attain the minimum
//                           is on it.
//                       Labels and int64 totals are those of the per-pixel kernels, bit for bit (tests compare both with
//                       the oracle).  Bytes per pass: 16 KB per occupied cell (64 MB when every cell is occupied).
//                       FUSE instances (dp_kmeans_hist_iterate): the workgroup that finishes last also runs the centre update,
//                       so that a Lloyd iteration on one device is ONE launch.
// Counts are 32-bit: a histogram holds fewer than 2^32 pixels (per rank).
#include <algorithm>

#include "dp_internal.h"
#include "wave_util.hip.h"
#include "kmeans_label.hip.h"

namespace dp {
namespace {

constexpr int kHistCells = 4096;
constexpr size_t kHistTableBytes = (size_t)4 << 24;          // 2^24 x uint32
constexpr size_t kHistInfoBytes = (size_t)4 * (2 * kHistCells + 4);  // pixels per cell | number of occupied cells | their list

// ---------------------------------------------------------------------------------------------------------------
// Building count[colour] by PARTITION.  One global atomic per pixel into the 64 MB table runs at 27 G atomics/s whatever the
// content (every one misses L2: a 128-byte line in and out per pixel, profiles/microbench/hist24_results.txt: 1.23 ms for the
// 33 M pixels of the C4 image), and a flat image serialises on one address (376 ms).  Instead (microbenchmark:
// profiles/microbench/hist24_partition.hip, 0.45 ms on noise):
//   A  hist_count_kernel    per-workgroup LDS histogram of the 4096 CELL ids -> pixels per cell (the passes need that anyway)
//   B  hist_plan_kernel     bucket bases (exclusive scan of the bucket CAPACITIES), the parts a cell's bucket is cut into
//                           (kPartEntries entries each), their scan
//   C  hist_scatter_kernel  every pixel's low 12 bits (r_lo, g_lo, b_lo) as a uint16 into its cell's bucket -- in whole 32-byte
//                           SECTORS of 16 entries (round 5): one persistent workgroup per CU keeps a 32-byte staging slot per cell in
//                           LDS (128 KB), a pixel's place in its cell's stream comes from an LDS atomic, and whenever a slot has
//                           16 entries the workgroup reserves a sector of the cell's bucket (one global cursor atomic per SECTOR)
//                           and stores it whole; what a workgroup has left in a slot at the end goes out as one more sector, padded
//                           with 0xffff.  No 128-byte line of a bucket is ever written two bytes at a time from eight XCDs
//                           (round 4: one cursor atomic and a run of ~4 two-byte stores per (16 K-pixel tile, cell) -- 568 MB
//                           written for 66 MB of entries, profiles/r04_pmc_legs.json).
//                           A build that does not accumulate also zeroes, in the scatter's prologue, the 16 KB slices D will not
//                           overwrite (cells without pixels; cells whose bucket is cut into several parts): no 64 MB memset.
//   D  hist_parts_kernel    one workgroup per part: LDS histogram of its bucket entries (LDS atomics), stored as the cell's
//                           16 KB slice of the table (a cell with one part, not accumulating: no read at all), added to it by a
//                           coalesced read-modify-write (one part, accumulating) or by atomics of the non-zero counts (several
//                           workgroups share the cell: image-like content, big cells)
// The buckets (2 bytes per pixel + the padding of the last sectors) and the plan live in a caller-provided workspace.
constexpr int kCountThreads = 1024;
constexpr int kScatterThreads = 1024;
constexpr int kTileGroups = kScatterThreads * 4;   // groups of 4 pixels per scatter tile (16 pixels per thread)
constexpr uint32_t kPartEntries = 16384;
constexpr uint32_t kSector = 16;                   // bucket entries per sector (32 bytes)
constexpr int kScatterMaxGrid = 256;               // scatter workgroups at most (each may leave one padded sector per cell)

// the workspace: [0] cell counts of this build | [1] bucket bases | [2] cursors | [3] part bases (4097) ... | buckets
constexpr size_t kPlanWords = 4 * 4096 + 64;

// scatter workgroups for n pixels: one per CU, fewer when there are fewer tiles than that
__host__ __device__ inline uint32_t scatter_grid(const int64_t n, const int cus)
{
    const int64_t tiles = ((n + 3) / 4 + kTileGroups - 1) / kTileGroups;
    int64_t g = tiles < (int64_t)cus ? tiles : (int64_t)cus;
    if (g > kScatterMaxGrid) g = kScatterMaxGrid;
    return (uint32_t)(g < 1 ? 1 : g);
}

// entries a cell's bucket must hold: its pixels in whole sectors, plus one partly filled sector per scatter workgroup that saw it
__host__ __device__ inline uint32_t bucket_capacity(const uint32_t cnt, const uint32_t grid)
{
    if (cnt == 0u) return 0u;
    return ((cnt / kSector) + (cnt < grid ? cnt : grid)) * kSector;
}

__device__ __forceinline__ void split_colour(const uint32_t v, uint32_t &cell, uint32_t &lo)
{
    cell = ((v & 0xf0u) << 4) | ((v & 0xf000u) >> 8) | ((v & 0xf00000u) >> 20);   // r' << 8 | g' << 4 | b'
    lo = ((v & 0xfu) << 8) | ((v & 0xf00u) >> 4) | ((v & 0xf0000u) >> 16);        // r_lo << 8 | g_lo << 4 | b_lo
}

// the four pixels of group gi as r | g << 8 | b << 16; cnt = how many of them exist
__device__ __forceinline__ void load_group(const uint8_t *__restrict__ px, const int64_t n, const int64_t gi, const bool aligned,
                                           uint32_t (&v)[4], int &cnt)
{
    const int64_t p0 = gi * 4;
    cnt = p0 < n ? (int)min<int64_t>(4, n - p0) : 0;
    v[0] = v[1] = v[2] = v[3] = 0u;
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
}

__global__ __launch_bounds__(kCountThreads) void hist_count_kernel(const uint8_t *__restrict__ px, const int64_t n, uint32_t *__restrict__ cell_count)
{
    __shared__ uint32_t s_cnt[kHistCells];
    for (int i = threadIdx.x; i < kHistCells; i += kCountThreads) s_cnt[i] = 0u;
    __syncthreads();
    const int64_t n_groups = (n + 3) / 4;
    const bool aligned = ((uintptr_t)px & 3) == 0;
    for (int64_t gi = (int64_t)blockIdx.x * kCountThreads + threadIdx.x; gi < n_groups; gi += (int64_t)gridDim.x * kCountThreads) {
        uint32_t v[4];
        int cnt;
        load_group(px, n, gi, aligned, v, cnt);
        // a lane's run of one cell costs one LDS atomic (flat regions: every lane the same address)
        uint32_t c[4], lo;
#pragma unroll
        for (int q = 0; q < 4; ++q) split_colour(v[q], c[q], lo);
        // (and a wave's lanes mostly share it on coherent content: while at least eight whole-group lanes agree with the first
        // pending one, one lane adds for all of them -- as hist_scatter_kernel does)
        const bool all4 = cnt == 4 && c[0] == c[1] && c[1] == c[2] && c[2] == c[3];
        bool done = false;
        unsigned long long pend = __ballot(all4);
        for (int round = 0; round < 3 && __popcll(pend) >= 8; ++round) {   // (wave-uniform)
            const int leader = __ffsll((long long)pend) - 1;
            const uint32_t lc = (uint32_t)__builtin_amdgcn_readlane((int)c[0], leader);
            const unsigned long long m = __ballot(all4 && !done && c[0] == lc);
            if ((int)(threadIdx.x & 63u) == leader) atomicAdd(&s_cnt[lc], 4u * (uint32_t)__popcll(m));
            if ((m >> (threadIdx.x & 63u)) & 1ull) done = true;
            pend &= ~m;
        }
        if (done) continue;
        uint32_t run = 1;
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            if (q < cnt) {
                if (c[q] == c[q - 1]) ++run;
                else {
                    atomicAdd(&s_cnt[c[q - 1]], run);
                    run = 1;
                }
            }
        }
        if (cnt > 0) atomicAdd(&s_cnt[c[cnt - 1]], run);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kHistCells; i += kCountThreads)
        if (s_cnt[i]) atomicAdd(&cell_count[i], s_cnt[i]);
}

// One block of 1024 threads (the tail of hist_plan_kernel): the occupied cells in ascending order (info[4096] = how many, info[4097 ...] = which; bit 31 of an
// entry: the cell holds 2^24 pixels or more -- its weighted sums need 64 bits), so that a pass hands its waves occupied cells only
// and a wave learns everything about its cell from ONE word.
__device__ __forceinline__ void occupied_list(uint32_t *__restrict__ info)
{
    __shared__ uint32_t s_part[16];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    uint32_t flag[4], mine = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        flag[i] = info[4 * t + i] != 0u ? 1u : 0u;
        mine += flag[i];
    }
    // inclusive prefix over the wave (row_shr ladder of wave_sum_to_lane63 is an inclusive scan), then over the 16 waves
    uint32_t incl = mine;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, off);
        if (lane >= off) incl += o;
    }
    if (lane == 63) s_part[wv] = incl;
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        base += w < wv ? s_part[w] : 0u;
        total += s_part[w];
    }
    uint32_t pos = base + incl - mine;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (flag[i]) info[kHistCells + 1 + pos++] = (uint32_t)(4 * t + i) | (info[4 * t + i] >= (1u << 24) ? 0x80000000u : 0u);
    if (t == 0) info[kHistCells] = total;
}

// One workgroup of 1024 threads, four cells each: bucket bases and cursors (exclusive scan of the bucket capacities), part bases
// (exclusive scan of ceil(capacity / kPartEntries)), the number of parts; and the passes' per-cell totals: info[cell] (+)= count.
// info[2 * 4096 + 1] becomes non-zero (and stays so) when an accumulating build carries a cell's 32-bit pixel count past 2^32.
__global__ __launch_bounds__(1024) void hist_plan_kernel(uint32_t *__restrict__ plan, uint32_t *__restrict__ info, const int accumulate,
                                                          const uint32_t grid)
{
    __shared__ uint32_t s_a[16], s_b[16];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    uint32_t cap[4], parts[4], sum_c = 0, sum_p = 0;
    bool wrapped = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t cnt = plan[4 * t + i];
        cap[i] = bucket_capacity(cnt, grid);
        parts[i] = (cap[i] + kPartEntries - 1) / kPartEntries;
        sum_c += cap[i];
        sum_p += parts[i];
        const uint32_t before = accumulate ? info[4 * t + i] : 0u;
        info[4 * t + i] = before + cnt;
        wrapped |= before + cnt < before;
    }
    if (!accumulate && t == 0) info[2 * kHistCells + 1] = 0u;
    if (wrapped) info[2 * kHistCells + 1] = 1u;
    uint32_t inc_c = sum_c, inc_p = sum_p;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t oc = (uint32_t)__shfl_up((int)inc_c, off), op = (uint32_t)__shfl_up((int)inc_p, off);
        if (lane >= off) {
            inc_c += oc;
            inc_p += op;
        }
    }
    if (lane == 63) {
        s_a[wv] = inc_c;
        s_b[wv] = inc_p;
    }
    __syncthreads();
    uint32_t base_c = 0, base_p = 0, total_p = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        base_c += w < wv ? s_a[w] : 0u;
        base_p += w < wv ? s_b[w] : 0u;
        total_p += s_b[w];
    }
    uint32_t at_c = base_c + inc_c - sum_c, at_p = base_p + inc_p - sum_p;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        plan[4096 + 4 * t + i] = at_c;       // bucket base (a multiple of kSector: every capacity is)
        plan[2 * 4096 + 4 * t + i] = at_c;   // cursor
        plan[3 * 4096 + 4 * t + i] = at_p;   // first part of the cell
        at_c += cap[i];
        at_p += parts[i];
    }
    if (t == 0) plan[4 * 4096] = total_p;    // part base of "cell 4096" = number of parts
    __syncthreads();   // (info[] above: read again below, by other threads' neighbours only through their own entries -- and s_a/s_b reuse)
    occupied_list(info);
}

// LDS of the scatter: 32-byte staging slot per cell | per cell: this tile's count, then (sector base | old fill) | per cell: the
// stream positions below which this tile's entries go straight to the reserved sectors | per cell: entries waiting in the slot
constexpr int kStageBytes = kHistCells * 32;
constexpr int kScatterLds = kStageBytes + kHistCells * 4 + kHistCells * 2 + kHistCells;

__global__ __launch_bounds__(kScatterThreads) void hist_scatter_kernel(const uint8_t *__restrict__ px, const int64_t n, uint32_t *__restrict__ plan,
                                                                        uint16_t *__restrict__ buckets, uint32_t *__restrict__ table,
                                                                        const int accumulate)
{
    __shared__ __align__(16) uint8_t s_raw[kScatterLds];
    uint16_t *s_stage = reinterpret_cast<uint16_t *>(s_raw);                               // [cell][16]
    uint32_t *s_a = reinterpret_cast<uint32_t *>(s_raw + kStageBytes);                     // [cell]
    uint16_t *s_lim = reinterpret_cast<uint16_t *>(s_raw + kStageBytes + kHistCells * 4);  // [cell]
    uint8_t *s_fill = s_raw + kStageBytes + kHistCells * 4 + kHistCells * 2;               // [cell]
    uint32_t *cursor = plan + 2 * 4096;
    const int64_t n_groups = (n + 3) / 4;
    const bool aligned = ((uintptr_t)px & 3) == 0;
    const int64_t tiles = (n_groups + kTileGroups - 1) / kTileGroups;
    const uint3 *px3 = reinterpret_cast<const uint3 *>(px);
    // a group's twelve bytes as one load when it is whole and the buffer is dword-aligned (else load_group's byte loads, at use)
    auto whole = [&](const int64_t gi) { return aligned && gi * 4 + 4 <= n; };
    uint3 nw[4];   // the next tile's groups, in flight while this tile is worked on
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t gi = (int64_t)blockIdx.x * kTileGroups + (int64_t)j * kScatterThreads + threadIdx.x;
        nw[j] = make_uint3(0u, 0u, 0u);
        if ((int64_t)blockIdx.x < tiles && whole(gi)) nw[j] = px3[gi];
    }
    for (int i = threadIdx.x; i < kHistCells; i += kScatterThreads) {
        s_a[i] = 0u;
        s_fill[i] = 0;
    }
    // A build that does not accumulate: the slices hist_parts_kernel will not overwrite -- cells without pixels, and cells whose
    // bucket is cut into several parts (their workgroups add with atomics) -- are zeroed here, each workgroup its share of the
    // cells, while its first pixels arrive; every other slice is stored whole by its one part (no 64 MB memset, no read).
    if (!accumulate) {
        for (uint32_t cell = blockIdx.x; cell < (uint32_t)kHistCells; cell += gridDim.x) {
            const uint32_t cap = bucket_capacity(plan[cell], gridDim.x);
            if (cap != 0u && cap <= kPartEntries) continue;   // one part
            uint4 *o = reinterpret_cast<uint4 *>(table + (size_t)cell * 4096);
            for (int i = threadIdx.x; i < 1024; i += kScatterThreads) o[i] = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    __syncthreads();
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        uint32_t key[16];   // cell << 12 | lo
        uint32_t rank[16];
        uint3 cw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) cw[j] = nw[j];
        {
            const int64_t next = tile + gridDim.x;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t gi = next * kTileGroups + (int64_t)j * kScatterThreads + threadIdx.x;
                if (next < tiles && whole(gi)) nw[j] = px3[gi];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t gi = tile * kTileGroups + (int64_t)j * kScatterThreads + threadIdx.x;
            uint32_t v[4];
            int cnt;
            if (whole(gi)) {
                cnt = 4;
                v[0] = cw[j].x & 0xffffffu;
                v[1] = __builtin_amdgcn_perm(cw[j].y, cw[j].x, 0x0c050403u);
                v[2] = __builtin_amdgcn_perm(cw[j].z, cw[j].y, 0x0c040302u);
                v[3] = cw[j].z >> 8;
            } else {
                load_group(px, n, gi, false, v, cnt);
            }
            uint32_t cell[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t lo;
                split_colour(v[q], cell[q], lo);
                key[4 * j + q] = (cell[q] << 12) | lo;
            }
            // Coherent content (an image, a flat frame): a lane's four pixels share a cell, and so do most lanes of the wave -- 64
            // LDS atomics on one address serialise.  Lanes whose four pixels share a cell take their ranks four at a time, and
            // while at least eight such lanes agree with the first pending one, ONE lane adds for all of them (three rounds at
            // most; on noise no lane qualifies and none of this runs).
            const bool all4 = cnt == 4 && cell[0] == cell[1] && cell[1] == cell[2] && cell[2] == cell[3];
            uint32_t base4 = 0xffffffffu;
            unsigned long long pend = __ballot(all4);
            for (int round = 0; round < 3 && __popcll(pend) >= 8; ++round) {   // (wave-uniform)
                const int leader = __ffsll((long long)pend) - 1;
                const uint32_t lc = (uint32_t)__builtin_amdgcn_readlane((int)cell[0], leader);
                const unsigned long long m = __ballot(all4 && base4 == 0xffffffffu && cell[0] == lc);
                uint32_t b = 0u;
                if ((int)(threadIdx.x & 63u) == leader) b = atomicAdd(&s_a[lc], 4u * (uint32_t)__popcll(m));
                b = (uint32_t)__builtin_amdgcn_readlane((int)b, leader);
                if ((m >> (threadIdx.x & 63u)) & 1ull)
                    base4 = b + 4u * __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                pend &= ~m;
            }
            if (all4 && base4 == 0xffffffffu) base4 = atomicAdd(&s_a[cell[0]], 4u);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                rank[4 * j + q] = all4 ? base4 + (uint32_t)q : (q < cnt ? atomicAdd(&s_a[cell[q]], 1u) : 0xffffffffu);
        }
        __syncthreads();
        // Per cell: the tile's entries continue the cell's stream behind the f entries waiting in its slot (stream position p = f +
        // rank); every 16 positions make a sector.  Positions below 16 complete the slot IN LDS, the slot leaves as one whole
        // 32-byte store, positions from the last sector boundary on (lim) start the slot afresh; only a cell that receives more
        // than a sector's worth in one tile (coherent content) has positions 16 .. lim - 1, which go straight to their
        // reserved sectors as runs of two-byte stores.
        {
            uint32_t f4[4], tot4[4], lim4[4], base4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int cell = threadIdx.x + k * kScatterThreads;
                f4[k] = s_fill[cell];
                tot4[k] = f4[k] + s_a[cell];
                lim4[k] = tot4[k] & ~(kSector - 1u);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {   // (the four reservations are in flight together: lim / 16 sectors, one atomic each)
                base4[k] = 0u;
                if (lim4[k] != 0u) base4[k] = atomicAdd(&cursor[threadIdx.x + k * kScatterThreads], lim4[k]);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int cell = threadIdx.x + k * kScatterThreads;
                s_a[cell] = base4[k] | f4[k];   // (bases are multiples of 16)
                s_lim[cell] = (uint16_t)lim4[k];
                s_fill[cell] = (uint8_t)(tot4[k] & (kSector - 1u));
            }
        }
        __syncthreads();
        uint32_t late = 0u;   // bit e: entry e starts the slot afresh (behind the flush)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            if (rank[e] == 0xffffffffu) continue;
            const uint32_t cell = key[e] >> 12;
            const uint32_t a = s_a[cell], lim = s_lim[cell];
            const uint32_t p = (a & (kSector - 1u)) + rank[e];
            const uint16_t lo = (uint16_t)(key[e] & 0xfffu);
            if (p < kSector) s_stage[cell * kSector + p] = lo;
            else if (p < lim) buckets[(size_t)(a & ~(kSector - 1u)) + p] = lo;
            else {
                late |= 1u << e;
                rank[e] = p - lim;   // its place in the fresh slot
            }
        }
        __syncthreads();
        for (int cell = threadIdx.x; cell < kHistCells; cell += kScatterThreads) {
            if (s_lim[cell] == 0) continue;
            const uint4 *slot = reinterpret_cast<const uint4 *>(s_stage + cell * kSector);
            uint4 *dst = reinterpret_cast<uint4 *>(buckets + (s_a[cell] & ~(kSector - 1u)));
            dst[0] = slot[0];
            dst[1] = slot[1];
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if ((late >> e) & 1u) s_stage[(key[e] >> 12) * kSector + rank[e]] = (uint16_t)(key[e] & 0xfffu);
        for (int i = threadIdx.x; i < kHistCells; i += kScatterThreads) s_a[i] = 0u;
        __syncthreads();
    }
    // what is left in the slots: one last sector per cell, padded with 0xffff (hist_parts_kernel skips those)
    for (int cell = threadIdx.x; cell < kHistCells; cell += kScatterThreads) {
        const uint32_t f = s_fill[cell];
        if (f == 0u) continue;
        const uint32_t base = atomicAdd(&cursor[cell], kSector);
        const uint32_t *slot = reinterpret_cast<const uint32_t *>(s_stage + cell * kSector);
        uint32_t w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t v = slot[k];
            const uint32_t e0 = 2u * k, e1 = 2u * k + 1u;
            w[k] = (e0 < f ? (v & 0xffffu) : 0xffffu) | (e1 < f ? (v & 0xffff0000u) : 0xffff0000u);
        }
        uint4 *dst = reinterpret_cast<uint4 *>(buckets + base);
        dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
        dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
}

__global__ __launch_bounds__(256) void hist_parts_kernel(const uint16_t *__restrict__ buckets, const uint32_t *__restrict__ plan,
                                                         uint32_t *__restrict__ table, const int accumulate)
{
    __shared__ uint32_t s_h[4096];
    __shared__ int s_cell;
    const uint32_t n_parts = plan[4 * 4096];
    for (uint32_t part = blockIdx.x; part < n_parts; part += gridDim.x) {   // (workgroup-uniform)
        for (int i = threadIdx.x; i < 4096; i += 256) s_h[i] = 0u;
        if (threadIdx.x == 0) {
            // the cell this part belongs to: the last cell whose first part is <= part (cells without pixels have no parts)
            const uint32_t *pb = plan + 3 * 4096;
            int lo = 0, hi = 4096;   // pb[4096] = n_parts > part
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (pb[mid] <= part) lo = mid;
                else hi = mid;
            }
            s_cell = lo;
        }
        __syncthreads();
        const int cell = s_cell;
        const uint32_t first = plan[3 * 4096 + cell], parts = plan[3 * 4096 + cell + 1] - first;
        // the bucket as the scatter left it: whole sectors from the base to the cursor, 0xffff where a last sector was padded
        const uint32_t used = plan[2 * 4096 + cell] - plan[4096 + cell];
        const uint32_t lo_e = (part - first) * kPartEntries, hi_e = min(used, lo_e + kPartEntries);
        const uint16_t *b = buckets + (size_t)plan[4096 + cell];
        bool peel = true;   // (wave-uniform) few colours in this part: see below; given up at the first block of 64 that has many
        for (uint32_t i0 = lo_e; i0 < hi_e; i0 += 256) {   // (workgroup-uniform trip count: the ballots below are whole-wave)
            const uint32_t i = i0 + threadIdx.x;
            const uint32_t e = i < hi_e ? b[i] : 0xffffu;
            bool mine = e < 4096u;
            if (peel) {
                // few colours (a flat frame: ONE): 64 LDS atomics on one address serialise -- while at least sixteen lanes hold the
                // colour of the first pending one, one lane counts them all (two rounds at most)
                unsigned long long pend = __ballot(mine);
                for (int round = 0; round < 2 && pend != 0ull; ++round) {
                    const int leader = __ffsll((long long)pend) - 1;
                    const uint32_t le = (uint32_t)__builtin_amdgcn_readlane((int)e, leader);
                    const unsigned long long m = __ballot(mine && e == le);
                    if (__popcll(m) < 16) {
                        peel = round != 0;   // not even the first colour is shared: noise -- stop looking
                        break;
                    }
                    if ((int)(threadIdx.x & 63u) == leader) atomicAdd(&s_h[le], (uint32_t)__popcll(m));
                    if ((m >> (threadIdx.x & 63u)) & 1ull) mine = false;
                    pend &= ~m;
                }
            }
            if (mine) atomicAdd(&s_h[e], 1u);
        }
        __syncthreads();
        uint32_t *slice = table + (size_t)cell * 4096;
        if (parts == 1) {
            uint4 *o = reinterpret_cast<uint4 *>(slice);
            for (int i = threadIdx.x; i < 1024; i += 256) {
                uint4 v = accumulate ? o[i] : make_uint4(0u, 0u, 0u, 0u);
                v.x += s_h[4 * i];
                v.y += s_h[4 * i + 1];
                v.z += s_h[4 * i + 2];
                v.w += s_h[4 * i + 3];
                o[i] = v;
            }
        } else {
            for (int i = threadIdx.x; i < 4096; i += 256) {
                const uint32_t c = s_h[i];
                if (c) __hip_atomic_fetch_add(&slice[i], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// float64 decision among the LISTED centres (ascending centre index, so "first minimum" means what it means over all K):
// same expressions as label_f64 (kmeans_label.hip.h); returns the winner's POSITION in the list.
__device__ __forceinline__ int label_f64_list(const double *s_c64, const int *list, const int n, const double *mean, const uint32_t r,
                                              const uint32_t g, const uint32_t b)
{
    double bd = __longlong_as_double(0x7ff0000000000000LL);
    int pos = 0;
    if (mean) {
        const double y0 = __dsub_rn((double)r, mean[0]), y1 = __dsub_rn((double)g, mean[1]), y2 = __dsub_rn((double)b, mean[2]);
        for (int i = 0; i < n; ++i) {
            const double *c = s_c64 + 4 * list[i];
            double acc = __dmul_rn(y0, c[0]);
            acc = __fma_rn(y1, c[1], acc);
            acc = __fma_rn(y2, c[2], acc);
            const double v = __dsub_rn(c[3], __dmul_rn(2.0, acc));
            if (v < bd) {
                bd = v;
                pos = i;
            }
        }
    } else {
        const double x0 = (double)r, x1 = (double)g, x2 = (double)b;
        for (int i = 0; i < n; ++i) {
            const double *c = s_c64 + 4 * list[i];
            const double a = __dsub_rn(x0, c[0]), d = __dsub_rn(x1, c[1]), e = __dsub_rn(x2, c[2]);
            const double dist = __dadd_rn(__dadd_rn(__dmul_rn(a, a), __dmul_rn(d, d)), __dmul_rn(e, e));
            if (dist < bd) {
                bd = dist;
                pos = i;
            }
        }
    }
    return pos;
}

constexpr int kHistMaxK = 256;
constexpr int kPassWaves = 4;   // waves per workgroup; every wave works on cells of its own
constexpr int kPassGrid = 1024; // workgroups (persistent: a wave takes every (4 * grid)-th occupied cell)

// FUSE: a whole Lloyd iteration in this one launch (single rank: nothing to all-reduce between the pass and the update).  The
// totals buffer is planar (sums | counts | squared norms, as dp_kmeans_update takes it) and ZERO on entry; every workgroup that
// has work adds its totals, fences, and takes a ticket; the last one runs the centre update with sklearn's stopping rules
// (lloyd_update_block, through atomic loads), then clears sums and counts and the ticket counter for the next iteration.
// No workgroup ever waits for another.
struct FuseArgs {
    double *centers_rw;
    long long *prev;
    double *status;
    uint32_t *ticket;
    double tol;
    int max_iter;
};

template <bool SQ, bool FUSE>
__global__ __launch_bounds__(64 * kPassWaves, 4) void hist_pass_kernel(const uint32_t *__restrict__ table, const uint32_t *__restrict__ info,
                                                                    const double *centers, const double *__restrict__ mean,
                                                                    const int K, unsigned long long *__restrict__ sums,
                                                                    unsigned long long *__restrict__ counts,
                                                                    unsigned long long *__restrict__ sumsq, const FuseArgs fuse, const int force_split)
{
    // a fused iteration launched past convergence (the host looks at the status every few iterations) has nothing to do
    if (FUSE) {
        const int done = (int)fuse.status[kStDone];
        if (done == 1 || done == 3) return;   // (uniform over the grid: only the last workgroup of a launch writes the status)
    }
    // the wave's first list entry is read next to the number of occupied cells, not after it: which entry that is depends on the
    // split (1, 2 or 4 waves per cell), so all three candidates are fetched (always inside the 4096-entry list; checked below)
    const uint32_t ui0 = (uint32_t)blockIdx.x * kPassWaves + (uint32_t)(threadIdx.x >> 6);
    const uint32_t spec1 = info[kHistCells + 1 + (ui0 & 4095u)], spec2 = info[kHistCells + 1 + ((ui0 >> 1) & 4095u)],
                   spec4 = info[kHistCells + 1 + ((ui0 >> 2) & 4095u)];
    const uint32_t n_occ = info[kHistCells];
    // Few occupied cells (image-like content): a cell is split over 2 or 4 waves (each takes 2 / 1 of the cell's four chunks
    // and builds the list for itself), so that the chip still has a few thousand waves to hide latency with.
    const int split_auto = n_occ * 4u <= (uint32_t)(kPassGrid * kPassWaves) ? 4 : (n_occ * 2u <= (uint32_t)(kPassGrid * kPassWaves) ? 2 : 1);
    const int split = force_split ? force_split : split_auto;
    const int chunks_per_unit = 4 / split;
    const uint32_t n_units = n_occ * (uint32_t)split;
    if ((uint32_t)blockIdx.x * kPassWaves >= n_units) return;  // (workgroup-uniform)
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    __shared__ float4 s_c[kHistMaxK];        // {x, y, z, |c|^2}: the list tests
    __shared__ float4 s_sc[kHistMaxK];       // {-2x, -2y, -2z, |c|^2 + BIAS}: the scores
    __shared__ double s_c64[4 * kHistMaxK];  // float64 records of the near-tie decision (stage_centre_f64)
    __shared__ double s_mean[3];
    __shared__ int s_surv[kPassWaves][kHistMaxK];
    __shared__ int s_list[kPassWaves][kHistMaxK];
    __shared__ uint32_t s_drop[kPassWaves];
    __shared__ unsigned long long s_tot[kHistMaxK * 5];  // per CENTRE: n, sum r, sum g, sum b, sum |x|^2 of this workgroup's cells
    if (t < 3 && mean) s_mean[t] = mean[t];
    for (int j = t; j < K; j += 64 * kPassWaves) {
        const double c0 = centers[3 * j], c1 = centers[3 * j + 1], c2 = centers[3 * j + 2];
        stage_centre_f64(s_c64 + 4 * j, c0, c1, c2, mean);
        const float x = (float)c0, y = (float)c1, z = (float)c2;
        s_c[j] = make_float4(x, y, z, x * x + y * y + z * z);
        s_sc[j] = make_float4((float)(-2.0 * c0), (float)(-2.0 * c1), (float)(-2.0 * c2),
                              (float)(c0 * c0 + c1 * c1 + c2 * c2 + (double)kScoreBias));
    }
    for (int i = t; i < K * 5; i += 64 * kPassWaves) s_tot[i] = 0ull;
    __syncthreads();  // from here on the waves go their own ways until the totals leave the workgroup
    const float inf = __int_as_float(0x7f800000);
    const int KM = (K + 63) >> 6;  // centres per lane in the list build
    int *surv = s_surv[wv], *list = s_list[wv];

    for (uint32_t ui = (uint32_t)blockIdx.x * kPassWaves + (uint32_t)wv; ui < n_units; ui += (uint32_t)gridDim.x * kPassWaves) {
        const uint32_t ci = split == 4 ? ui >> 2 : (split == 2 ? ui >> 1 : ui);
        const int c_first = (int)(ui - ci * (uint32_t)split) * chunks_per_unit, c_end = c_first + chunks_per_unit;
        const uint32_t entry = ui == ui0 ? (split == 4 ? spec4 : (split == 2 ? spec2 : spec1)) : info[kHistCells + 1 + ci];
        const int cell = (int)(entry & 0xfffu);
        // a lane's 64 counts, 16 at a time: chunk c = rows r_lo = 4c .. 4c+3 (a wave-uniform r per register), g_lo = lane >> 2,
        // b_lo = 4 (lane & 3) + k.  The next chunk is in flight while this one is worked on.
        const uint4 *tb = reinterpret_cast<const uint4 *>(table) + (size_t)cell * 1024 + lane;
        // (two chunks ahead was measured: the 16 more registers spill at 4 waves per SIMD -- 39 -> 47 us on noise -- and at 3 waves per
        // SIMD without spills it is 45 us: the pass is not waiting on this chain, profiles/experiments/r04_kmeans_hist_split.txt)
        uint4 d[4], dn[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) d[jj] = tb[(4 * c_first + jj) * 64];

        // ---- the cell's candidate list (this wave only)
        const float lo0 = (float)((cell >> 8) << 4), lo1 = (float)(((cell >> 4) & 15) << 4), lo2 = (float)((cell & 15) << 4);
        const float hi0 = lo0 + 15.f, hi1 = lo1 + 15.f, hi2 = lo2 + 15.f;
        float near2[4];
        float far_min = inf;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            near2[m] = inf;
            const int j = lane + 64 * m;
            if (m < KM && j < K) {
                const float4 c = s_c[j];
                const float f0 = fmaxf(fabsf(c.x - lo0), fabsf(c.x - hi0)), f1 = fmaxf(fabsf(c.y - lo1), fabsf(c.y - hi1)),
                            f2 = fmaxf(fabsf(c.z - lo2), fabsf(c.z - hi2));
                far_min = fminf(far_min, f0 * f0 + f1 * f1 + f2 * f2);
                const float n0 = fmaxf(fmaxf(lo0 - c.x, c.x - hi0), 0.f), n1 = fmaxf(fmaxf(lo1 - c.y, c.y - hi1), 0.f),
                            n2 = fmaxf(fmaxf(lo2 - c.z, c.z - hi2), 0.f);
                near2[m] = n0 * n0 + n1 * n1 + n2 * n2;
            }
        }
        // a centre survives when its smallest distance to the box does not exceed the smallest "largest distance" (+ slack:
        // float32 arithmetic on values below 4e5, errors far below 1.0)
        const float U = wave_min_to_all(far_min) + 1.0f;
        int cnt = 0;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (m < KM) {
                const bool sv = near2[m] <= U;  // (inf for lanes without a centre)
                const unsigned long long mk = __ballot(sv);
                if (sv) surv[cnt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u))] = lane + 64 * m;
                cnt += __popcll(mk);
            }
        }
        int n_list;
        if (cnt <= 16) {
            // pairwise: entry a leaves if a listed b is closer on the WHOLE box by more than the slack:
            // max over the box of |x - c_b|^2 - |x - c_a|^2 = 2 x.(c_a - c_b) + |c_b|^2 - |c_a|^2  <  -1
            if (lane == 0) s_drop[wv] = 0u;
            uint32_t dropbits = 0u;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int a = (lane >> 4) + 4 * q, b = lane & 15;
                if (a < cnt && b < cnt && a != b) {
                    const float4 ca = s_c[surv[a]], cb = s_c[surv[b]];
                    const float d0 = ca.x - cb.x, d1 = ca.y - cb.y, d2 = ca.z - cb.z;
                    const float mm = 2.f * ((d0 > 0.f ? hi0 : lo0) * d0 + (d1 > 0.f ? hi1 : lo1) * d1 + (d2 > 0.f ? hi2 : lo2) * d2) + (cb.w - ca.w);
                    if (mm < -1.0f) dropbits |= 1u << a;
                }
            }
            if (dropbits) atomicOr(&s_drop[wv], dropbits);
            const uint32_t keep = ~s_drop[wv] & ((1u << cnt) - 1u);
            n_list = __popc(keep);
            int sv = 0;
            if (lane < cnt) sv = surv[lane];
            if (lane < cnt && ((keep >> lane) & 1u)) list[__popc(keep & ((1u << lane) - 1u))] = sv;
        } else {
            n_list = cnt;
            for (int i = lane; i < cnt; i += 64) list[i] = surv[i];
        }

        const uint32_t g = (uint32_t)(((cell >> 4) & 15) << 4) + (uint32_t)(lane >> 2);
        const uint32_t b0 = (uint32_t)((cell & 15) << 4) + 4u * (uint32_t)(lane & 3);
        const float fg = (float)g;
        // a cell with fewer than 2^24 pixels: every weighted partial sum below fits 32 bits (count x 255)
        const bool small = (entry >> 31) == 0u;
        // the wave's totals of the counts m[][] of chunk rows rbase .. rbase+3 (already masked by label) go to centre j
        auto add_to = [&](const int j, const uint32_t(&m)[4][4], const uint32_t rbase) {
            uint32_t rows[4], cols[4] = {0u, 0u, 0u, 0u}, nl = 0u;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                rows[jj] = (m[jj][0] + m[jj][1]) + (m[jj][2] + m[jj][3]);
                nl += rows[jj];
#pragma unroll
                for (int k = 0; k < 4; ++k) cols[k] += m[jj][k];
            }
            if (__ballot(nl != 0u) == 0ull) return;  // (wave-uniform)
            unsigned long long *tot = s_tot + 5 * j;
            if (small) {
                uint32_t rl = 0u, bl = 0u;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) rl += rows[jj] * (rbase + (uint32_t)jj);
#pragma unroll
                for (int k = 0; k < 4; ++k) bl += cols[k] * (b0 + (uint32_t)k);
                uint32_t N = nl, R = rl, G = nl * g, B = bl;
                wave_sum4_to_lane63(N, R, G, B);
                if (lane == 63) {
                    atomicAdd(&tot[0], (unsigned long long)N);
                    atomicAdd(&tot[1], (unsigned long long)R);
                    atomicAdd(&tot[2], (unsigned long long)G);
                    atomicAdd(&tot[3], (unsigned long long)B);
                }
            } else if (nl != 0u) {  // a giant cell (>= 16.7 M pixels of near-identical colour): 64-bit partials, lane by lane
                unsigned long long rl = 0ull, bl = 0ull;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) rl += (unsigned long long)rows[jj] * (rbase + (uint32_t)jj);
#pragma unroll
                for (int k = 0; k < 4; ++k) bl += (unsigned long long)cols[k] * (b0 + (uint32_t)k);
                atomicAdd(&tot[0], (unsigned long long)nl);
                atomicAdd(&tot[1], rl);
                atomicAdd(&tot[2], (unsigned long long)nl * g);
                atomicAdd(&tot[3], bl);
            }
            if (SQ && nl != 0u) {  // (the first pass of a fit only)
                unsigned long long q = (unsigned long long)nl * (g * g);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) q += (unsigned long long)rows[jj] * ((rbase + (uint32_t)jj) * (rbase + (uint32_t)jj));
#pragma unroll
                for (int k = 0; k < 4; ++k) q += (unsigned long long)cols[k] * ((b0 + (uint32_t)k) * (b0 + (uint32_t)k));
                atomicAdd(&tot[4], q);
            }
        };

        const uint32_t r0 = (uint32_t)((cell >> 8) << 4);
        if (n_list == 1) {
            // ONE candidate: no distance needed, and the four chunks are summed per lane before anything crosses lanes
            uint32_t cols[4] = {0u, 0u, 0u, 0u};
            unsigned long long one_r = 0ull, one_q = 0ull;  // count x r (and x r^2), weighted chunk by chunk
#pragma unroll 1
            for (int c = c_first; c < c_end; ++c) {
                if (c + 1 < c_end) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) dn[jj] = tb[(4 * (c + 1) + jj) * 64];
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const uint32_t rr = r0 + (uint32_t)(4 * c + jj);
                    const uint32_t row = (d[jj].x + d[jj].y) + (d[jj].z + d[jj].w);
                    one_r += (unsigned long long)row * rr;
                    if (SQ) one_q += (unsigned long long)row * (rr * rr);
                    cols[0] += d[jj].x;
                    cols[1] += d[jj].y;
                    cols[2] += d[jj].z;
                    cols[3] += d[jj].w;
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) d[jj] = dn[jj];
            }
            const uint32_t nl = (cols[0] + cols[1]) + (cols[2] + cols[3]);
            unsigned long long *tot = s_tot + 5 * list[0];
            if (small) {
                uint32_t bl = 0u;
#pragma unroll
                for (int k = 0; k < 4; ++k) bl += cols[k] * (b0 + (uint32_t)k);
                uint32_t N = nl, R = (uint32_t)one_r, G = nl * g, B = bl;
                wave_sum4_to_lane63(N, R, G, B);
                if (lane == 63) {
                    atomicAdd(&tot[0], (unsigned long long)N);
                    atomicAdd(&tot[1], (unsigned long long)R);
                    atomicAdd(&tot[2], (unsigned long long)G);
                    atomicAdd(&tot[3], (unsigned long long)B);
                }
            } else if (nl != 0u) {
                unsigned long long bl = 0ull;
#pragma unroll
                for (int k = 0; k < 4; ++k) bl += (unsigned long long)cols[k] * (b0 + (uint32_t)k);
                atomicAdd(&tot[0], (unsigned long long)nl);
                atomicAdd(&tot[1], one_r);
                atomicAdd(&tot[2], (unsigned long long)nl * g);
                atomicAdd(&tot[3], bl);
            }
            if (SQ && nl != 0u) {
                unsigned long long q = one_q + (unsigned long long)nl * (g * g);
#pragma unroll
                for (int k = 0; k < 4; ++k) q += (unsigned long long)cols[k] * ((b0 + (uint32_t)k) * (b0 + (uint32_t)k));
                atomicAdd(&tot[4], q);
            }
            continue;
        }
#pragma unroll 1
        for (int c = c_first; c < c_end; ++c) {
            if (c + 1 < c_end) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) dn[jj] = tb[(4 * (c + 1) + jj) * 64];
            }
            const uint32_t rbase = r0 + 4u * (uint32_t)c;
            uint32_t cn[4][4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                cn[jj][0] = d[jj].x;
                cn[jj][1] = d[jj].y;
                cn[jj][2] = d[jj].z;
                cn[jj][3] = d[jj].w;
            }
            // scores: |c|^2 + BIAS - 2 c.x through three v_fma_f32, g first (per lane), then b (per lane and k), then r (a
            // wave-uniform value per register): one fma per (colour, candidate).  Within 0.15 of the exact score like the
            // per-pixel kernels' (the same three roundings + the record's), keys 256 apart per ulp of 0.0625.
            int k0[4][4], k1[4][4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int k = 0; k < 4; ++k) k0[jj][k] = k1[jj][k] = 0x7fffffff;
            for (int i = 0; i < n_list; ++i) {
                const float4 cc = s_sc[list[i]];
                const float gs = fmaf(cc.y, fg, cc.w);
                float bk[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) bk[k] = fmaf(cc.z, (float)(b0 + (uint32_t)k), gs);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float fr = (float)(rbase + (uint32_t)jj);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float sc = fmaf(cc.x, fr, bk[k]);
                        const int key = (int)((__float_as_uint(sc) << 8) + (uint32_t)i);
                        k1[jj][k] = med3_s32(k0[jj][k], k1[jj][k], key);
                        k0[jj][k] = min(k0[jj][k], key);
                    }
                }
            }
            bool near = false;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int k = 0; k < 4; ++k) near |= cn[jj][k] != 0u && (k1[jj][k] - k0[jj][k] <= (6 << 8) + 255);
            // from here on k0 holds the label (the winner's list position)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool nt = k1[jj][k] - k0[jj][k] <= (6 << 8) + 255;
                    k0[jj][k] = (k0[jj][k] & 255) | (nt && cn[jj][k] != 0u ? 0x100 : 0);
                }
            if (__ballot(near) != 0ull) {  // (wave-uniform: the float64 code stays off the common path)
                if (near) {
                    const double *mp = mean ? s_mean : nullptr;
#pragma unroll 1
                    for (int e = 0; e < 16; ++e) {
                        const int jj = e >> 2, k = e & 3;
                        int kk = 0;
                        // (dynamic indexing of the register arrays would spill: select)
#pragma unroll
                        for (int a = 0; a < 4; ++a)
#pragma unroll
                            for (int b = 0; b < 4; ++b)
                                if (a == jj && b == k) kk = k0[a][b];
                        if (kk & 0x100) {
                            const int p = label_f64_list(s_c64, list, n_list, mp, rbase + (uint32_t)jj, g, b0 + (uint32_t)k);
#pragma unroll
                            for (int a = 0; a < 4; ++a)
#pragma unroll
                                for (int b = 0; b < 4; ++b)
                                    if (a == jj && b == k) k0[a][b] = p;
                        }
                    }
                }
            }
            for (int i = 0; i < n_list; ++i) {
                uint32_t m[4][4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int k = 0; k < 4; ++k) m[jj][k] = (k0[jj][k] & 255) == i ? cn[jj][k] : 0u;
                add_to(list[i], m, rbase);
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) d[jj] = dn[jj];
        }
    }
    __syncthreads();
    for (int i = t; i < K * 5; i += 64 * kPassWaves) {
        const int j = i / 5, what = i - 5 * j;
        const unsigned long long v = s_tot[i];
        if (v == 0ull || (what == 4 && !SQ)) continue;
        if (what == 0) atomicAdd(&counts[j], v);
        else if (what == 4) atomicAdd(&sumsq[j], v);
        else atomicAdd(&sums[3 * j + (what - 1)], v);
    }
    if (FUSE) {
        // (`centers` is NOT __restrict__: in the FUSE instances it is the same memory as fuse.centers_rw, which the last
        // workgroup rewrites below -- after every workgroup, itself included, has copied the centres into LDS.)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "hist_pass_kernel<FUSE>: the fence-free ticket protocol below relies on gfx950 performing device-scope atomics at the coherence point; on another target use pass + all-reduce + update (dp_kmeans_hist_step / dp_kmeans_update)"
#endif
        __shared__ uint32_t s_ticket;
        // Ordering without a fence (outside the HIP memory model, deliberately, and only for gfx950 -- the #error above): an agent-scope fence writes back and invalidates the whole L2 of this XCD (1024 workgroups
        // doing that while the others stream the table: measured 2x slower than three separate launches).  The totals are
        // atomics, performed at the device's coherence point; each thread waits until its own are acknowledged (vmcnt), the
        // barrier collects the workgroup, and only then the ticket is taken -- so the workgroup that draws the last ticket
        // finds every total performed, and reads them back with atomic loads (never through its L2).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const uint32_t n_active = min((uint32_t)gridDim.x, (n_units + kPassWaves - 1) / kPassWaves);
        if (t == 0) s_ticket = __hip_atomic_fetch_add(fuse.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (s_ticket != n_active - 1u) return;  // (workgroup-uniform)
        // the totals, written by every workgroup's atomics, come in through atomic loads -- all of them in flight together (a
        // chain of dependent ones costs microseconds each) -- into LDS, where the update reads them like any other buffer
        long long *s_totals = reinterpret_cast<long long *>(s_tot);   // 5K <= 1280 words: the accumulators are spent
        double *s_red = s_c64;                                        // 256 doubles of scratch: so are the centre records
        __syncthreads();
        for (int i = t; i < 5 * K; i += 64 * kPassWaves)
            s_totals[i] = (long long)__hip_atomic_load(sums + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        lloyd_update_block<false>(s_totals, fuse.centers_rw, fuse.prev, fuse.status, K, fuse.tol, fuse.max_iter, s_red);
        __syncthreads();
        for (int i = t; i < 4 * K; i += 64 * kPassWaves) sums[i] = 0ull;  // sums | counts (planar): the next iteration adds to zero
        if (t == 0) __hip_atomic_store(fuse.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace

size_t kmeans_hist_bytes() { return kHistTableBytes + kHistInfoBytes; }

// plan + buckets: 2 bytes per pixel, plus one padded sector per (scatter workgroup, cell) at most
size_t kmeans_hist_ws_bytes(int64_t n)
{
    const size_t px = (size_t)(n > 0 ? n : 0);
    const size_t entries = px + (size_t)kSector * kHistCells * scatter_grid(n, kScatterMaxGrid);
    return kPlanWords * 4 + ((entries * 2 + 255) & ~(size_t)255) + 256;
}

int launch_kmeans_hist_build(const uint8_t *px, int64_t n, void *hist, int accumulate, void *ws, hipStream_t s)
{
    uint32_t *table = static_cast<uint32_t *>(hist);
    uint32_t *info = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(hist) + kHistTableBytes);
    uint32_t *plan = static_cast<uint32_t *>(ws);
    uint16_t *buckets = reinterpret_cast<uint16_t *>(static_cast<uint8_t *>(ws) + kPlanWords * 4);
    DP_HIP(hipMemsetAsync(plan, 0, 4096 * sizeof(uint32_t), s));
    int cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const int64_t groups = (n + 3) / 4;
    const uint32_t sgrid = scatter_grid(n, cus);
    ProfMark *pm = prof_begin(s);
    if (n > 0) {
        const unsigned cblocks = (unsigned)std::min<int64_t>((groups + kCountThreads - 1) / kCountThreads, (int64_t)cus * 2);
        hipLaunchKernelGGL(hist_count_kernel, dim3(cblocks), dim3(kCountThreads), 0, s, px, n, plan);
    }
    hipLaunchKernelGGL(hist_plan_kernel, dim3(1), dim3(1024), 0, s, plan, info, accumulate, sgrid);
    if (n == 0 && !accumulate) DP_HIP(hipMemsetAsync(table, 0, kHistTableBytes, s));
    if (n > 0) {
        hipLaunchKernelGGL(hist_scatter_kernel, dim3(sgrid), dim3(kScatterThreads), 0, s, px, n, plan, buckets, table, accumulate);
        // (the number of parts is on the device: capacity / kPartEntries + 4096 at most; a persistent grid takes them in turn)
        const int64_t max_parts = (n + (int64_t)kSector * kHistCells * sgrid) / (int64_t)kPartEntries + 4096;
        const unsigned pblocks = (unsigned)std::min<int64_t>(max_parts, (int64_t)cus * 16);
        hipLaunchKernelGGL(hist_parts_kernel, dim3(pblocks), dim3(256), 0, s, buckets, plan, table, accumulate);
    }
    prof_end(pm, s);
    DP_HIP(hipGetLastError());
    return DP_OK;
}

int launch_kmeans_hist_step(const void *hist, const double *centers, const double *mean, int K, int64_t *sums, int64_t *counts,
                            int64_t *sumsq, hipStream_t s)
{
    if (K > kHistMaxK) {
        set_error("dp_kmeans_hist_step: more than %d clusters (use dp_kmeans_step_u8)", kHistMaxK);
        return DP_EUNSUPPORTED;
    }
    if (counts == sums + 3 * (size_t)K && (sumsq == nullptr || sumsq == counts + K)) {
        DP_HIP(hipMemsetAsync(sums, 0, sizeof(int64_t) * (size_t)K * (sumsq ? 5 : 4), s));
    } else {
        DP_HIP(hipMemsetAsync(sums, 0, sizeof(int64_t) * 3 * (size_t)K, s));
        DP_HIP(hipMemsetAsync(counts, 0, sizeof(int64_t) * (size_t)K, s));
        if (sumsq) DP_HIP(hipMemsetAsync(sumsq, 0, sizeof(int64_t) * (size_t)K, s));
    }
    const uint32_t *table = static_cast<const uint32_t *>(hist);
    const uint32_t *cellinfo = reinterpret_cast<const uint32_t *>(static_cast<const uint8_t *>(hist) + kHistTableBytes);
    ProfMark *pm = prof_begin(s);
    const FuseArgs none{nullptr, nullptr, nullptr, nullptr, 0.0, 0};
    const char *fse = exp_env("DP_KMEANS_HIST_SPLIT");  // experiments: waves per cell (1, 2, 4)
    const int fs = fse ? atoi(fse) : 0;
    if (sumsq)
        hipLaunchKernelGGL((hist_pass_kernel<true, false>), dim3(kPassGrid), dim3(64 * kPassWaves), 0, s, table, cellinfo, centers, mean, K,
                           reinterpret_cast<unsigned long long *>(sums), reinterpret_cast<unsigned long long *>(counts),
                           reinterpret_cast<unsigned long long *>(sumsq), none, fs);
    else
        hipLaunchKernelGGL((hist_pass_kernel<false, false>), dim3(kPassGrid), dim3(64 * kPassWaves), 0, s, table, cellinfo, centers, mean, K,
                           reinterpret_cast<unsigned long long *>(sums), reinterpret_cast<unsigned long long *>(counts), nullptr, none, fs);
    prof_end(pm, s);
    DP_HIP(hipGetLastError());
    return DP_OK;
}

int launch_kmeans_hist_iterate(const void *hist, double *centers, const double *mean, int K, int64_t *totals, int64_t *prev,
                               double *status, uint32_t *ticket, double tol, int max_iter, int first, hipStream_t s)
{
    if (K > kHistMaxK) {
        set_error("dp_kmeans_hist_iterate: more than %d clusters", kHistMaxK);
        return DP_EUNSUPPORTED;
    }
    const uint32_t *table = static_cast<const uint32_t *>(hist);
    const uint32_t *cellinfo = reinterpret_cast<const uint32_t *>(static_cast<const uint8_t *>(hist) + kHistTableBytes);
    unsigned long long *sums = reinterpret_cast<unsigned long long *>(totals);
    const FuseArgs fa{centers, reinterpret_cast<long long *>(prev), status, ticket, tol, max_iter};
    ProfMark *pm = prof_begin(s);
    if (first)
        hipLaunchKernelGGL((hist_pass_kernel<true, true>), dim3(kPassGrid), dim3(64 * kPassWaves), 0, s, table, cellinfo, centers, mean, K,
                           sums, sums + 3 * (size_t)K, sums + 4 * (size_t)K, fa, 0);
    else
        hipLaunchKernelGGL((hist_pass_kernel<false, true>), dim3(kPassGrid), dim3(64 * kPassWaves), 0, s, table, cellinfo, centers, mean, K,
                           sums, sums + 3 * (size_t)K, nullptr, fa, 0);
    prof_end(pm, s);
    DP_HIP(hipGetLastError());
    return DP_OK;
}

}  // namespace dp
