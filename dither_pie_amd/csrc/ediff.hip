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
// ed_wavefront_kernel (serpentine off): one workgroup per frame, up to 16 waves.  A wave owns a band of
//   64 image rows (lane = row) and walks it on the anti-diagonal schedule: lane L works on
//   x = t - skew*L at step t, so every source pixel of a row above was finished `skew` steps
//   earlier.  Bands of one frame are pipelined across the waves of the workgroup: band b+1 behaves
//   like "lanes 64.." of band b, trailing it by 63*skew+3 steps; it spins on a progress word in LDS
//   that the producing wave publishes after its stores are acknowledged.  Nothing on the dependency
//   chain touches global memory: all global traffic happens at 16-step period boundaries -- each lane
//   fetches the 48 bytes of its next 16 pixels one full period ahead (unaligned dword loads), flushes
//   the 48 output bytes of the finished period, rows 62/63 flush their staged errors to a global row
//   buffer (L2) and lanes prefetch the previous band's two boundary rows into a 64-column LDS ring --
//   so every wait is for operations issued a period earlier.  Errors of the band's own rows live in an
//   8-deep per-row LDS ring.
// ed_rowserial_kernel (any scan, used for serpentine): rows are strictly sequential under a serpentine
//   scan (the first pixel of row y+1 needs the last pixel of row y), so only the latency of one pixel
//   step counts: one wave per frame, error rows in LDS, lane-parallel palette scan (see the kernel).
// ed_serial_kernel: lane = frame, error rows interleaved across lanes in global memory; the fallback for
//   rows too wide for the LDS rows of ed_rowserial_kernel.
//
// Nearest colour: a float32 scan keeps the two smallest distances; if they are separated by more than
// the float32 error margin the float32 winner is provably the float64 winner.  Otherwise (near ties)
// brute force in float64 with the KD-tree's arithmetic, and an exact tie between the two smallest
// distances (they do occur: diffused errors are dyadic) is resolved by replaying scipy's traversal
// (tree_query<1>) unless the palette fits one leaf (then the lowest index wins).
#include <algorithm>
#include <cstdlib>
#include <cmath>
#include <limits>
#include <thread>
#include <vector>

#include "dp_internal.h"
#include "tree_query.hip.h"
#include "wave_util.hip.h"
#include "ed_nearest.hip.h"

namespace dp {
namespace {

constexpr int kMaxTaps = 16;
constexpr int kRing = 8;    // per-row error ring depth (positions): skew*dy+dx <= 8 for every supported tap set
constexpr int kMaxWaves = 16;
constexpr int kRingStride = kRing * 3 + 3;  // words per row of the LDS error ring (see ed_wavefront_kernel)

struct Taps {
    int n;
    int skew;
    int dx[kMaxTaps], dy[kMaxTaps];
    float wq[kMaxTaps];
    double wq64[kMaxTaps];  // numba arithmetic: (double)float32(weight) / divisor
    // numba arithmetic of the HYBRID diffuser (_hybrid_numba, dithering_lib.py:1396-1494): the error is split into its luminance
    // part and the rest, scaled by lum_factor / col_factor, before it is pushed with the Floyd-Steinberg weights
    int hybrid;
    double hyb_lum, hyb_col;
};

// ---- the reference's OTHER error-diffusion arithmetic: _error_diffusion_numba (dithering_lib.py:213-308), which the
// reference takes instead of the pure-Python loop when numba is installed (dispatch at :638-653).  Same scan, same taps,
// different numbers.  TYPED PER NUMBA'S UNIFICATION RULE (fixtures pending: numba is not installable in the build image,
// so nothing the reference produced pins this branch; the CPU restatement orc_error_diffusion_numba_u8 is checked against
// an independent numpy transcription only): a variable has one type, the unification of all assignments to it, and
// `r = work_2d[y, x, 0]` (float32, :239) is re-assigned the float64 literals `0.0` / `255.0` (:242-245) -- so r, g, b are
// float64, and with them
// (1) the scan: dr = r - palette_arr[i, 0] (float64 - float32 = float64), dist = (dr*dr + dg*dg) + db*db in float64 with
//     separately rounded products and sums (no fastmath, no contraction), strict `<` against best_dist = 1e20: the FIRST
//     minimum of the float64 distances (not the KD-tree's traversal order, and not a float32 ranking, which collapses
//     near-equal candidates onto the lowest index);
// (2) the error: err0 = r - chosen0 stays float64 (it is NOT rounded to float32: |r - c| > |r| loses bits in float32);
// (3) the push: work[ny, nx] = float32( float64(work[ny, nx]) + err0 * (float64(w_f32) / divisor) ), product and sum in
//     float64, one rounding on the store.
// The error rings of the NB kernel instances therefore hold doubles (6 words per pixel instead of 3).  Rounds 1-3
// implemented a float32 scan and a float32 error; no numba release is known to type the function that way.
__device__ __forceinline__ int nearest_numba_f64(const float4 *__restrict__ cand, const int K, const float o0, const float o1,
                                                 const float o2)
{
    const double r = (double)o0, g = (double)o1, b = (double)o2;
    double best = 1e20;
    int j0 = 0;
    for (int j = 0; j < K; ++j) {
        const float4 c = cand[j];
        const double dr = __dsub_rn(r, (double)c.x), dg = __dsub_rn(g, (double)c.y), db = __dsub_rn(b, (double)c.z);
        const double dist = __dadd_rn(__dadd_rn(__dmul_rn(dr, dr), __dmul_rn(dg, dg)), __dmul_rn(db, db));
        if (dist < best) {
            best = dist;
            j0 = j;
        }
    }
    return j0;
}

__device__ __forceinline__ float push_numba(const float acc, const double err, const double w)
{
    return (float)__dadd_rn((double)acc, __dmul_rn(err, w));
}

// _hybrid_numba's error transform (dithering_lib.py:1444-1451), every variable float64 (err0 = r - chosen0 with r unified to
// float64; the literals and lum_factor / col_factor are float64), the operations in source order, no contraction:
//   lum_err_val = (0.299 * err0 + 0.587 * err1) + 0.114 * err2;  lum_c = w_c * lum_err_val;  fe_c = lum_factor * lum_c + col_factor * (err_c - lum_c)
__device__ __forceinline__ void hybrid_error_f64(double &e0, double &e1, double &e2, const double lf, const double cf)
{
    const double l = __dadd_rn(__dadd_rn(__dmul_rn(0.299, e0), __dmul_rn(0.587, e1)), __dmul_rn(0.114, e2));
    const double l0 = __dmul_rn(0.299, l), l1 = __dmul_rn(0.587, l), l2 = __dmul_rn(0.114, l);
    e0 = __dadd_rn(__dmul_rn(lf, l0), __dmul_rn(cf, __dsub_rn(e0, l0)));
    e1 = __dadd_rn(__dmul_rn(lf, l1), __dmul_rn(cf, __dsub_rn(e1, l1)));
    e2 = __dadd_rn(__dmul_rn(lf, l2), __dmul_rn(cf, __dsub_rn(e2, l2)));
}

// the type an error is kept in: float32 (the pure-Python branch) or float64 (the numba branch)
template <bool NB> struct ErrT { typedef float type; };
template <> struct ErrT<true> { typedef double type; };
template <bool NB> __device__ __forceinline__ typename ErrT<NB>::type err_of(const float o, const float c)
{
    if (NB) return (typename ErrT<NB>::type)__dsub_rn((double)o, (double)c);
    return (typename ErrT<NB>::type)__fsub_rn(o, c);
}

// min(max(v, 0), 255) for every non-NaN v, in one instruction
__device__ __forceinline__ float clamp255(const float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 255.0f); }

// ---- candidate lists for the nearest-colour search (palettes of 9..256 colours) -------------------------
// The query points of error diffusion are arbitrary float32 triples in [0,255]^3.  For every 8x8x8 cell of that
// cube the table lists the entries that are nearest to at least one (real) point of the cell; a superset is
// enough and is found geometrically: entry j qualifies if its smallest distance to the closed box does not exceed
// the smallest of all entries' largest distances to the box.  Whatever the float32 scan below decides among the
// listed entries is then validated exactly as in the full scan (the true nearest entry, and every entry tied with
// it, is on the list).  ~3 entries per cell for 256 random colours, at most 15 stored.
// (kEdCells = 32^3: host_logic.h)

__global__ __launch_bounds__(256) void ed_cells_kernel(const PalDev pal, uint4 *__restrict__ cells)
{
    const int cell = blockIdx.x * 256 + threadIdx.x;
    if (cell >= kEdCells) return;
    const double lo[3] = {(double)((cell & 31) * 8), (double)(((cell >> 5) & 31) * 8), (double)((cell >> 10) * 8)};
    const int K = pal.K;
    double bound = __longlong_as_double(0x7ff0000000000000LL);
    for (int j = 0; j < K; ++j) {
        double far2 = 0.0;
        for (int k = 0; k < 3; ++k) {
            const double c = pal.pts[3 * j + k];
            const double a = fabs(c - lo[k]), b = fabs(c - (lo[k] + 8.0));
            const double m = a > b ? a : b;
            far2 += m * m;
        }
        bound = far2 < bound ? far2 : bound;
    }
    bound = bound * (1.0 + 1e-12) + 1e-9;  // the sums above are rounded
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    int n = 0;
    for (int j = 0; j < K; ++j) {
        double near2 = 0.0;
        for (int k = 0; k < 3; ++k) {
            const double c = pal.pts[3 * j + k];
            double m = lo[k] - c;
            const double m2 = c - (lo[k] + 8.0);
            m = m > m2 ? m : m2;
            m = m > 0.0 ? m : 0.0;
            near2 += m * m;
        }
        if (near2 <= bound) {
            ++n;
            if (K <= 256) {
                if (n <= 15) w[n >> 2] |= (uint32_t)j << (8 * (n & 3));
            } else if (n <= 12) {   // 257..1024 colours: ten bits per entry from bit 8 on (host_logic.h: ed_list_put)
                const int off = 8 + 10 * (n - 1), wi = off >> 5, sh = off & 31;
                w[wi] |= (uint32_t)j << sh;
                if (sh > 22 && wi < 3) w[wi + 1] |= (uint32_t)j >> (32 - sh);
            }
        }
    }
    w[0] |= n <= (K <= 256 ? 15 : 12) ? (uint32_t)n : 255u;
    cells[cell] = make_uint4(w[0], w[1], w[2], w[3]);
}

// I/O period of the wavefront kernel: all global traffic happens at period boundaries.  A band follows the one above it at the
// dependency distance (63 * skew + 2 steps) PLUS three periods of hand-off (its boundary errors are stored at the end of their period,
// acknowledged one period later when the stores have landed, and fetched one period before use): with 16 steps that is 48 of ~175
// steps per band, 33 times per 4K frame -- 18 % of a lone frame.  The instances of the few-frames schedule (<= 4 waves, every wave
// with a SIMD to itself: their time IS that chain) therefore run 8-step periods; the sixteen-wave instances, bound by the
// instructions they issue, keep 16 (half as many period boundaries per step).
constexpr int kPeriod = 16;
constexpr int kPeriodFew = 8;
// progress words per frame when a frame's bands are spread over several workgroups (the last one: give-up flag)
constexpr int kEdProgWords = 64;

// NT: tap slots compiled in (taps.n <= NT); EXACT: taps.n == NT, the slots carry no test (the reference's eight tap sets
// have 3, 4, 6, 7, 10 or 12 taps), so all LDS reads of a step's taps are in flight together
// MAXW: waves per workgroup this instance is built for -- 16 (one workgroup per frame) or 4 (the few-frames schedule: a
// frame's bands over up to 16 workgroups of <= 4 waves, one wave per SIMD).  The small one has the registers of a
// 256-thread workgroup (no spills) and LDS to spare, which it uses for the 16^3-cell lists of palettes above 16 colours.
// NB: the numba arithmetic (above)
// PALS: which palettes the instance has code for -- 1: at most 16 colours only (the nibble lists of the 16^3 cells, `coarse`),
// 2: more than 16 only (16^3 lists, hierarchical table), 0: both (large-queue palettes, i.e. degenerate trees).  With everything in one
// instance the hierarchical table's walk cost the 16-colour configuration (C3) 4 % (12.9 -> 13.5 ms per 256 4K frames), and the
// nibble-list code cost 256 colours as much (launch_error_diffusion has the same-process A/B, tools/bench_scripts/ab_libs.py).
template <int CAP, int NT, bool EXACT, int MAXW, bool NB = false, int PALS = 0>
__global__ __launch_bounds__(64 * MAXW) void ed_wavefront_kernel(const uint8_t *__restrict__ in,
                                                                      uint8_t *__restrict__ out, const int h,
                                                                      const int w, const PalDev pal, const Taps taps,
                                                                      void *__restrict__ bnd_all_v, const int G,
                                                                      uint32_t *__restrict__ gprog_all, const int test_giveup,
                                                                      const uint32_t n_frames)
{
    typedef typename ErrT<NB>::type E;  // an error: float, or double with the numba arithmetic
    constexpr int PERIOD = MAXW <= 4 ? kPeriodFew : kPeriod;   // steps per I/O period (a multiple of 4)
    constexpr int PW = PERIOD * 3 / 4;                          // dwords of a period's pixels
    static_assert(PERIOD % 4 == 0 && PERIOD >= 4 && PERIOD <= 16, "the boundary fetch covers 32 columns per period");
    E *__restrict__ bnd_all = reinterpret_cast<E *>(bnd_all_v);
    // G == 1 with progress words given: the REPAIR launch that follows a G > 1 launch on the stream -- only the frames
    // whose give-up flag is set are done again, one workgroup per frame (the others return at once).
    if (G == 1 && gprog_all != nullptr &&
        __hip_atomic_load(&gprog_all[(size_t)blockIdx.x * kEdProgWords + kEdProgWords - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
        return;
    if (G != 1 && test_giveup) {  // tests: every workgroup gives up before it has written a pixel
        if (threadIdx.x == 0)
            __hip_atomic_store(&gprog_all[(size_t)(blockIdx.x / (unsigned)G) * kEdProgWords + kEdProgWords - 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // G > 1: a frame's bands are spread over G workgroups (few frames in flight: more CUs per frame, fewer waves per
    // CU).  Waves of different workgroups then meet through progress words in global memory instead of s_prog, and the
    // boundary rows are written with agent-scope stores (the workgroups may sit on different XCDs, i.e. L2s).
    // errors of the band's own rows (last 8 columns).  A row's ring is padded to 27 words: at step t lane L reads (and
    // writes) slot (t - skew*L - dx) & 7 of a row, so with the natural stride of 24 words the lanes L, L+8, L+16, ... --
    // eight of them -- meet in one LDS bank on every ring access (SQ_LDS_BANK_CONFLICT: 72 % of the LDS cycles, the LDS
    // pipe 80 % busy with 16 waves per CU); 27 makes the accesses conflict-free at skew 2 and two-way at skew 3.
    __shared__ E s_ring[MAXW][64][kRingStride];
    __shared__ E s_vring[MAXW][2][64][3];                // errors of the two rows above the band (64-column ring)
    __shared__ E s_bout[MAXW][2][PERIOD][3];            // this period's errors of rows 62/63, flushed to global
    __shared__ uint8_t s_lut[256];
    __shared__ volatile uint32_t s_prog[MAXW];           // (running band number << 16) | (acknowledged column of row 63 + 1024)
    // {x, y, z, out_rgb bits} of the palette; palettes of 9..16 colours keep the candidate lists of the 16^3 cells
    // (4096 words) behind their 16 entries
    __shared__ float4 s_pal[DP_MAX_COLORS + 16];
    uint32_t *s_coarse = reinterpret_cast<uint32_t *>(s_pal + 16);
    __shared__ E s_zero[4];                              // the "error" of pixels that do not exist
    typedef const __attribute__((address_space(3))) E lds_float_t;
    const int L = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int NW = blockDim.x >> 6;
    // PERSISTENT (plain one-workgroup launches, G == 1 without progress words in global memory): the workgroup does the frames
    // blockIdx.x, blockIdx.x + gridDim.x, ... and its waves take the bands of ALL of them round-robin by a running band number
    // -- a wave that has finished its last band of one frame starts on the next frame (whose first band waits for nobody)
    // instead of idling through the tail: 34 bands of a 4K frame over 16 waves are two full rounds and one with two waves busy.
    const bool persist = G == 1 && gprog_all == nullptr;
    const size_t f0 = blockIdx.x / (unsigned)G;
    const int NWT = NW * G;                                // waves working on this frame
    const int gw = (int)(blockIdx.x % (unsigned)G) * NW + wv;  // this wave's number among them
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_lut[i] = pal.lut_in ? pal.lut_in[i] : (uint8_t)i;
    for (int i = threadIdx.x; i < pal.K; i += blockDim.x) s_pal[i] = pal.fcand[i];
    constexpr bool kSmall = PALS != 2, kLarge = PALS != 1;   // code for palettes of <= 16 / > 16 colours
    if (kSmall && pal.ed_coarse)
        for (int i = threadIdx.x; i < 4096; i += blockDim.x) s_coarse[i] = pal.ed_coarse[i];
    const uint32_t *coarse = (kSmall && pal.ed_coarse) ? s_coarse : nullptr;
    // ... and their expanded records for the key scan (ed_nearest.hip.h, ed_key_expanded): |c|^2 rounded once, from float64
    __shared__ float4 s_expanded[16];
    if (kSmall && pal.ed_coarse && threadIdx.x < 16) {
        const float4 c = threadIdx.x < (unsigned)pal.K ? pal.fcand[threadIdx.x] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        const double n2 = (double)c.x * c.x + (double)c.y * c.y + (double)c.z * c.z + (double)kEdExpandedBias;
        s_expanded[threadIdx.x] = make_float4(-2.0f * c.x, -2.0f * c.y, -2.0f * c.z, (float)n2);
    }
    __shared__ uint4 s_lists16[(MAXW <= 4 && kLarge) ? kEdH4LdsWords / 4 : 1];   // the 16^3 lists (64 KB) or the hierarchical table (<= 108 KB)
    const uint4 *lists16 = nullptr;
    const uint32_t *h4 = nullptr;
    if (!kLarge) {
        // (no tables of larger palettes)
    } else if (MAXW <= 4 && pal.ed_h4 && pal.ed_h4_words <= pal.ed_h4_lds_words) {
        // the hierarchical <= 4-entry table instead of the 16^3 lists (same LDS): one group of four candidates per step for every
        // lane; what it cannot answer goes to the 8^3 lists in L2
        uint32_t *s_h4 = reinterpret_cast<uint32_t *>(s_lists16);
        for (int i = threadIdx.x; i < pal.ed_h4_words; i += blockDim.x) s_h4[i] = pal.ed_h4[i];
        h4 = s_h4;
    } else if (MAXW <= 4 && pal.ed_lists16) {
        for (int i = threadIdx.x; i < 4096; i += blockDim.x) s_lists16[i] = pal.ed_lists16[i];
        lists16 = s_lists16;
    } else if (MAXW > 4 && pal.ed_h4 && pal.ed_h4_global) {
        h4 = pal.ed_h4;   // sixteen waves of rings fill LDS: the table stays in global memory (<= 256 KB, L2-resident)
    }
    // (MAXW <= 4: h4 is an LDS pointer or null and nothing else -- with one assignment from global memory next to it the compiler
    // no longer knows the address space and walks the table with flat_load instead of ds_read: measured 9.97 -> 10.73 ms per 4K
    // frame at 256 colours.  A table larger than LDS read from L2 by these instances was tried that way and lost; see
    // profiles/experiments/r05_priced_structures.md section 2.)
    if (threadIdx.x < MAXW) s_prog[threadIdx.x] = 0;
    if (threadIdx.x < 4) s_zero[threadIdx.x] = (E)0;
    const long frame_bytes = (long)h * w * 3;
    const int skew = taps.skew;
    const int n_bands = (h + 63) / 64;
    const int n_mine = persist ? (int)((n_frames - (uint32_t)f0 + gridDim.x - 1u) / gridDim.x) : 1;  // frames of this workgroup
    const int n_gbands = n_mine * n_bands;  // (< 65536: the launcher sizes the grid for it)
    __syncthreads();  // the only workgroup barrier: from here on waves only meet through s_prog

    // gb: running band number over this workgroup's frames (= the band, with one frame); the progress words carry it
    for (int gb = gw; gb < n_gbands; gb += NWT) {
        const int fi = gb / n_bands, band = gb - fi * n_bands;
        const size_t f = f0 + (size_t)fi * gridDim.x;
        const uint8_t *fin = in + f * (size_t)h * w * 3;
        uint8_t *fout = out + f * (size_t)h * w * 3;
        E *bnd = bnd_all + f * (size_t)4 * w * 3;  // [2 buffers][2 rows][w][3]
        uint32_t *gprog = gprog_all + f * (size_t)kEdProgWords;   // [NWT] progress words + [kEdProgWords-1] give-up flag
        const int r = band * 64 + L;
        const E *bprev = bnd + (size_t)((band + 1) & 1) * 2 * w * 3;  // written by band-1
        E *bnext = bnd + (size_t)(band & 1) * 2 * w * 3;
        const int rows_here = min(64, h - band * 64);
        const int steps = w + skew * (rows_here - 1);
        const int pw = (gw + NWT - 1) % NWT;  // wave that owns band-1
        const bool row_ok = r < h;
        const long row_byte = (long)r * w * 3;

        // Where each tap of this lane's row finds its source row: the ring of a row of the band (8 columns), the ring of
        // the two rows above the band (64 columns), or -- the row does not exist (top of the image) -- the zero slot.
        // Adding a zero error is exact, so the step needs no row test and one column test per tap.
        lds_float_t *tbase[NT];
        uint32_t tmask[NT];
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int rel = L - taps.dy[k];
            const bool exists = (EXACT || k < taps.n) && r - taps.dy[k] >= 0;
            const E *row = rel >= 0 ? &s_ring[wv][rel & 63][0] : &s_vring[wv][(rel + 2) & 1][0][0];
            tbase[k] = exists ? (lds_float_t *)row : (lds_float_t *)s_zero;
            tmask[k] = exists ? (rel >= 0 ? (uint32_t)(kRing - 1) : 63u) : 0u;
        }
        uint32_t pix[PW], cur[PW + 1], outb[PW + 1];  // a period's pixels in flight / being consumed / being produced (raw bytes)
        E pb0 = 0, pb1 = 0, pb2 = 0;  // boundary errors in flight (one column per lane)
        int pb_col = 0;
        bool pb_valid = false;
#pragma unroll
        for (int k = 0; k < PW; ++k) pix[k] = 0;
#pragma unroll
        for (int k = 0; k < PW + 1; ++k) cur[k] = outb[k] = 0;

        for (int t0 = -PERIOD; t0 < steps + PERIOD; t0 += PERIOD) {
            // ================= period boundary: everything issued one period ago has landed =================
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (L == 0) {
                // boundary columns produced in steps < t0 - PERIOD are acknowledged
                int ack = t0 - PERIOD - 63 * skew + 1024;
                ack = ack < 0 ? 0 : ack;
                const uint32_t word = ((uint32_t)gb << 16) | (uint32_t)ack;
                if (G == 1) s_prog[wv] = word;
                else __hip_atomic_store(&gprog[gw], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // park the boundary errors fetched during the previous period
            if (pb_valid) {
                E *dst = &s_vring[wv][L >> 5][pb_col & 63][0];
                dst[0] = pb0;
                dst[1] = pb1;
                dst[2] = pb2;
            }
            // ---- flush the outputs of the period that just ended: columns [xs, xs+PERIOD) of row r
            {
                const int xs = (t0 - PERIOD) - skew * L;
                const int lo = xs < 0 ? 0 : xs, hi = xs + PERIOD > w ? w : xs + PERIOD;
                if (row_ok && hi > lo) {
                    const long B = row_byte + (long)xs * 3;
                    if (lo == xs && hi == xs + PERIOD) {
#pragma unroll
                        for (int k = 0; k < PW; ++k) *reinterpret_cast<uint32_t *>(fout + B + 4 * k) = outb[k];
                    } else {
#pragma unroll
                        for (int i = 0; i < PERIOD; ++i) {
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
                for (int k = 0; k < PW + 1; ++k) outb[k] = 0;
                // rows 62/63: this band's boundary errors of the same period go to the global row buffer
                if (L >= 62 && row_ok && hi > lo) {
                    for (int i = lo - xs; i < hi - xs; ++i) {
                        E *b = bnext + ((size_t)(L - 62) * w + (xs + i)) * 3;
                        if (G == 1) {
                            b[0] = s_bout[wv][L - 62][i][0];
                            b[1] = s_bout[wv][L - 62][i][1];
                            b[2] = s_bout[wv][L - 62][i][2];
                        } else {
                            __hip_atomic_store(b, s_bout[wv][L - 62][i][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(b + 1, s_bout[wv][L - 62][i][1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(b + 2, s_bout[wv][L - 62][i][2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
            }
            // ---- the pixels fetched during the previous period become current; fetch the next period's
#pragma unroll
            for (int k = 0; k < PW; ++k) cur[k] = pix[k];
            {
                const int xn = (t0 + PERIOD) - skew * L;  // first column of the NEXT period
                const long B = row_byte + (long)xn * 3;
                if (row_ok && xn + PERIOD > 0 && xn < w) {
                    if (B >= 0 && B + 4 * PW <= frame_bytes) {
#pragma unroll
                        for (int k = 0; k < PW; ++k) pix[k] = *reinterpret_cast<const uint32_t *>(fin + B + 4 * k);
                    } else {
#pragma unroll
                        for (int k = 0; k < PW; ++k) {
                            uint32_t v = 0;
#pragma unroll
                            for (int bb = 0; bb < 4; ++bb) {
                                const long a = B + 4 * k + bb;
                                if (a >= 0 && a < frame_bytes) v |= (uint32_t)fin[a] << (8 * bb);
                            }
                            pix[k] = v;
                        }
                    }
                }
            }
            // ---- boundary rows of the band above: the 32 columns that end two past lane 0's next period, [x0n+PERIOD+2-32, x0n+PERIOD+2)
            pb_valid = false;
            if (band > 0) {
                const int x0n = t0 + PERIOD;       // lane 0's first column of the next period (wave-uniform)
                constexpr int kFirst = PERIOD + 2 - 32;   // (-14 with 16 steps: lane 1's taps reach back skew + 2 columns)
                int need = x0n + PERIOD + 2;       // one past the last column fetched
                need = need > w ? w : need;
                if (x0n + kFirst < w && need > 0) {
                    const uint32_t want = (uint32_t)(need + 1024);
                    for (uint32_t spins = 0;; ++spins) {
                        const uint32_t v = G == 1 ? s_prog[pw] : __hip_atomic_load(&gprog[pw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((int)(v >> 16) > gb - 1 || ((int)(v >> 16) == gb - 1 && (v & 0xffffu) >= want)) break;
                        __builtin_amdgcn_s_sleep(4);
                        // across workgroups the producer is another workgroup of the grid: never wait for it forever
                        // (a give-up flag stays in the workspace; the launch ends instead of hanging)
                        if (G != 1 && (spins > (1u << 24) || (spins % 1024u == 1023u &&
                                                               __hip_atomic_load(&gprog[kEdProgWords - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u))) {
                            if (L == 0) __hip_atomic_store(&gprog[kEdProgWords - 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            return;
                        }
                    }
                    const int col = x0n + kFirst + (L & 31);
                    if (col >= 0 && col < w) {
                        // lanes 0..31 -> row -2 (lane 62 of band-1), lanes 32..63 -> row -1 (lane 63); the buffer was
                        // written by another wave of this workgroup and is reused every 2 bands: bypass L1
                        const E *b = bprev + ((size_t)(L >> 5) * w + col) * 3;
                        pb0 = __hip_atomic_load(b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        pb1 = __hip_atomic_load(b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        pb2 = __hip_atomic_load(b + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        pb_col = col;
                        pb_valid = true;
                    } else if (col >= -2 && col < w + 2) {  // the two columns either side of the image: zero errors
                        pb0 = pb1 = pb2 = (E)0;
                        pb_col = col;
                        pb_valid = true;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();

            // ================= PERIOD steps that touch LDS and registers only =================
            // Four steps per round of the loop: the round's pixels are bytes 0..11 of cur[] and its colours become the LAST twelve
            // bytes of outb[] at fixed positions, then both buffers move down by three whole registers -- a per-step rotation by
            // three bytes costs 24 funnel shifts per step.
            for (int i4 = 0; i4 < PERIOD; i4 += 4) {
              uint32_t cb[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int i = i4 + q;
                const int t = t0 + i;
                const int x = t - skew * L;
                const bool act = row_ok && (uint32_t)x < (uint32_t)w;  // (0 <= t < steps follows for the rows of the band)
                E e0 = 0, e1 = 0, e2 = 0;
                uint32_t cbytes = 0;
                if (act) {
                    // this step's pixel: bytes 3q..3q+2 of the round's twelve
                    const uint32_t pxv = q == 0 ? cur[0] : (q == 1 ? __funnelshift_r(cur[0], cur[1], 24)
                                                               : (q == 2 ? __funnelshift_r(cur[1], cur[2], 16) : (cur[2] >> 8)));
                    float a0, a1, a2;
                    if (pal.lut_in) {
                        a0 = (float)s_lut[pxv & 255u];
                        a1 = (float)s_lut[(pxv >> 8) & 255u];
                        a2 = (float)s_lut[(pxv >> 16) & 255u];
                    } else {  // no gamma table: v_cvt_f32_ubyte0/1/2 straight from the packed pixel
                        a0 = (float)(pxv & 255u);
                        a1 = (float)((pxv >> 8) & 255u);
                        a2 = (float)((pxv >> 16) & 255u);
                    }
                    // fully unrolled with constant indices: the tap parameters stay in scalar registers instead of
                    // being re-read from the kernel arguments at every step
                    // no column test: the rings hold zero errors for the two columns either side of the image (below)
                    if (!NB && EXACT) {
                        // the reads of every tap first, then the sums in the reference's order: written as one loop the compiler
                        // waits for each tap's two reads before it issues the next tap's (eight LDS round trips in a row per step)
                        float v0[NT], v1[NT], v2[NT];
#pragma unroll
                        for (int k = 0; k < NT; ++k) {
                            lds_float_t *src = tbase[k] + ((x - taps.dx[k]) & (int)tmask[k]) * 3;
                            v0[k] = (float)src[0];
                            v1[k] = (float)src[1];
                            v2[k] = (float)src[2];
                        }
#pragma unroll
                        for (int k = 0; k < NT; ++k) {
                            const float wq = taps.wq[k];
                            a0 = __fadd_rn(a0, __fmul_rn(v0[k], wq));
                            a1 = __fadd_rn(a1, __fmul_rn(v1[k], wq));
                            a2 = __fadd_rn(a2, __fmul_rn(v2[k], wq));
                        }
                    } else {
#pragma unroll
                    for (int k = 0; k < NT; ++k) {
                        if (EXACT || k < taps.n) {
                            lds_float_t *src = tbase[k] + ((x - taps.dx[k]) & (int)tmask[k]) * 3;
                            if (NB) {
                                const double w64 = taps.wq64[k];
                                a0 = push_numba(a0, (double)src[0], w64);
                                a1 = push_numba(a1, (double)src[1], w64);
                                a2 = push_numba(a2, (double)src[2], w64);
                            } else {
                                const float wq = taps.wq[k];
                                a0 = __fadd_rn(a0, __fmul_rn((float)src[0], wq));
                                a1 = __fadd_rn(a1, __fmul_rn((float)src[1], wq));
                                a2 = __fadd_rn(a2, __fmul_rn((float)src[2], wq));
                            }
                        }
                    }
                    }
                    const float o0 = clamp255(a0), o1 = clamp255(a1), o2 = clamp255(a2);
                    const int j = NB ? nearest_numba_f64(s_pal, pal.K, o0, o1, o2)
                                     : (pal.ed_cells ? nearest_color_cells<CAP, true, CAP == kQueueLarge>(pal, s_pal, coarse, o0, o1, o2, lists16, s_expanded, h4)
                                                     : nearest_color<CAP>(pal, s_pal, o0, o1, o2));
                    const float4 pj = s_pal[j];
                    e0 = err_of<NB>(o0, pj.x);
                    e1 = err_of<NB>(o1, pj.y);
                    e2 = err_of<NB>(o2, pj.z);
                    if (NB && taps.hybrid) {
                        double h0 = (double)e0, h1 = (double)e1, h2 = (double)e2;
                        hybrid_error_f64(h0, h1, h2, taps.hyb_lum, taps.hyb_col);
                        e0 = (E)h0;
                        e1 = (E)h1;
                        e2 = (E)h2;
                    }
                    cbytes = __float_as_uint(pj.w);
                }
                cb[q] = cbytes;
                // every pull of this step precedes the ring writes below (slot x&7 still holds column x-8, which
                // the row two below reads in this very step for a dx=+2 tap)
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // columns -2, -1, w and w+1 are written too, with zero errors (e is 0 without a pixel): a tap that reaches
                // past either end of a row reads them instead of testing its column
                if (row_ok && x >= -2 && x < w + 2) {
                    E *slot = &s_ring[wv][L][(x & (kRing - 1)) * 3];
                    slot[0] = e0;
                    slot[1] = e1;
                    slot[2] = e2;
                    if (act && L >= 62) {
                        s_bout[wv][L - 62][i][0] = e0;
                        s_bout[wv][L - 62][i][1] = e1;
                        s_bout[wv][L - 62][i][2] = e2;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
              }
              // the round's twelve bytes are consumed / produced: both buffers move down by three registers; the colour of
              // step i will have travelled to byte 3*i by the time the period is flushed
#pragma unroll
              for (int k = 0; k < PW - 3; ++k) {
                  cur[k] = cur[k + 3];
                  outb[k] = outb[k + 3];
              }
              cur[PW - 3] = cur[PW - 2] = cur[PW - 1] = 0u;
              outb[PW - 3] = cb[0] | (cb[1] << 24);
              outb[PW - 2] = (cb[1] >> 8) | (cb[2] << 16);
              outb[PW - 1] = (cb[2] >> 16) | (cb[3] << 8);
            }
        }
        // band finished: once its boundary stores are acknowledged the next band may read any column
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (L == 0) {
            if (G == 1) s_prog[wv] = ((uint32_t)(gb + 1) << 16);
            else __hip_atomic_store(&gprog[gw], (uint32_t)(gb + 1) << 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// lane = frame; err rows: ring[3][w][3][n_frames] floats (frame index fastest => coalesced)
template <int CAP, bool NB = false>
__global__ __launch_bounds__(64) void ed_serial_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                       const int64_t n_frames, const int h, const int w,
                                                       const PalDev pal, const Taps taps, const int serpentine,
                                                       void *__restrict__ ring_v)
{
    typedef typename ErrT<NB>::type E;
    E *__restrict__ ring = reinterpret_cast<E *>(ring_v);
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
                const E *e = ring + (((size_t)(sr % 3) * w + sxp) * 3) * nf + f;
                if (NB) {
                    const double w64 = taps.wq64[k];
                    a0 = push_numba(a0, (double)e[0], w64);
                    a1 = push_numba(a1, (double)e[nf], w64);
                    a2 = push_numba(a2, (double)e[2 * nf], w64);
                } else {
                    const float wq = taps.wq[k];
                    a0 = __fadd_rn(a0, __fmul_rn((float)e[0], wq));
                    a1 = __fadd_rn(a1, __fmul_rn((float)e[nf], wq));
                    a2 = __fadd_rn(a2, __fmul_rn((float)e[2 * nf], wq));
                }
            }
            const float o0 = clamp255(a0), o1 = clamp255(a1), o2 = clamp255(a2);
            const int j = NB ? nearest_numba_f64(pal.fcand, pal.K, o0, o1, o2)
                             : (pal.ed_cells ? nearest_color_cells<CAP, false, CAP == kQueueLarge>(pal, pal.fcand, nullptr, o0, o1, o2, nullptr, nullptr, pal.ed_h4)
                                             : nearest_color<CAP>(pal, pal.fcand, o0, o1, o2));
            E *e = ring + (((size_t)(y % 3) * w + x) * 3) * nf + f;
            E v0 = err_of<NB>(o0, (float)pal.pts[3 * j]), v1 = err_of<NB>(o1, (float)pal.pts[3 * j + 1]),
              v2 = err_of<NB>(o2, (float)pal.pts[3 * j + 2]);
            if (NB && taps.hybrid) {
                double h0 = (double)v0, h1 = (double)v1, h2 = (double)v2;
                hybrid_error_f64(h0, h1, h2, taps.hyb_lum, taps.hyb_col);
                v0 = (E)h0;
                v1 = (E)h1;
                v2 = (E)h2;
            }
            e[0] = v0;
            e[nf] = v1;
            e[2 * nf] = v2;
            const uint32_t c = pal.out_rgb[j];
            uint8_t *o = fout + ((size_t)y * w + x) * 3;
            o[0] = (uint8_t)c;
            o[1] = (uint8_t)(c >> 8);
            o[2] = (uint8_t)(c >> 16);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Serpentine scans: one WAVE per frame.  Rows (and the pixels of a row) are strictly sequential, so the
// only thing that matters is the latency of one pixel step; the lane = frame kernel above spends five
// dependent global-memory round trips per step (~1.4 us).  Here everything on the chain stays in
// registers: the three newest error rows live in LDS and are only read lane-parallel, 64 pixels ahead of
// the walk (each lane prepares the pixel value plus all contributions from the rows above for "its"
// pixel of the next 64-step block); the walk itself adds the two same-row taps from registers, evaluates
// the palette lane-parallel (entry l + 64 m in lane l), reduces the two smallest distances with DPP row
// shifts and picks the winner's colour up with v_readlane (or one LDS broadcast read when a lane holds
// several entries).  Same float32 prefilter / float64 fallback / tie replay as everywhere else.
// ---------------------------------------------------------------------------------------------
template <int CAP, int M>  // M: palette entries per lane (K <= 64 * M)
__global__ __launch_bounds__(64) void ed_rowserial_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                          const int h, const int w, const PalDev pal, const Taps taps,
                                                          const int serpentine)
{
    extern __shared__ __align__(16) float s_dyn[];  // err[3][w][3], then (M > 1) K x {x, y, z, out_rgb}
    __shared__ uint8_t s_lut[256];
    float *s_err = s_dyn;
    const float4 *s_pal = reinterpret_cast<const float4 *>(s_dyn + (((size_t)9 * w + 3) & ~(size_t)3));
    const int lane = threadIdx.x;
    const size_t f = blockIdx.x;
    const uint8_t *fin = in + f * (size_t)h * w * 3;
    uint8_t *fout = out + f * (size_t)h * w * 3;
    const float inf = __int_as_float(0x7f800000);
    const int K = pal.K;

    float4 pc[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const int j = lane + 64 * m;
        pc[m] = j < K ? pal.fcand[j] : make_float4(1e30f, 1e30f, 1e30f, 0.f);  // distance overflows to +inf
    }
    if (M > 1)
        for (int j = lane; j < K; j += 64) const_cast<float4 *>(s_pal)[j] = pal.fcand[j];
    for (int i = lane; i < 256; i += 64) s_lut[i] = pal.lut_in ? pal.lut_in[i] : (uint8_t)i;
    __syncthreads();

    // taps arrive sorted (dy descending, dx descending): the rows above first, then the same row (dx = 2, then 1)
    int n_above = 0;
    float w1 = 0.f, w2 = 0.f;
    for (int k = 0; k < taps.n; ++k) {
        if (taps.dy[k] > 0) n_above = k + 1;
        else if (taps.dx[k] == 1) w1 = taps.wq[k];
        else if (taps.dx[k] == 2) w2 = taps.wq[k];
    }

    // raw bytes of this lane's pixel in block (y, step0): loaded one block ahead of the walk
    auto load_px = [&](const int y, const int step0) -> uint32_t {
        const int step = step0 + lane;
        if (y >= h || step >= w) return 0u;
        const bool rv = serpentine && (y & 1);
        const int x = rv ? (w - 1 - step) : step;
        const uint8_t *px = fin + ((size_t)y * w + x) * 3;
        return (uint32_t)px[0] | ((uint32_t)px[1] << 8) | ((uint32_t)px[2] << 16);
    };
    // value of pixel (y, step) before the same-row taps: input (through lut_in) + contributions of the rows above,
    // in the reference's order; `rev`: direction of row y
    auto prepare = [&](const uint32_t raw, const int y, const int step, const bool rev, float &p0, float &p1, float &p2) {
        p0 = p1 = p2 = 0.f;
        if (y >= h || step >= w) return;
        const int x = rev ? (w - 1 - step) : step;
        float a0 = (float)s_lut[raw & 255u], a1 = (float)s_lut[(raw >> 8) & 255u], a2 = (float)s_lut[raw >> 16];
        for (int k = 0; k < n_above; ++k) {
            const int sr = y - taps.dy[k];
            if (sr < 0) continue;
            const int sdir = (serpentine && (sr & 1)) ? -1 : 1;
            const int sxp = x - taps.dx[k] * sdir;
            if (sxp < 0 || sxp >= w) continue;
            const float *e = s_err + ((size_t)(sr % 3) * w + sxp) * 3;
            const float wq = taps.wq[k];
            a0 = __fadd_rn(a0, __fmul_rn(e[0], wq));
            a1 = __fadd_rn(a1, __fmul_rn(e[1], wq));
            a2 = __fadd_rn(a2, __fmul_rn(e[2], wq));
        }
        p0 = a0;
        p1 = a1;
        p2 = a2;
    };

    uint32_t raw_next = load_px(0, 0);
    for (int y = 0; y < h; ++y) {
        const bool rev = serpentine && (y & 1);
        float e1x = 0.f, e1y = 0.f, e1z = 0.f, e2x = 0.f, e2y = 0.f, e2z = 0.f;  // errors of the previous two pixels
        for (int step0 = 0; step0 < w; step0 += 64) {
            const uint32_t raw = raw_next;
            raw_next = step0 + 64 < w ? load_px(y, step0 + 64) : load_px(y + 1, 0);  // in flight during the walk
            // rows above are complete (their LDS writes precede this point in program order)
            float p0, p1, p2;
            prepare(raw, y, step0 + lane, rev, p0, p1, p2);
            float my_e0 = 0.f, my_e1 = 0.f, my_e2 = 0.f;
            uint32_t my_c = 0;
            const int nstep = min(64, w - step0);
            for (int i = 0; i < nstep; ++i) {
                float a0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p0), i));
                float a1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p1), i));
                float a2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p2), i));
                // same-row taps (an absent source contributes fl(0*w) = 0, which leaves the sum unchanged)
                a0 = __fadd_rn(__fadd_rn(a0, __fmul_rn(e2x, w2)), __fmul_rn(e1x, w1));
                a1 = __fadd_rn(__fadd_rn(a1, __fmul_rn(e2y, w2)), __fmul_rn(e1y, w1));
                a2 = __fadd_rn(__fadd_rn(a2, __fmul_rn(e2z, w2)), __fmul_rn(e1z, w1));
                const float o0 = clamp255(a0), o1 = clamp255(a1), o2 = clamp255(a2);
                // lane-parallel palette scan
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
                // the second smallest distance is only needed for the margin test "b1 > b0 * 1.000002": it fails iff
                // some other entry lies within the margin, i.e. more than one lane passes, or a lane's own runner-up
                const float lim = B0 * 1.000002f;
                const unsigned long long close = __ballot(b0 <= lim);
                const bool near_tie = (close & (close - 1ull)) != 0ull || (M > 1 && __ballot(b1 <= lim) != 0ull);
                int j = M == 1 ? winner : __builtin_amdgcn_readlane(i0, winner);
                float cx, cy, cz;
                uint32_t cw;
                if (!near_tie) {
                    if (M == 1) {
                        cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pc[0].x), winner));
                        cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pc[0].y), winner));
                        cz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pc[0].z), winner));
                        cw = (uint32_t)__builtin_amdgcn_readlane(__float_as_int(pc[0].w), winner);
                    } else {
                        const float4 c = s_pal[j];
                        cx = c.x;
                        cy = c.y;
                        cz = c.z;
                        cw = __float_as_uint(c.w);
                    }
                } else {  // near tie: float64 scan, scipy's traversal on exact ties
                    j = nearest_f64<CAP>(pal, o0, o1, o2);
                    const float4 c = pal.fcand[j];
                    cx = c.x;
                    cy = c.y;
                    cz = c.z;
                    cw = __float_as_uint(c.w);
                }
                const float ex = __fsub_rn(o0, cx), ey = __fsub_rn(o1, cy), ez = __fsub_rn(o2, cz);
                if (lane == i) {  // the lane that prepared this pixel keeps its error and colour for the block's stores
                    my_e0 = ex;
                    my_e1 = ey;
                    my_e2 = ez;
                    my_c = cw;
                }
                e2x = e1x;
                e2y = e1y;
                e2z = e1z;
                e1x = ex;
                e1y = ey;
                e1z = ez;
            }
            if (lane < nstep) {
                const int x = rev ? (w - 1 - (step0 + lane)) : (step0 + lane);
                float *e = s_err + ((size_t)(y % 3) * w + x) * 3;
                e[0] = my_e0;
                e[1] = my_e1;
                e[2] = my_e2;
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

}  // namespace

size_t error_diffusion_ws_bytes(int64_t n_frames, int h, int w)
{
    (void)h;
    // wavefront: 4 boundary rows per frame; serial: 3 error rows per frame; up to 6 words per column and row (the
    // variable-weight diffusers of vardiff.hip carry an extra value per error: 4 floats; the numba arithmetic keeps its
    // errors in float64: 3 doubles)
    // + 256 bytes per frame of progress words (few frames in flight: a frame's bands spread over workgroups)
    return (size_t)n_frames * (size_t)w * 6 * sizeof(float) * 4 + 512 + (size_t)n_frames * 256;
}

int build_ed_cells(PalDev &dev, const double *pts, void **blob_out)
{
    *blob_out = nullptr;
    const int K = dev.K;
    static_assert(sizeof(U4) == sizeof(uint4), "U4 mirrors uint4");
    uint4 *cells = nullptr;
    // room behind the lists (and the nodes): the 16^3-cell table(s) -- 4096 quads (K > 16) or 2 x 4096 words (K <= 16) -- and, for
    // K > 16, the hierarchical table of up to kEdH4MaxWords words
    constexpr size_t kCoarseQuads = 4096 + kEdH4MaxWords / 4;   // 16^3 lists (or nibble tables) | hierarchical table
    DP_HIP(hipMalloc((void **)&cells, sizeof(uint4) * (kEdCells + kCoarseQuads)));
    hipLaunchKernelGGL(ed_cells_kernel, dim3(kEdCells / 256), dim3(256), 0, 0, dev, cells);
    hipError_t e = hipGetLastError();
    std::vector<U4> host(kEdCells);
    if (e == hipSuccess) e = hipMemcpy(host.data(), cells, sizeof(uint4) * kEdCells, hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        (void)hipFree(cells);
        return hip_fail(e, "error-diffusion candidate lists");
    }
    // Everything between the kernel's geometric lists and the finished tables is host work (host_logic.h): the pairwise
    // sharpening, the refinement of overflowing cells (palettes extracted from an image crowd their colours into a few
    // cells) into an octree down to unit cubes, the 16^3-cell lists for LDS.
    EdTables tb;
    ed_tables_refine(pts, K, host, tb);
    if (exp_env("DP_ED_H4_REPORT")) {
        fprintf(stderr, "ed tables: K %d, h4 %zu words (%zu wanted, limit %zu, in LDS up to %zu), mean depth at the palette's colours %.2f (no answer at %.3f of them), nodes %zu\n", K, tb.h4.size(), tb.h4_wanted, kEdH4MaxWords, kEdH4LdsWords, tb.h4_depth, tb.h4_none, tb.nodes.size());
        if (!tb.coarse.empty()) {   // K <= 16: how long the 16^3 nibble lists are (a wave pays for the longest among its 64 lanes)
            int hist[16] = {0};
            for (uint32_t w : tb.coarse) ++hist[w & 15u];
            fprintf(stderr, "ed tables: K %d, 16^3 nibble lists by length:", K);
            for (int n = 0; n < 16; ++n)
                if (hist[n]) fprintf(stderr, " %d:%d", n, hist[n]);
            fprintf(stderr, "\n");
        }
    }
    if (!tb.nodes.empty()) {  // one allocation: cells, then the nodes, then the 16^3-cell table(s)
        uint4 *both = nullptr;
        e = hipMalloc((void **)&both, sizeof(uint4) * (kEdCells + tb.nodes.size() + kCoarseQuads));
        (void)hipFree(cells);
        cells = both;
        if (e != hipSuccess) return hip_fail(e, "error-diffusion candidate lists");
        e = hipMemcpy(cells + kEdCells, tb.nodes.data(), sizeof(uint4) * tb.nodes.size(), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMemcpy(cells, host.data(), sizeof(uint4) * kEdCells, hipMemcpyHostToDevice);
    uint4 *tail = cells + kEdCells + tb.nodes.size();
    if (e == hipSuccess && !tb.l16.empty()) e = hipMemcpy(tail, tb.l16.data(), sizeof(uint4) * 4096, hipMemcpyHostToDevice);
    uint32_t *d_coarse = reinterpret_cast<uint32_t *>(tail);
    if (e == hipSuccess && !tb.coarse.empty()) e = hipMemcpy(d_coarse, tb.coarse.data(), sizeof(uint32_t) * 4096, hipMemcpyHostToDevice);
    if (e == hipSuccess && !tb.ext.empty()) e = hipMemcpy(d_coarse + 4096, tb.ext.data(), sizeof(uint32_t) * 4096, hipMemcpyHostToDevice);
    uint32_t *d_h4 = reinterpret_cast<uint32_t *>(tail + 4096);
    if (e == hipSuccess && !tb.h4.empty()) e = hipMemcpy(d_h4, tb.h4.data(), sizeof(uint32_t) * tb.h4.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(cells);
        return hip_fail(e, "error-diffusion candidate lists");
    }
    dev.ed_cells = cells;
    dev.ed_nodes = tb.nodes.empty() ? nullptr : cells + kEdCells;
    dev.ed_lists16 = tb.l16.empty() ? nullptr : tail;
    dev.ed_coarse = tb.coarse.empty() ? nullptr : d_coarse;
    dev.ed_coarse_ext = tb.ext.empty() ? nullptr : d_coarse + 4096;
    // (dev.ed_ext16 / ed_ext_nodes: build_ed_ext below, when an unclamped diffuser first meets the palette)
    dev.ed_h4 = tb.h4.empty() ? nullptr : d_h4;
    dev.ed_h4_words = (int)tb.h4.size();
    dev.ed_h4_shallow = (!tb.h4.empty() && tb.h4_depth <= 0.25) ? 1 : 0;
    // The few-frames instances (table in LDS) take it unless the palette is so crowded that the 8^3 lists themselves overflowed
    // into their octree (tb.nodes) or the walk is deep where the colours are: median cut of DARK content (colours a unit or two
    // apart) ran 10-20 % slower on the table than on the lists, of smooth content 3-30 % faster
    // (tools/bench_scripts/ed_crowded_lds.py: depths 1.9-2.8 with nodes against 0.3-1.6 without)
    // (small palettes have short lists to begin with: the walk must be shallower to pay -- median cut 32 of dark content, depth 1.50,
    // ran 15 % slower on the table)
    if (!tb.h4.empty() && (!tb.nodes.empty() || tb.h4_depth > std::min(1.75, (double)K / 25.0)) && !exp_env("DP_ED_H4_ALWAYS")) {
        dev.ed_h4 = nullptr;
        dev.ed_h4_words = 0;
    }
    *blob_out = cells;
    return DP_OK;
}

// The extended 16^3 lists of the unclamped diffusers (host_logic.h: EdTables::ext16 / ext_nodes), 17..1024 colours: built when such a
// diffuser first meets the palette (5-16 ms of host time that plain error diffusion never needs).  No policy switch: a wave that
// holds a lane without a usable list beyond the cube takes the whole-palette scan with ALL its lanes (vardiff.hip: nearest_ext16),
// so a palette crowded at a face of the cube -- every palette under use_gamma is, at the dark end -- never costs more than without.
int build_ed_ext(PalDev &dev, const double *pts, void **blob_out)
{
    *blob_out = nullptr;
    const int K = dev.K;
    if (K <= 16) return DP_OK;
    std::vector<U4> none;
    EdTables tb;
    ed_tables_refine(pts, K, none, tb, 2);
    if (tb.ext16.empty()) return DP_OK;
    uint4 *blob = nullptr;
    DP_HIP(hipMalloc((void **)&blob, sizeof(uint4) * (4096 + tb.ext_nodes.size())));
    hipError_t e = hipMemcpy(blob, tb.ext16.data(), sizeof(uint4) * 4096, hipMemcpyHostToDevice);
    if (e == hipSuccess && !tb.ext_nodes.empty())
        e = hipMemcpy(blob + 4096, tb.ext_nodes.data(), sizeof(uint4) * tb.ext_nodes.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(blob);
        return hip_fail(e, "extended candidate lists");
    }
    if (exp_env("DP_ED_H4_REPORT")) fprintf(stderr, "ed tables: K %d, extended 16^3 lists with %zu refinement nodes\n", K, tb.ext_nodes.size() / 8);
    dev.ed_ext16 = blob;
    dev.ed_ext_nodes = tb.ext_nodes.empty() ? nullptr : blob + 4096;
    *blob_out = blob;
    return DP_OK;
}

int launch_error_diffusion(const uint8_t *in, uint8_t *out, int64_t n_frames, int h, int w, const PalDev &pal_in,
                           const int32_t *dx, const int32_t *dy, const float *wq, int ntaps, int serpentine,
                           void *ws, size_t ws_bytes, hipStream_t s, const double *wq64, const double *hybrid)
{
    PalDev pal = pal_in;   // (by-value snapshot; the experiments build may switch a table off for an A/B)
    if (exp_env("DP_ED_NO_H4")) pal.ed_h4 = nullptr;
    // The sixteen-wave instances (rings fill their LDS) read the table from L2 only when it is SHALLOW where the palette's colours
    // are -- palettes spread over the cube: 256 frames 9-22 % faster than on the 8^3 lists; a palette extracted from the content
    // sends most lanes down two or three dependent L2 reads per step and runs 3-31 % SLOWER than on the lists
    // (tools/bench_scripts/ed_h4_global_policy.py: mean depth 0.00-0.03 against 0.34-2.83)
    pal.ed_h4_global = exp_env("DP_ED_H4_LDS_ONLY") ? 0 : (exp_env("DP_ED_H4_GLOBAL") ? 1 : pal.ed_h4_shallow);
    pal.ed_h4_lds_words = (int)kEdH4LdsWords;
    if (const char *e = exp_env("DP_ED_H4_LDS_WORDS")) pal.ed_h4_lds_words = std::min((int)kEdH4LdsWords, atoi(e));
    // wq64 != nullptr: the numba arithmetic (see nearest_numba_f32) with these float64 tap weights
    const bool numba = wq64 != nullptr;
    Taps t;
    t.n = ntaps;
    // hybrid != nullptr (with wq64): {lum_factor, col_factor} of _hybrid_numba
    t.hybrid = (numba && hybrid) ? 1 : 0;
    t.hyb_lum = hybrid ? hybrid[0] : 0.0;
    t.hyb_col = hybrid ? hybrid[1] : 0.0;
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
        t.wq64[i] = numba ? wq64[order[i]] : 0.0;
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
        t.wq64[i] = 0.0;
    }
    ProfMark *pm = prof_begin(s);
    if (!serpentine && skew * 2 + 2 <= kRing && w < 60000) {
        if (n_frames > 0x7fffffff) {
            set_error("dp_error_diffusion_u8: too many frames for one launch");
            return DP_EINVAL;
        }
        const int n_bands = (h + 63) / 64;
        int nw = n_bands < kMaxWaves ? n_bands : kMaxWaves;
        if (const char *e = exp_env("DP_ED_WAVES")) {  // experiments: waves per one-workgroup frame
            const int v = atoi(e);
            if (v >= 1 && v <= kMaxWaves && v <= n_bands) nw = v;
        }
        // Few frames in flight: spread each frame's bands over G workgroups so that the batch covers the CUs (one
        // workgroup per frame leaves all but n_frames CUs idle and packs 16 waves onto one CU's four SIMDs) and up to
        // 32 bands of a frame advance together.  The workgroups of a frame meet through progress words in the workspace.
        int G = 1;
        int cus = 0, dev_id = 0;
        if (hipGetDevice(&dev_id) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev_id) != hipSuccess) cus = 0;
        uint32_t *gprog = nullptr;
        {
            const size_t prog_off = ((size_t)n_frames * (size_t)w * (numba ? 96 : 48) + 255) & ~(size_t)255;
            const size_t prog_bytes = (size_t)n_frames * kEdProgWords * sizeof(uint32_t);
            if (n_frames * 2 <= cus && n_bands >= 4 && w >= 64 && prog_off + prog_bytes <= ws_bytes && !exp_env("DP_ED_ONE_WG")) {
                const int nwt = n_bands < 32 ? n_bands : 32;
                while (G * 2 <= 16 && n_frames * (G * 2) <= cus && G * 2 <= nwt) G *= 2;
                if (G > 1) {
                    nw = (nwt + G - 1) / G;
                    gprog = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(ws) + prog_off);
                    DP_HIP(hipMemsetAsync(gprog, 0, prog_bytes, s));
                }
            }
        }
        // A wave of a G > 1 launch that waits too long for another workgroup (a preempted or CU-masked GPU) sets its
        // frame's give-up flag and ends.  The repair launch right behind it (same stream, one workgroup per frame,
        // G = 1, frames without the flag return immediately) does those frames again, so the call never reports success
        // over partly written frames.
        const int test_giveup = exp_env("DP_ED_TEST_GIVEUP") ? 1 : 0;
        const int nw1 = n_bands < kMaxWaves ? n_bands : kMaxWaves;
        // More frames than CUs, one workgroup per frame and CU (more than 4 waves: the instances that fill a CU's LDS): a
        // PERSISTENT grid of one workgroup per CU, each doing every grid-th frame with its waves running on into the next frame
        // (ed_wavefront_kernel) -- the tail of a frame, when most of its bands are done, overlaps the head of the next.
        // The running band number has 16 bits in the progress words: enough workgroups that none counts past them.
        int64_t pgrid = n_frames * G;
        if (G == 1 && nw > 4 && cus > 0 && n_frames > cus && n_bands <= 4096 && !exp_env("DP_ED_NO_PERSIST")) {
            pgrid = cus;
            const int64_t need = (n_frames * n_bands + 59999) / 60000;
            if (pgrid < need) pgrid = need;
        }
        if (const char *e = exp_env("DP_ED_GRID")) {  // experiments / tests: any grid (several frames per workgroup on small batches)
            const int64_t v = atoll(e);
            if (G == 1 && v >= 1 && v <= n_frames && (n_frames + v - 1) / v * n_bands < 60000) pgrid = v;
        }
        const uint32_t nfr = (uint32_t)n_frames;
#define DP_EDW(C, N, X, S4, S)                                                                                               \
    do {                                                                                                                 \
        if (nw <= 4)                                                                                                     \
            hipLaunchKernelGGL((ed_wavefront_kernel<C, N, X, 4, false, S4>), dim3((unsigned)pgrid), dim3(64 * nw), 0, s, in, out, h, w, \
                               pal, t, ws, G, gprog, test_giveup, nfr);                            \
        else                                                                                                             \
            hipLaunchKernelGGL((ed_wavefront_kernel<C, N, X, kMaxWaves, false, S>), dim3((unsigned)pgrid), dim3(64 * nw), 0, s, in,  \
                               out, h, w, pal, t, ws, G, gprog, test_giveup, nfr);                 \
        if (G > 1)                                                                                                       \
            hipLaunchKernelGGL((ed_wavefront_kernel<C, N, X, kMaxWaves, false, S>), dim3((unsigned)n_frames), dim3(64 * nw1), 0, s, in, out,  \
                               h, w, pal, t, ws, 1, gprog, 0, nfr);                                \
    } while (0)
        // (a palette of at most 16 colours has at most 15 inner nodes: its instances are the small-queue ones)
        // Measured (ab_libs.py, one process, bytes identical; <= 16-only / > 16-only instance against one with everything):
        // sixteen waves, 16 colours 12.98 against 13.61 ms per 256 4K frames, 256 colours 21.6 against 22.8; one frame, 256 colours
        // 9.65 against 10.02 ms, 16 colours with 7-12 taps 2-8 % faster, with 3-6 taps within +-2 % (the sign follows code placement).
        const bool small_k = pal.ed_coarse != nullptr && pal.n_inner <= kQueueSmall;
#define DP_EDN(N, X)                                                                                                      \
    do {                                                                                                                 \
        if (pal.n_inner > kQueueSmall || pal.K > 256) DP_EDW(kQueueLarge, N, X, 0, 0);   /* (wide lists: these instances) */ \
        else if (small_k) DP_EDW(kQueueSmall, N, X, 1, 1);                                                                \
        else DP_EDW(kQueueSmall, N, X, 2, 2);                                                                             \
    } while (0)
        if (numba) {  // one general instance per workgroup size (a full float64 palette scan per pixel, no candidate lists);
            // float64 error rings: 8 waves per workgroup fill LDS (158 KB)
            constexpr int kNbWaves = 8;
            if (nw > kNbWaves) nw = kNbWaves;  // (bands are dealt round-robin over whatever waves there are)
            const int nwr = nw1 < kNbWaves ? nw1 : kNbWaves;
            if (nw <= 4)
                hipLaunchKernelGGL((ed_wavefront_kernel<kQueueSmall, kMaxTaps, false, 4, true>), dim3((unsigned)pgrid), dim3(64 * nw),
                                   0, s, in, out, h, w, pal, t, ws, G, gprog, test_giveup, nfr);
            else
                hipLaunchKernelGGL((ed_wavefront_kernel<kQueueSmall, kMaxTaps, false, kNbWaves, true>), dim3((unsigned)pgrid),
                                   dim3(64 * nw), 0, s, in, out, h, w, pal, t, ws, G, gprog, test_giveup, nfr);
            if (G > 1)
                hipLaunchKernelGGL((ed_wavefront_kernel<kQueueSmall, kMaxTaps, false, kNbWaves, true>), dim3((unsigned)n_frames),
                                   dim3(64 * nwr), 0, s, in, out, h, w, pal, t, ws, 1, gprog, 0, nfr);
        } else
        switch (ntaps) {  // the tap counts of the reference's kernels get a test-free instance
        case 3: DP_EDN(3, true); break;
        case 4: DP_EDN(4, true); break;
        case 6: DP_EDN(6, true); break;
        case 7: DP_EDN(7, true); break;
        case 10: DP_EDN(10, true); break;
        case 12: DP_EDN(12, true); break;
        default: DP_EDN(kMaxTaps, false); break;
        }
#undef DP_EDN
#undef DP_EDW
    } else if (!numba && n_frames <= 0x7fffffff &&
               ((size_t)9 * w + 4) * sizeof(float) + (pal.K > 64 ? (size_t)pal.K * 16 : 0) + 512 <= (size_t)158 * 1024) {
        // any scan direction, one wave per frame (three error rows in LDS)
        const size_t lds = ((((size_t)9 * w + 3) & ~(size_t)3)) * sizeof(float) + (pal.K > 64 ? (size_t)pal.K * 16 : 0);
        const bool big = pal.n_inner > kQueueSmall;
#define DP_EDR(C, MM)                                                                                                   \
    do {                                                                                                               \
        auto kern = ed_rowserial_kernel<C, MM>;                                                                        \
        DP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                   (int)lds));                                                                         \
        hipLaunchKernelGGL(kern, dim3((unsigned)n_frames), dim3(64), lds, s, in, out, h, w, pal, t, serpentine);        \
    } while (0)
        if (pal.K <= 64) {
            if (big) DP_EDR(kQueueLarge, 1); else DP_EDR(kQueueSmall, 1);
        } else if (pal.K <= 256) {
            if (big) DP_EDR(kQueueLarge, 4); else DP_EDR(kQueueSmall, 4);
        } else {
            if (big) DP_EDR(kQueueLarge, 16); else DP_EDR(kQueueSmall, 16);
        }
#undef DP_EDR
    } else {
        const int64_t blocks = (n_frames + 63) / 64;
        if (numba)  // serpentine scan with the numba arithmetic: the frame-parallel kernel (lane = frame)
            hipLaunchKernelGGL((ed_serial_kernel<kQueueSmall, true>), dim3((unsigned)blocks), dim3(64), 0, s, in, out, n_frames, h,
                               w, pal, t, serpentine, ws);
        else if (pal.n_inner > kQueueSmall || pal.K > 256)
            hipLaunchKernelGGL(ed_serial_kernel<kQueueLarge>, dim3((unsigned)blocks), dim3(64), 0, s, in, out, n_frames, h,
                               w, pal, t, serpentine, ws);
        else
            hipLaunchKernelGGL(ed_serial_kernel<kQueueSmall>, dim3((unsigned)blocks), dim3(64), 0, s, in, out, n_frames, h,
                               w, pal, t, serpentine, ws);
    }
    prof_end(pm, s);
    DP_HIP(hipGetLastError());
    return DP_OK;
}

}  // namespace dp
