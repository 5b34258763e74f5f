// Internal declarations shared by the translation units of libditherpie_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "../../include/ditherpie_hip.h"
#include "host_logic.h"

namespace dp {

constexpr int kQueueSmall = 64;   // traversal queue entries: every balanced tree of K <= 256 has <= 51 inner nodes
constexpr int kQueueLarge = 256;  // ... larger palettes (K <= 1024) and degenerate trees use the large instantiation
constexpr int kQueueTiles = 65536;  // wave tiles (256 px) with a flagged pixel that the fix-up pass visits directly (queue in the workspace)
constexpr int kIdxBits = 10;      // palette index bits packed under the distance key (brute-force kernels)
constexpr int kLocalBits = 8;     // byte offset of a candidate inside its block (cell-table kernel)

void set_error(const char *fmt, ...);

// Experiment switches (DP_* environment variables: forcing a table, a kernel or a schedule for tests, A/B measurements and
// tools/bench_scripts).  Only the library variant built with -DDP_EXPERIMENTS (libditherpie_hip_exp.so, what tests and
// tools load) looks at the environment; the default library (libditherpie_hip.so) has no getenv() on any path.
inline const char *exp_env(const char *name)
{
#ifdef DP_EXPERIMENTS
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}
int hip_fail(hipError_t e, const char *what);

#define DP_HIP(call)                                   \
    do {                                               \
        hipError_t e__ = (call);                       \
        if (e__ != hipSuccess) return dp::hip_fail(e__, #call); \
    } while (0)

// Device view of a prepared palette (plain pointers; passed to kernels by value).
struct PalDev {
    int K;
    int n_nodes;
    int n_inner;               // inner nodes of the tree = upper bound of the traversal queue
    int is_integer;
    const uint32_t *p4;        // K: r | g<<8 | b<<16          (integer palettes)
    const int32_t *nkey;       // K: (|p|^2 << kIdxBits) | j   (integer palettes)
    const double *pts;         // K*3 float64 coordinates as the tree sees them
    const float *pts_f32;      // the same values as float32 (they are float32 values to begin with)
    const uint32_t *out_rgb;   // K: output bytes r | g<<8 | b<<16
    const uint8_t *lut_in;     // 256 or nullptr
    // tree
    const int32_t *indices;    // K
    const int32_t *split_dim;  // n_nodes (-1 = leaf)
    const double *split;       // n_nodes
    const int32_t *start, *end, *less, *greater;
    double mins[3], maxes[3];
    // search accelerator (accel.hip); cell_tab == nullptr when absent
    const uint32_t *cell_tab;   // [4096 cells][8] packed colours r | g<<8 | b<<16 (sorted by palette index),
                                // then [n_split][8 sub-cells][8]; word 0 of a block with bit 31 set is a marker:
                                // 0x80000000|split index (cell is split) or 0xC0000000 (resolve in the fix-up pass)
    int tab_words;              // words of cell_tab that the dither kernels stage in LDS ...
    int tab_total;              // ... of tab_total: with clustered palettes the deepest split nodes stay in global memory
                                // (only pixels of split cells, on the deferred path, ever read them)
    // small palettes: the same table with 4-entry blocks (16 bytes per cell), nullptr when too many cells overflow;
    // the lean kernel prefers it (half the candidate work)
    const uint32_t *cell_tab4;
    int tab4_words;
    // staging orders for ordered_fast_kernel (accel.hip, assemble_table): per cell block (by cell_slot) which entry of
    // the index-ordered block goes to which LDS slot, nearest set first; nullptr = the fast kernel is not used.
    const uint32_t *cell_perm;  // for cell_tab (3-bit fields); every unsplit cell's nearest set fits near_slots entries
    const uint32_t *cell_perm4; // for cell_tab4 (2-bit fields)
    int near_slots;
    // ... and, for the cells that are split, the whole candidate list of the cell as a flat block of 16 entries in
    // index order (perm word 0xff000000 | list number): n_wide lists of 16 words
    const uint32_t *cell_wide, *cell_wide4;
    int n_wide, n_wide4;
    int adapt;                  // the palette crowds a few cells of cell_tab: use the kernel instantiation that adapts per wave
    // crowded palettes (extracted from an image): a table over WARPED cells -- cell coordinates are
    // (warp_lut[r], warp_lut[256 + g], warp_lut[512 + b]) instead of (r, g, b) -- or nullptr.  Lean kernels only.
    const uint32_t *warp_tab;   // blocks of warp_bw entries (4 or 8), same structure as cell_tab
    const uint8_t *warp_lut;    // 768 bytes
    int warp_words, warp_total; // staged / all words (as tab_words / tab_total)
    int warp_bw;
    int warp_adapt;
    // crowded palettes: the table the adaptive lean kernel would use (plain or warped 8-entry blocks), one BYTE per entry
    // (index into p4; accel.hip: compact_table) -- ordered_compact_kernel keeps all of it in LDS; nullptr when absent
    const uint32_t *comp_tab;
    int comp_words;             // (4096 + 8 * nodes) * 2
    int comp_warp;              // its cells are the warped ones (warp_lut)
    int n_split;
    int n_slow_blocks;
    int n_split_cells;          // 16^3 cells of cell_tab that are split (a palette crowded into few cells has many)
    int max_cell;
    const uint32_t *code1;      // 2 bits per colour: tie outcome of the k=1 query
    const uint32_t *code2;      // 4 bits per colour: ... of the k=2 query (accel.hip)
    // the same for float (gamma) palettes: blocks of 8 byte offsets (16 * j) into fcand
    const uint32_t *ftab;
    int ftab_words;             // words staged in LDS ...
    int ftab_total;             // ... of ftab_total (deep split nodes of clustered palettes stay in global memory)
    const float4 *fcand;        // K: {x, y, z, out_rgb bits}
    // error diffusion: for each 8x8x8 cell of the cube the entries that can be nearest to some point of the cell
    // (count in byte 0, up to 15 indices; count 255 = more than that: scan the whole palette); nullptr if absent
    const uint4 *ed_cells;
    const uint4 *ed_nodes;      // refinement of overflowing cells: 8 entries per node (count byte 254 = refined further)
    const uint32_t *ed_coarse;  // palettes of 9..16 colours: lists of the 16^3 cells, count | 7 index nibbles (count 15: too long)
    const uint32_t *ed_coarse_ext;  // the same for the diffusers that do not clamp (vardiff.hip): the outermost cells stand for the half-spaces beyond the cube
    const uint4 *ed_ext16;       // 17..256 colours: the 16^3 lists whose outermost cells are unbounded (EdTables::ext16), for the unclamped diffusers; or null
    const uint4 *ed_ext_nodes;   // the octree below its outermost cells whose list is too long (EdTables::ext_nodes), or null
    const uint4 *ed_lists16;    // palettes of 17..256 colours: lists of the 16^3 cells, count byte | up to 15 index bytes (255: too long)
    const uint32_t *ed_h4;      // palettes of 17..256 colours: hierarchical table of <= 4 entries per leaf (host_logic.h EdTables::h4), or null
    int ed_h4_words;
    int ed_h4_shallow;          // mean depth of the table at the palette's own colours <= 0.25: worth reading from L2 (ediff.hip)
    int ed_h4_global;           // the sixteen-wave instances read it from global memory (set per launch)
    int ed_h4_lds_words;        // the few-frames instances stage it in LDS up to this size (set per launch)
    const uint4 *exc;           // colours whose outcome no code expresses, sorted by colour:
    int n_exc;                  //   {colour, k=2 indices i0 | i1<<16, k=1 index, 0}; n_exc < 0: list overflowed
};

struct ThrDev {
    int th_h, th_w;
    const float *f32;     // th_h*th_w
    const uint32_t *m;    // integer form: t = m / 2^sh (nullptr when not representable)
    int sh;
    // lean ordered kernels: rows of tw_pad = th_w + 3 entries, the last three repeating the row from its start, so
    // that the 4 thresholds of a lane's pixels are 4 consecutive entries whatever the starting column.
    // fpad: the float32 values; mpad: the integer form (needs sh >= 0).  pow2: both sizes are powers of two
    // (positions by masking; otherwise by an exact float64 reciprocal, inv_h / inv_w).
    const float *fpad;
    const uint32_t *mpad;
    // Classes of the wave tiles (ordered kernels): nibble row * th_w + col of cls_nib, bit q set = when lane 0 of a wave
    // sits at (row, col) of the table and its 64 lanes x 4 pixels lie in one image row, no lane's pixel q has a threshold
    // below 1/2.  Held BY VALUE (kernel arguments: scalar loads, no memory dependence) for tables of up to 256 entries
    // that have such slots at all (has_cls); larger tables (blue noise) have none worth testing for.
    uint32_t cls_nib[32];
    int has_cls;
    int tw_pad;
    int pow2;
    double inv_h, inv_w;
};

}  // namespace dp

struct dp_palette {
    dp::PalDev dev;
    void *blob;  // one device allocation backing every pointer in dev
    size_t blob_bytes;
    void *accel_blob;  // cell lists + tie codes (may be null)
    void *ed_blob;     // dev.ed_cells (may be null)
    void *ext_blob;    // dev.ed_ext16 / ed_ext_nodes (may be null)
    size_t accel_bytes;
    bool accel_tried, same_out;
    std::vector<uint32_t> p4_host;
    std::vector<float> pal_host;   // K*3 as given (float palettes: for the accelerator build)
    std::vector<uint8_t> lut_host; // 256 or empty
    bool float_accel;              // a float palette the cell-table accelerator handles
    int device;
    // the candidate tables of the diffusion kernels (dev.ed_*) are built at the first diffusion call with the palette
    // (host.cpp: ensure_ed_tables): ordered-only users -- one palette per image in the CLI -- never pay for them
    std::vector<double> pts_host;  // K*3 float64 (KD-tree points)
    bool ed_tried;
    bool ext_tried;
    // `dev` is what the kernels are launched with.  The accelerator and the diffusion tables are added to it lazily,
    // possibly while other host threads are launching with the same palette (ctypes drops the GIL): builders serialise
    // on build_mu, fill a private copy and publish it with one assignment under dev_mu; every launch works on a
    // by-value snapshot taken under dev_mu (host.cpp: snapshot / publish).  Pointers of an older snapshot stay valid
    // until dp_palette_destroy.
    std::mutex build_mu, dev_mu;
};

struct dp_thresholds {
    dp::ThrDev dev;
    void *blob;
    void *blob_pad;  // dev.fpad / dev.mpad (may be null)
    int device;
};

// per-thread HIP-event profiling (dp_profile_enable / dp_profile_read)
namespace dp {
struct ProfMark {
    hipEvent_t ev[3];  // start, after main kernel, after fix-up
    int n;
};
bool prof_on();
// returns nullptr when profiling is off; otherwise a mark whose events the launcher records
ProfMark *prof_begin(hipStream_t s);
void prof_mid(ProfMark *m, hipStream_t s);
void prof_end(ProfMark *m, hipStream_t s);
}  // namespace dp

// launchers (defined in the .hip files)
namespace dp {
int build_accel(PalDev &dev, const std::vector<uint32_t> &p4_host, void **blob_out, size_t *blob_bytes);
int build_accel_float(PalDev &dev, const float *pal_f32, const uint8_t *lut_host, void **blob_out, size_t *blob_bytes);
int build_ed_cells(PalDev &dev, const double *pts_host, void **blob_out);
int build_ed_ext(PalDev &dev, const double *pts_host, void **blob_out);   // the unclamped diffusers' extended lists (ediff.hip)
int launch_ordered(const uint8_t *in, uint8_t *out, int64_t n_frames, int h, int w, int y0, int x0,
                   const PalDev &pal, int mode, const ThrDev *thr, float ign_scale, int ign_seed, void *ws,
                   size_t ws_bytes, hipStream_t s);
int launch_ign_thresholds(float *out, int h, int w, int y0, int x0, float scale, int seed, hipStream_t s);
int launch_blue_noise(int size, uint32_t seed, float *out_dev, void *scratch_dev, hipStream_t s);
size_t blue_noise_scratch_bytes(int size);
int launch_error_diffusion(const uint8_t *in, uint8_t *out, int64_t n_frames, int h, int w, const PalDev &pal,
                           const int32_t *dx, const int32_t *dy, const float *wq, int ntaps, int serpentine,
                           void *ws, size_t ws_bytes, hipStream_t s, const double *wq64 = nullptr, const double *hybrid = nullptr);
size_t error_diffusion_ws_bytes(int64_t n_frames, int h, int w);
int launch_kmeans_step(const uint8_t *px, int64_t n, const double *centers, const double *mean, int K, int64_t *sums, int64_t *counts,
                       int64_t *sumsq, hipStream_t s);
size_t distinct_first_ws_bytes(int64_t n);
int launch_distinct_first(const uint8_t *px, int64_t n, uint8_t *out, long long *n_distinct, void *ws, hipStream_t s);
size_t kmeans_hist_bytes();
size_t kmeans_hist_ws_bytes(int64_t n);
int launch_kmeans_hist_build(const uint8_t *px, int64_t n, void *hist, int accumulate, void *ws, hipStream_t s);
int launch_kmeans_hist_step(const void *hist, const double *centers, const double *mean, int K, int64_t *sums, int64_t *counts,
                            int64_t *sumsq, hipStream_t s);
int launch_kmeans_hist_iterate(const void *hist, double *centers, const double *mean, int K, int64_t *totals, int64_t *prev,
                               double *status, uint32_t *ticket, double tol, int max_iter, int first, hipStream_t s);
int launch_kmeans_pp(const uint8_t *sample, int n, int K, int first, const double *uniforms, int n_trials, int *out_ids,
                     double *out_centers, hipStream_t s);
int launch_kmeans_update(const int64_t *totals, double *centers, int64_t *prev, double *status, int K, double tol, int max_iter,
                         hipStream_t s);
int launch_variance_gate(const uint8_t *in, uint8_t *gate, int64_t n_frames, int h, int w, const PalDev &pal, float thr,
                         int radius, void *ws, hipStream_t s);
size_t variance_gate_ws_bytes(int64_t n_frames, int h, int w);
int launch_variable_diffusion(const uint8_t *in, uint8_t *out, int64_t n_frames, int h, int w, const PalDev &pal, int model,
                              float p0, float p1, int serpentine, const uint8_t *gate, const float *coef, void *ws,
                              hipStream_t s);
int launch_resize_nearest(const uint8_t *in, uint8_t *out, int64_t n_frames, int h, int w, int oh, int ow,
                          hipStream_t s);
}  // namespace dp
