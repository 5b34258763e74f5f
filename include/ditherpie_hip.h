/*
 * ditherpie_hip.h -- C ABI of libditherpie_hip.so, the MI355X (gfx950) backend for the
 * per-pixel hot path of dobrosketchkun/dither_pie.
 *
 * The reference is pure Python; it has no FFI of its own.  Each entry point below names the
 * reference interface (file:line under the reference repo) whose arithmetic it replaces.  The
 * binding a maintainer adds on the reference side is a ctypes stub; see INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 on success, a DP_E* code otherwise; dp_last_error() gives the
 *     text for the calling thread.  Nothing aborts.
 *   - "dev" pointers are device (HBM) pointers owned by the caller (e.g. torch tensors);
 *     "host" pointers are ordinary host memory.  The library never frees caller memory.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  All work is
 *     enqueued asynchronously on it; the caller synchronises.
 *   - images are packed uint8 RGB, HWC, row pitch = 3*w bytes, frames back to back
 *     (frame stride = 3*h*w bytes).  Output has the same layout.
 *   - the library uses the calling thread's current HIP device.
 *   - no mutable global state besides the thread-local error text and profiling marks.  One call is a SEQUENCE of
 *     launches that use the caller's workspace in stream order (flag bitmap, progress words): concurrent calls are fine
 *     on different streams with different workspaces; calls that share a workspace and stream must not be issued from
 *     two host threads at once (the Python layer serialises them per (device, stream)).  Tables a palette acquires after
 *     creation (dp_palette_build_accel, the candidate tables of the first diffusion call) are built into a private copy
 *     of the palette's device record under a per-palette mutex and published with one assignment; every launch works on
 *     a by-value snapshot of that record, so other threads may keep launching with the palette while it is being built --
 *     they simply still run without the new table.
 *   - the library reads NO environment variable (nm -D shows no getenv).  The experiment / test switches that force a
 *     table, kernel or schedule exist only in the twin build libditherpie_hip_exp.so (-DDP_EXPERIMENTS, same sources);
 *     INTEGRATION.md lists them.
 *   - dp_version() returns DP_ABI_VERSION; a binding checks it at load time (a changed argument list bumps it).
 */
#ifndef DITHERPIE_HIP_H
#define DITHERPIE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DP_OK 0
#define DP_EINVAL 1      /* bad argument (NULL pointer, size out of range, ...) */
#define DP_EUNSUPPORTED 2 /* e.g. palette larger than DP_MAX_COLORS */
#define DP_EHIP 3        /* a HIP runtime call failed; see dp_last_error() */
#define DP_ENOMEM 4
#define DP_EWORKSPACE 5  /* workspace too small; see the *_workspace_bytes query */

#define DP_MAX_COLORS 1024

/* 100: rounds 1-2.  101: dp_kmeans_step_u8 takes mean_dev (round 3).  102: round 4's surface (dp_distinct_first_u8,
 * dp_kmeans_hist_*, dp_hybrid_numba_u8, dp_error_diffusion_numba_u8 computing the float64 reading of the numba branch,
 * dp_error_diffusion_workspace_bytes at 6 words per column) and round 5's larger dp_kmeans_hist_workspace_bytes (whole
 * 32-byte sectors per scatter workgroup and cell) with the overflow word in the histogram's info block. */
#define DP_ABI_VERSION 102

#define DP_MODE_NEAREST 0 /* NoDitherStrategy                     dithering_lib.py:333-341 */
#define DP_MODE_MATRIX 1  /* MatrixDitherStrategy (Bayer, blue)   dithering_lib.py:346-378 */
#define DP_MODE_IGN 2     /* InterleavedGradientNoiseDitherStrategy dithering_lib.py:502-571 */

typedef struct dp_palette dp_palette;       /* prepared palette, device resident */
typedef struct dp_thresholds dp_thresholds; /* threshold matrix, device resident */

/* library / device ---------------------------------------------------------------------- */
int dp_version(void);
const char *dp_last_error(void);
/* number of HIP devices visible and the gcnArchName of the current one (buf may be NULL) */
int dp_device_info(int *n_devices, char *arch_buf, size_t arch_buf_len);

/* palette ----------------------------------------------------------------------------------
 * Replaces `KDTree(palette_arr)` at dithering_lib.py:339, 358, 554, 655 plus the palette /
 * output conversions of ImageDitherer.apply_dithering (dithering_lib.py:1970-1974, 1984-1990).
 *   pal_f32     host, K x 3 float32: the colours the KD-tree sees (sRGB ints, or the linearised
 *               non-integer values when use_gamma is on)
 *   out_colors  host, K x 3 uint8: the bytes written when entry i is chosen
 *   lut_in      host, NULL or 256 uint8: per-channel map applied to every input byte before the
 *               search (the uint8-quantised sRGB->linear map of dithering_lib.py:1957-1959)
 * Builds scipy.spatial.KDTree's structure (leafsize 10, sliding-midpoint rules, libstdc++
 * nth_element order) on the host, uploads it, and picks the exact-integer fast path when every
 * palette value is an integer in [0,255] and lut_in is NULL.  Costs ~0.05 ms (0.1 ms at 1024 colours).
 * The candidate tables of the diffusion kernels (9..256 colours) are NOT built here: the first
 * dp_error_diffusion_* / dp_variable_diffusion_u8 call with the palette builds them on the host (up to 8
 * threads: 1.4 ms at 16 colours, 1.9 ms at 256; under a per-palette mutex, so concurrent first calls are safe) and uploads them
 * synchronously; ordered-only users never pay for them. */
int dp_palette_create(const float *pal_f32, const uint8_t *out_colors, int K, const uint8_t *lut_in,
                      dp_palette **out);
void dp_palette_destroy(dp_palette *p);
int dp_palette_info(const dp_palette *p, int *K, int *is_integer, int *n_nodes);
/* Search accelerator (accel.hip): per-cell candidate lists kept in LDS by the dither kernels; for integer
 * palettes (4..1024 colours whose output colours are the palette colours) also per-colour tie codes and an
 * exception list, so that scipy's tie order needs no tree traversal; float palettes (use_gamma, 8..256
 * colours) get the same table over the lut_in-mapped pixel values.  Building it scans all 2^24 colours once
 * (~3.5 ms up to 256 colours, ~25 ms at 1024, synchronous), which pays off only after ~1.4e10 / K pixels (56 Mpixel at 256 colours: the
 * brute-force kernels need ~9 K vector instructions per pixel, the table kernels ~70); without it dp_ordered_u8 runs
 * the brute-force kernels.  dp_palette_build_accel is idempotent and returns DP_OK
 * without building when the palette does not qualify or the table would not fit LDS.
 * dp_palette_accel_info: size of the LDS table in 32-bit words and the longest exact candidate list; both 0
 * when there is no accelerator. */
int dp_palette_build_accel(dp_palette *p);
int dp_palette_accel_info(const dp_palette *p, int *pool_entries, int *max_cell);

/* Host-only (no GPU needed): the KD-tree build used by dp_palette_create, exported so that it can
 * be checked against scipy on any machine.  Arrays: indices[K]; nodes up to 2*K entries each.
 * Returns the number of nodes through *n_nodes. */
int dp_kdtree_build_host(const double *pts, int K, int32_t *indices, int32_t *split_dim, double *split,
                         int32_t *start, int32_t *end, int32_t *less, int32_t *greater, int *n_nodes);

/* thresholds -------------------------------------------------------------------------------
 * dp_thresholds_create: upload a th_h x th_w float32 matrix (Bayer tables
 * dithering_lib.py:1705-1768, polka-dot, any MatrixDitherStrategy matrix).
 * dp_thresholds_blue_noise: generate_blue_noise(size, seed) (dithering_lib.py:381-399) computed on
 * the device, including numpy's legacy RandomState(seed).shuffle.
 * dp_thresholds_download copies the float32 matrix to host memory (th_h*th_w floats). */
int dp_thresholds_create(const float *thr_host, int th_h, int th_w, dp_thresholds **out);
int dp_thresholds_blue_noise(int size, uint32_t seed, void *stream, dp_thresholds **out);
int dp_thresholds_shape(const dp_thresholds *t, int *th_h, int *th_w, int *is_integer_form);
int dp_thresholds_download(const dp_thresholds *t, float *thr_host);
void dp_thresholds_destroy(dp_thresholds *t);

/* IGN threshold field (dithering_lib.py:539-549) written to out_dev[h*w] float32; (y0,x0) are the
 * global coordinates of out_dev[0]. */
int dp_ign_thresholds(float *out_dev, int h, int w, int y0, int x0, float scale, int seed, void *stream);

/* ordered / nearest family -------------------------------------------------------------------
 * Replaces NoDitherStrategy.dither (dithering_lib.py:337-341), MatrixDitherStrategy.dither
 * (:355-378) and InterleavedGradientNoiseDitherStrategy.dither (:551-568) together with the
 * uint8 conversions around them (:1953, 1977, 1984).
 *   in_dev/out_dev  n_frames x h x w x 3 uint8
 *   (y0,x0)         global coordinates of pixel (0,0) of every frame (row-band / tile sharding)
 *   thr             required for DP_MODE_MATRIX, ignored otherwise
 *   workspace_dev   at least dp_ordered_workspace_bytes(n_frames,h,w) bytes, 16-byte aligned
 * Results are bit-identical to the reference including scipy's tie order. */
size_t dp_ordered_workspace_bytes(int64_t n_frames, int h, int w);
int dp_ordered_u8(const uint8_t *in_dev, uint8_t *out_dev, int64_t n_frames, int h, int w, int y0, int x0,
                  const dp_palette *pal, int mode, const dp_thresholds *thr, float ign_scale, int ign_seed,
                  void *workspace_dev, size_t workspace_bytes, void *stream);

/* error diffusion ------------------------------------------------------------------------------
 * Replaces ErrorDiffusionDitherStrategy.dither, pure-Python branch (dithering_lib.py:655-690).
 *   dx,dy,wq  host arrays of ntaps entries in the reference's list order (ErrorDiffusionKernel,
 *             dithering_lib.py:107-188); wq[k] = (float)(weight/divisor)
 *   serpentine  0/1 (dithering_lib.py:659-664) */
size_t dp_error_diffusion_workspace_bytes(int64_t n_frames, int h, int w);
int dp_error_diffusion_u8(const uint8_t *in_dev, uint8_t *out_dev, int64_t n_frames, int h, int w,
                          const dp_palette *pal, const int32_t *dx, const int32_t *dy, const float *wq,
                          int ntaps, int serpentine, void *workspace_dev, size_t workspace_bytes,
                          void *stream);

/* The same with the arithmetic of the reference's numba branch (_error_diffusion_numba, dithering_lib.py:213-308, taken
 * at :638-653 when numba is importable), typed per numba's unification rule: `r` is assigned a float32 array element
 * (:239) and the float64 literals 0.0 / 255.0 (:242-245), so r, g, b -- and everything computed from them -- are float64:
 * nearest entry = first minimum of a float64 linear scan ((dr*dr + dg*dg) + db*db, strict <), err = float64(r) -
 * float64(chosen) kept in float64, pushed as float32(float64(v) + err * (float64(weights[k]) / divisor)).
 *   weights  host array of ntaps float32 (the reference's np.float32 weight values), divisor as the reference passes it
 * Parity status: restated in the CPU oracle (orc_error_diffusion_numba_u8) and in numpy; fixtures pending -- NOT pinned
 * by reference output (numba cannot be installed in the build image).  (Rounds 1-3 implemented a float32 scan and a
 * float32 error; ABI 101 carries the float64 reading.) */
int dp_error_diffusion_numba_u8(const uint8_t *in_dev, uint8_t *out_dev, int64_t n_frames, int h, int w,
                                const dp_palette *pal, const int32_t *dx, const int32_t *dy, const float *weights,
                                double divisor, int ntaps, int serpentine, void *workspace_dev, size_t workspace_bytes,
                                void *stream);

/* HybridDitherStrategy's numba branch (_hybrid_numba, dithering_lib.py:1396-1494, taken at :1114-1125 when numba is importable):
 * unlike the pure-Python branch of the same strategy (dp_variable_diffusion_u8, model 2) it CLAMPS the value to [0, 255] before
 * the search, takes the first minimum of a float64 linear scan, and -- typed per numba's unification rule exactly as
 * dp_error_diffusion_numba_u8 -- keeps the error in float64: err = r - chosen, lum_err_val = (0.299 err0 + 0.587 err1) + 0.114 err2,
 * lum_c = w_c lum_err_val, fe_c = lum_factor lum_c + col_factor (err_c - lum_c), pushed with the Floyd-Steinberg weights as
 * float32(float64(v) + fe_c * (7/16 | 3/16 | 5/16 | 1/16)).  Workspace: dp_error_diffusion_workspace_bytes.
 * Parity status: restated in the CPU oracle (orc_hybrid_numba_u8) and in numpy; fixtures pending, NOT pinned (no numba here). */
int dp_hybrid_numba_u8(const uint8_t *in_dev, uint8_t *out_dev, int64_t n_frames, int h, int w, const dp_palette *pal,
                       double lum_factor, double col_factor, void *workspace_dev, size_t workspace_bytes, void *stream);

/* variable-weight diffusers -------------------------------------------------------
 * Replaces the pure-Python branches of PerceptualDitherStrategy.dither (dithering_lib.py:1030-1066, model 1),
 * HybridDitherStrategy.dither (:1111-1155, model 2; p0 = lum_factor, p1 = col_factor),
 * AdaptiveVarianceDitherStrategy.dither (:984-1017, model 3; gate_dev from dp_variance_gate_u8) and
 * OstromoukhovDitherStrategy.dither (:1229-1266, model 4; coef_dev = 256 x 3 float32 c_k/(c0+c1+c2),
 * serpentine 0/1).  Workspace: dp_error_diffusion_workspace_bytes(n_frames, h, w). */
#define DP_DIFFUSER_PERCEPTUAL 1
#define DP_DIFFUSER_HYBRID 2
#define DP_DIFFUSER_ADAPTIVE_VARIANCE 3
#define DP_DIFFUSER_OSTROMOUKHOV 4
int dp_variable_diffusion_u8(const uint8_t *in_dev, uint8_t *out_dev, int64_t n_frames, int h, int w,
                             const dp_palette *pal, int model, float p0, float p1, int serpentine,
                             const uint8_t *gate_dev, const float *coef_dev, void *workspace_dev,
                             size_t workspace_bytes, void *stream);
/* The variance gate of AdaptiveVarianceDitherStrategy (dithering_lib.py:988-992, 1019-1025): gate_dev[f][y][x] =
 * (max(0, uniform_filter(gray^2) - uniform_filter(gray)^2) >= var_threshold), scipy.ndimage.uniform_filter's
 * arithmetic (float32 passes, double running sums, mode 'nearest') on the palette's (LUT-mapped) input. */
size_t dp_variance_gate_workspace_bytes(int64_t n_frames, int h, int w);
int dp_variance_gate_u8(const uint8_t *in_dev, uint8_t *gate_dev, int64_t n_frames, int h, int w, const dp_palette *pal,
                        float var_threshold, int window_radius, void *workspace_dev, size_t workspace_bytes,
                        void *stream);

/* k-means palette extraction ---------------------------------------------------------------------
 * One Lloyd pass of the KMeans fit at dithering_lib.py:1854-1856 over n uint8 RGB pixels: nearest
 * centre in float64 and exact int64 per-cluster totals: channel sums
 * sums_dev[K*3], member counts counts_dev[K] and squared norms sumsq_dev[K] (sum of r^2+g^2+b^2),
 * all overwritten.  Being integers, the totals all-reduce exactly across ranks (RCCL, any order);
 * the caller updates the centres and derives the inertia
 *   sum_k ( sumsq_k - 2 c_k . sums_k + counts_k |c_k|^2 ).
 * Images of 2^19 pixels and more with K <= 256 go through per-cell candidate lists that the call rebuilds from
 * centers_dev first (one small extra launch; 64 KB of library-owned device memory per (device, stream) that has
 * run such a pass, kept until the library is unloaded; a call enqueues its two launches under that entry's mutex, so
 * host threads that share a stream cannot interleave them); same totals.
 *   centers_dev  K*3 float64.  PRECONDITION: every coordinate within [0, 255] (means of uint8 pixels and k-means++
 *                seeds always are): the float32 ranking in front of the float64 decision packs biased scores into one
 *                binary exponent and is only valid for centres inside the colour cube.  kmeans.lloyd() checks
 *                caller-supplied initial centres; the C ABI does not (the values are on the device).
 *   mean_dev     3 float64 or NULL.  Given (the mean of ALL the fit's pixels, sum / n per channel), a pixel that is
 *                equidistant from two centres gets the label sklearn gives it: the argmin of sklearn's own float64
 *                expression |c'|^2 - 2 x'.c' on mean-centred data (KMeans.fit centres the data; rounding decides exact
 *                ties, reproducibly -- see label_f64 in kmeans.hip for the operation order: it is the order of the x86
 *                AVX-512 / FMA OpenBLAS + numpy configuration the fixtures were recorded on, tests/golden/kat.json
 *                "kmeans_fixture_host"; sklearn on another BLAS may break exact ties differently).  With it a fit of
 *                <= 10 000 pixels, where the reference is deterministic, reproduces the reference's centres to 1e-9 on
 *                every fixture incl. structured images (tests/golden kmx_*); the truncated palette is the reference's
 *                except where a cluster mean is an exact integer: sklearn's threaded float64 sums land on it or 1 ulp
 *                below, so the reference returns colour or colour - 1 from run to run (tests/golden kmf_*), the exact
 *                integer totals here always the colour.  NULL: lowest index on exact ties.
 *   sumsq_dev    may be NULL: the squared norms are then not accumulated -- their total is a constant of the data, only
 *                the first pass of a fit needs it. */
int dp_kmeans_step_u8(const uint8_t *px_dev, int64_t n, const double *centers_dev, const double *mean_dev, int K,
                      int64_t *sums_dev, int64_t *counts_dev, int64_t *sumsq_dev, void *stream);
/* The same pass over the COLOUR HISTOGRAM of the pixels instead of the pixels (kmeans_hist.hip).  A pixel's label depends on
 * its colour only and the totals are sums of count x colour, so a fit reads its pixels ONCE (3 B/pixel) into count[colour]
 * over all 2^24 colours and every Lloyd pass of dithering_lib.py:1854-1856 then costs 16 KB per occupied 16^3 cell of the
 * colour cube (64 MB when every cell is occupied), whatever the number of pixels.  Same labels (float64 decision, mean_dev as
 * above), same int64 totals as dp_kmeans_step_u8, bit for bit.
 *   dp_kmeans_hist_bytes     size of the caller-owned histogram buffer (2^24 uint32 counts, cell-major, + 4096 cell totals)
 *   dp_kmeans_hist_build_u8  adds n pixels to hist_dev (16-byte aligned); accumulate = 0 clears it first, 1 keeps what it
 *                            holds (several buffers into one histogram).  Fewer than 2^32 pixels in total (32-bit counts):
 *                            an accumulating build that carries a cell's pixel count past 2^32 sets the uint32 word at byte
 *                            offset 4 * (2^24 + 8193) of hist_dev to 1 (it stays set until a build with accumulate = 0) --
 *                            the counts have wrapped and the histogram is then wrong; callers that add more than 2^32 - 1
 *                            pixels must check it (the Python wrapper refuses such totals up front).  Accumulating builds
 *                            into ONE histogram must be ordered on one stream: two concurrent builds race on the table.
 *                            By partition, not by one global atomic per pixel: the pixels are bucketed by their 16^3 cell
 *                            (2 bytes per pixel in workspace_dev, written in whole 32-byte sectors: each of up to 256
 *                            scatter workgroups may leave one padded sector per cell, so dp_kmeans_hist_workspace_bytes(n)
 *                            is 2 n + up to 32 MB; 16-byte aligned), and every bucket becomes its cell's table slice
 *                            through an LDS histogram.
 *   dp_kmeans_hist_step      one pass; K <= 256 (DP_EUNSUPPORTED above: use dp_kmeans_step_u8); centers_dev inside the colour
 *                            cube as for dp_kmeans_step_u8; outputs as there (sumsq_dev may be NULL).  Each workgroup builds
 *                            its cell's candidate list from centers_dev itself: no scratch, no second launch.
 * With ranks, each rank histograms its own pixels and the totals are all-reduced per pass exactly as before. */
size_t dp_kmeans_hist_bytes(void);
size_t dp_kmeans_hist_workspace_bytes(int64_t n);
int dp_kmeans_hist_build_u8(const uint8_t *px_dev, int64_t n, void *hist_dev, int accumulate, void *workspace_dev,
                            size_t workspace_bytes, void *stream);
int dp_kmeans_hist_step(const void *hist_dev, const double *centers_dev, const double *mean_dev, int K, int64_t *sums_dev,
                        int64_t *counts_dev, int64_t *sumsq_dev, void *stream);
/* One WHOLE Lloyd iteration over the histogram in one launch, for a fit that lives on one device (nothing to all-reduce between
 * the pass and the update): dp_kmeans_hist_step into totals_dev followed by dp_kmeans_update, the update run by whichever
 * workgroup finishes the pass last (no workgroup waits for another).
 *   totals_dev  5K int64, planar as dp_kmeans_update takes it, ALL ZERO before the first iteration; the call leaves the sums
 *               and counts zero again for the next one
 *   ticket_dev  one uint32, zero before the first iteration (left zero)
 *   first       non-zero for the first iteration of a fit (it also accumulates the squared norms)
 *   centers_dev / prev_dev / status_dev / tol / max_iter   as dp_kmeans_update
 * An empty histogram (no pixels) leaves everything untouched. */
int dp_kmeans_hist_iterate(const void *hist_dev, double *centers_dev, const double *mean_dev, int K, int64_t *totals_dev,
                           int64_t *prev_dev, double *status_dev, uint32_t *ticket_dev, double tol, int max_iter, int first,
                           void *stream);
/*
 * The centre update of one Lloyd iteration ON THE DEVICE, so that a host loop can launch iterations back to back
 * (pass, all-reduce of the totals across ranks, update) and look at the status only every few iterations -- sklearn's
 * _kmeans_single_lloyd as dithering_lib.py:1854-1856 runs it: empty clusters keep their centre, stop when the
 * assignments (sums and counts) did not change or the squared centre shift is <= tol * mean(var(X)) or after
 * max_iter iterations.
 *   totals_dev   5K int64 (already summed over all ranks), planar: sums [K][3] | counts [K] | squared norms [K] --
 *                dp_kmeans_step_u8 called with sums_dev = totals_dev, counts_dev = totals_dev + 3K, sumsq_dev =
 *                totals_dev + 4K writes exactly this (the last part has to be valid in the first iteration only)
 *   centers_dev  K*3 float64, updated in place
 *   prev_dev     [K][4] int64 scratch, owned by the fit
 *   status_dev   8 float64, zeroed by the caller before the first iteration:
 *                [0] done: 0 running, 1 assignments unchanged (centres kept), 2 tolerance / max_iter reached (centres
 *                    updated; run ONE more pass + update to get their inertia: done becomes 3), 3 finished
 *                [1] iterations so far   [2] inertia of the centres the last pass used   [3] squared centre shift
 *                [4] tol * mean(var)     [5] total of the squared norms
 * A call that finds done = 1 or 3 changes nothing. */
int dp_kmeans_update(const int64_t *totals_dev, double *centers_dev, int64_t *prev_dev, double *status_dev, int K,
                     double tol, int max_iter, void *stream);

/* k-means++ seeding (sklearn.cluster._kmeans._kmeans_plusplus, reached from KMeans.fit as
 * dithering_lib.py:1854-1856 calls it: greedy, n_trials = 2 + int(ln K) local trials per centre) on a SAMPLE of the
 * pixels held on the device, one workgroup, nothing read back inside the loop.
 *   sample_dev    n packed uint8 RGB points, n <= 16384 (DP_EUNSUPPORTED above: seed on the host)
 *   first         index of the first centre (the host's RandomState.choice(n))
 *   uniforms_dev  (K-1) * n_trials float64 in [0,1): RandomState.uniform(size=n_trials) per centre, in order (they do
 *                 not depend on the data)
 *   ids_dev       out: K int32 sample indices of the centres;  centers_dev  out: K*3 float64 (the points themselves)
 * All distances and potentials are integers below 2^53, so the reference's float64 arithmetic is reproduced exactly. */
int dp_kmeans_plusplus_u8(const uint8_t *sample_dev, int n, int K, int first, const double *uniforms_dev, int n_trials,
                          int32_t *ids_dev, double *centers_dev, void *stream);

/* The distinct colours of n packed RGB pixels in order of FIRST OCCURRENCE, on the device: the device side of the reference's
 * `set(image.getdata())` (ColorReducer.reduce_colors, dithering_lib.py:1835-1843 -- a set built from all pixels is the set
 * built from the first occurrences in that order), so that only the distinct colours cross PCIe for dp_median_cut_host.
 *   out_dev         room for 3 * n bytes; the first 3 * (*n_distinct_dev) are written
 *   n_distinct_dev  one int64 on the device
 *   workspace_dev   dp_distinct_first_workspace_bytes(n) bytes (a 2^24-entry first-index table + flag words), 16-byte aligned
 * n < 2^32 - 16.  Four launches on `stream`, nothing read back. */
size_t dp_distinct_first_workspace_bytes(int64_t n);
int dp_distinct_first_u8(const uint8_t *px_dev, int64_t n, uint8_t *out_dev, int64_t *n_distinct_dev, void *workspace_dev,
                         size_t workspace_bytes, void *stream);

/* Median cut of the reference (ColorReducer.reduce_colors, dithering_lib.py:1813-1843:
 * `median_cut(list(set(image.getdata())), depth)`), host side, no GPU involved.
 *   rgb_host     n colours, 3 bytes each, in order of insertion into the reference's set (duplicates allowed: the image's
 *                pixels in raster order, or only its distinct colours in order of first occurrence -- same set)
 *   depth        int(log2(num_colors)); the palette has at most 2^depth entries (an empty bucket yields one (0,0,0))
 *   palette_out  room for 3 * 2^depth int32; *n_out receives the number of entries written
 * The stable sort of the cut makes the iteration order of the Python set observable; the call replays CPython's set
 * (tuple hash + open addressing + growth policy of CPython 3.8 ... 3.12) to obtain it.  dp_pyset_order_host alone returns that
 * order (index of the first occurrence of every distinct colour, in the order `list(set(...))` yields them). */
int dp_pyset_order_host(const uint8_t *rgb_host, int64_t n, uint32_t *order_out, int64_t *n_distinct);
int dp_median_cut_host(const uint8_t *rgb_host, int64_t n, int depth, int32_t *palette_out, int *n_out);

/* NEAREST resize of packed RGB frames (pixelize_regular / final upscale,
 * video_processor.py:563-577, 393-420), bit-identical to Pillow's Image.resize(..., NEAREST): source indices
 * come from Pillow's double-accumulated coordinate tables. */
int dp_resize_nearest_u8(const uint8_t *in_dev, uint8_t *out_dev, int64_t n_frames, int h, int w, int oh,
                         int ow, void *stream);

/* measurement -----------------------------------------------------------------------------------
 * Per-thread kernel timing with HIP events recorded on the stream the kernels run on.  While
 * enabled, every dp_ordered_u8 / dp_error_diffusion_u8 / dp_kmeans_step_u8 launch made by the calling
 * thread is bracketed by events; dp_profile_read synchronises them, returns the summed milliseconds
 * of the main kernel (pass 1 for dp_ordered_u8) and of the fix-up pass, and the number of main-kernel
 * launches, then clears the record. */
int dp_profile_enable(int on);
int dp_profile_read(double *main_ms, double *fixup_ms, int64_t *n_launches);

#ifdef __cplusplus
}
#endif
#endif /* DITHERPIE_HIP_H */
