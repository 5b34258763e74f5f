// Host side of libditherpie_hip.so: error text, KD-tree construction, palette / threshold objects
// and the extern "C" entry points declared in include/ditherpie_hip.h.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>

#include "dp_internal.h"

namespace dp {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char *what)
{
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return e == hipErrorOutOfMemory ? DP_ENOMEM : DP_EHIP;
}

// (the KD-tree construction -- build_tree -- lives in host_logic.h: pure C++, also built under the CPU sanitizers)

// ---- profiling ------------------------------------------------------------------------------
static thread_local bool g_prof = false;
static thread_local std::vector<ProfMark> *g_marks = nullptr;

bool prof_on() { return g_prof; }

ProfMark *prof_begin(hipStream_t s)
{
    if (!g_prof) return nullptr;
    if (!g_marks) g_marks = new std::vector<ProfMark>();
    if (g_marks->capacity() < 65536) g_marks->reserve(65536);  // keep pointers stable
    if (g_marks->size() >= 65536) return nullptr;
    ProfMark m;
    m.n = 0;
    for (auto &e : m.ev) {
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
    }
    g_marks->push_back(m);
    ProfMark *p = &g_marks->back();
    (void)hipEventRecord(p->ev[0], s);
    p->n = 1;
    return p;
}

void prof_mid(ProfMark *m, hipStream_t s)
{
    if (!m) return;
    (void)hipEventRecord(m->ev[1], s);
    m->n = 2;
}

void prof_end(ProfMark *m, hipStream_t s)
{
    if (!m) return;
    if (m->n == 1) prof_mid(m, s);
    (void)hipEventRecord(m->ev[2], s);
    m->n = 3;
}

}  // namespace dp

using namespace dp;

// ---------------------------------------------------------------------------------------------
extern "C" {

int dp_version(void) { return DP_ABI_VERSION; }

const char *dp_last_error(void) { return g_err; }

int dp_device_info(int *n_devices, char *arch_buf, size_t arch_buf_len)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        n = 0;
        (void)hipGetLastError();
    }
    if (n_devices) *n_devices = n;
    if (arch_buf && arch_buf_len) {
        arch_buf[0] = 0;
        if (n > 0) {
            int dev = 0;
            DP_HIP(hipGetDevice(&dev));
            hipDeviceProp_t prop;
            DP_HIP(hipGetDeviceProperties(&prop, dev));
            snprintf(arch_buf, arch_buf_len, "%s", prop.gcnArchName);
        }
    }
    return DP_OK;
}

int dp_kdtree_build_host(const double *pts, int K, int32_t *indices, int32_t *split_dim, double *split,
                         int32_t *start, int32_t *end, int32_t *less, int32_t *greater, int *n_nodes)
{
    if (!pts || K < 1 || !indices || !split_dim || !split || !start || !end || !less || !greater || !n_nodes) {
        set_error("dp_kdtree_build_host: bad argument");
        return DP_EINVAL;
    }
    HostTree t;
    build_tree(pts, K, t);
    const size_t n = t.split_dim.size();
    std::memcpy(indices, t.indices.data(), sizeof(int32_t) * (size_t)K);
    std::memcpy(split_dim, t.split_dim.data(), sizeof(int32_t) * n);
    std::memcpy(split, t.split.data(), sizeof(double) * n);
    std::memcpy(start, t.start.data(), sizeof(int32_t) * n);
    std::memcpy(end, t.end.data(), sizeof(int32_t) * n);
    std::memcpy(less, t.less.data(), sizeof(int32_t) * n);
    std::memcpy(greater, t.greater.data(), sizeof(int32_t) * n);
    *n_nodes = (int)n;
    return DP_OK;
}

// ---- palette ---------------------------------------------------------------------------------
int dp_palette_create(const float *pal_f32, const uint8_t *out_colors, int K, const uint8_t *lut_in,
                      dp_palette **out)
{
    if (!pal_f32 || !out_colors || !out || K < 1) {
        set_error("dp_palette_create: bad argument");
        return DP_EINVAL;
    }
    if (K > DP_MAX_COLORS) {
        set_error("dp_palette_create: %d colours requested, this backend supports at most %d", K,
                  DP_MAX_COLORS);
        return DP_EUNSUPPORTED;
    }
    std::vector<double> pts((size_t)K * 3);
    bool integer = (lut_in == nullptr);
    for (int i = 0; i < K * 3; ++i) {
        const float v = pal_f32[i];
        if (!std::isfinite(v)) {
            set_error("dp_palette_create: non-finite palette value");
            return DP_EINVAL;
        }
        pts[i] = (double)v;
        if (!(v >= 0.0f && v <= 255.0f && v == std::floor(v))) integer = false;
    }
    HostTree t;
    build_tree(pts.data(), K, t);
    const int nn = (int)t.split_dim.size();
    int inner = 0;
    for (int i = 0; i < nn; ++i) inner += t.split_dim[i] >= 0;
    if (inner > kQueueLarge) {
        set_error("dp_palette_create: KD-tree with %d inner nodes exceeds the device traversal queue (%d)", inner,
                  kQueueLarge);
        return DP_EUNSUPPORTED;
    }

    // pack everything into one blob
    std::vector<uint8_t> blob;
    auto put = [&](const void *src, size_t bytes) {
        size_t off = (blob.size() + 15) & ~size_t(15);
        blob.resize(off + bytes);
        if (bytes) std::memcpy(blob.data() + off, src, bytes);
        return off;
    };
    std::vector<uint32_t> p4(K), orgb(K);
    std::vector<int32_t> nkey(K);
    for (int j = 0; j < K; ++j) {
        orgb[j] = (uint32_t)out_colors[3 * j] | ((uint32_t)out_colors[3 * j + 1] << 8) |
                  ((uint32_t)out_colors[3 * j + 2] << 16);
        if (integer) {
            const uint32_t r = (uint32_t)pal_f32[3 * j], g = (uint32_t)pal_f32[3 * j + 1],
                           b = (uint32_t)pal_f32[3 * j + 2];
            p4[j] = r | (g << 8) | (b << 16);
            nkey[j] = (int32_t)(((r * r + g * g + b * b) << kIdxBits) | (uint32_t)j);
        } else {
            p4[j] = 0;
            nkey[j] = 0;
        }
    }
    const size_t o_p4 = put(p4.data(), sizeof(uint32_t) * K);
    const size_t o_nk = put(nkey.data(), sizeof(int32_t) * K);
    const size_t o_pts = put(pts.data(), sizeof(double) * 3 * K);
    const size_t o_pf = put(pal_f32, sizeof(float) * 3 * K);
    const size_t o_org = put(orgb.data(), sizeof(uint32_t) * K);
    const size_t o_lut = lut_in ? put(lut_in, 256) : 0;
    std::vector<float> fcand((size_t)K * 4);
    for (int j = 0; j < K; ++j) {
        fcand[4 * j] = pal_f32[3 * j];
        fcand[4 * j + 1] = pal_f32[3 * j + 1];
        fcand[4 * j + 2] = pal_f32[3 * j + 2];
        std::memcpy(&fcand[4 * j + 3], &orgb[j], sizeof(float));
    }
    const size_t o_fc = put(fcand.data(), sizeof(float) * 4 * K);
    const size_t o_idx = put(t.indices.data(), sizeof(int32_t) * K);
    const size_t o_sd = put(t.split_dim.data(), sizeof(int32_t) * nn);
    const size_t o_sp = put(t.split.data(), sizeof(double) * nn);
    const size_t o_st = put(t.start.data(), sizeof(int32_t) * nn);
    const size_t o_en = put(t.end.data(), sizeof(int32_t) * nn);
    const size_t o_le = put(t.less.data(), sizeof(int32_t) * nn);
    const size_t o_gr = put(t.greater.data(), sizeof(int32_t) * nn);

    dp_palette *p = new (std::nothrow) dp_palette();
    if (!p) return DP_ENOMEM;
    p->blob = nullptr;
    p->blob_bytes = blob.size();
    hipError_t e = hipGetDevice(&p->device);
    if (e == hipSuccess) e = hipMalloc(&p->blob, blob.size());
    if (e == hipSuccess) e = hipMemcpy(p->blob, blob.data(), blob.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (p->blob) (void)hipFree(p->blob);
        delete p;
        return hip_fail(e, "dp_palette_create upload");
    }
    const uint8_t *base = (const uint8_t *)p->blob;
    PalDev &d = p->dev;
    d.K = K;
    d.n_nodes = nn;
    d.n_inner = inner;
    d.is_integer = integer ? 1 : 0;
    d.p4 = (const uint32_t *)(base + o_p4);
    d.nkey = (const int32_t *)(base + o_nk);
    d.pts = (const double *)(base + o_pts);
    d.pts_f32 = (const float *)(base + o_pf);
    d.out_rgb = (const uint32_t *)(base + o_org);
    d.lut_in = lut_in ? (base + o_lut) : nullptr;
    d.fcand = (const float4 *)(base + o_fc);
    d.ftab = nullptr;
    d.ftab_words = 0;
    d.ftab_total = 0;
    d.indices = (const int32_t *)(base + o_idx);
    d.split_dim = (const int32_t *)(base + o_sd);
    d.split = (const double *)(base + o_sp);
    d.start = (const int32_t *)(base + o_st);
    d.end = (const int32_t *)(base + o_en);
    d.less = (const int32_t *)(base + o_le);
    d.greater = (const int32_t *)(base + o_gr);
    for (int c = 0; c < 3; ++c) {
        d.mins[c] = t.mins[c];
        d.maxes[c] = t.maxes[c];
    }
    d.cell_tab = nullptr;
    d.tab_words = 0;
    d.tab_total = 0;
    d.cell_tab4 = nullptr;
    d.tab4_words = 0;
    d.cell_perm = d.cell_perm4 = nullptr;
    d.near_slots = 0;
    d.cell_wide = d.cell_wide4 = nullptr;
    d.n_wide = d.n_wide4 = 0;
    d.adapt = 0;
    d.warp_tab = nullptr;
    d.comp_tab = nullptr;
    d.comp_words = d.comp_warp = 0;
    d.warp_lut = nullptr;
    d.warp_words = d.warp_total = d.warp_bw = d.warp_adapt = 0;
    d.n_split = d.n_slow_blocks = 0;
    d.n_split_cells = 0;
    d.max_cell = 0;
    d.code1 = d.code2 = nullptr;
    d.exc = nullptr;
    d.n_exc = 0;
    d.ed_cells = nullptr;
    d.ed_nodes = nullptr;
    d.ed_coarse = nullptr;
    d.ed_lists16 = nullptr;
    d.ed_h4 = nullptr;
    d.ed_h4_words = 0;
    d.ed_h4_shallow = 0;
    d.ed_h4_global = 0;
    d.ed_h4_lds_words = 0;
    d.ed_coarse_ext = nullptr;
    d.ed_ext16 = nullptr;
    d.ed_ext_nodes = nullptr;
    p->ed_blob = nullptr;
    p->ext_blob = nullptr;
    p->accel_blob = nullptr;
    p->accel_bytes = 0;
    p->accel_tried = false;
    p->p4_host = p4;
    p->same_out = integer && K >= 4 && K <= DP_MAX_COLORS;  // what the accelerator handles
    for (int j = 0; j < K && p->same_out; ++j) p->same_out = (orgb[j] == p4[j]);
    // float palettes (use_gamma): coordinates within [0, 255], as the reference's clip guarantees
    p->float_accel = !integer && K >= 8 && K <= 256;
    for (int i = 0; i < 3 * K && p->float_accel; ++i) p->float_accel = pal_f32[i] >= 0.0f && pal_f32[i] <= 255.0f;
    if (p->float_accel) {
        p->pal_host.assign(pal_f32, pal_f32 + 3 * K);
        if (lut_in) p->lut_host.assign(lut_in, lut_in + 256);
    }
    p->pts_host = pts;  // for the candidate tables of the diffusion kernels, built when first needed (ensure_ed_tables)
    p->ed_tried = false;
    p->ext_tried = false;
    *out = p;
    return DP_OK;
}

// The candidate lists of the diffusion kernels (a 512 KB table from one small kernel, sharpened and padded on the host,
// plus the 16^3 tables: 3 / 5 / 19 ms at 16 / 64 / 256 colours): built once, at the first diffusion call with the palette.
static dp::PalDev snapshot(const dp_palette *pal_c)
{
    dp_palette *p = const_cast<dp_palette *>(pal_c);
    std::lock_guard<std::mutex> lock(p->dev_mu);
    return p->dev;
}

static void publish(dp_palette *p, const dp::PalDev &d)
{
    std::lock_guard<std::mutex> lock(p->dev_mu);
    p->dev = d;
}

static int ensure_ed_tables(const dp_palette *pal_c)
{
    dp_palette *p = const_cast<dp_palette *>(pal_c);
    std::lock_guard<std::mutex> lock(p->build_mu);
    if (p->ed_tried) return DP_OK;
    p->ed_tried = true;
    if (p->dev.K > 8 && p->dev.K <= DP_MAX_COLORS) {   // (257..1024 colours: ten-bit list entries, host_logic.h)
        dp::PalDev d = snapshot(p);
        const int rc = build_ed_cells(d, p->pts_host.data(), &p->ed_blob);
        if (rc != DP_OK) return rc;
        publish(p, d);
    }
    return DP_OK;
}

// the extended lists of the unclamped diffusers: only when one of them meets the palette (build_ed_ext)
static int ensure_ed_ext(const dp_palette *pal_c)
{
    dp_palette *p = const_cast<dp_palette *>(pal_c);
    std::lock_guard<std::mutex> lock(p->build_mu);
    if (p->ext_tried) return DP_OK;
    p->ext_tried = true;
    if (p->dev.K > 16) {
        dp::PalDev d = snapshot(p);
        const int rc = build_ed_ext(d, p->pts_host.data(), &p->ext_blob);
        if (rc != DP_OK) return rc;
        publish(p, d);
    }
    return DP_OK;
}

void dp_palette_destroy(dp_palette *p)
{
    if (!p) return;
    if (p->ed_blob) (void)hipFree(p->ed_blob);
    if (p->ext_blob) (void)hipFree(p->ext_blob);
    if (p->blob) (void)hipFree(p->blob);
    if (p->accel_blob) (void)hipFree(p->accel_blob);
    delete p;
}

int dp_palette_info(const dp_palette *p, int *K, int *is_integer, int *n_nodes)
{
    if (!p) {
        set_error("dp_palette_info: NULL palette");
        return DP_EINVAL;
    }
    if (K) *K = p->dev.K;
    if (is_integer) *is_integer = p->dev.is_integer;
    if (n_nodes) *n_nodes = p->dev.n_nodes;
    return DP_OK;
}

int dp_palette_build_accel(dp_palette *p)
{
    if (!p) {
        set_error("dp_palette_build_accel: NULL palette");
        return DP_EINVAL;
    }
    std::lock_guard<std::mutex> lock(p->build_mu);
    if (p->accel_tried || !(p->same_out || p->float_accel)) return DP_OK;
    p->accel_tried = true;
    // built into a private copy and published whole: a thread launching with this palette meanwhile sees either the
    // palette without the accelerator or with all of it, never a table pointer without its sizes
    dp::PalDev d = snapshot(p);
    int rc;
    if (p->float_accel)
        rc = build_accel_float(d, p->pal_host.data(), p->lut_host.empty() ? nullptr : p->lut_host.data(), &p->accel_blob,
                               &p->accel_bytes);
    else  // cell lists + tie codes: integer palettes whose output bytes are the palette colours themselves
        rc = build_accel(d, p->p4_host, &p->accel_blob, &p->accel_bytes);
    if (rc == DP_OK) publish(p, d);
    return rc;
}

int dp_palette_accel_info(const dp_palette *p, int *pool_entries, int *max_cell)
{
    if (!p) {
        set_error("dp_palette_accel_info: NULL palette");
        return DP_EINVAL;
    }
    const dp::PalDev d = snapshot(p);
    if (pool_entries) *pool_entries = d.cell_tab ? d.tab_words : (d.cell_tab4 ? d.tab4_words : (d.ftab ? d.ftab_words : 0));
    if (max_cell) *max_cell = (d.cell_tab || d.cell_tab4 || d.ftab) ? d.max_cell : 0;
    return DP_OK;
}

// ---- thresholds --------------------------------------------------------------------------------
static int thresholds_from_device_f32(float *dev_f32, int th_h, int th_w, const float *host_copy,
                                      dp_thresholds **out)
{
    // integer form t = m / 2^sh with sh <= 13 and 0 <= t <= 1 lets the kernel decide in uint32
    const int n = th_h * th_w;
    int sh = -1;
    std::vector<uint32_t> m;
    for (int s = 0; s <= 13 && sh < 0; ++s) {
        bool ok = true;
        for (int i = 0; i < n && ok; ++i) {
            const double v = (double)host_copy[i] * (double)(1u << s);
            ok = host_copy[i] >= 0.0f && host_copy[i] <= 1.0f && v == std::floor(v);
        }
        if (ok) sh = s;
    }
    dp_thresholds *t = new (std::nothrow) dp_thresholds();
    if (!t) return DP_ENOMEM;
    t->blob = dev_f32;
    (void)hipGetDevice(&t->device);
    t->dev.th_h = th_h;
    t->dev.th_w = th_w;
    t->dev.f32 = dev_f32;
    t->dev.m = nullptr;
    t->dev.sh = 0;
    t->dev.fpad = nullptr;
    t->dev.mpad = nullptr;
    for (uint32_t &wd : t->dev.cls_nib) wd = 0;
    t->dev.has_cls = 0;
    t->dev.tw_pad = 0;
    t->blob_pad = nullptr;
    if (sh >= 0) {
        m.resize(n);
        for (int i = 0; i < n; ++i) m[i] = (uint32_t)((double)host_copy[i] * (double)(1u << sh));
        // the integer table lives behind the f32 table in the same allocation
        uint32_t *dm = (uint32_t *)(dev_f32 + n);
        hipError_t e = hipMemcpy(dm, m.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            delete t;
            return hip_fail(e, "thresholds integer upload");
        }
        t->dev.m = dm;
        t->dev.sh = sh;
    }
    t->dev.pow2 = ((th_h & (th_h - 1)) == 0 && (th_w & (th_w - 1)) == 0) ? 1 : 0;
    t->dev.inv_h = 1.0 / (double)th_h;
    t->dev.inv_w = 1.0 / (double)th_w;
    if ((int64_t)th_h * (th_w + 3) <= (1 << 18)) {
        const int twp = th_w + 3;
        const size_t np = (size_t)th_h * twp;
        std::vector<uint32_t> pad(2 * np);  // float32 bit patterns, then the integer form
        for (int y = 0; y < th_h; ++y)
            for (int x = 0; x < twp; ++x) {
                const size_t src = (size_t)y * th_w + (x % th_w);
                std::memcpy(&pad[(size_t)y * twp + x], &host_copy[src], sizeof(float));
                pad[np + (size_t)y * twp + x] = sh >= 0 ? m[src] : 0u;
            }
        // classes (ordered kernels): a wave covers 256 consecutive pixels of a row, lane l the pixels 4l .. 4l+3.
        // nibble (r, p) bit q: with lane 0 at column p of threshold row r, no lane's pixel q has a threshold below 1/2 --
        // such pixels always take the nearest entry (dithering_lib.py:361-376: factor <= 1/2).
        if (n <= 256 && t->dev.pow2) {
            bool any = false;
            for (int r = 0; r < th_h; ++r)
                for (int p0 = 0; p0 < th_w; ++p0) {
                    uint32_t bits = 0;
                    for (int q = 0; q < 4; ++q) {
                        bool all = true;
                        for (int l = 0; l < 64 && all; ++l) all = host_copy[(size_t)r * th_w + (size_t)((p0 + 4 * l + q) % th_w)] >= 0.5f;
                        bits |= all ? (1u << q) : 0u;
                    }
                    const int idx = r * th_w + p0;
                    t->dev.cls_nib[idx >> 3] |= bits << ((idx & 7) * 4);
                    any |= bits != 0;
                }
            t->dev.has_cls = any ? 1 : 0;
        }
        void *dp = nullptr;
        hipError_t e = hipMalloc(&dp, sizeof(uint32_t) * pad.size());
        if (e == hipSuccess) e = hipMemcpy(dp, pad.data(), sizeof(uint32_t) * pad.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            if (dp) (void)hipFree(dp);
            delete t;
            return hip_fail(e, "thresholds padded upload");
        }
        t->blob_pad = dp;
        t->dev.fpad = (const float *)dp;
        t->dev.mpad = sh >= 0 ? (const uint32_t *)dp + np : nullptr;
        t->dev.tw_pad = twp;
    }
    *out = t;
    return DP_OK;
}

int dp_thresholds_create(const float *thr_host, int th_h, int th_w, dp_thresholds **out)
{
    if (!thr_host || !out || th_h < 1 || th_w < 1 || (int64_t)th_h * th_w > (1 << 20)) {
        set_error("dp_thresholds_create: bad argument");
        return DP_EINVAL;
    }
    const int n = th_h * th_w;
    for (int i = 0; i < n; ++i)
        if (std::isnan(thr_host[i])) {
            set_error("dp_thresholds_create: NaN threshold");
            return DP_EINVAL;
        }
    float *d = nullptr;
    DP_HIP(hipMalloc((void **)&d, sizeof(float) * 2 * (size_t)n));
    hipError_t e = hipMemcpy(d, thr_host, sizeof(float) * n, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(d);
        return hip_fail(e, "dp_thresholds_create upload");
    }
    int rc = thresholds_from_device_f32(d, th_h, th_w, thr_host, out);
    if (rc != DP_OK) (void)hipFree(d);
    return rc;
}

int dp_thresholds_blue_noise(int size, uint32_t seed, void *stream, dp_thresholds **out)
{
    if (!out || size < 2 || size > 256) {
        set_error("dp_thresholds_blue_noise: size must be in [2,256]");
        return DP_EINVAL;
    }
    const int n = size * size;
    float *d = nullptr;
    void *scratch = nullptr;
    DP_HIP(hipMalloc((void **)&d, sizeof(float) * 2 * (size_t)n));
    hipError_t e = hipMalloc(&scratch, blue_noise_scratch_bytes(size));
    if (e != hipSuccess) {
        (void)hipFree(d);
        return hip_fail(e, "blue-noise scratch");
    }
    hipStream_t s = (hipStream_t)stream;
    int rc = launch_blue_noise(size, seed, d, scratch, s);
    std::vector<float> host(n);
    if (rc == DP_OK) {
        e = hipMemcpyAsync(host.data(), d, sizeof(float) * n, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) rc = hip_fail(e, "blue-noise download");
    }
    (void)hipFree(scratch);
    if (rc == DP_OK) rc = thresholds_from_device_f32(d, size, size, host.data(), out);
    if (rc != DP_OK) (void)hipFree(d);
    return rc;
}

int dp_thresholds_shape(const dp_thresholds *t, int *th_h, int *th_w, int *is_integer_form)
{
    if (!t) {
        set_error("dp_thresholds_shape: NULL");
        return DP_EINVAL;
    }
    if (th_h) *th_h = t->dev.th_h;
    if (th_w) *th_w = t->dev.th_w;
    if (is_integer_form) *is_integer_form = t->dev.m != nullptr;
    return DP_OK;
}

int dp_thresholds_download(const dp_thresholds *t, float *thr_host)
{
    if (!t || !thr_host) {
        set_error("dp_thresholds_download: NULL");
        return DP_EINVAL;
    }
    DP_HIP(hipMemcpy(thr_host, t->dev.f32, sizeof(float) * (size_t)t->dev.th_h * t->dev.th_w,
                     hipMemcpyDeviceToHost));
    return DP_OK;
}

void dp_thresholds_destroy(dp_thresholds *t)
{
    if (!t) return;
    if (t->blob) (void)hipFree(t->blob);
    if (t->blob_pad) (void)hipFree(t->blob_pad);
    delete t;
}

// ---- compute entry points (argument checks here, kernels in the .hip files) ------------------------
int dp_ign_thresholds(float *out_dev, int h, int w, int y0, int x0, float scale, int seed, void *stream)
{
    if (!out_dev || h < 1 || w < 1) {
        set_error("dp_ign_thresholds: bad argument");
        return DP_EINVAL;
    }
    return launch_ign_thresholds(out_dev, h, w, y0, x0, scale, seed, (hipStream_t)stream);
}

size_t dp_ordered_workspace_bytes(int64_t n_frames, int h, int w)
{
    if (n_frames < 0 || h < 0 || w < 0) return 0;
    // one flag bit per pixel, written as 4 x u64 per 256-pixel wave tile (+ slack for the tail)
    const int64_t npx = n_frames * (int64_t)h * w;
    const int64_t tiles = (npx + 255) / 256;
    return (size_t)(tiles * 32 + 1024 + 4 * ((size_t)dp::kQueueTiles + 8));  // + dirty count and tile queue (ordered.hip)
}

int dp_ordered_u8(const uint8_t *in_dev, uint8_t *out_dev, int64_t n_frames, int h, int w, int y0, int x0,
                  const dp_palette *pal, int mode, const dp_thresholds *thr, float ign_scale, int ign_seed,
                  void *workspace_dev, size_t workspace_bytes, void *stream)
{
    if (n_frames == 0 && pal && h >= 1 && w >= 1) return DP_OK;  // nothing to do (pointers may be null)
    if (!in_dev || !out_dev || !pal || n_frames < 0 || h < 1 || w < 1 || y0 < 0 || x0 < 0) {
        set_error("dp_ordered_u8: bad argument");
        return DP_EINVAL;
    }
    if (mode != DP_MODE_NEAREST && mode != DP_MODE_MATRIX && mode != DP_MODE_IGN) {
        set_error("dp_ordered_u8: unknown mode %d", mode);
        return DP_EINVAL;
    }
    if (mode == DP_MODE_MATRIX && !thr) {
        set_error("dp_ordered_u8: DP_MODE_MATRIX needs a threshold matrix");
        return DP_EINVAL;
    }
    if ((int64_t)h * w > (int64_t)1 << 30 || (int64_t)y0 + h > (int64_t)1 << 30 || (int64_t)x0 + w > (int64_t)1 << 30) {
        set_error("dp_ordered_u8: frame too large");
        return DP_EINVAL;
    }
    if (n_frames == 0) return DP_OK;
    if (!workspace_dev || workspace_bytes < dp_ordered_workspace_bytes(n_frames, h, w) ||
        ((uintptr_t)workspace_dev & 15)) {
        set_error("dp_ordered_u8: workspace too small or misaligned (need %zu bytes, 16-byte aligned)",
                  dp_ordered_workspace_bytes(n_frames, h, w));
        return DP_EWORKSPACE;
    }
    return launch_ordered(in_dev, out_dev, n_frames, h, w, y0, x0, snapshot(pal), mode, thr ? &thr->dev : nullptr,
                          ign_scale, ign_seed, workspace_dev, workspace_bytes, (hipStream_t)stream);
}

size_t dp_error_diffusion_workspace_bytes(int64_t n_frames, int h, int w)
{
    if (n_frames < 0 || h < 0 || w < 0) return 0;
    return error_diffusion_ws_bytes(n_frames, h, w);
}

static int error_diffusion_common(const uint8_t *in_dev, uint8_t *out_dev, int64_t n_frames, int h, int w,
                                  const dp_palette *pal, const int32_t *dx, const int32_t *dy, const float *wq,
                                  const double *wq64, int ntaps, int serpentine, void *workspace_dev, size_t workspace_bytes,
                                  void *stream, const double *hybrid = nullptr);

int dp_error_diffusion_u8(const uint8_t *in_dev, uint8_t *out_dev, int64_t n_frames, int h, int w,
                          const dp_palette *pal, const int32_t *dx, const int32_t *dy, const float *wq,
                          int ntaps, int serpentine, void *workspace_dev, size_t workspace_bytes,
                          void *stream)
{
    return error_diffusion_common(in_dev, out_dev, n_frames, h, w, pal, dx, dy, wq, nullptr, ntaps, serpentine, workspace_dev,
                                  workspace_bytes, stream);
}

int dp_error_diffusion_numba_u8(const uint8_t *in_dev, uint8_t *out_dev, int64_t n_frames, int h, int w,
                                const dp_palette *pal, const int32_t *dx, const int32_t *dy, const float *weights,
                                double divisor, int ntaps, int serpentine, void *workspace_dev, size_t workspace_bytes,
                                void *stream)
{
    if (ntaps < 0 || ntaps > 16 || (ntaps && !weights) || !(divisor > 0.0)) {
        set_error("dp_error_diffusion_numba_u8: bad argument");
        return DP_EINVAL;
    }
    float wq[16];
    double wq64[16];
    for (int k = 0; k < ntaps; ++k) {
        wq64[k] = (double)weights[k] / divisor;  // numba: float32 weight / float64 divisor
        wq[k] = (float)wq64[k];
    }
    return error_diffusion_common(in_dev, out_dev, n_frames, h, w, pal, dx, dy, wq, wq64, ntaps, serpentine, workspace_dev,
                                  workspace_bytes, stream);
}

int dp_hybrid_numba_u8(const uint8_t *in_dev, uint8_t *out_dev, int64_t n_frames, int h, int w, const dp_palette *pal,
                       double lum_factor, double col_factor, void *workspace_dev, size_t workspace_bytes, void *stream)
{
    // _hybrid_numba's Floyd-Steinberg pushes (dithering_lib.py:1453-1468): (x+1, y) 7/16, (x-1, y+1) 3/16, (x, y+1) 5/16, (x+1, y+1) 1/16,
    // the weights float64 constants
    static const int32_t dx[4] = {1, -1, 0, 1}, dy[4] = {0, 1, 1, 1};
    static const double wq64[4] = {7.0 / 16.0, 3.0 / 16.0, 5.0 / 16.0, 1.0 / 16.0};
    static const float wq[4] = {7.0f / 16.0f, 3.0f / 16.0f, 5.0f / 16.0f, 1.0f / 16.0f};
    const double hybrid[2] = {lum_factor, col_factor};
    return error_diffusion_common(in_dev, out_dev, n_frames, h, w, pal, dx, dy, wq, wq64, 4, 0, workspace_dev, workspace_bytes, stream,
                                  hybrid);
}

static int error_diffusion_common(const uint8_t *in_dev, uint8_t *out_dev, int64_t n_frames, int h, int w,
                                  const dp_palette *pal, const int32_t *dx, const int32_t *dy, const float *wq,
                                  const double *wq64, int ntaps, int serpentine, void *workspace_dev, size_t workspace_bytes,
                                  void *stream, const double *hybrid)
{
    if (n_frames == 0 && pal && h >= 1 && w >= 1) return DP_OK;  // nothing to do (pointers may be null)
    if (!in_dev || !out_dev || !pal || n_frames < 0 || h < 1 || w < 1 || ntaps < 0 || ntaps > 16 ||
        (ntaps && (!dx || !dy || !wq))) {
        set_error("dp_error_diffusion_u8: bad argument");
        return DP_EINVAL;
    }
    for (int k = 0; k < ntaps; ++k) {
        const bool forward = dy[k] > 0 || (dy[k] == 0 && dx[k] > 0);
        if (!forward || dy[k] > 2 || dx[k] < -2 || dx[k] > 2) {
            set_error("dp_error_diffusion_u8: tap %d (dx=%d, dy=%d) is outside the supported causal window", k,
                      dx[k], dy[k]);
            return DP_EUNSUPPORTED;
        }
    }
    if (n_frames == 0) return DP_OK;
    if (!workspace_dev || workspace_bytes < dp_error_diffusion_workspace_bytes(n_frames, h, w)) {
        set_error("dp_error_diffusion_u8: workspace too small (need %zu bytes)",
                  dp_error_diffusion_workspace_bytes(n_frames, h, w));
        return DP_EWORKSPACE;
    }
    const int rc_tab = ensure_ed_tables(pal);
    if (rc_tab != DP_OK) return rc_tab;
    return launch_error_diffusion(in_dev, out_dev, n_frames, h, w, snapshot(pal), dx, dy, wq, ntaps, serpentine,
                                  workspace_dev, workspace_bytes, (hipStream_t)stream, wq64, hybrid);
}

int dp_variable_diffusion_u8(const uint8_t *in_dev, uint8_t *out_dev, int64_t n_frames, int h, int w,
                             const dp_palette *pal, int model, float p0, float p1, int serpentine,
                             const uint8_t *gate_dev, const float *coef_dev, void *workspace_dev,
                             size_t workspace_bytes, void *stream)
{
    if (n_frames == 0 && pal && h >= 1 && w >= 1) return DP_OK;
    if (!in_dev || !out_dev || !pal || n_frames < 0 || h < 1 || w < 1 || model < 1 || model > 4 ||
        (model == DP_DIFFUSER_ADAPTIVE_VARIANCE && !gate_dev) || (model == DP_DIFFUSER_OSTROMOUKHOV && !coef_dev)) {
        set_error("dp_variable_diffusion_u8: bad argument");
        return DP_EINVAL;
    }
    if (!workspace_dev || workspace_bytes < dp_error_diffusion_workspace_bytes(n_frames, h, w)) {
        set_error("dp_variable_diffusion_u8: workspace too small (need %zu bytes)",
                  dp_error_diffusion_workspace_bytes(n_frames, h, w));
        return DP_EWORKSPACE;
    }
    const int rc_tab = ensure_ed_tables(pal);
    if (rc_tab != DP_OK) return rc_tab;
    if (model != DP_DIFFUSER_OSTROMOUKHOV) {   // (Ostromoukhov clamps its values: the plain lists)
        const int rc_ext = ensure_ed_ext(pal);
        if (rc_ext != DP_OK) return rc_ext;
    }
    return launch_variable_diffusion(in_dev, out_dev, n_frames, h, w, snapshot(pal), model, p0, p1, serpentine ? 1 : 0,
                                     gate_dev, coef_dev, workspace_dev, (hipStream_t)stream);
}

size_t dp_variance_gate_workspace_bytes(int64_t n_frames, int h, int w)
{
    if (n_frames < 0 || h < 0 || w < 0) return 0;
    return variance_gate_ws_bytes(n_frames, h, w);
}

int dp_variance_gate_u8(const uint8_t *in_dev, uint8_t *gate_dev, int64_t n_frames, int h, int w, const dp_palette *pal,
                        float var_threshold, int window_radius, void *workspace_dev, size_t workspace_bytes,
                        void *stream)
{
    if (n_frames == 0 && pal && h >= 1 && w >= 1) return DP_OK;
    if (!in_dev || !gate_dev || !pal || n_frames < 0 || h < 1 || w < 1 || window_radius < 0 || window_radius > 64) {
        set_error("dp_variance_gate_u8: bad argument");
        return DP_EINVAL;
    }
    if (!workspace_dev || workspace_bytes < variance_gate_ws_bytes(n_frames, h, w)) {
        set_error("dp_variance_gate_u8: workspace too small (need %zu bytes)", variance_gate_ws_bytes(n_frames, h, w));
        return DP_EWORKSPACE;
    }
    return launch_variance_gate(in_dev, gate_dev, n_frames, h, w, snapshot(pal), var_threshold, window_radius, workspace_dev,
                                (hipStream_t)stream);
}

int dp_kmeans_step_u8(const uint8_t *px_dev, int64_t n, const double *centers_dev, const double *mean_dev, int K,
                      int64_t *sums_dev, int64_t *counts_dev, int64_t *sumsq_dev, void *stream)
{
    if ((!px_dev && n > 0) || n < 0 || !centers_dev || K < 1 || K > 1024 || !sums_dev || !counts_dev) {
        set_error("dp_kmeans_step_u8: bad argument");
        return DP_EINVAL;
    }
    return launch_kmeans_step(px_dev, n, centers_dev, mean_dev, K, sums_dev, counts_dev, sumsq_dev, (hipStream_t)stream);
}

size_t dp_kmeans_hist_bytes(void) { return kmeans_hist_bytes(); }

size_t dp_kmeans_hist_workspace_bytes(int64_t n) { return n < 0 ? 0 : kmeans_hist_ws_bytes(n); }

int dp_kmeans_hist_build_u8(const uint8_t *px_dev, int64_t n, void *hist_dev, int accumulate, void *workspace_dev,
                            size_t workspace_bytes, void *stream)
{
    if ((!px_dev && n > 0) || n < 0 || n > (int64_t)0xfffffff0LL || !hist_dev || ((uintptr_t)hist_dev & 15)) {
        set_error("dp_kmeans_hist_build_u8: bad argument (n must be below 2^32 - 16, hist_dev 16-byte aligned)");
        return DP_EINVAL;
    }
    if (!workspace_dev || ((uintptr_t)workspace_dev & 15) || workspace_bytes < kmeans_hist_ws_bytes(n)) {
        set_error("dp_kmeans_hist_build_u8: workspace too small or not 16-byte aligned (need %zu bytes)", kmeans_hist_ws_bytes(n));
        return DP_EWORKSPACE;
    }
    return launch_kmeans_hist_build(px_dev, n, hist_dev, accumulate ? 1 : 0, workspace_dev, (hipStream_t)stream);
}

int dp_kmeans_hist_step(const void *hist_dev, const double *centers_dev, const double *mean_dev, int K, int64_t *sums_dev,
                        int64_t *counts_dev, int64_t *sumsq_dev, void *stream)
{
    if (!hist_dev || !centers_dev || K < 1 || !sums_dev || !counts_dev) {
        set_error("dp_kmeans_hist_step: bad argument");
        return DP_EINVAL;
    }
    return launch_kmeans_hist_step(hist_dev, centers_dev, mean_dev, K, sums_dev, counts_dev, sumsq_dev, (hipStream_t)stream);
}

int dp_kmeans_hist_iterate(const void *hist_dev, double *centers_dev, const double *mean_dev, int K, int64_t *totals_dev,
                           int64_t *prev_dev, double *status_dev, uint32_t *ticket_dev, double tol, int max_iter, int first,
                           void *stream)
{
    if (!hist_dev || !centers_dev || !totals_dev || !prev_dev || !status_dev || !ticket_dev || K < 1 || max_iter < 1 || !(tol >= 0.0)) {
        set_error("dp_kmeans_hist_iterate: bad argument");
        return DP_EINVAL;
    }
    return launch_kmeans_hist_iterate(hist_dev, centers_dev, mean_dev, K, totals_dev, prev_dev, status_dev, ticket_dev, tol, max_iter,
                                      first ? 1 : 0, (hipStream_t)stream);
}

int dp_kmeans_update(const int64_t *totals_dev, double *centers_dev, int64_t *prev_dev, double *status_dev, int K,
                     double tol, int max_iter, void *stream)
{
    if (!totals_dev || !centers_dev || !prev_dev || !status_dev || K < 1 || K > 1024 || max_iter < 1 || !(tol >= 0.0)) {
        set_error("dp_kmeans_update: bad argument");
        return DP_EINVAL;
    }
    return launch_kmeans_update(totals_dev, centers_dev, prev_dev, status_dev, K, tol, max_iter, (hipStream_t)stream);
}

int dp_kmeans_plusplus_u8(const uint8_t *sample_dev, int n, int K, int first, const double *uniforms_dev, int n_trials,
                          int32_t *ids_dev, double *centers_dev, void *stream)
{
    if (!sample_dev || !uniforms_dev || !ids_dev || !centers_dev || n < 1 || K < 1 || K > n || first < 0 || first >= n ||
        n_trials < 1) {
        set_error("dp_kmeans_plusplus_u8: bad argument");
        return DP_EINVAL;
    }
    return launch_kmeans_pp(sample_dev, n, K, first, uniforms_dev, n_trials, ids_dev, centers_dev, (hipStream_t)stream);
}

size_t dp_distinct_first_workspace_bytes(int64_t n) { return n < 0 ? 0 : distinct_first_ws_bytes(n); }

int dp_distinct_first_u8(const uint8_t *px_dev, int64_t n, uint8_t *out_dev, int64_t *n_distinct_dev, void *workspace_dev,
                         size_t workspace_bytes, void *stream)
{
    if ((!px_dev && n > 0) || n < 0 || n > (int64_t)0xfffffff0LL || (!out_dev && n > 0) || !n_distinct_dev) {
        set_error("dp_distinct_first_u8: bad argument (n must be below 2^32 - 16)");
        return DP_EINVAL;
    }
    if (n > 0 && (!workspace_dev || ((uintptr_t)workspace_dev & 15) || workspace_bytes < distinct_first_ws_bytes(n))) {
        set_error("dp_distinct_first_u8: workspace too small or not 16-byte aligned (need %zu bytes)", distinct_first_ws_bytes(n));
        return DP_EWORKSPACE;
    }
    return launch_distinct_first(px_dev, n, out_dev, reinterpret_cast<long long *>(n_distinct_dev), workspace_dev, (hipStream_t)stream);
}

int dp_pyset_order_host(const uint8_t *rgb_host, int64_t n, uint32_t *order_out, int64_t *n_distinct)
{
    if (!rgb_host || n < 0 || n > ((int64_t)1 << 31) - 2 || !order_out || !n_distinct) {
        set_error("dp_pyset_order_host: bad argument");
        return DP_EINVAL;
    }
    try {   // (no exception may cross the C ABI: ctypes would see std::terminate)
        std::vector<uint32_t> order;
        pyset_order(rgb_host, (size_t)n, order);
        std::copy(order.begin(), order.end(), order_out);
        *n_distinct = (int64_t)order.size();
    } catch (const std::exception &e) {
        set_error("dp_pyset_order_host: %s", e.what());
        return DP_ENOMEM;
    }
    return DP_OK;
}

int dp_median_cut_host(const uint8_t *rgb_host, int64_t n, int depth, int32_t *palette_out, int *n_out)
{
    if (!rgb_host || n < 0 || n > ((int64_t)1 << 31) - 2 || depth < 0 || depth > 10 || !palette_out || !n_out) {
        set_error("dp_median_cut_host: bad argument");
        return DP_EINVAL;
    }
    try {   // (bad_alloc out of a vector, also from the cut's helper threads -- carried across their join: host_logic.h)
        std::vector<uint32_t> order;
        pyset_order(rgb_host, (size_t)n, order);
        std::vector<uint32_t> colours(order.size() + 1), scratch(order.size() + 1);
        for (size_t i = 0; i < order.size(); ++i) {
            const uint8_t *c = rgb_host + 3 * (size_t)order[i];
            colours[i] = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16);
        }
        std::vector<int32_t> out;
        median_cut_u32(colours.data(), scratch.data(), order.size(), depth, out, (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency())));
        std::copy(out.begin(), out.end(), palette_out);
        *n_out = (int)(out.size() / 3);
    } catch (const std::exception &e) {
        set_error("dp_median_cut_host: %s", e.what());
        return DP_ENOMEM;
    }
    return DP_OK;
}

int dp_resize_nearest_u8(const uint8_t *in_dev, uint8_t *out_dev, int64_t n_frames, int h, int w, int oh,
                         int ow, void *stream)
{
    if (n_frames == 0 && h >= 1 && w >= 1 && oh >= 1 && ow >= 1) return DP_OK;
    if (!in_dev || !out_dev || n_frames < 0 || h < 1 || w < 1 || oh < 1 || ow < 1) {
        set_error("dp_resize_nearest_u8: bad argument");
        return DP_EINVAL;
    }
    if (n_frames == 0) return DP_OK;
    return launch_resize_nearest(in_dev, out_dev, n_frames, h, w, oh, ow, (hipStream_t)stream);
}

int dp_profile_enable(int on)
{
    g_prof = on != 0;
    return DP_OK;
}

int dp_profile_read(double *main_ms, double *fixup_ms, int64_t *n_launches)
{
    double a = 0, b = 0;
    int64_t n = 0;
    if (g_marks) {
        for (auto &m : *g_marks) {
            if (m.n == 3) {
                float t1 = 0, t2 = 0;
                DP_HIP(hipEventSynchronize(m.ev[2]));
                DP_HIP(hipEventElapsedTime(&t1, m.ev[0], m.ev[1]));
                DP_HIP(hipEventElapsedTime(&t2, m.ev[1], m.ev[2]));
                a += t1;
                b += t2;
                ++n;
            }
            for (auto &e : m.ev) (void)hipEventDestroy(e);
        }
        g_marks->clear();
    }
    if (main_ms) *main_ms = a;
    if (fixup_ms) *fixup_ms = b;
    if (n_launches) *n_launches = n;
    return DP_OK;
}

}  // extern "C"
