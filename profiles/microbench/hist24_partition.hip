// Variant C of hist24.hip: count[colour] by PARTITION instead of one global atomic per pixel.
//   A  per-workgroup LDS histogram of the 4096 cell ids -> global cell counts
//   B  exclusive scan -> bucket bases
//   C  scatter: every pixel's low 12 bits (r_lo, g_lo, b_lo) as uint16 into its cell's bucket; ranks from LDS atomics, one global
//      cursor atomic per (tile, non-empty cell)
//   D  one workgroup per cell: LDS histogram of its bucket, written out as the cell's 16 KB table slice (plain stores)
// usage: hist24_partition ; contents: noise / smooth+grain / flat.  Checks the table against variant A's (atomics).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <cmath>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void split(uint32_t v, uint32_t &cell, uint32_t &lo)
{
    cell = ((v & 0xf0u) << 4) | ((v & 0xf000u) >> 8) | ((v & 0xf00000u) >> 20);   // r' << 8 | g' << 4 | b'
    lo = ((v & 0xfu) << 8) | ((v & 0xf00u) >> 4) | ((v & 0xf0000u) >> 16);        // r_lo << 8 | g_lo << 4 | b_lo
}
__device__ __forceinline__ void load4(const uint32_t *px, size_t gi, uint32_t (&v)[4])
{
    const uint32_t w0 = px[3 * gi], w1 = px[3 * gi + 1], w2 = px[3 * gi + 2];
    v[0] = w0 & 0xffffffu; v[1] = (w0 >> 24) | ((w1 & 0xffffu) << 8); v[2] = (w1 >> 16) | ((w2 & 0xffu) << 16); v[3] = w2 >> 8;
}

constexpr int TB = 1024;            // threads
constexpr int TILE = TB * 4 * 4;    // pixels per tile (4 groups of 4 per thread)

__global__ __launch_bounds__(TB) void count_kernel(const uint32_t *__restrict__ px, const size_t n_groups, uint32_t *__restrict__ cell_count)
{
    __shared__ uint32_t s_cnt[4096];
    for (int i = threadIdx.x; i < 4096; i += TB) s_cnt[i] = 0;
    __syncthreads();
    for (size_t gi = (size_t)blockIdx.x * TB + threadIdx.x; gi < n_groups; gi += (size_t)gridDim.x * TB) {
        uint32_t v[4]; load4(px, gi, v);
#pragma unroll
        for (int q = 0; q < 4; ++q) { uint32_t c, lo; split(v[q], c, lo); atomicAdd(&s_cnt[c], 1u); }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += TB) if (s_cnt[i]) atomicAdd(&cell_count[i], s_cnt[i]);
}

__global__ __launch_bounds__(1024) void scan_kernel(const uint32_t *__restrict__ cell_count, uint32_t *__restrict__ base, uint32_t *__restrict__ cursor)
{
    __shared__ uint32_t s[4096];
    for (int i = threadIdx.x; i < 4096; i += 1024) s[i] = cell_count[i];
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t acc = 0; for (int i = 0; i < 4096; ++i) { const uint32_t c = s[i]; s[i] = acc; acc += c; } }
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 1024) { base[i] = s[i]; cursor[i] = s[i]; }
}

__global__ __launch_bounds__(TB) void scatter_kernel(const uint32_t *__restrict__ px, const size_t n_groups, uint32_t *__restrict__ cursor,
                                                     uint16_t *__restrict__ buckets)
{
    __shared__ uint32_t s_cnt[4096];   // count, then the tile's base in the cell's bucket
    const size_t tiles = (n_groups + TB * 4 - 1) / (TB * 4);
    for (size_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        for (int i = threadIdx.x; i < 4096; i += TB) s_cnt[i] = 0;
        __syncthreads();
        uint32_t cell[16], lo[16], rank[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const size_t gi = tile * (TB * 4) + (size_t)j * TB + threadIdx.x;
            uint32_t v[4] = {0, 0, 0, 0};
            const bool ok = gi < n_groups;
            if (ok) load4(px, gi, v);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                split(v[q], cell[4 * j + q], lo[4 * j + q]);
                rank[4 * j + q] = ok ? atomicAdd(&s_cnt[cell[4 * j + q]], 1u) : 0xffffffffu;
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 4096; i += TB) { const uint32_t c = s_cnt[i]; if (c) s_cnt[i] = atomicAdd(&cursor[i], c); }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (rank[e] != 0xffffffffu) buckets[s_cnt[cell[e]] + rank[e]] = (uint16_t)lo[e];
        __syncthreads();
    }
}

// variant 2: big tiles (tile_groups groups of 4 pixels per workgroup), two passes over the tile's pixels (the second one hits L2):
// count per cell -> one cursor atomic per (tile, cell) -> rank by a second LDS counter + store.  Runs of tile/4096 entries per cell.
__global__ __launch_bounds__(TB) void scatter2_kernel(const uint32_t *__restrict__ px, const size_t n_groups, const size_t tile_groups,
                                                      uint32_t *__restrict__ cursor, uint16_t *__restrict__ buckets)
{
    __shared__ uint32_t s_cnt[4096];
    __shared__ uint32_t s_base[4096];
    const size_t tiles = (n_groups + tile_groups - 1) / tile_groups;
    for (size_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        for (int i = threadIdx.x; i < 4096; i += TB) s_cnt[i] = 0;
        __syncthreads();
        const size_t g0 = tile * tile_groups, g1 = g0 + tile_groups < n_groups ? g0 + tile_groups : n_groups;
        for (size_t gi = g0 + threadIdx.x; gi < g1; gi += TB) {
            uint32_t v[4]; load4(px, gi, v);
#pragma unroll
            for (int q = 0; q < 4; ++q) { uint32_t c, lo; split(v[q], c, lo); atomicAdd(&s_cnt[c], 1u); }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 4096; i += TB) { const uint32_t c = s_cnt[i]; s_base[i] = c ? atomicAdd(&cursor[i], c) : 0u; s_cnt[i] = 0; }
        __syncthreads();
        for (size_t gi = g0 + threadIdx.x; gi < g1; gi += TB) {
            uint32_t v[4]; load4(px, gi, v);
#pragma unroll
            for (int q = 0; q < 4; ++q) { uint32_t c, lo; split(v[q], c, lo); const uint32_t r = atomicAdd(&s_cnt[c], 1u); buckets[s_base[c] + r] = (uint16_t)lo; }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void cell_hist_kernel(const uint16_t *__restrict__ buckets, const uint32_t *__restrict__ base,
                                                        const uint32_t *__restrict__ cell_count, uint32_t *__restrict__ table)
{
    __shared__ uint32_t s_h[4096];
    const int cell = blockIdx.x;
    for (int i = threadIdx.x; i < 4096; i += 256) s_h[i] = 0;
    __syncthreads();
    const uint32_t b = base[cell], n = cell_count[cell];
    for (uint32_t i = threadIdx.x; i < n; i += 256) atomicAdd(&s_h[buckets[b + i]], 1u);
    __syncthreads();
    uint4 *out = reinterpret_cast<uint4 *>(table + (size_t)cell * 4096);
    for (int i = threadIdx.x; i < 1024; i += 256) out[i] = make_uint4(s_h[4 * i], s_h[4 * i + 1], s_h[4 * i + 2], s_h[4 * i + 3]);
}

__global__ __launch_bounds__(256) void hist_atomic(const uint32_t *__restrict__ px, const size_t n_groups, uint32_t *__restrict__ table)
{
    for (size_t gi = (size_t)blockIdx.x * 256 + threadIdx.x; gi < n_groups; gi += (size_t)gridDim.x * 256) {
        uint32_t v[4]; load4(px, gi, v);
#pragma unroll
        for (int q = 0; q < 4; ++q) { uint32_t c, lo; split(v[q], c, lo); __hip_atomic_fetch_add(&table[(c << 12) | lo], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
}

int main()
{
    const size_t n = (size_t)7680 * 4320;
    std::vector<uint8_t> h(n * 3);
    uint8_t *d; uint32_t *table, *ref, *cell_count, *base, *cursor; uint16_t *buckets;
    CK(hipMalloc(&d, n * 3)); CK(hipMalloc(&table, (size_t)1 << 26)); CK(hipMalloc(&ref, (size_t)1 << 26));
    CK(hipMalloc(&cell_count, 16384)); CK(hipMalloc(&base, 16384)); CK(hipMalloc(&cursor, 16384)); CK(hipMalloc(&buckets, n * 2));
    hipEvent_t e[6]; for (auto &x : e) CK(hipEventCreate(&x));
    const char *names[3] = {"noise", "smooth+grain", "flat"};
    for (int content = 0; content < 3; ++content) {
        uint32_t s = 12345u; const size_t W = 7680;
        for (size_t i = 0; i < n; ++i) {
            s = s * 1664525u + 1013904223u; const uint32_t rnd = s >> 8; const size_t x = i % W, y = i / W;
            if (content == 0) { h[3 * i] = rnd & 255u; h[3 * i + 1] = (rnd >> 8) & 255u; h[3 * i + 2] = (rnd >> 16) & 255u; }
            else if (content == 1) {
                const int g0 = (int)(rnd & 7u) - 3, g1 = (int)((rnd >> 3) & 7u) - 3, g2 = (int)((rnd >> 6) & 7u) - 3;
                const double r = 80 + 60 * sin(x / 1200.0) + 40 * (y / 4320.0), g = 110 + 50 * cos(y / 800.0) + 20 * sin(x / 388.0), b = 160 + 70 * (y / 4320.0) + 10 * sin((x + y) / 200.0);
                auto cl = [](double v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
                h[3 * i] = cl(r + g0); h[3 * i + 1] = cl(g + g1); h[3 * i + 2] = cl(b + g2);
            } else { h[3 * i] = 17; h[3 * i + 1] = 99; h[3 * i + 2] = 200; }
        }
        CK(hipMemcpy(d, h.data(), n * 3, hipMemcpyHostToDevice));
        const uint32_t *px = reinterpret_cast<const uint32_t *>(d);
        if (content != 2) { CK(hipMemset(ref, 0, (size_t)1 << 26)); hist_atomic<<<2048, 256>>>(px, n / 4, ref); CK(hipDeviceSynchronize()); }
        for (int variant = 0; variant < 4; ++variant)
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemsetAsync(cell_count, 0, 16384));
            CK(hipEventRecord(e[0]));
            count_kernel<<<512, TB>>>(px, n / 4, cell_count);
            CK(hipEventRecord(e[1]));
            scan_kernel<<<1, 1024>>>(cell_count, base, cursor);
            CK(hipEventRecord(e[2]));
            if (variant == 0) scatter_kernel<<<1024, TB>>>(px, n / 4, cursor, buckets);
            else scatter2_kernel<<<512, TB>>>(px, n / 4, (size_t)(variant == 1 ? 32768 : (variant == 2 ? 16384 : 65536)), cursor, buckets);
            CK(hipEventRecord(e[3]));
            cell_hist_kernel<<<4096, 256>>>(buckets, base, cell_count, table);
            CK(hipEventRecord(e[4]));
            CK(hipEventSynchronize(e[4]));
            float a, b, c, dd, all;
            CK(hipEventElapsedTime(&a, e[0], e[1])); CK(hipEventElapsedTime(&b, e[1], e[2])); CK(hipEventElapsedTime(&c, e[2], e[3]));
            CK(hipEventElapsedTime(&dd, e[3], e[4])); CK(hipEventElapsedTime(&all, e[0], e[4]));
            printf("%-13s scatter variant %d: count %.3f  scan %.3f  scatter %.3f  cell histograms %.3f  = %.3f ms\n", names[content], variant, a, b, c, dd, all);
        }
        if (content != 2) {
            std::vector<uint32_t> t1((size_t)1 << 24), t2((size_t)1 << 24);
            CK(hipMemcpy(t1.data(), table, (size_t)1 << 26, hipMemcpyDeviceToHost)); CK(hipMemcpy(t2.data(), ref, (size_t)1 << 26, hipMemcpyDeviceToHost));
            size_t bad = 0; for (size_t i = 0; i < t1.size(); ++i) bad += t1[i] != t2[i];
            printf("%-13s tables differ in %zu entries\n", names[content], bad);
        }
    }
    return 0;
}
