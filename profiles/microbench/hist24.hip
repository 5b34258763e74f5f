// How fast can count[colour] over all 2^24 colours be built from packed RGB pixels on an MI355X?  (k-means over the colour
// histogram: VERDICT round 3, item 3.)  Variants:
//   A  one agent-scope no-return atomic add per pixel into a 64 MB table (index = cell-major colour)
//   B  the same with a wave-level merge of equal neighbours first (runs of one colour cost one atomic)
//   C  two levels: pixels partitioned by 16^3 cell into per-tile sorted runs, then one workgroup per cell gathers its runs
//      into an LDS histogram (not built here: see DESIGN)
// usage: hist24 [megapixels] ; content: noise / smooth+grain / flat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t colour_index(uint32_t r, uint32_t g, uint32_t b)
{
    // cell-major: (r>>4, g>>4, b>>4) then the low nibbles
    return ((r >> 4) << 20) | ((g >> 4) << 16) | ((b >> 4) << 12) | ((r & 15u) << 8) | ((g & 15u) << 4) | (b & 15u);
}

// 4 pixels (12 bytes) per lane and round
__global__ __launch_bounds__(256) void hist_a(const uint32_t *__restrict__ px, const size_t n_groups, uint32_t *__restrict__ table)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_groups; i += stride) {
        const uint32_t w0 = px[3 * i], w1 = px[3 * i + 1], w2 = px[3 * i + 2];
        const uint32_t c0 = colour_index(w0 & 255u, (w0 >> 8) & 255u, (w0 >> 16) & 255u);
        const uint32_t c1 = colour_index(w0 >> 24, w1 & 255u, (w1 >> 8) & 255u);
        const uint32_t c2 = colour_index((w1 >> 16) & 255u, w1 >> 24, w2 & 255u);
        const uint32_t c3 = colour_index((w2 >> 8) & 255u, (w2 >> 16) & 255u, w2 >> 24);
        __hip_atomic_fetch_add(&table[c0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&table[c1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&table[c2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&table[c3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// B: a lane's four pixels merged when equal; plus lanes whose first colour equals the previous lane's last are NOT merged
// (kept simple): measures what runs of one colour gain
__global__ __launch_bounds__(256) void hist_b(const uint32_t *__restrict__ px, const size_t n_groups, uint32_t *__restrict__ table)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_groups; i += stride) {
        const uint32_t w0 = px[3 * i], w1 = px[3 * i + 1], w2 = px[3 * i + 2];
        uint32_t c[4];
        c[0] = colour_index(w0 & 255u, (w0 >> 8) & 255u, (w0 >> 16) & 255u);
        c[1] = colour_index(w0 >> 24, w1 & 255u, (w1 >> 8) & 255u);
        c[2] = colour_index((w1 >> 16) & 255u, w1 >> 24, w2 & 255u);
        c[3] = colour_index((w2 >> 8) & 255u, (w2 >> 16) & 255u, w2 >> 24);
        uint32_t run = 1;
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            if (c[k] == c[k - 1]) ++run;
            else {
                __hip_atomic_fetch_add(&table[c[k - 1]], run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                run = 1;
            }
        }
        __hip_atomic_fetch_add(&table[c[3]], run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// reading the dense table once (what a Lloyd pass over the histogram has to do at least)
__global__ __launch_bounds__(256) void table_read(const uint4 *__restrict__ t, const size_t n16, uint32_t *__restrict__ sink)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const uint4 v = t[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char **argv)
{
    const size_t mp = argc > 1 ? strtoull(argv[1], nullptr, 10) : 33;
    const size_t n = (mp == 33 ? (size_t)7680 * 4320 : mp * 1000000) / 4 * 4;
    std::vector<uint8_t> h(n * 3);
    uint8_t *d;
    uint32_t *table, *sink;
    CK(hipMalloc(&d, n * 3));
    CK(hipMalloc(&table, (size_t)1 << 26));
    CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char *names[3] = {"noise", "smooth+grain", "flat"};
    for (int content = 0; content < 3; ++content) {
        uint32_t s = 12345u;
        const size_t W = 7680;
        for (size_t i = 0; i < n; ++i) {
            s = s * 1664525u + 1013904223u;
            const uint32_t rnd = s >> 8;
            const size_t x = i % W, y = i / W;
            if (content == 0) {
                h[3 * i] = rnd & 255u; h[3 * i + 1] = (rnd >> 8) & 255u; h[3 * i + 2] = (rnd >> 16) & 255u;
            } else if (content == 1) {
                const int g0 = (int)(rnd & 7u) - 3, g1 = (int)((rnd >> 3) & 7u) - 3, g2 = (int)((rnd >> 6) & 7u) - 3;
                const double r = 80 + 60 * sin(x / 1200.0) + 40 * (y / 4320.0), g = 110 + 50 * cos(y / 800.0) + 20 * sin(x / 388.0),
                             b = 160 + 70 * (y / 4320.0) + 10 * sin((x + y) / 200.0);
                auto cl = [](double v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
                h[3 * i] = cl(r + g0); h[3 * i + 1] = cl(g + g1); h[3 * i + 2] = cl(b + g2);
            } else {
                h[3 * i] = 17; h[3 * i + 1] = 99; h[3 * i + 2] = 200;
            }
        }
        CK(hipMemcpy(d, h.data(), n * 3, hipMemcpyHostToDevice));
        for (int variant = 0; variant < 2; ++variant) {
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemset(table, 0, (size_t)1 << 26));
                CK(hipDeviceSynchronize());
                float ms;
                CK(hipEventRecord(e0));
                if (variant == 0) hist_a<<<2048, 256>>>(reinterpret_cast<const uint32_t *>(d), n / 4, table);
                else hist_b<<<2048, 256>>>(reinterpret_cast<const uint32_t *>(d), n / 4, table);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                printf("%-13s variant %c: %.3f ms  (%.1f Gpx/s, %zu px)\n", names[content], 'A' + variant, ms, n / ms * 1e-6, n);
            }
        }
    }
    for (int rep = 0; rep < 3; ++rep) {
        float ms;
        CK(hipEventRecord(e0));
        table_read<<<2048, 256>>>(reinterpret_cast<const uint4 *>(table), ((size_t)1 << 26) / 16, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("dense table read (64 MB): %.4f ms\n", ms);
        CK(hipEventRecord(e0));
        CK(hipMemsetAsync(table, 0, (size_t)1 << 26));
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("table memset (64 MB): %.4f ms\n", ms);
    }
    return 0;
}
