// Feasibility microbenchmark: ordered dithering as one 4-byte gather per pixel from a 2^24-entry table
// (nearest | second << 8 | critical threshold << 16), on white-noise frames vs. smooth frames with grain.
// build: hipcc --offload-arch=gfx950 -O3 -o lut_gather lut_gather.hip ; run: ./lut_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

__global__ void fill_kernel(uint8_t *f, int n_frames, int h, int w, int mode, int grain)
{
    const size_t n = (size_t)n_frames * h * w;
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
        const int x = p % w, y = (p / w) % h, fr = p / ((size_t)w * h);
        uint32_t r, g, b;
        const uint32_t hh = hash32((uint32_t)p * 3u + 12345u);
        if (mode == 0) { r = hh & 255; g = (hh >> 8) & 255; b = (hh >> 16) & 255; }
        else {
            const float u = x / (float)w, v = y / (float)h, t = fr * 0.13f;
            float fr_ = 128 + 100 * __sinf(6.3f * (u * 1.3f + t)) * __cosf(4.1f * v);
            float fg_ = 128 + 100 * __sinf(5.1f * (v * 1.7f - t) + 2 * u);
            float fb_ = 128 + 100 * __cosf(3.3f * (u + v) + t);
            int n0 = (int)(hh % (2 * grain + 1)) - grain, n1 = (int)((hh >> 8) % (2 * grain + 1)) - grain, n2 = (int)((hh >> 16) % (2 * grain + 1)) - grain;
            r = min(255, max(0, (int)fr_ + n0)); g = min(255, max(0, (int)fg_ + n1)); b = min(255, max(0, (int)fb_ + n2));
        }
        f[p * 3] = r; f[p * 3 + 1] = g; f[p * 3 + 2] = b;
    }
}

template <int LAYOUT>
__device__ __forceinline__ uint32_t lut_index(uint32_t x)
{
    if (LAYOUT == 0) return x;  // r fastest
    // 4x4x4 colour cubes contiguous (256 B)
    const uint32_t lo = (x & 3u) | ((x >> 6) & 0xcu) | ((x >> 12) & 0x30u);
    const uint32_t hi = ((x >> 2) & 0x3fu) | ((x >> 4) & 0xfc0u) | ((x >> 6) & 0x3f000u);
    return (hi << 6) | lo;
}

template <int LAYOUT>
__global__ __launch_bounds__(256) void lut_dither_kernel(const uint3 *__restrict__ in, uint3 *__restrict__ out,
                                                         const uint32_t *__restrict__ lut, const uint32_t *__restrict__ pal,
                                                         const uint16_t *__restrict__ thr, uint32_t n_groups, int w, int h)
{
    __shared__ uint32_t s_pal[256];
    __shared__ uint16_t s_thr[64];
    s_pal[threadIdx.x] = pal[threadIdx.x];
    if (threadIdx.x < 64) s_thr[threadIdx.x] = thr[threadIdx.x];
    __syncthreads();
    for (uint32_t gidx = blockIdx.x * 256u + threadIdx.x; gidx < n_groups; gidx += gridDim.x * 256u) {
        const uint3 wc = in[gidx];
        uint32_t xq[4];
        xq[0] = wc.x & 0xffffffu;
        xq[1] = __builtin_amdgcn_perm(wc.y, wc.x, 0x0c050403u);
        xq[2] = __builtin_amdgcn_perm(wc.z, wc.y, 0x0c040302u);
        xq[3] = wc.z >> 8;
        uint32_t e[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) e[q] = lut[lut_index<LAYOUT>(xq[q])];
        const uint32_t p = gidx * 4u;
        const uint32_t px = p % (uint32_t)w, py = (p / (uint32_t)w) % (uint32_t)h;
        const uint2 t4 = *reinterpret_cast<const uint2 *>(&s_thr[(py & 7u) * 8u + (px & 4u)]);
        uint32_t col[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t tw = (q & 2) ? t4.y : t4.x;
            const uint32_t mt = (q & 1) ? (tw >> 16) : (tw & 0xffffu);
            const uint32_t idx = (mt >= (e[q] >> 16)) ? (e[q] & 255u) : ((e[q] >> 8) & 255u);
            col[q] = s_pal[idx];
        }
        uint3 wo;
        wo.x = __builtin_amdgcn_perm(col[1], col[0], 0x04020100u);
        wo.y = __builtin_amdgcn_perm(col[2], col[1], 0x05040201u);
        wo.z = __builtin_amdgcn_perm(col[3], col[2], 0x06050402u);
        out[gidx] = wo;
    }
}

int main()
{
    const int F = 24, H = 2160, W = 3840;
    const size_t npx = (size_t)F * H * W;
    uint8_t *in, *out; uint32_t *lut, *pal; uint16_t *thr;
    CK(hipMalloc(&in, npx * 3)); CK(hipMalloc(&out, npx * 3)); CK(hipMalloc(&lut, sizeof(uint32_t) << 24));
    CK(hipMalloc(&pal, 1024)); CK(hipMalloc(&thr, 128));
    std::vector<uint32_t> hl(1u << 24);
    uint32_t s = 1;
    for (auto &v : hl) { s = s * 1664525u + 1013904223u; v = (s >> 8) & 0x3fffffu; }
    CK(hipMemcpy(lut, hl.data(), sizeof(uint32_t) << 24, hipMemcpyHostToDevice));
    std::vector<uint32_t> hp(256); for (int i = 0; i < 256; ++i) hp[i] = (i * 0x010307u) & 0xffffffu;
    CK(hipMemcpy(pal, hp.data(), 1024, hipMemcpyHostToDevice));
    std::vector<uint16_t> ht(64); for (int i = 0; i < 64; ++i) ht[i] = (uint16_t)((i * 37) & 63);
    CK(hipMemcpy(thr, ht.data(), 128, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t n_groups = (uint32_t)(npx / 4);
    for (int mode = 0; mode < 5; ++mode) {
        const int grain = mode == 0 ? 0 : (mode == 1 ? 0 : (mode == 2 ? 2 : (mode == 3 ? 6 : 16)));
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, in, F, H, W, mode == 0 ? 0 : 1, grain);
        CK(hipDeviceSynchronize());
        for (int layout = 0; layout < 2; ++layout) {
            for (int grid : {1024, 4096, 16384}) {
                float best = 1e9f;
                for (int it = 0; it < 6; ++it) {
                    CK(hipEventRecord(e0));
                    if (layout == 0) hipLaunchKernelGGL(lut_dither_kernel<0>, dim3(grid), dim3(256), 0, 0, (const uint3 *)in, (uint3 *)out, lut, pal, thr, n_groups, W, H);
                    else hipLaunchKernelGGL(lut_dither_kernel<1>, dim3(grid), dim3(256), 0, 0, (const uint3 *)in, (uint3 *)out, lut, pal, thr, n_groups, W, H);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (it > 0 && ms < best) best = ms;
                }
                printf("data=%s grain=%d layout=%d grid=%d: %.3f ms = %.1f Gpx/s = %.0f GB/s (6 B/px)\n", mode == 0 ? "noise" : "smooth", grain, layout, grid, best,
                       npx / best * 1e-6, npx * 6.0 / best * 1e-6);
            }
        }
    }
    return 0;
}
