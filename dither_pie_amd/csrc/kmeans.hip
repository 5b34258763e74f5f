// One Lloyd pass of the k-means palette extractor (ColorReducer.generate_kmeans_palette,
// dithering_lib.py:1845-1857 -> sklearn KMeans) over packed uint8 RGB pixels.
//
// HBM-read bound by construction: 3 B/pixel in, nothing out but K*(3+1+1) int64 totals.  Each lane
// owns 4 consecutive pixels (12 B, three coalesced dword loads); centres sit in LDS as float64 and
// are read as wave-wide broadcasts; per-cluster channel sums / counts / squared norms accumulate in
// LDS as uint32 (a workgroup never sees more than 2^20 pixels) and leave the workgroup as one int64
// atomic per non-empty entry.  All totals are integers, so any rank count / reduction order gives
// identical results; the float64 inertia is derived from them on the host.
#include "dp_internal.h"

namespace dp {
namespace {

constexpr int kBlock = 256;
constexpr int kPxPerBlock = 1 << 16;  // pixels per workgroup (<< 2^20 keeps uint32 sums exact)

__global__ __launch_bounds__(kBlock) void kmeans_step_kernel(const uint8_t *__restrict__ px, const int64_t n,
                                                             const double *__restrict__ centers, const int K,
                                                             unsigned long long *__restrict__ sums,
                                                             unsigned long long *__restrict__ counts,
                                                             unsigned long long *__restrict__ sumsq)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double *s_c = reinterpret_cast<double *>(smem);                                      // K*3
    unsigned long long *s_sq = reinterpret_cast<unsigned long long *>(s_c + 3 * K);      // K
    uint32_t *s_sum = reinterpret_cast<uint32_t *>(s_sq + K);                            // K*3
    uint32_t *s_cnt = s_sum + 3 * K;                                                     // K

    for (int i = threadIdx.x; i < 3 * K; i += kBlock) {
        s_c[i] = centers[i];
        s_sum[i] = 0;
    }
    for (int i = threadIdx.x; i < K; i += kBlock) {
        s_cnt[i] = 0;
        s_sq[i] = 0;
    }
    __syncthreads();

    const int64_t base = (int64_t)blockIdx.x * kPxPerBlock;
    const int64_t lim = min(n, base + kPxPerBlock);
    const bool aligned = ((uintptr_t)px & 3) == 0;
    for (int64_t p0 = base + (int64_t)threadIdx.x * 4; p0 < lim; p0 += kBlock * 4) {
        uint32_t v[4];
        int cnt = (int)min<int64_t>(4, lim - p0);
        if (aligned && cnt == 4) {
            const uint32_t *q = reinterpret_cast<const uint32_t *>(px + p0 * 3);
            const uint32_t w0 = q[0], w1 = q[1], w2 = q[2];
            v[0] = w0 & 0xffffffu;
            v[1] = (w0 >> 24) | ((w1 & 0xffffu) << 8);
            v[2] = (w1 >> 16) | ((w2 & 0xffu) << 16);
            v[3] = w2 >> 8;
        } else {
            for (int q = 0; q < 4; ++q) {
                v[q] = 0;
                if (q < cnt) {
                    const uint8_t *b = px + (p0 + q) * 3;
                    v[q] = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16);
                }
            }
        }
        for (int q = 0; q < cnt; ++q) {
            const uint32_t r = v[q] & 255u, g = (v[q] >> 8) & 255u, b = v[q] >> 16;
            const double x0 = (double)r, x1 = (double)g, x2 = (double)b;
            double bd = __longlong_as_double(0x7ff0000000000000LL);
            int best = 0;
            for (int j = 0; j < K; ++j) {
                const double a = __dsub_rn(x0, s_c[3 * j]), c = __dsub_rn(x1, s_c[3 * j + 1]),
                             e = __dsub_rn(x2, s_c[3 * j + 2]);
                const double d = __dadd_rn(__dadd_rn(__dmul_rn(a, a), __dmul_rn(c, c)), __dmul_rn(e, e));
                if (d < bd) {
                    bd = d;
                    best = j;
                }
            }
            atomicAdd(&s_sum[3 * best], r);
            atomicAdd(&s_sum[3 * best + 1], g);
            atomicAdd(&s_sum[3 * best + 2], b);
            atomicAdd(&s_cnt[best], 1u);
            atomicAdd(&s_sq[best], (unsigned long long)(r * r + g * g + b * b));
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * K; i += kBlock)
        if (s_sum[i]) atomicAdd(&sums[i], (unsigned long long)s_sum[i]);
    for (int i = threadIdx.x; i < K; i += kBlock)
        if (s_cnt[i]) {
            atomicAdd(&counts[i], (unsigned long long)s_cnt[i]);
            atomicAdd(&sumsq[i], s_sq[i]);
        }
}

}  // namespace

int launch_kmeans_step(const uint8_t *px, int64_t n, const double *centers, int K, int64_t *sums, int64_t *counts,
                       int64_t *sumsq, hipStream_t s)
{
    DP_HIP(hipMemsetAsync(sums, 0, sizeof(int64_t) * 3 * (size_t)K, s));
    DP_HIP(hipMemsetAsync(counts, 0, sizeof(int64_t) * (size_t)K, s));
    DP_HIP(hipMemsetAsync(sumsq, 0, sizeof(int64_t) * (size_t)K, s));
    if (n == 0) return DP_OK;
    const int64_t blocks = (n + kPxPerBlock - 1) / kPxPerBlock;
    if (blocks > 0x7fffffff) {
        set_error("dp_kmeans_step_u8: too many pixels for one launch");
        return DP_EINVAL;
    }
    ProfMark *pm = prof_begin(s);
    const size_t smem = sizeof(double) * 3 * K + sizeof(unsigned long long) * K + sizeof(uint32_t) * 4 * K;
    hipLaunchKernelGGL(kmeans_step_kernel, dim3((unsigned)blocks), dim3(kBlock), smem, s, px, n, centers, K,
                       reinterpret_cast<unsigned long long *>(sums), reinterpret_cast<unsigned long long *>(counts),
                       reinterpret_cast<unsigned long long *>(sumsq));
    prof_end(pm, s);
    DP_HIP(hipGetLastError());
    return DP_OK;
}

}  // namespace dp
