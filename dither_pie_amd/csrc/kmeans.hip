// One Lloyd pass of the k-means palette extractor (ColorReducer.generate_kmeans_palette,
// dithering_lib.py:1845-1857 -> sklearn KMeans) over packed uint8 RGB pixels.
//
// HBM-read bound by construction: 3 B/pixel in, nothing out but K*(3+1+1) int64 totals.  Each lane
// owns 4 consecutive pixels (12 B, three coalesced dword loads); centres sit in LDS (float64 and a
// float32 copy) and are read as wave-wide broadcasts.  The label is found with a float32 scan that
// keeps the two smallest distances; when they are closer than the float32 error bound the float64 scan
// decides (lowest index on exact ties).  Per-cluster totals accumulate in LDS as three packed 64-bit
// words (r | g<<24, b | count<<24, sum of squares; a workgroup sees 2^16 pixels, so every field fits)
// and leave the workgroup as one int64 atomic per non-empty entry.  All totals are integers, so any rank count / reduction order gives
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
    double *s_c = reinterpret_cast<double *>(smem);                                   // K*3
    unsigned long long *s_rg = reinterpret_cast<unsigned long long *>(s_c + 3 * K);   // K: sum r | sum g << 24
    unsigned long long *s_bn = s_rg + K;                                              // K: sum b | count << 24
    unsigned long long *s_sq = s_bn + K;                                              // K: sum of r^2+g^2+b^2
    float *s_cf = reinterpret_cast<float *>(s_sq + K);                                // K*3

    for (int i = threadIdx.x; i < 3 * K; i += kBlock) {
        s_c[i] = centers[i];
        s_cf[i] = (float)centers[i];
    }
    for (int i = threadIdx.x; i < K; i += kBlock) s_rg[i] = s_bn[i] = s_sq[i] = 0;
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
        // centre loop outermost: each centre is read from LDS once for the lane's four pixels
        float fr[4], fg[4], fb[4], b0[4], b1[4];
        int best[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            fr[q] = (float)(v[q] & 255u);
            fg[q] = (float)((v[q] >> 8) & 255u);
            fb[q] = (float)(v[q] >> 16);
            b0[q] = b1[q] = __int_as_float(0x7f800000);
            best[q] = 0;
        }
        for (int j = 0; j < K; ++j) {
            const float c0 = s_cf[3 * j], c1 = s_cf[3 * j + 1], c2 = s_cf[3 * j + 2];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float a = fr[q] - c0, c = fg[q] - c1, e = fb[q] - c2;
                const float d = a * a + c * c + e * e;
                const bool lt0 = d < b0[q];
                b1[q] = lt0 ? b0[q] : (d < b1[q] ? d : b1[q]);
                best[q] = lt0 ? j : best[q];
                b0[q] = lt0 ? d : b0[q];
            }
        }
        for (int q = 0; q < cnt; ++q) {
            const uint32_t r = v[q] & 255u, g = (v[q] >> 8) & 255u, b = v[q] >> 16;
            int lab = best[q];
            // float32 centres are off by <= 255*2^-24, distances by <= ~5e-5*sqrt(d) + 4e-7*d <= 0.05 + 1e-6 d
            if (!(b1[q] - b0[q] > 0.05f + 1e-6f * b1[q])) {
                const double x0 = (double)r, x1 = (double)g, x2 = (double)b;
                double bd = __longlong_as_double(0x7ff0000000000000LL);
                for (int j = 0; j < K; ++j) {
                    const double a = __dsub_rn(x0, s_c[3 * j]), c = __dsub_rn(x1, s_c[3 * j + 1]),
                                 e = __dsub_rn(x2, s_c[3 * j + 2]);
                    const double d = __dadd_rn(__dadd_rn(__dmul_rn(a, a), __dmul_rn(c, c)), __dmul_rn(e, e));
                    if (d < bd) {
                        bd = d;
                        lab = j;
                    }
                }
            }
            atomicAdd(&s_rg[lab], (unsigned long long)r | ((unsigned long long)g << 24));
            atomicAdd(&s_bn[lab], (unsigned long long)b | (1ull << 24));
            atomicAdd(&s_sq[lab], (unsigned long long)(r * r + g * g + b * b));
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K; i += kBlock) {
        const unsigned long long rg = s_rg[i], bn = s_bn[i];
        const unsigned long long c = bn >> 24;
        if (c) {
            atomicAdd(&sums[3 * i], rg & 0xffffffull);
            atomicAdd(&sums[3 * i + 1], rg >> 24);
            atomicAdd(&sums[3 * i + 2], bn & 0xffffffull);
            atomicAdd(&counts[i], c);
            atomicAdd(&sumsq[i], s_sq[i]);
        }
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
    const size_t smem = sizeof(double) * 3 * K + sizeof(unsigned long long) * 3 * K + sizeof(float) * 3 * K;
    hipLaunchKernelGGL(kmeans_step_kernel, dim3((unsigned)blocks), dim3(kBlock), smem, s, px, n, centers, K,
                       reinterpret_cast<unsigned long long *>(sums), reinterpret_cast<unsigned long long *>(counts),
                       reinterpret_cast<unsigned long long *>(sumsq));
    prof_end(pm, s);
    DP_HIP(hipGetLastError());
    return DP_OK;
}

}  // namespace dp
