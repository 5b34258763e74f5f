// generate_blue_noise(size, seed) (dithering_lib.py:381-399) on the device.
//
// One workgroup of 1024 lanes per matrix.  Lane 0 replays numpy's legacy
// RandomState(seed).shuffle (MT19937 init_genrand + masked rejection sampling) over the
// row-major coordinate list; then size^2 void-filling rounds run with every lane owning the list
// positions p = lane + 1024*k: each round fuses "update min_dist against the point just placed"
// with "first maximum of min_dist in list order" (max value, then smallest list position), reduced
// across the wave with DPP shuffles and across waves through LDS.
#include "dp_internal.h"
#include "wave_util.hip.h"

namespace dp {
namespace {

constexpr int kThreads = 1024;

struct Best {
    float v;
    uint32_t pos;
};

__device__ __forceinline__ Best better(const Best a, const Best b)
{
    // larger value wins; equal values: earlier list position wins (python max() keeps the first)
    return (b.v > a.v || (b.v == a.v && b.pos < a.pos)) ? b : a;
}

__device__ uint32_t mt_next(uint32_t *mt, int &pos)
{
    if (pos == 624) {
        for (int i = 0; i < 624; ++i) {
            const uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
            mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        pos = 0;
    }
    uint32_t y = mt[pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// scratch layout: coords[n] uint32 (r*size+c in shuffled order), md[n] float, alive handled by md = -1.  IN_LDS (sizes up to
// 128: 8 bytes per position, 128 KB): both arrays live in dynamic LDS -- lane 0's serial shuffle swaps 4095 ... 16383 pairs and
// every round reads and writes md; from global memory those dependent accesses were most of the kernel (64 x 64: 10.6 ms).
template <bool IN_LDS>
__global__ __launch_bounds__(kThreads) void blue_noise_kernel(const int size, const uint32_t seed,
                                                              float *__restrict__ out, uint32_t *__restrict__ coords_g,
                                                              float *__restrict__ md_g)
{
    extern __shared__ __align__(16) uint8_t s_dyn[];
    uint32_t *coords = IN_LDS ? reinterpret_cast<uint32_t *>(s_dyn) : coords_g;
    float *md = IN_LDS ? reinterpret_cast<float *>(s_dyn + (size_t)size * size * sizeof(uint32_t)) : md_g;
    __shared__ uint32_t s_mt[624];
    __shared__ Best s_part[kThreads / 64];
    __shared__ Best s_best;
    const int n = size * size;
    const int tid = threadIdx.x;
    const float inf = __int_as_float(0x7f800000);

    for (int i = tid; i < n; i += kThreads) {
        coords[i] = (uint32_t)i;
        md[i] = inf;
    }
    __syncthreads();
    if (tid == 0) {
        s_mt[0] = seed;
        for (int i = 1; i < 624; ++i) s_mt[i] = 1812433253u * (s_mt[i - 1] ^ (s_mt[i - 1] >> 30)) + (uint32_t)i;
        int pos = 624;
        for (int i = n - 1; i >= 1; --i) {
            uint32_t mask = (uint32_t)i;
            mask |= mask >> 1;
            mask |= mask >> 2;
            mask |= mask >> 4;
            mask |= mask >> 8;
            mask |= mask >> 16;
            uint32_t j;
            do {
                j = mt_next(s_mt, pos) & mask;
            } while (j > (uint32_t)i);
            const uint32_t t = coords[i];
            coords[i] = coords[j];
            coords[j] = t;
        }
    }
    __threadfence_block();
    __syncthreads();
    // (row, column) split once, not in every round: row << 16 | column
    for (int i = tid; i < n; i += kThreads) {
        const uint32_t c = coords[i];
        const uint32_t rr = c / (uint32_t)size;
        coords[i] = (rr << 16) | (c - rr * (uint32_t)size);
    }
    __syncthreads();

    const double denom = (double)(n - 1) + 1e-9;
    int br = 0, bc = 0;
    for (int it = 0; it < n; ++it) {
        // update against the previous pick (none in round 0) and find this round's first maximum
        Best mine;
        mine.v = -1.0f;
        mine.pos = 0xffffffffu;
        for (int p = tid; p < n; p += kThreads) {
            float v = md[p];
            if (v < 0.0f) continue;  // already placed
            if (it > 0) {
                const uint32_t c = coords[p];
                const int rr = (int)(c >> 16), cc = (int)(c & 0xffffu);
                const float d2 = (float)((rr - br) * (rr - br) + (cc - bc) * (cc - bc));
                if (d2 < v) {
                    v = d2;
                    md[p] = v;
                }
            }
            Best cand;
            cand.v = v;
            cand.pos = (uint32_t)p;
            mine = better(mine, cand);
        }
        for (int off = 32; off >= 1; off >>= 1) {
            Best o;
            o.v = lane_xor_f32(mine.v, off);
            o.pos = lane_xor_u32(mine.pos, off);
            mine = better(mine, o);
        }
        if ((tid & 63) == 0) s_part[tid >> 6] = mine;
        __syncthreads();
        if (tid < 64) {
            Best b;
            b.v = -1.0f;
            b.pos = 0xffffffffu;
            if (tid < kThreads / 64) b = s_part[tid];
            for (int off = 8; off >= 1; off >>= 1) {
                Best o;
                o.v = lane_xor_f32(b.v, off);
                o.pos = lane_xor_u32(b.pos, off);
                b = better(b, o);
            }
            if (tid == 0) s_best = b;
        }
        __syncthreads();
        const uint32_t bp = s_best.pos;
        const uint32_t c = coords[bp];
        br = (int)(c >> 16);
        bc = (int)(c & 0xffffu);
        if (tid == 0) {
            out[br * size + bc] = (float)((double)it / denom);
            md[bp] = -1.0f;
        }
        __threadfence_block();
        __syncthreads();
    }
}

}  // namespace

size_t blue_noise_scratch_bytes(int size) { return (size_t)size * size * (sizeof(uint32_t) + sizeof(float)); }

int launch_blue_noise(int size, uint32_t seed, float *out_dev, void *scratch_dev, hipStream_t s)
{
    const size_t n = (size_t)size * size;
    uint32_t *coords = reinterpret_cast<uint32_t *>(scratch_dev);
    float *md = reinterpret_cast<float *>(coords + n);
    const size_t lds = n * (sizeof(uint32_t) + sizeof(float));
    if (lds <= 128 * 1024) {
        DP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&blue_noise_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((blue_noise_kernel<true>), dim3(1), dim3(kThreads), lds, s, size, seed, out_dev, coords, md);
    } else {
        hipLaunchKernelGGL((blue_noise_kernel<false>), dim3(1), dim3(kThreads), 0, s, size, seed, out_dev, coords, md);
    }
    DP_HIP(hipGetLastError());
    return DP_OK;
}

}  // namespace dp
