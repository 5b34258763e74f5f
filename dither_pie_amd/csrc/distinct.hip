// The distinct colours of an image in order of FIRST OCCURRENCE, on the device -- the device side of the reference's
// `set(image.getdata())` (ColorReducer.reduce_colors, dithering_lib.py:1835-1843): adding a colour that is already in a set
// changes nothing, so the set the reference builds from all pixels is the set built from the first occurrences, in that order,
// and only those (a few hundred thousand colours instead of 8 M pixels) have to cross PCIe for the host's replay of CPython's
// set order and the median cut (dp_median_cut_host).
//   first_index_kernel   first[colour] = min(pixel index) over all 2^24 colours (uint32 table, 64 MB, preset to 0xffffffff).
//                        As in the histogram build (kmeans_hist.hip) a workgroup merges its window of pixels in an LDS hash
//                        table first -- (colour, smallest index) -- so runs of one colour cost one global atomic, not one per
//                        pixel; what leaves the workgroup is one agent-scope atomicMin per distinct colour of the window.
//   first_flags_kernel   pixel i is a first occurrence iff first[colour(i)] == i: flags as ballot words (4 x 64 bits per 256
//                        pixels) + the number of flags per block of 2048 pixels.
//   block_scan_kernel    exclusive prefix sum over the blocks' counts (one workgroup), the total to n_distinct.
//   emit_kernel          the flagged pixels' bytes, compacted in pixel order: block base + rank inside the block from the
//                        ballot words (no second look at the table).
// 3 B/pixel read three times + two random 4-byte table accesses per pixel; bit-exact by construction (integers, min is
// order-independent).
#include <algorithm>

#include "dp_internal.h"
#include "wave_util.hip.h"

namespace dp {
namespace {

constexpr size_t kFirstTableBytes = (size_t)4 << 24;
constexpr uint32_t kNone = 0xffffffffu;
constexpr int kBlockPx = 2048;  // pixels per compaction block (one wave, 8 rounds of 256)

__device__ __forceinline__ void load4(const uint8_t *__restrict__ px, const int64_t n, const int64_t gi, const bool aligned, uint32_t (&c)[4], int &cnt)
{
    const int64_t p0 = gi * 4;
    cnt = p0 < n ? (int)min<int64_t>(4, n - p0) : 0;
    c[0] = c[1] = c[2] = c[3] = 0u;
    if (aligned && cnt == 4) {
        const uint3 w = reinterpret_cast<const uint3 *>(px)[gi];
        c[0] = w.x & 0xffffffu;
        c[1] = __builtin_amdgcn_perm(w.y, w.x, 0x0c050403u);
        c[2] = __builtin_amdgcn_perm(w.z, w.y, 0x0c040302u);
        c[3] = w.z >> 8;
    } else {
        for (int q = 0; q < cnt; ++q) {
            const uint8_t *b = px + (p0 + q) * 3;
            c[q] = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16);
        }
    }
}

constexpr int kFB = 256;
constexpr int kFSlots = 4096;
constexpr int kFMaxWindow = 16;

__global__ __launch_bounds__(kFB) void first_index_kernel(const uint8_t *__restrict__ px, const int64_t n, uint32_t *__restrict__ first)
{
    __shared__ uint32_t s_key[kFSlots];
    __shared__ uint32_t s_min[kFSlots];
    __shared__ uint32_t s_occ;
    for (int i = threadIdx.x; i < kFSlots; i += kFB) {
        s_key[i] = kNone;
        s_min[i] = kNone;
    }
    if (threadIdx.x == 0) s_occ = 0u;
    __syncthreads();
    const int64_t n_groups = (n + 3) / 4;
    const bool aligned = ((uintptr_t)px & 3) == 0;
    int window = 1, in_window = 0;
    auto insert = [&](const uint32_t colour, const uint32_t idx) {
        uint32_t slot = (colour * 0x9E3779B1u) >> 20;
#pragma unroll 1
        for (int probe = 0; probe < 8; ++probe) {
            const uint32_t old = atomicCAS(&s_key[slot], kNone, colour);
            if (old == kNone || old == colour) {
                atomicMin(&s_min[slot], idx);
                return;
            }
            slot = (slot + 1u) & (uint32_t)(kFSlots - 1);
        }
        __hip_atomic_fetch_min(&first[colour], idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto flush = [&]() -> uint32_t {
        __syncthreads();
        uint32_t occ = 0;
        for (int i = threadIdx.x; i < kFSlots; i += kFB) {
            const uint32_t k = s_key[i];
            if (k != kNone) {
                __hip_atomic_fetch_min(&first[k], s_min[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_key[i] = kNone;
                s_min[i] = kNone;
                ++occ;
            }
        }
        occ = wave_sum_to_lane63(occ);
        if ((threadIdx.x & 63) == 63) atomicAdd(&s_occ, occ);
        __syncthreads();
        const uint32_t total = s_occ;
        __syncthreads();
        if (threadIdx.x == 0) s_occ = 0u;
        return total;
    };
    for (int64_t g0 = (int64_t)blockIdx.x * kFB; g0 < n_groups; g0 += (int64_t)gridDim.x * kFB) {
        const int64_t gi = g0 + threadIdx.x;
        uint32_t c[4];
        int cnt;
        load4(px, n, gi, aligned, c, cnt);
        const uint32_t i0 = (uint32_t)(gi * 4);
        // a run of one colour inside the lane: its first pixel carries the smallest index
        int start = 0;
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            if (q < cnt && c[q] != c[q - 1]) {
                insert(c[q - 1], i0 + (uint32_t)start);
                start = q;
            }
        }
        if (cnt > 0) insert(c[cnt - 1], i0 + (uint32_t)start);
        if (++in_window >= window) {  // (block-uniform)
            const uint32_t occ = flush();
            in_window = 0;
            if (occ < (uint32_t)kFSlots / 8 && window < kFMaxWindow) window *= 2;
            else if (occ > (uint32_t)kFSlots / 3 && window > 1) window /= 2;
        }
    }
    if (in_window) flush();
}

// one wave per block of kBlockPx pixels; flags: [block][round 0..7][q 0..3] 64-bit ballots (bit = lane), pixel = 256 round + 4 lane + q
__global__ __launch_bounds__(64) void first_flags_kernel(const uint8_t *__restrict__ px, const int64_t n, const uint32_t *__restrict__ first,
                                                         unsigned long long *__restrict__ flags, uint32_t *__restrict__ block_counts)
{
    const int lane = threadIdx.x;
    const bool aligned = ((uintptr_t)px & 3) == 0;
    const int64_t gbase = (int64_t)blockIdx.x * (kBlockPx / 4);
    uint32_t total = 0;
#pragma unroll 1
    for (int round = 0; round < kBlockPx / 256; ++round) {
        const int64_t gi = gbase + round * 64 + lane;
        uint32_t c[4];
        int cnt;
        load4(px, n, gi, aligned, c, cnt);
        uint32_t f[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) f[q] = q < cnt ? first[c[q]] : kNone - 1u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned long long m = __ballot(q < cnt && f[q] == (uint32_t)(gi * 4 + q));
            if (lane == 0) flags[((size_t)blockIdx.x * (kBlockPx / 256) + round) * 4 + q] = m;
            total += (uint32_t)__popcll(m);
        }
    }
    if (lane == 0) block_counts[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void block_scan_kernel(uint32_t *__restrict__ block_counts, const int64_t n_blocks, long long *__restrict__ n_distinct)
{
    __shared__ uint32_t s_part[16];
    __shared__ unsigned long long s_carry;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (t == 0) s_carry = 0ull;
    __syncthreads();
    for (int64_t base = 0; base < n_blocks; base += 1024) {
        const int64_t i = base + t;
        const uint32_t mine = i < n_blocks ? block_counts[i] : 0u;
        uint32_t incl = mine;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)incl, off);
            if (lane >= off) incl += o;
        }
        if (lane == 63) s_part[wv] = incl;
        __syncthreads();
        uint32_t before = 0, chunk = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            before += w < wv ? s_part[w] : 0u;
            chunk += s_part[w];
        }
        const unsigned long long carry = s_carry;
        // (block bases are 32-bit: n < 2^32 pixels)
        if (i < n_blocks) block_counts[i] = (uint32_t)(carry + before + incl - mine);
        __syncthreads();
        if (t == 0) s_carry = carry + chunk;
        __syncthreads();
    }
    if (t == 0) *n_distinct = (long long)s_carry;
}

__global__ __launch_bounds__(64) void emit_kernel(const uint8_t *__restrict__ px, const int64_t n, const unsigned long long *__restrict__ flags,
                                                  const uint32_t *__restrict__ block_base, uint8_t *__restrict__ out)
{
    const int lane = threadIdx.x;
    const bool aligned = ((uintptr_t)px & 3) == 0;
    const int64_t gbase = (int64_t)blockIdx.x * (kBlockPx / 4);
    size_t at = block_base[blockIdx.x];
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll 1
    for (int round = 0; round < kBlockPx / 256; ++round) {
        const unsigned long long *fw = flags + ((size_t)blockIdx.x * (kBlockPx / 256) + round) * 4;
        const unsigned long long m0 = fw[0], m1 = fw[1], m2 = fw[2], m3 = fw[3];
        if ((m0 | m1 | m2 | m3) == 0ull) continue;  // (wave-uniform)
        const int64_t gi = gbase + round * 64 + lane;
        uint32_t c[4];
        int cnt;
        load4(px, n, gi, aligned, c, cnt);
        // pixels are ordered lane-major (4 lane + q): everything flagged in lower lanes, then this lane's lower q
        uint32_t rank = (uint32_t)(__popcll(m0 & lt) + __popcll(m1 & lt) + __popcll(m2 & lt) + __popcll(m3 & lt));
        const unsigned long long mm[4] = {m0, m1, m2, m3};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if ((mm[q] >> lane) & 1ull) {
                uint8_t *o = out + (at + rank) * 3;
                o[0] = (uint8_t)c[q];
                o[1] = (uint8_t)(c[q] >> 8);
                o[2] = (uint8_t)(c[q] >> 16);
                ++rank;
            }
        }
        at += (size_t)(__popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3));
    }
}

inline int64_t n_blocks_of(const int64_t n) { return (n + kBlockPx - 1) / kBlockPx; }
inline size_t flags_bytes(const int64_t n) { return (size_t)n_blocks_of(n) * (kBlockPx / 256) * 4 * sizeof(unsigned long long); }

}  // namespace

size_t distinct_first_ws_bytes(int64_t n)
{
    return kFirstTableBytes + flags_bytes(n) + (((size_t)n_blocks_of(n) * sizeof(uint32_t) + 255) & ~(size_t)255) + 256;
}

int launch_distinct_first(const uint8_t *px, int64_t n, uint8_t *out, long long *n_distinct, void *ws, hipStream_t s)
{
    uint32_t *first = static_cast<uint32_t *>(ws);
    unsigned long long *flags = reinterpret_cast<unsigned long long *>(static_cast<uint8_t *>(ws) + kFirstTableBytes);
    uint32_t *block_counts = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(flags) + flags_bytes(n));
    const int64_t nb = n_blocks_of(n);
    if (n == 0) {
        DP_HIP(hipMemsetAsync(n_distinct, 0, sizeof(long long), s));
        return DP_OK;
    }
    DP_HIP(hipMemsetAsync(first, 0xff, kFirstTableBytes, s));
    int cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const int64_t groups = (n + 3) / 4;
    const unsigned blocks = (unsigned)std::min<int64_t>((groups + kFB - 1) / kFB, (int64_t)cus * 4);
    ProfMark *pm = prof_begin(s);
    hipLaunchKernelGGL(first_index_kernel, dim3(blocks), dim3(kFB), 0, s, px, n, first);
    hipLaunchKernelGGL(first_flags_kernel, dim3((unsigned)nb), dim3(64), 0, s, px, n, first, flags, block_counts);
    hipLaunchKernelGGL(block_scan_kernel, dim3(1), dim3(1024), 0, s, block_counts, nb, n_distinct);
    hipLaunchKernelGGL(emit_kernel, dim3((unsigned)nb), dim3(64), 0, s, px, n, flags, block_counts, out);
    prof_end(pm, s);
    DP_HIP(hipGetLastError());
    return DP_OK;
}

}  // namespace dp
