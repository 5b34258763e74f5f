// Per-palette search accelerator for integer palettes (built once in dp_palette_create).
//
// The nearest / second-nearest palette entry of a colour x can only be an entry of
//     T(x) = { j : d_j(x) <= d_(2)(x) }        (everything at least as close as the 2nd nearest),
// and which of several equidistant entries scipy reports depends only on the order in which its
// KD-tree traversal meets the members of T(x).  Both facts are properties of the 2^24 colours, not
// of an image, so they are tabulated per palette:
//
//   cell lists   for each of the 16x16x16 cells of the RGB cube, the exact union of T(x) over the
//                cell's 4096 colours, sorted by palette index and padded to a multiple of 4 with
//                further (harmless) palette entries.  ~5 entries per cell for a 256-colour palette;
//                the whole table (~100 KB) lives in LDS in the dither kernel.
//   tie codes    2 bits per colour and per query kind (k=1, k=2): for colours whose three smallest
//                distances contain a tie, the outcome of scipy's traversal (tree_query) expressed
//                relative to the candidates sorted by (distance, index):
//                  k=2: 0 -> (c0,c1)  1 -> (c1,c0)  2 -> (c0,c2)  3 -> none of these (rare; the
//                       dither kernel flags the pixel for the generic fix-up pass)
//                  k=1: 0 -> c0  1 -> c1  2 -> c2  3 -> other
//                "equal" is judged on the output colour, so duplicate palette entries never need the
//                fix-up pass.  4 MB per kind, touched only by tied pixels (~0.2 %).
#include <algorithm>
#include <vector>

#include "dp_internal.h"
#include "tree_query.cuh"

namespace dp {
namespace {

__device__ __forceinline__ int med3i(const int a, const int b, const int c)
{
    return max(min(a, b), min(max(a, b), c));
}

__global__ __launch_bounds__(256) void accel_scan_kernel(const PalDev pal, uint32_t *__restrict__ masks,
                                                         uint32_t *__restrict__ code1, uint32_t *__restrict__ code2)
{
    __shared__ uint32_t s_mask[8];
    if (threadIdx.x < 8) s_mask[threadIdx.x] = 0;
    __syncthreads();
    const int cell = blockIdx.x;
    const int rc = cell >> 8, gc = (cell >> 4) & 15, bc = cell & 15;
    const int K = pal.K;
    constexpr int kBig = 0x7fffffff;
    constexpr int IM = (1 << kIdxBits) - 1;

    for (int t = 0; t < 16; ++t) {
        const int id = threadIdx.x + 256 * t;
        const uint32_t r = rc * 16 + (id >> 8), g = gc * 16 + ((id >> 4) & 15), b = bc * 16 + (id & 15);
        const uint32_t x4 = r | (g << 8) | (b << 16);
        int m0 = kBig, m1 = kBig, m2 = kBig, m3 = kBig;
        for (int j = 0; j < K; ++j) {
            const int dot = (int)__builtin_amdgcn_udot4(x4, pal.p4[j], 0u, false);
            const int key = pal.nkey[j] - (dot << (kIdxBits + 1));
            const int n3 = med3i(m2, m3, key);
            const int n2 = med3i(m1, m2, key);
            const int n1 = med3i(m0, m1, key);
            m0 = min(m0, key);
            m1 = n1;
            m2 = n2;
            m3 = n3;
        }
        const int d0 = m0 >> kIdxBits, d1 = m1 >> kIdxBits, d2 = m2 >> kIdxBits, d3 = m3 >> kIdxBits;
        const int c0 = m0 & IM, c1 = m1 & IM, c2 = m2 & IM;
        atomicOr(&s_mask[c0 >> 5], 1u << (c0 & 31));
        if (K > 1) atomicOr(&s_mask[c1 >> 5], 1u << (c1 & 31));
        if (K > 2 && d2 == d1) atomicOr(&s_mask[c2 >> 5], 1u << (c2 & 31));
        if (K > 3 && d3 == d1) {  // four or more at the second distance: take every one of them
            for (int j = 0; j < K; ++j) {
                const int dot = (int)__builtin_amdgcn_udot4(x4, pal.p4[j], 0u, false);
                const int dj = (pal.nkey[j] - (dot << (kIdxBits + 1))) >> kIdxBits;
                if (dj <= d1) atomicOr(&s_mask[j >> 5], 1u << (j & 31));
            }
        }
        if (K < 2) continue;
        const bool tie01 = d0 == d1, tie12 = (K > 2) && d1 == d2;
        if (!(tie01 || tie12)) continue;
        const uint32_t o0 = pal.out_rgb[c0], o1 = pal.out_rgb[c1], o2 = K > 2 ? pal.out_rgb[c2] : 0xffffffffu;
        double dd[2];
        int ii[2];
        {
            tree_query<2>(pal, (double)r, (double)g, (double)b, dd, ii);
            const uint32_t on = pal.out_rgb[ii[0]], os = pal.out_rgb[ii[1]];
            uint32_t code = 3;
            if (on == o0 && os == o1) code = 0;
            else if (on == o1 && os == o0) code = 1;
            else if (on == o0 && os == o2) code = 2;
            if (code) atomicOr(&code2[x4 >> 4], code << ((x4 & 15u) * 2));
        }
        if (tie01) {
            tree_query<1>(pal, (double)r, (double)g, (double)b, dd, ii);
            const uint32_t on = pal.out_rgb[ii[0]];
            uint32_t code = 3;
            if (on == o0) code = 0;
            else if (on == o1) code = 1;
            else if (on == o2) code = 2;
            if (code) atomicOr(&code1[x4 >> 4], code << ((x4 & 15u) * 2));
        }
    }
    __syncthreads();
    if (threadIdx.x < 8) masks[cell * 8 + threadIdx.x] = s_mask[threadIdx.x];
}

}  // namespace

// Builds the accelerator for an integer palette whose output colours equal its search colours.
// On success fills the accel fields of `dev` and returns the device allocation through *blob_out.
int build_accel(PalDev &dev, const std::vector<uint32_t> &p4_host, void **blob_out, size_t *blob_bytes)
{
    *blob_out = nullptr;
    *blob_bytes = 0;
    const int K = dev.K;
    constexpr size_t kCodeWords = (size_t)1 << 20;  // 2^24 colours x 2 bits
    constexpr int kCells = 4096;
    uint32_t *d_masks = nullptr;
    uint8_t *blob = nullptr;
    // layout: code1 | code2 | desc[4096] | pool[cap]
    constexpr int kPoolCap = 36 * 1024;  // entries; desc(16 KB) + pool(144 KB) stays inside 160 KB of LDS
    const size_t bytes = sizeof(uint32_t) * (2 * kCodeWords + kCells + kPoolCap);
    DP_HIP(hipMalloc((void **)&blob, bytes));
    hipError_t e = hipMalloc((void **)&d_masks, sizeof(uint32_t) * kCells * 8);
    if (e == hipSuccess) e = hipMemset(blob, 0, sizeof(uint32_t) * 2 * kCodeWords);
    if (e != hipSuccess) {
        (void)hipFree(blob);
        if (d_masks) (void)hipFree(d_masks);
        return hip_fail(e, "accelerator allocation");
    }
    uint32_t *code1 = reinterpret_cast<uint32_t *>(blob);
    uint32_t *code2 = code1 + kCodeWords;
    uint32_t *d_desc = code2 + kCodeWords;
    uint32_t *d_pool = d_desc + kCells;
    hipLaunchKernelGGL(accel_scan_kernel, dim3(kCells), dim3(256), 0, 0, dev, d_masks, code1, code2);
    std::vector<uint32_t> masks((size_t)kCells * 8);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(masks.data(), d_masks, sizeof(uint32_t) * masks.size(), hipMemcpyDeviceToHost);
    (void)hipFree(d_masks);
    if (e != hipSuccess) {
        (void)hipFree(blob);
        return hip_fail(e, "accelerator scan");
    }

    std::vector<uint32_t> desc(kCells), pool;
    pool.reserve(kPoolCap);
    std::vector<int> list, extra;
    int max_cnt = 0;
    for (int cell = 0; cell < kCells; ++cell) {
        list.clear();
        for (int j = 0; j < K; ++j)
            if (masks[(size_t)cell * 8 + (j >> 5)] >> (j & 31) & 1u) list.push_back(j);
        const int want = ((int)list.size() + 3) & ~3;
        if (want > K || want > 252) {  // cannot pad with distinct entries
            (void)hipFree(blob);
            return DP_OK;              // no accelerator: the brute-force kernel stays in charge
        }
        if ((int)list.size() < want) {
            // pad with the unused entries closest to the cell centre (any real entry is harmless)
            const int cr = (cell >> 8) * 16 + 8, cg = ((cell >> 4) & 15) * 16 + 8, cb = (cell & 15) * 16 + 8;
            extra.clear();
            for (int j = 0; j < K; ++j)
                if (!(masks[(size_t)cell * 8 + (j >> 5)] >> (j & 31) & 1u)) extra.push_back(j);
            auto dist = [&](int j) {
                const int r = p4_host[j] & 255, g = (p4_host[j] >> 8) & 255, b = (p4_host[j] >> 16) & 255;
                return (r - cr) * (r - cr) + (g - cg) * (g - cg) + (b - cb) * (b - cb);
            };
            std::stable_sort(extra.begin(), extra.end(), [&](int a, int b) { return dist(a) < dist(b); });
            for (int i = 0; (int)list.size() < want; ++i) list.push_back(extra[i]);
            std::sort(list.begin(), list.end());
        }
        if (pool.size() + list.size() > (size_t)kPoolCap) {
            (void)hipFree(blob);
            return DP_OK;
        }
        desc[cell] = (uint32_t)pool.size() | ((uint32_t)list.size() << 20);
        for (int j : list) pool.push_back(p4_host[j]);
        max_cnt = std::max(max_cnt, (int)list.size());
    }
    e = hipMemcpy(d_desc, desc.data(), sizeof(uint32_t) * kCells, hipMemcpyHostToDevice);
    if (e == hipSuccess && !pool.empty())
        e = hipMemcpy(d_pool, pool.data(), sizeof(uint32_t) * pool.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(blob);
        return hip_fail(e, "accelerator upload");
    }
    dev.cell_desc = d_desc;
    dev.cell_pool = d_pool;
    dev.pool_entries = (int)pool.size();
    dev.max_cell = max_cnt;
    dev.code1 = code1;
    dev.code2 = code2;
    *blob_out = blob;
    *blob_bytes = bytes;
    return DP_OK;
}

}  // namespace dp
