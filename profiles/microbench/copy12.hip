// Calibration of FETCH_SIZE / WRITE_SIZE for the access pattern of the ordered kernels: 12 bytes per lane
// (global_load_dwordx3 / global_store_dwordx3, a wave covers 768 contiguous bytes), over a known byte count.
// MI355X_MICROARCH.md (HBM): "Other access widths are uncalibrated: calibrate on a known byte count in your own access
// pattern before trusting an absolute."  usage: copy12 [bytes]   (default: the C2 batch, 24 x 2160 x 3840 x 3)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

struct U3 { unsigned x, y, z; };

__global__ __launch_bounds__(1024) void copy12_kernel(const U3 *__restrict__ src, U3 *__restrict__ dst, const size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const U3 v = src[i];
        dst[i] = v;
    }
}

// the same with reads only (one store per workgroup) and writes only, to separate the two counters
__global__ __launch_bounds__(1024) void read12_kernel(const U3 *__restrict__ src, unsigned *__restrict__ sink, const size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const U3 v = src[i];
        acc ^= v.x ^ v.y ^ v.z;
    }
    if (acc == 0x12345u) sink[0] = acc;
}

__global__ __launch_bounds__(1024) void write12_kernel(U3 *__restrict__ dst, const size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = U3{(unsigned)i, 1u, 2u};
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const size_t bytes = argc > 1 ? strtoull(argv[1], nullptr, 10) : (size_t)24 * 2160 * 3840 * 3;
    const size_t n = bytes / 12;
    U3 *a, *b;
    unsigned *sink;
    CK(hipMalloc(&a, n * 12));
    CK(hipMalloc(&b, n * 12));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 1, n * 12));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 4; ++rep) {
        float ms[3];
        CK(hipEventRecord(e0));
        copy12_kernel<<<256 * 2, 1024>>>(a, b, n);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms[0], e0, e1));
        CK(hipEventRecord(e0));
        read12_kernel<<<256 * 2, 1024>>>(a, sink, n);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms[1], e0, e1));
        CK(hipEventRecord(e0));
        write12_kernel<<<256 * 2, 1024>>>(b, n);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms[2], e0, e1));
        printf("bytes %zu copy %.4f ms (%.0f GB/s r+w) read %.4f ms (%.0f GB/s) write %.4f ms (%.0f GB/s)\n", n * 12, ms[0],
               2.0 * n * 12 / ms[0] * 1e-6, ms[1], n * 12 / ms[1] * 1e-6, ms[2], n * 12 / ms[2] * 1e-6);
    }
    return 0;
}
