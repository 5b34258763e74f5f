// Are unaligned dword global loads/stores exact on gfx950?  (needed for 48-byte pixel bursts at 3-byte granularity)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void k(const unsigned char *in, unsigned char *out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;  // each thread copies 12 bytes from offset 3*i*4+off
    for (int off = 0; off < 4; ++off) {
        const unsigned char *s = in + off * (size_t)n * 16 + (size_t)i * 12 + off;
        unsigned char *d = out + off * (size_t)n * 16 + (size_t)i * 12 + off;
        if (i < n) {
            unsigned a = *reinterpret_cast<const unsigned *>(s), b = *reinterpret_cast<const unsigned *>(s + 4),
                     c = *reinterpret_cast<const unsigned *>(s + 8);
            *reinterpret_cast<unsigned *>(d) = a;
            *reinterpret_cast<unsigned *>(d + 4) = b;
            *reinterpret_cast<unsigned *>(d + 8) = c;
        }
    }
}
int main()
{
    const int n = 100000;
    const size_t bytes = (size_t)4 * n * 16 + 64;
    std::vector<unsigned char> h(bytes), o(bytes, 0);
    for (size_t i = 0; i < bytes; ++i) h[i] = (unsigned char)(i * 2654435761u >> 13);
    unsigned char *di, *dout;
    hipMalloc(&di, bytes);
    hipMalloc(&dout, bytes);
    hipMemcpy(di, h.data(), bytes, hipMemcpyHostToDevice);
    hipMemset(dout, 0, bytes);
    k<<<(n + 255) / 256, 256>>>(di, dout, n);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(o.data(), dout, bytes, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (int off = 0; off < 4; ++off)
        for (size_t i = 0; i < (size_t)n * 12; ++i) {
            size_t p = off * (size_t)n * 16 + i + off;
            bad += o[p] != h[p];
        }
    printf("sync=%d mismatching bytes=%zu\n", (int)e, bad);
    return bad != 0;
}
