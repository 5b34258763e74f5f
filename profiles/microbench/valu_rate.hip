// Issue-rate microbenchmark for the VALU ops of the ordered kernel (gfx950).
// hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned seed)
{
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    const unsigned b = blockIdx.x * 2654435761u + seed;
    const int s = (int)seed;
    for (int i = 0; i < ITER; ++i) {
#define R(x)                                                                                   \
    if (OP == 0) asm volatile("v_dot4_u32_u8 %0, %0, %1, 0" : "+v"(x) : "v"(b));                 \
    if (OP == 1) asm volatile("v_med3_i32 %0, %0, %1, %1" : "+v"(x) : "v"(b));                   \
    if (OP == 2) asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(x) : "s"(s));                \
    if (OP == 3) asm volatile("v_min_i32 %0, %0, %1" : "+v"(x) : "v"(b));                        \
    if (OP == 4) asm volatile("v_lshl_add_u32 %0, %0, 8, %1" : "+v"(x) : "v"(b));                \
    if (OP == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));                        \
    if (OP == 6) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(b));                     \
    if (OP == 7) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(b));                    \
    if (OP == 8) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(b));                    \
    if (OP == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b));               \
    if (OP == 10) asm volatile("v_dot2_i32_i16 %0, %0, %1, %0" : "+v"(x) : "v"(b));              \
    if (OP == 11) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(x) : "v"(b));                    \
    if (OP == 12) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(x) : "v"(b));                  \
    if (OP == 13) asm volatile("v_bfe_u32 %0, %0, 4, 20" : "+v"(x));                             \
    if (OP == 14) asm volatile("v_sad_u8 %0, %0, %1, %0" : "+v"(x) : "v"(b));
        R(a0) R(a1) R(a2) R(a3) R(a4) R(a5) R(a6) R(a7)
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
template <int OP>
void run(const char *name, unsigned *d, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd;  // 256 CUs x (4 waves per block => 1 wave/SIMD per block)
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, 2);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD = waves_per_simd * ITER * 8
    const double inst = (double)waves_per_simd * ITER * 8;
    printf("%-18s waves/SIMD=%d  %.3f ms  => %.2f ns per wave-instruction per SIMD (%.2f cycles @2.4GHz)\n", name,
           waves_per_simd, ms, ms * 1e6 / inst, ms * 1e6 / inst * 2.4);
}
int main()
{
    unsigned *d;
    hipMalloc(&d, 256 * 256 * 8 * 4);
    for (int w : {1, 4}) {
        run<0>("v_dot4_u32_u8", d, w);
        run<1>("v_med3_i32", d, w);
        run<2>("v_mad_i32_i24", d, w);
        run<3>("v_min_i32", d, w);
        run<4>("v_lshl_add_u32", d, w);
        run<5>("v_add_u32", d, w);
        run<6>("v_mul_lo_u32", d, w);
        run<7>("v_fma_f32", d, w);
        run<8>("v_mul_u32_u24", d, w);
        run<9>("v_cndmask_b32", d, w);
        run<10>("v_dot2_i32_i16", d, w);
        run<11>("v_pk_sub_i16", d, w);
        run<12>("v_perm_b32", d, w);
        run<13>("v_bfe_u32", d, w);
        run<14>("v_sad_u8", d, w);
    }
    return 0;
}
