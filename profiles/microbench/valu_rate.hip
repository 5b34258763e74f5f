// Issue-rate microbenchmark for the VALU ops of the ordered kernel (gfx950).
// hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
// Part 1: one op, 8 independent chains per wave, 1/2/4/8 waves per SIMD: cycles per wave-instruction per SIMD.
// Part 2: one op, ONE dependent chain per wave (latency), 1 and 8 waves per SIMD.
// Part 3: the candidate network of ordered.hip (cand8: 16 dot4 + 8 lshl_add + 8 mad + 18 min/med/max = 50 ops)
//         and the one-op-per-key network (8 dot4-with-accumulate + 18) in a loop, 1/2/4 waves per SIMD.
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
    if (OP == 14) asm volatile("v_sad_u8 %0, %0, %1, %0" : "+v"(x) : "v"(b));                    \
    if (OP == 15) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(x) : "v"(b));               \
    if (OP == 16) asm volatile("v_min3_i32 %0, %0, %1, %1" : "+v"(x) : "v"(b));                  \
    if (OP == 17) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(x) : "v"(b));                \
    if (OP == 18) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc");           \
    if (OP == 19) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(x) : "v"(b));                 \
    if (OP == 20) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(x));                             \
    if (OP == 21) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(b));                           \
    if (OP == 22) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(x) : "v"(b));                    \
    if (OP == 23) asm volatile("v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x));
        R(a0) R(a1) R(a2) R(a3) R(a4) R(a5) R(a6) R(a7)
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
template <int OP>
__global__ __launch_bounds__(256) void kdep(unsigned *out, unsigned seed)
{
    unsigned a0 = threadIdx.x + seed;
    const unsigned b = blockIdx.x * 2654435761u + seed;
    const int s = (int)seed;
    for (int i = 0; i < ITER; ++i) {
        R(a0) R(a0) R(a0) R(a0) R(a0) R(a0) R(a0) R(a0)
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0;
}

// the candidate network of ordered.hip, as it stands (50 ops)
__device__ __forceinline__ void cand8(const unsigned x, const uint4 ca, const uint4 cb, const int neg2, int &m0, int &m1, int &m2)
{
    int n0, n1, n2, n3, n4, n5, n6, n7, p0, p1, p2, p3, p4, p5, p6, p7;
    asm volatile(
        "v_dot4_u32_u8 %[n0], %[c0], %[c0], 0\n\t"
        "v_dot4_u32_u8 %[p0], %[x], %[c0], 0\n\t"
        "v_dot4_u32_u8 %[n1], %[c1], %[c1], 0\n\t"
        "v_dot4_u32_u8 %[p1], %[x], %[c1], 0\n\t"
        "v_dot4_u32_u8 %[n2], %[c2], %[c2], 0\n\t"
        "v_dot4_u32_u8 %[p2], %[x], %[c2], 0\n\t"
        "v_dot4_u32_u8 %[n3], %[c3], %[c3], 0\n\t"
        "v_dot4_u32_u8 %[p3], %[x], %[c3], 0\n\t"
        "v_dot4_u32_u8 %[n4], %[c4], %[c4], 0\n\t"
        "v_dot4_u32_u8 %[p4], %[x], %[c4], 0\n\t"
        "v_dot4_u32_u8 %[n5], %[c5], %[c5], 0\n\t"
        "v_dot4_u32_u8 %[p5], %[x], %[c5], 0\n\t"
        "v_dot4_u32_u8 %[n6], %[c6], %[c6], 0\n\t"
        "v_dot4_u32_u8 %[p6], %[x], %[c6], 0\n\t"
        "v_dot4_u32_u8 %[n7], %[c7], %[c7], 0\n\t"
        "v_dot4_u32_u8 %[p7], %[x], %[c7], 0\n\t"
        "v_lshl_add_u32 %[n0], %[n0], 8, 0\n\t"
        "v_lshl_add_u32 %[n1], %[n1], 8, 4\n\t"
        "v_lshl_add_u32 %[n2], %[n2], 8, 8\n\t"
        "v_lshl_add_u32 %[n3], %[n3], 8, 12\n\t"
        "v_lshl_add_u32 %[n4], %[n4], 8, 16\n\t"
        "v_lshl_add_u32 %[n5], %[n5], 8, 20\n\t"
        "v_lshl_add_u32 %[n6], %[n6], 8, 24\n\t"
        "v_lshl_add_u32 %[n7], %[n7], 8, 28\n\t"
        "v_mad_i32_i24 %[n0], %[p0], %[ng], %[n0]\n\t"
        "v_mad_i32_i24 %[n1], %[p1], %[ng], %[n1]\n\t"
        "v_mad_i32_i24 %[n2], %[p2], %[ng], %[n2]\n\t"
        "v_mad_i32_i24 %[n3], %[p3], %[ng], %[n3]\n\t"
        "v_mad_i32_i24 %[n4], %[p4], %[ng], %[n4]\n\t"
        "v_mad_i32_i24 %[n5], %[p5], %[ng], %[n5]\n\t"
        "v_mad_i32_i24 %[n6], %[p6], %[ng], %[n6]\n\t"
        "v_mad_i32_i24 %[n7], %[p7], %[ng], %[n7]\n\t"
        "v_min3_i32 %[m0], %[n0], %[n1], %[n2]\n\t"
        "v_med3_i32 %[m1], %[n0], %[n1], %[n2]\n\t"
        "v_max3_i32 %[m2], %[n0], %[n1], %[n2]\n\t"
        "v_med3_i32 %[m2], %[m1], %[m2], %[n3]\n\t"
        "v_med3_i32 %[m1], %[m0], %[m1], %[n3]\n\t"
        "v_min_i32 %[m0], %[m0], %[n3]\n\t"
        "v_med3_i32 %[m2], %[m1], %[m2], %[n4]\n\t"
        "v_med3_i32 %[m1], %[m0], %[m1], %[n4]\n\t"
        "v_min_i32 %[m0], %[m0], %[n4]\n\t"
        "v_med3_i32 %[m2], %[m1], %[m2], %[n5]\n\t"
        "v_med3_i32 %[m1], %[m0], %[m1], %[n5]\n\t"
        "v_min_i32 %[m0], %[m0], %[n5]\n\t"
        "v_med3_i32 %[m2], %[m1], %[m2], %[n6]\n\t"
        "v_med3_i32 %[m1], %[m0], %[m1], %[n6]\n\t"
        "v_min_i32 %[m0], %[m0], %[n6]\n\t"
        "v_med3_i32 %[m2], %[m1], %[m2], %[n7]\n\t"
        "v_med3_i32 %[m1], %[m0], %[m1], %[n7]\n\t"
        "v_min_i32 %[m0], %[m0], %[n7]\n\t"
        : [n0] "=&v"(n0), [n1] "=&v"(n1), [n2] "=&v"(n2), [n3] "=&v"(n3), [n4] "=&v"(n4), [n5] "=&v"(n5),
          [n6] "=&v"(n6), [n7] "=&v"(n7), [p0] "=&v"(p0), [p1] "=&v"(p1), [p2] "=&v"(p2), [p3] "=&v"(p3),
          [p4] "=&v"(p4), [p5] "=&v"(p5), [p6] "=&v"(p6), [p7] "=&v"(p7), [m0] "=&v"(m0), [m1] "=&v"(m1),
          [m2] "=&v"(m2)
        : [x] "v"(x), [c0] "v"(ca.x), [c1] "v"(ca.y), [c2] "v"(ca.z), [c3] "v"(ca.w), [c4] "v"(cb.x),
          [c5] "v"(cb.y), [c6] "v"(cb.z), [c7] "v"(cb.w), [ng] "s"(neg2));
}
// one op per key: key = dot4(x', c) + a (26 ops)
__device__ __forceinline__ void cand8b(const unsigned x, const uint4 ca, const uint4 cb, const uint4 aa, const uint4 ab, int &m0, int &m1, int &m2)
{
    int n0, n1, n2, n3, n4, n5, n6, n7;
    asm volatile(
        "v_dot4_u32_u8 %[n0], %[x], %[c0], %[a0]\n\t"
        "v_dot4_u32_u8 %[n1], %[x], %[c1], %[a1]\n\t"
        "v_dot4_u32_u8 %[n2], %[x], %[c2], %[a2]\n\t"
        "v_dot4_u32_u8 %[n3], %[x], %[c3], %[a3]\n\t"
        "v_dot4_u32_u8 %[n4], %[x], %[c4], %[a4]\n\t"
        "v_dot4_u32_u8 %[n5], %[x], %[c5], %[a5]\n\t"
        "v_dot4_u32_u8 %[n6], %[x], %[c6], %[a6]\n\t"
        "v_dot4_u32_u8 %[n7], %[x], %[c7], %[a7]\n\t"
        "v_min3_i32 %[m0], %[n0], %[n1], %[n2]\n\t"
        "v_med3_i32 %[m1], %[n0], %[n1], %[n2]\n\t"
        "v_max3_i32 %[m2], %[n0], %[n1], %[n2]\n\t"
        "v_med3_i32 %[m2], %[m1], %[m2], %[n3]\n\t"
        "v_med3_i32 %[m1], %[m0], %[m1], %[n3]\n\t"
        "v_min_i32 %[m0], %[m0], %[n3]\n\t"
        "v_med3_i32 %[m2], %[m1], %[m2], %[n4]\n\t"
        "v_med3_i32 %[m1], %[m0], %[m1], %[n4]\n\t"
        "v_min_i32 %[m0], %[m0], %[n4]\n\t"
        "v_med3_i32 %[m2], %[m1], %[m2], %[n5]\n\t"
        "v_med3_i32 %[m1], %[m0], %[m1], %[n5]\n\t"
        "v_min_i32 %[m0], %[m0], %[n5]\n\t"
        "v_med3_i32 %[m2], %[m1], %[m2], %[n6]\n\t"
        "v_med3_i32 %[m1], %[m0], %[m1], %[n6]\n\t"
        "v_min_i32 %[m0], %[m0], %[n6]\n\t"
        "v_med3_i32 %[m2], %[m1], %[m2], %[n7]\n\t"
        "v_med3_i32 %[m1], %[m0], %[m1], %[n7]\n\t"
        "v_min_i32 %[m0], %[m0], %[n7]\n\t"
        : [n0] "=&v"(n0), [n1] "=&v"(n1), [n2] "=&v"(n2), [n3] "=&v"(n3), [n4] "=&v"(n4), [n5] "=&v"(n5),
          [n6] "=&v"(n6), [n7] "=&v"(n7), [m0] "=&v"(m0), [m1] "=&v"(m1), [m2] "=&v"(m2)
        : [x] "v"(x), [c0] "v"(ca.x), [c1] "v"(ca.y), [c2] "v"(ca.z), [c3] "v"(ca.w), [c4] "v"(cb.x),
          [c5] "v"(cb.y), [c6] "v"(cb.z), [c7] "v"(cb.w), [a0] "v"(aa.x), [a1] "v"(aa.y), [a2] "v"(aa.z), [a3] "v"(aa.w),
          [a4] "v"(ab.x), [a5] "v"(ab.y), [a6] "v"(ab.z), [a7] "v"(ab.w));
}
#define NITER 1024
template <int V>
__global__ __launch_bounds__(256) void knet(unsigned *out, unsigned seed)
{
    unsigned x = threadIdx.x * 2654435761u + seed;
    uint4 ca = make_uint4(x * 3, x * 5, x * 7, x * 11), cb = make_uint4(x * 13, x * 17, x * 19, x * 23);
    uint4 aa = make_uint4(x * 29, x * 31, x * 37, x * 41), ab = make_uint4(x * 43, x * 47, x * 53, x * 59);
    const int neg2 = -512 + (int)(seed & 0);
    unsigned acc = 0;
    for (int i = 0; i < NITER; ++i) {
        int m0, m1, m2;
        if (V == 0) cand8(x, ca, cb, neg2, m0, m1, m2);
        else cand8b(x, ca, cb, aa, ab, m0, m1, m2);
        acc += (unsigned)(m0 ^ m1 ^ m2);
        x += acc;  // the next iteration depends on this one (as a pixel loop does not: see V >= 2)
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int V>
__global__ __launch_bounds__(256) void knet4(unsigned *out, unsigned seed)
{  // four independent pixels per iteration, as in the kernel
    unsigned x = threadIdx.x * 2654435761u + seed;
    uint4 ca = make_uint4(x * 3, x * 5, x * 7, x * 11), cb = make_uint4(x * 13, x * 17, x * 19, x * 23);
    uint4 aa = make_uint4(x * 29, x * 31, x * 37, x * 41), ab = make_uint4(x * 43, x * 47, x * 53, x * 59);
    const int neg2 = -512 + (int)(seed & 0);
    unsigned acc = 0;
    for (int i = 0; i < NITER / 4; ++i) {
        int m0[4], m1[4], m2[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (V == 0) cand8(x + q, ca, cb, neg2, m0[q], m1[q], m2[q]);
            else cand8b(x + q, ca, cb, aa, ab, m0[q], m1[q], m2[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc += (unsigned)(m0[q] ^ m1[q] ^ m2[q]);
        x += 4;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

static unsigned *d;
static float time_launch(void (*kern)(unsigned *, unsigned), int blocks)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 2u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}
template <int OP>
void run(const char *name)
{
    printf("%-18s", name);
    for (int w : {1, 2, 4, 8}) {
        const float ms = time_launch(k<OP>, 256 * w);  // 256 CUs x (4 waves per block => 1 wave/SIMD per block)
        const double inst = (double)w * ITER * 8;      // wave-instructions per SIMD
        printf("  w%d: %5.2f cyc", w, ms * 1e6 / inst * 2.4);
    }
    {
        const float ms = time_launch(kdep<OP>, 256);
        printf("  | dep w1: %5.2f", ms * 1e6 / ((double)ITER * 8) * 2.4);
        const float ms8 = time_launch(kdep<OP>, 256 * 8);
        printf("  dep w8: %5.2f", ms8 * 1e6 / ((double)8 * ITER * 8) * 2.4);
    }
    printf("   (cycles @2.4GHz per wave-instruction per SIMD)\n");
}
template <int V>
void runnet(const char *name, int ops)
{
    printf("%-28s", name);
    for (int w : {1, 2, 4, 8}) {
        const float ms = time_launch(knet<V>, 256 * w);
        printf("  w%d: %6.1f cyc/px (%4.2f/op)", w, ms * 1e6 / ((double)w * NITER) * 2.4, ms * 1e6 / ((double)w * NITER) * 2.4 / ops);
    }
    printf("\n%-28s", "  4 independent px/iter");
    for (int w : {1, 2, 4, 8}) {
        const float ms = time_launch(knet4<V>, 256 * w);
        printf("  w%d: %6.1f cyc/px (%4.2f/op)", w, ms * 1e6 / ((double)w * NITER) * 2.4, ms * 1e6 / ((double)w * NITER) * 2.4 / ops);
    }
    printf("\n");
}
int main()
{
    hipMalloc(&d, 256 * 256 * 8 * 4);
    // warm the clocks
    for (int i = 0; i < 20; ++i) time_launch(k<5>, 256 * 8);
    run<0>("v_dot4_u32_u8");
    run<15>("v_dot4_u32_u8 acc");
    run<1>("v_med3_i32");
    run<16>("v_min3_i32");
    run<2>("v_mad_i32_i24");
    run<3>("v_min_i32");
    run<4>("v_lshl_add_u32");
    run<5>("v_add_u32");
    run<6>("v_mul_lo_u32");
    run<7>("v_fma_f32");
    run<8>("v_mul_u32_u24");
    run<9>("v_cndmask_b32");
    run<10>("v_dot2_i32_i16");
    run<11>("v_pk_sub_i16");
    run<12>("v_perm_b32");
    run<13>("v_bfe_u32");
    run<14>("v_sad_u8");
    run<17>("v_and_or_b32");
    run<18>("v_cmp_lt_u32");
    run<19>("v_pk_mul_lo_u16");
    run<20>("v_lshlrev_b32");
    run<21>("v_mov_b32");
    run<22>("v_pk_min_i16");
    run<23>("v_min_u32_dpp");
    runnet<0>("cand8 (50 ops)", 50);
    runnet<1>("cand8b one-op keys (26 ops)", 26);
    return 0;
}
