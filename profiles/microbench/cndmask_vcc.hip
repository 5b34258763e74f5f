// Is v_cndmask_b32 reading VCC really ~5x slower than reading another SGPR pair on gfx950?  (valu_rate*.hip say 22.8 cycles
// against 4.6.)  Variants: VOP2 (implicit vcc), VOP3 naming vcc, VOP3 with s[20:21], and the compiler's usual pair
// v_cmp -> mask -> v_cndmask with the mask in vcc or in an SGPR pair; v_addc_co_u32 (carry-in from vcc) for comparison.
// hipcc --offload-arch=gfx950 -O3 cndmask_vcc.hip -o cndmask_vcc
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
#define REP8(X) X(a0) X(a1) X(a2) X(a3) X(a4) X(a5) X(a6) X(a7)
template <int OP> __global__ __launch_bounds__(256) void k(unsigned *out, unsigned seed)
{
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    const unsigned b = blockIdx.x * 2654435761u + seed;
    asm volatile("s_mov_b32 s20, 0x55555555\n\ts_mov_b32 s21, 0x33333333\n\tv_cmp_lt_u32 vcc, %0, %1" ::"v"(a0), "v"(b) : "vcc", "s20", "s21");
    for (int i = 0; i < ITER; ++i) {
#define V0(a) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc", "s20", "s21");
#define V1(a) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc", "s20", "s21");
#define V2(a) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a) : "v"(b) : "vcc", "s20", "s21");
#define V3(a) asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc", "s20", "s21");
#define V4(a) asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a) : "v"(b) : "vcc", "s20", "s21");
#define V5(a) asm volatile("v_addc_co_u32_e32 %0, vcc, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc", "s20", "s21");
#define V6(a) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a) : "v"(b) : "vcc", "s20", "s21");
#define V7(a) asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1" : "+v"(a) : "v"(b) : "vcc", "s20", "s21");
#define V8(a) asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %1" : "+v"(a) : "v"(b) : "vcc", "s20", "s21");
#define V9(a) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n\ts_nop 0" : "+v"(a) : "v"(b) : "vcc", "s20", "s21");
        if (OP == 0) { REP8(V0) }
        if (OP == 1) { REP8(V1) }
        if (OP == 2) { REP8(V2) }
        if (OP == 3) { REP8(V3) }
        if (OP == 4) { REP8(V4) }
        if (OP == 5) { REP8(V5) }
        if (OP == 6) { REP8(V6) }
        if (OP == 7) { REP8(V7) }
        if (OP == 8) { REP8(V8) }
        if (OP == 9) { REP8(V9) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
static unsigned *d;
template <int OP> float timek(int blocks)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 1);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(e0);
        k<OP><<<blocks, 256>>>(d, 2);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}
template <int OP> void run(const char *name, int per)
{
    printf("%-52s", name);
    for (int w : {1, 2, 4, 8}) {
        const float ms = timek<OP>(256 * w);
        printf("  w%d: %6.2f", w, ms * 1e6 / ((double)w * ITER * 8) * 2.4);
    }
    printf("   cycles @2.4GHz per wave per SIMD for the %d-instruction unit\n", per);
}
int main()
{
    (void)hipMalloc(&d, 256 * 256 * 8 * 4);
    for (int i = 0; i < 20; ++i) timek<6>(2048);
    run<6>("v_min_u32", 1);
    run<0>("v_cndmask_b32_e32 (VOP2, implicit vcc)", 1);
    run<1>("v_cndmask_b32_e64 ..., vcc", 1);
    run<2>("v_cndmask_b32_e64 ..., s[20:21]", 1);
    run<9>("v_cndmask_b32_e64 ..., s[20:21] ; s_nop 0", 2);
    run<7>("v_cmp_lt_u32_e32 vcc", 1);
    run<8>("v_cmp_lt_u32_e64 s[20:21]", 1);
    run<3>("v_cmp_lt_u32_e32 vcc ; v_cndmask_b32_e32 vcc", 2);
    run<4>("v_cmp_lt_u32_e64 s[20:21] ; v_cndmask_b32_e64 s[20:21]", 2);
    run<5>("v_addc_co_u32_e32 vcc", 1);
    return 0;
}
