// Wave-level helpers shared by the one-wave-per-frame (serpentine) kernels of ediff.hip and vardiff.hip.
#pragma once
#include "dp_internal.h"

namespace dp {

// Minimum of a non-negative float over the 64 lanes, returned to every lane.  Non-negative floats order like
// their bit patterns, so the reduction runs on unsigned integers with one fused v_min_u32_dpp per step: inclusive
// prefix minimum inside each row of 16 lanes (row_shr 1, 2, 4, 8), then row_bcast:15 / row_bcast:31 across the
// rows; lane 63 ends up with the minimum.  A lane without a source, or outside the row mask, keeps its value.
// (A VGPR written by a VALU instruction needs two wait states before a DPP read.)
__device__ __forceinline__ float wave_min_to_all(const float vf)
{
    uint32_t v = __float_as_uint(vf);
    asm volatile(
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 0"
        : "+v"(v));
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)v, 63));
}

// Sum of a 32-bit value over the 64 lanes, valid in lane 63 only: the same row_shr / row_bcast ladder with
// v_add_u32_dpp (a lane without a source keeps its value, which is what a prefix sum wants there).
__device__ __forceinline__ uint32_t wave_sum_to_lane63(uint32_t v)
{
    asm volatile(
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 0"
        : "+v"(v));
    return v;
}

// Four sums at once: the four ladders interleaved, so that each v_add_u32_dpp finds its source written three instructions
// earlier and the wait states between a VALU write and a DPP read are filled with work instead of s_nop.  Lane 63 only.
__device__ __forceinline__ void wave_sum4_to_lane63(uint32_t &a, uint32_t &b, uint32_t &c, uint32_t &d)
{
#define DP_STEP4(ctrl)                                        \
    "v_add_u32_dpp %0, %0, %0 " ctrl "\n\t"                   \
    "v_add_u32_dpp %1, %1, %1 " ctrl "\n\t"                   \
    "v_add_u32_dpp %2, %2, %2 " ctrl "\n\t"                   \
    "v_add_u32_dpp %3, %3, %3 " ctrl "\n\t"
    asm volatile("s_nop 1\n\t" DP_STEP4("row_shr:1 row_mask:0xf bank_mask:0xf") DP_STEP4("row_shr:2 row_mask:0xf bank_mask:0xf")
                     DP_STEP4("row_shr:4 row_mask:0xf bank_mask:0xf") DP_STEP4("row_shr:8 row_mask:0xf bank_mask:0xf")
                         DP_STEP4("row_bcast:15 row_mask:0xa bank_mask:0xf") DP_STEP4("row_bcast:31 row_mask:0xc bank_mask:0xf") "s_nop 0"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef DP_STEP4
}

// Value of lane (lane ^ mask) -- the butterfly exchange of wave-wide reductions over pairs (value, position), for which
// no fused DPP minimum exists; one ds_bpermute_b32 (LDS crossbar, no memory).
__device__ __forceinline__ uint32_t lane_xor_u32(const uint32_t v, const int mask)
{
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    return (uint32_t)__builtin_amdgcn_ds_bpermute((lane ^ mask) << 2, (int)v);
}
__device__ __forceinline__ float lane_xor_f32(const float v, const int mask)
{
    return __uint_as_float(lane_xor_u32(__float_as_uint(v), mask));
}

}  // namespace dp
