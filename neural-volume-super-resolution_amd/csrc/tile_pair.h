// Shared machinery of the two-tiles-per-wave kernels (render2.hip, render_bwd2.hip): one wave per SIMD, a 2 x 64 KB LDS ring of weight
// chunks, and the MFMA block loop that calls a side-work functor in every 64-cycle MFMA shadow.
#pragma once
#include "decode_core.h"

#ifndef R2_ABLATE
#define R2_ABLATE 0      // timing experiments only: 1 no compositing, 2 no exposed Y epilogue, 4 no exposed prologue gather, 8 no ring barrier
#endif

namespace nvsr {

__device__ __forceinline__ void ring2_sync() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if !(R2_ABLATE & 8)
    __syncthreads();
#endif
}

constexpr int TPB2 = 256;                         // 4 waves, one per SIMD, one workgroup per CU
constexpr int NW2 = TPB2 / 64;
constexpr int RAYS2 = NW2 * 64;                   // 256 rays per workgroup: 2 tiles x 32 rays per wave
constexpr int SLOT2 = 16384;                      // 64 KB ring slot
constexpr int LDS2_SMALL = 2 * SLOT2;             // heads / biases behind the two ring slots

struct Ring2 {
    const float* packed;
    float* lds;
    int slot;
    int wave, lane;
    unsigned voff;
};

template <int BLOCKS>
__device__ __forceinline__ const float* ring2_issue(Ring2& rs, int packed_off) {
    float* dst = rs.lds + rs.slot * SLOT2;
    stage_chunk<NW2, BLOCKS>(rs.packed + packed_off, dst, rs.voff, rs.wave);
    rs.slot ^= 1;
    return dst;
}

// NG groups of 4 MFMAs into acc[g & 3] (same fragment order as decode_core.h::mfma_groups) with side work after each group.
// In-order issue: while MFMA k executes (64 cycles) the wave may issue other instructions, but it blocks at MFMA k+1 until the matrix
// pipe is free -- so whatever sits between two MFMAs has exactly ONE 64-cycle shadow.  Side work is therefore called after EVERY MFMA
// (slot j = 0..3 of group g) in slices of a few VALU instructions, and an LDS read issued in one slot is consumed two or three slots
// later, never in the slot that issued it.
// (`#pragma unroll` gives up silently when a side functor makes the body too large -- "loop not unrolled" -- and a runtime group index
//  then sends the whole tile state to scratch: check the remark output after touching a block.  A template-recursive loop that cannot
//  fail to unroll was tried; the register allocator then spills 80 VGPRs where this form spills none.)
template <int NG, int NACC = 4, class BFn, class Side>
__device__ __forceinline__ void mfma_block(const float* wl, int lane, f32x16 (&acc)[NACC], BFn b, Side side) {
    const f32x4* wv = reinterpret_cast<const f32x4*>(wl) + lane;
    f32x4 a = wv[0];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int ib = g % NACC;
        acc[ib] = mfma32(a[0], b(g, 0), acc[ib]);
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 an = wv[(g + 1 < NG ? g + 1 : g) * 64];
        side(g, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 1; j < 4; ++j) {
            acc[ib] = mfma32(a[j], b(g, j), acc[ib]);
            __builtin_amdgcn_sched_barrier(0);
            side(g, j);
            __builtin_amdgcn_sched_barrier(0);
        }
        a = an;
    }
}

}  // namespace nvsr
