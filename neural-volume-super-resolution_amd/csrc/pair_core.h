// Helpers shared by the two-tiles-per-wave limb kernels: the fused render pass (render3.hip) and the training forward on tile pairs
// (decode_pair.hip) -- rolling plane gathers, the LDS-DMA weight ring with its resident region, the bias + ReLU slices.
#pragma once
#include <type_traits>
#include <utility>

#ifndef R3_DMA_AUX
#define R3_DMA_AUX 0      // cache-policy bits of the weight copies (experiment: 2 = nt, 1 = sc0, 16 = sc1)
#endif
#ifndef R3_ABLATE
#define R3_ABLATE 0   // timing experiments (wrong results): 1 no gather loads, 2 / 4 see gather_roll, 32 no weight copies, 64 no ring barriers, 128 no A-fragment reads, 1024 ReLU reads VGPRs (no v_accvgpr_read), 2048 no bias + ReLU, 4096 no limb split
#endif
#include "limb_core.h"
#include "side_work.h"

#ifndef R3_GATHER_NT
#define R3_GATHER_NT 0
#endif
#ifndef R3_STAMP
#define R3_STAMP 0    // debug builds: raw_out[ray, s = 0..1, :] of tile X = cycles per sample spent in 7 sections of the step (tools/limb_stamp.py)
#endif
#if R3_STAMP == 4
#define R3_MARK(i)
#define R3_MARKB(i)
#define R3_RESET
#define R3_MARKH(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamp[i] += (float)(t_ - tprev); tprev = t_; __builtin_amdgcn_sched_barrier(0); }
#define R3_RESETH { tprev = __builtin_amdgcn_s_memtime(); }
#elif R3_STAMP == 3
#define R3_MARKH(i)
#define R3_RESETH
#define R3_MARK(i)
#define R3_MARKB(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamp[i] += (float)(t_ - tprev); tprev = t_; __builtin_amdgcn_sched_barrier(0); }
#define R3_RESET { tprev = __builtin_amdgcn_s_memtime(); }
#elif R3_STAMP
#define R3_MARKB(i)
#define R3_RESET
#define R3_MARKH(i)
#define R3_RESETH
#define R3_MARK(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamp[i] += (float)(t_ - tprev); tprev = t_; __builtin_amdgcn_sched_barrier(0); }
#else
#define R3_MARK(i)
#define R3_MARKB(i)
#define R3_RESET
#define R3_MARKH(i)
#define R3_RESETH
#endif
#ifndef R3_RELU_PK
#define R3_RELU_PK 0      // f16-limb ReLU: 1 = pairs through v_pk_fma_f32, 0 = two v_fma_f32.  Same-box A/B: the packed form is SLOWER (88.1 vs 83.6 ms
#endif                    // fine pass): a v_pk_fma_f32 costs the wave more issue time than the two v_fma_f32 it replaces
#ifndef R3_BLEND_PK
#define R3_BLEND_PK 0     // bilinear blends of the f16-limb kernels through v_pk_mul_f32 / v_pk_fma_f32 (A/B)
#endif
#ifndef R3_NO_VIEW_HOIST
#define R3_NO_VIEW_HOIST 1
#endif

namespace nvsr {

// per-ray constants of a tile: ro, rd, |rd|, near | view-plane taps | far.  3 bf16 limbs: 20 floats per ray (far = rc[16], 3 spare);
// f16 limbs: 16 floats per ray + far in its own array (3 KB less: what the resident weights below need to fit)
// f16 limbs: the weights of the first 9 K-blocks of a step -- the view plane's and planes 0 and 1's share of rgb layer 0, 72 KB -- stay in LDS
// for the whole launch.  Those are the blocks that issue the plane gathers: weight copies issued beside the gathers go through the same
// texture path and cost the step ~9 k of its 94 k cycles (stamps with -DR3_ABLATE=32); resident, the gather blocks issue no copy at all and the
// step streams 432 instead of 504 KB.  (3 bf16 limbs: 108 KB would not fit beside the 96 KB ring.)
template <int LIMBS>
struct Lds3 {
    static constexpr int SLOT = 4 * kb_words(LIMBS);                 // words: 48 KB (3 limbs) / 32 KB
    static constexpr int SMALL = 2 * SLOT;
    static constexpr int RAYS = SMALL + SMALL_FLOATS;
    // ray cache: RAY_FLOATS per ray (origin, direction, norm, near) + TAP_FLOATS (the view plane's four tap offsets and weights).  f16 limbs: the
    // taps live in a region of their own, 2 KB per wave, that is dead after the prologue -- the wave's accumulator bounce slot (relu_bias_step)
    static constexpr int RAY_FLOATS = LIMBS == 2 ? 8 : 20;
    static constexpr int TAP_FLOATS = LIMBS == 2 ? 8 : 0;
    static constexpr int VTAPS = RAYS + RAYS2 * RAY_FLOATS;
    static constexpr int FAR = LIMBS == 2 ? VTAPS + RAYS2 * TAP_FLOATS : -1;
    static constexpr int RES = VTAPS + RAYS2 * TAP_FLOATS + (LIMBS == 2 ? RAYS2 : 0);
    static constexpr int RES_KB = LIMBS == 2 ? 9 : 0;
    static constexpr int TOTAL = RES + RES_KB * kb_words(LIMBS);
};
static_assert(Lds3<3>::TOTAL * 4 <= 160 * 1024 && Lds3<2>::TOTAL * 4 <= 160 * 1024, "LDS budget");

struct Tile3 {
    f32x16 acc[4];   // layer accumulators (AGPRs), written by MFMAs only
    f32x16 act[4];   // max(acc + bias, 0) of the finished layer (VGPRs)
    float D[HALF_C], F[HALF_C];
    float V[HALF_C];  // view-plane features: the same for every sample of a ray, gathered once
    float T, cr, cg, cb, dep, ac, zc, zn;
    float raw[4];
};

// four tap buffers (96 registers; the activation sets are dead while planes are gathered): every load of a gather is issued in the
// first slots of a block and blended in its last quarter
struct RawTaps4 { f32x4 r[4][HALF_C / 4]; };
constexpr int GATHER_STEPS = 12 + HALF_C;
__device__ __forceinline__ void gather4_load(int k, const GatherJob& job, int h, RawTaps4& rt) {           // k 0..11: tap k/3, 2 loads
    const int tap = k / 3, i0 = 2 * (k % 3);
#if R3_ABLATE & 1
    rt.r[tap][i0] = f32x4{job.t.nw, job.t.ne, job.t.sw, job.t.se};
    rt.r[tap][i0 + 1] = f32x4{job.t.se, job.t.ne, job.t.sw, job.t.nw};
    return;
#endif
    const int off = tap == 0 ? job.t.o00 : tap == 1 ? job.t.o01 : tap == 2 ? job.t.o10 : job.t.o11;
    const f32x4* p = reinterpret_cast<const f32x4*>(job.plane + off + HALF_C * h);
    rt.r[tap][i0] = p[i0];
    rt.r[tap][i0 + 1] = p[i0 + 1];
}
__device__ __forceinline__ void gather4_blend(int c, const GatherJob& job, const RawTaps4& rt, float (&F)[HALF_C]) {   // channel c
    const int i = c >> 2, j = c & 3;
    F[c] = fmaf(rt.r[3][i][j], job.t.se, fmaf(rt.r[2][i][j], job.t.sw, fmaf(rt.r[1][i][j], job.t.ne, rt.r[0][i][j] * job.t.nw)));
}

// Rolling gather.  Block j issues the 24 loads of gather j, one per 3 of the block's 72 virtual steps (a wave64 dwordx4 load with 64
// distinct addresses holds the vector-memory issue for ~80 cycles while the 4 waves of a CU gather together: issued back to back they stall
// the MFMA stream), tap by tap (the 6 loads of a tap read the same cache lines).  The blend is accumulated per tap,
//     F = T0 nw;  F = fma(T1, ne, F);  F = fma(T2, sw, F);  F = fma(T3, se, F)          (the same operations as gather24)
// 54 steps behind the loads: pass 0 of gather j in the last quarter of block j, passes 1..3 in the first three quarters of block j + 1,
// each just before the next gather's loads of that tap reuse the registers.
template <int NS, bool LOADS, bool BLENDS, bool PK = false>
__device__ __forceinline__ void gather_roll(int slot, const GatherJob& jl, float (&Fl)[HALF_C], const GatherJob& jb, float (&Fb)[HALF_C], int h,
                                            RawTaps4& rt) {
    spread<72, 0, NS>(slot, [&](int v) {
        if (LOADS && v % 3 == 0) {
            const int tap = (v / 3) / 6, piece = (v / 3) % 6;
#if R3_ABLATE & 1
            rt.r[tap][piece] = f32x4{jl.t.nw, jl.t.ne, jl.t.sw, jl.t.se};
#else
            int off = tap == 0 ? jl.t.o00 : tap == 1 ? jl.t.o01 : tap == 2 ? jl.t.o10 : jl.t.o11;
#if R3_ABLATE & 2
            off = __builtin_amdgcn_readfirstlane(off);          // every lane reads the first ray's texel (timing experiment)
#endif
#if R3_ABLATE & 4
            off = off & 0xffff;                                  // all gathers inside the first 256 KB of the plane (timing experiment)
            off -= off % 48;
#endif
#if R3_GATHER_NT            // timing experiment: non-temporal plane gathers (do they leave more of the L2 to the weight stream?)
            rt.r[tap][piece] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(jl.plane + off + HALF_C * h) + piece);
#else
            rt.r[tap][piece] = reinterpret_cast<const f32x4*>(jl.plane + off + HALF_C * h)[piece];
#endif
#endif
        }
        const int w = v / 18, u = v % 18;
        if constexpr (PK) {
            // f16-limb kernels: channel pairs through v_pk_mul_f32 / v_pk_fma_f32 (12 pairs over the 18 steps of a quarter)
            if (u % 3 != 2) {
                const int c = 2 * ((u / 3) * 2 + u % 3);
                if (LOADS && w == 3) {
                    f32x2_t m = f32x2_t{rt.r[0][c >> 2][c & 3], rt.r[0][c >> 2][(c & 3) + 1]} * f32x2_t{jl.t.nw, jl.t.nw};
                    Fl[c] = m[0]; Fl[c + 1] = m[1];
                }
                if (BLENDS && w < 3) {
                    const float wt = w == 0 ? jb.t.ne : w == 1 ? jb.t.sw : jb.t.se;
                    f32x2_t f = f32x2_t{Fb[c], Fb[c + 1]};
                    f32x2_t m = __builtin_elementwise_fma(f32x2_t{rt.r[w + 1][c >> 2][c & 3], rt.r[w + 1][c >> 2][(c & 3) + 1]}, f32x2_t{wt, wt}, f);
                    Fb[c] = m[0]; Fb[c + 1] = m[1];
                }
            }
        } else {
#pragma unroll
        for (int c = (u * 4) / 3; c < ((u + 1) * 4) / 3; ++c) {                         // 24 channels over the 18 steps of a quarter
            if (LOADS && w == 3) Fl[c] = rt.r[0][c >> 2][c & 3] * jl.t.nw;
            if (BLENDS && w < 3) Fb[c] = fmaf(rt.r[w + 1][c >> 2][c & 3], w == 0 ? jb.t.ne : w == 1 ? jb.t.sw : jb.t.se, Fb[c]);
        }
        }
    });
}

// ring wait with N younger vector-memory operations allowed in flight (vmcnt counts in issue order: the chunk issued before them has landed)
template <int N>
__device__ __forceinline__ void ring3_sync() {
#if R3_ABLATE & 256       // timing experiment: the ring waits do not wait for the copies
#elif R3_ABLATE & 9
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
#if !(R3_ABLATE & 64)      // timing experiment: no workgroup barrier at the ring waits
    __syncthreads();
#endif
}

// act = max(acc + bias, 0): 64 elements in 68 steps (a bias quad is read 4 steps before its first use)
constexpr int RELU_STEPS = 68;
#ifndef R3_BOUNCE
#define R3_BOUNCE 0       // f16-limb ReLU: 1 = accumulators reach the VALU through LDS (ds_write_b128 straight from AGPRs, ds_read_b128 back) instead of
#endif                    // one v_accvgpr_read per element: 2 LDS instructions per 4 elements replace 4 VALU instructions (7 548 -> 7 293 VALU
                          // in the fine-pass kernel).  Same-box A/B: SLOWER, fine pass 79.6 -> 83.6 ms, training forward S=128 0.433 -> 0.452 ms:
                          // 512 more 1-KB LDS transfers per wave and step queue behind the fragment reads, and the 176 extra waits cost an
                          // issue slot each
typedef __attribute__((address_space(3))) f32x4* lds_f32x4_p;
struct BiasPend4 { f32x4 v[2]; f32x4 a[2]; lds_f32x4_p slot; };
__device__ __forceinline__ lds_f32x4_p bounce_slot(float* p) { return (lds_f32x4_p)p; }
// LIMBS = 2 (f16 limbs): act = relu(acc 2^-SW + bias 2^SX) must let a NaN through -- an operand beyond the f16 range turns a layer's
// accumulators into NaNs (the matrix pipe always returns the default NaN 0xFFC00000, whatever the sign of a NaN it was fed), and v_max_f32
// would return its non-NaN operand: the overflow would render as a finite, wrong pixel.  So the ReLU is an INTEGER max on the bits (equal to
// max(x, 0) for every number: negative floats are negative integers; a POSITIVE NaN is a large positive integer and survives), and the FMA in
// front of it computes (-acc) (-2^-SW) + bias -- the same value, but the source-negation modifier turns the pipe's negative NaN into a
// positive one.  `nsc` = {-2^-SW, -2^-SW} in a scalar register pair (opaque to the compiler, which would fold the two negations away).
template <int LIMBS>
__device__ __forceinline__ void relu_bias_step(int k, const float* bias, int h, const f32x16 (&acc)[4], f32x16 (&act)[4], BiasPend4& pend, f32x2_t nsc) {
    if ((k & 3) == 0 && k < 64) pend.v[(k >> 2) & 1] = *reinterpret_cast<const f32x4*>(bias + (k >> 2) * 8 + h * 4);
#if R3_BOUNCE
    if constexpr (LIMBS == 2) {
        // quad j = k / 4 leaves the accumulators at step 4 j, comes back at 4 j + 2 and is used at 4 j + 4 .. 4 j + 7 (one wave's LDS operations
        // execute in order: the slot is free again when the next quad is written)
        if ((k & 3) == 0 && k < 64) {
            const int q = k >> 2;
            *pend.slot = f32x4{acc[q >> 2][4 * (q & 3)], acc[q >> 2][4 * (q & 3) + 1], acc[q >> 2][4 * (q & 3) + 2], acc[q >> 2][4 * (q & 3) + 3]};
        }
        if ((k & 3) == 2 && k < 66) {
            asm volatile("" : "+v"(pend.slot));        // (opaque: no store-to-load forwarding, which would be the v_accvgpr_read again)
            pend.a[(k >> 2) & 1] = *pend.slot;
        }
        if (k >= 4) {
            const int r = k - 4;
            float v;
            asm("v_fma_f32 %0, -%1, %2, %3\n\tv_max_i32 %0, 0, %0" : "=v"(v) : "v"(pend.a[(r >> 2) & 1][r & 3]), "s"(nsc[0]), "v"(pend.v[(r >> 2) & 1][r & 3]));
            act[r >> 4][r & 15] = v;
        }
        return;
    }
#endif
#if R3_ABLATE & 2048      // (one element of every accumulator block keeps the MFMAs alive)
    if (k >= 4 && ((k - 4) & 15) != 0) return;
#endif
    if (k >= 4) {
        const int r = k - 4;
        if constexpr (LIMBS == 2) {
#if R3_ABLATE & 1024
            float v;
            asm("v_fma_f32 %0, -%1, %2, %3\n\tv_max_i32 %0, 0, %0" : "=v"(v) : "v"((r & 15) == 0 ? acc[r >> 4][0] : act[r >> 4][r & 15]), "s"(nsc[0]), "v"(pend.v[(r >> 2) & 1][r & 3]));
            act[r >> 4][r & 15] = v;
#elif R3_RELU_PK
            if (r & 1) {
                const int q = r - 1;
                const f32x2_t a = f32x2_t{acc[q >> 4][q & 15], acc[r >> 4][r & 15]}, b = f32x2_t{pend.v[(q >> 2) & 1][q & 3], pend.v[(r >> 2) & 1][r & 3]};
                f32x2_t v;
                asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(v) : "v"(a), "s"(nsc), "v"(b));
                act[q >> 4][q & 15] = __int_as_float(max(__float_as_int(v[0]), 0));
                act[r >> 4][r & 15] = __int_as_float(max(__float_as_int(v[1]), 0));
            }
#else
            float v;            // (written out: the compiler would fold the two negations away -- and both instructions in ONE statement: behind
                                // an asm statement hipcc pads an s_nop in front of a dependent instruction, 600 of them per step)
            asm("v_fma_f32 %0, -%1, %2, %3\n\tv_max_i32 %0, 0, %0" : "=v"(v) : "v"(acc[r >> 4][r & 15]), "s"(nsc[0]), "v"(pend.v[(r >> 2) & 1][r & 3]));
            act[r >> 4][r & 15] = v;
#endif
        } else act[r >> 4][r & 15] = fmaxf(acc[r >> 4][r & 15] + pend.v[(r >> 2) & 1][r & 3], 0.0f);
    }
}

// Weight ring.  The chunks are copied with `buffer_load_dwordx4 ... lds` (MUBUF LDS-DMA), not with global_load_lds: hipcc's wait-count
// pass treats the FLAT-encoded form as an access to both memories ("pending flat") and answers the first vector-memory dependence after
// it with s_waitcnt vmcnt(0) -- which here would wait for the chunk just issued and for every gather load in flight.
template <int LIMBS>
struct Ring3 {
    __amdgpu_buffer_rsrc_t rsrc;   // fragment region of the packed blob
    unsigned* lds;
    int slot;
    int wave, lane;
    unsigned voff;
};
template <int LIMBS, int NKB>
__device__ __forceinline__ const unsigned* ring3_issue(Ring3<LIMBS>& rs, int kb0) {
    unsigned* dst = rs.lds + rs.slot * Lds3<LIMBS>::SLOT;
    constexpr int BLOCKS = NKB * 4 * LIMBS;                 // 1-KiB pieces, NW2 per round
    static_assert(BLOCKS % NW2 == 0, "chunk must split evenly over the waves");
#pragma unroll
    for (int i = 0; i < BLOCKS / NW2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs.rsrc, (__attribute__((address_space(3))) void*)(dst + (i * NW2 + rs.wave) * 256), 16,
                                             (int)rs.voff, kb0 * kb_words(LIMBS) * 4 + i * (NW2 * 1024), 0, R3_DMA_AUX);
    rs.slot ^= 1;
    return dst;
}

// NKB K-blocks from kb0 on into the resident region `dst` (once per launch; waited for by the first ring wait)
template <int LIMBS, int NKB>
__device__ __forceinline__ void ring3_load_resident(const Ring3<LIMBS>& rs, unsigned* dst, int kb0) {
    constexpr int BLOCKS = NKB * 4 * LIMBS;                 // 1-KiB pieces, NW2 per round
    static_assert(BLOCKS % NW2 == 0, "region must split evenly over the waves");
#pragma unroll
    for (int i = 0; i < BLOCKS / NW2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs.rsrc, (__attribute__((address_space(3))) void*)(dst + (i * NW2 + rs.wave) * 256), 16,
                                                 (int)rs.voff, kb0 * kb_words(LIMBS) * 4 + i * (NW2 * 1024), 0, R3_DMA_AUX);
}

// The same copy issued piece by piece from the gaps of the block that follows the ring wait: a chunk is 36-48 KB, the 4 waves push it
// through the texture path at its 64 B/clk -- issued as a burst that is ~700 cycles in which the wave issues nothing else (12 chunks of
// hidden layers per sample: 8 k cycles); one piece every third gap overlaps with the MFMAs.
template <int LIMBS>
__device__ __forceinline__ unsigned* ring3_take(Ring3<LIMBS>& rs) {
    unsigned* dst = rs.lds + rs.slot * Lds3<LIMBS>::SLOT;
    rs.slot ^= 1;
    return dst;
}
template <int LIMBS, int NKB>
__device__ __forceinline__ void dma_side(int slot, const Ring3<LIMBS>& rs, unsigned* dst, int kb0) {
    constexpr int PIECES = NKB * LIMBS;                     // per wave
#if R3_ABLATE & 32         // timing experiment: no weight copies (stale LDS contents)
    return;
#endif
    if (slot % 3 == 2 && slot / 3 < PIECES) {
        const int i = slot / 3;
#if R3_ABLATE & 512        // timing experiment: the same number of copy instructions, a quarter of the bytes (dword instead of dwordx4)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs.rsrc, (__attribute__((address_space(3))) void*)(dst + (i * NW2 + rs.wave) * 256), 4,
                                                 (int)rs.voff, kb0 * kb_words(LIMBS) * 4 + i * (NW2 * 1024), 0, R3_DMA_AUX);
#else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs.rsrc, (__attribute__((address_space(3))) void*)(dst + (i * NW2 + rs.wave) * 256), 16,
                                                 (int)rs.voff, kb0 * kb_words(LIMBS) * 4 + i * (NW2 * 1024), 0, R3_DMA_AUX);
#endif
    }
}
// the last piece of a chunk is issued in slot 3 * PIECES - 1 <= 35 of a block that also issues a gather: at least 12 of that gather's loads
// and the 24 of the next block's are younger
constexpr int YOUNGER_THAN_CHUNK = 32;

}  // namespace nvsr
