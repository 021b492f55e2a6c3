// f32 GEMM on the bf16 matrix pipe: operands split into bf16 limbs (render3.hip).
//
// v_mfma_f32_32x32x2_f32 retires 32 MAC / cycle / SIMD, v_mfma_f32_32x32x16_bf16 512.  An f32 value is EXACTLY the sum of three bf16
// values taken by truncation (8 + 8 + 8 significant bits):  x = xh + xm + xl, and a product of two bf16 values is exact in f32.  So
//     W x  =  Wh(xh + xm + xl) + Wm(xh + xm) + Wl xh   +  [Wm xl + Wl xm + Wl xl]
// With truncation limbs |m| < 2^-7 |v| and |l| < 2^-15 |v|, so the dropped bracket is < (2^-22 + 2^-22 + 2^-30) |W||x| = 2^-21 + 2^-30 per
// product -- up to four f32 roundings, and ONE-SIDED (the limbs carry the operand's sign: the bracket has the sign of W x and adds up
// over K).  Six bf16 MFMAs with f32 accumulation give the product at 16/6 = 2.7x the rate of the f32 MFMA (LIMBS = 3); the running sum
// takes one f32 rounding per MFMA (6 K / 16 of them).  Measured worst case at K = 192 on adversarial mantissas: 9.8e-7 of sum |W||x|
// (exact-f32 kernel: 5.6e-7), tests/test_hip_round2.py::test_limb_error_bound -- close to, not bit-grade, f32.  LIMBS = 2 keeps 16 bits
// per operand (second limb rounded to nearest, 3 products, error <= 2^-15 |W||x| per product) at 5.3x.
//
// Fragment layout: A = weights [32 out x 16 in], B = activations [16 in x 32 points].  Lane l = (m | n = l & 31, h = l >> 5) holds the
// 8 k-values 8h .. 8h+7 of its row / column as 4 words {bf16 k even (low half), bf16 k odd (high half)}.  The C/D layout of a 32x32
// tile (register r of lane (n, h) = row 8(r>>2) + 4h + (r&3)) makes the 8 registers acc[ib][8q .. 8q+7] of a lane the B operand of
// K-block (ib, q) of the next layer once the weights are packed with the same k-order -- layers chain through registers.
#pragma once
#include "decode_core.h"

namespace nvsr {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

// (always_inline: inside a very large kernel the inliner otherwise leaves these as CALLS, and an operand index that is not a compile-time
//  constant turns every fragment selection into a chain of v_cndmask over a merged register array)
#define NVSR_CX __host__ __device__ __attribute__((always_inline)) constexpr
// side-work lambdas take their step index as an argument and use it as a register index or an instruction immediate: they must be inlined
// whatever their size (left to its cost model, the inliner keeps the larger ones as calls and the index stops being a constant)
#define NVSR_INL __attribute__((always_inline))
NVSR_CX int limb_products(int limbs) { return limbs == 3 ? 6 : 3; }
// product p of a (weight limb, activation limb) set, small terms first
NVSR_CX int limb_w(int limbs, int p) { return limbs == 3 ? (p < 3 ? 0 : p < 5 ? 1 : 2) : (p < 2 ? 0 : 1); }
NVSR_CX int limb_x(int limbs, int p) { return limbs == 3 ? (p == 0 ? 2 : p == 1 ? 1 : p == 2 ? 0 : p == 3 ? 1 : 0) : (p == 0 ? 1 : 0); }

// K-blocks (16 input channels each) of the decoder in consumption order
constexpr int KB_RGB0 = 0;          // 4 planes x 3
constexpr int KB_RGB1 = 12;         // 3 layers x 8
constexpr int KB_DEN0 = 36;         // 3
constexpr int KB_DEN1 = 39;         // 3 layers x 8
constexpr int KB_TOTAL = 63;
constexpr int kb_words(int limbs) { return 4 * limbs * 256; }                 // [ob 0..3][limb][lane][4 words]
constexpr int P_LIMB3 = NVSR_DECODER_PACKED_F32_FLOATS;                       // fragment regions behind the f32 blob
constexpr int P_LIMB2 = P_LIMB3 + KB_TOTAL * kb_words(3);
static_assert(P_LIMB2 + KB_TOTAL * kb_words(2) == NVSR_DECODER_PACKED_FLOATS, "packed blob size");
constexpr int limb_region(int limbs) { return limbs == 3 ? P_LIMB3 : P_LIMB2; }

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// {bf16 trunc(e1), bf16 trunc(e0)}: the high halves of two f32 registers in one v_perm_b32
__device__ __forceinline__ unsigned trunc_pair(float e1, float e0) {
    return __builtin_amdgcn_perm(__float_as_uint(e1), __float_as_uint(e0), 0x07060302u);
}
// x minus its leading 8 significant bits (exact)
__device__ __forceinline__ float limb_rest(float x) { return x - __uint_as_float(__float_as_uint(x) & 0xffff0000u); }
__device__ __forceinline__ unsigned round_pair(float e1, float e0) {          // v_cvt_pk_bf16_f32, round to nearest even
    const bf16x2_t v = __builtin_convertvector(f32x2_t{e0, e1}, bf16x2_t);
    return __builtin_bit_cast(unsigned, v);
}

// ---- LIMBS = 2: two f16 limbs, round to nearest, statically scaled ----------------------------------------------------------------
// hi = RN_f16(x) keeps 11 significant bits, x - hi is exact in f32 (|x - hi| <= 2^-11 |x|, either sign) and lo = RN_f16(x - hi) leaves
// |x - hi - lo| <= 2^-23 |x| (one f32 ulp; 0 unless the 13-bit remainder is odd at full width).  W x = Wh xh + Wh xl + Wl xh + [Wl xl]:
// three v_mfma_f32_32x32x16_f16 per f32 product block, total error <= (2^-23 + 2^-23 + 2^-22) |W||x| = 2^-21 |W||x| per product in the
// worst case, signs random (no bias that adds up over K, unlike truncation limbs).  f16 has 5 exponent bits, so both operands carry a
// static power-of-two scale (exact): weights are packed as W 2^F16_SW, activations and features live in registers as x 2^F16_SX, and the
// accumulator of a layer holds 2^(SW+SX) W x -- undone for free in act = max(fma(acc, 2^-SW, bias 2^SX), 0).  Both limbs are normal f16
// numbers for 2^-10 <= |W| < 255 and 2^-6 <= |x| < 4094; below, the low limb is a subnormal with an absolute error <= 2^-25-SW resp.
// 2^-25-SX (1.2e-10 |x| / 1.9e-9 |W|: below an f32 ulp of any activation > 0.03); above, the conversion overflows to inf and the pixel
// comes out NaN (loud, never a wrong number).
constexpr int F16_SW = 8, F16_SX = 4;
constexpr float F16_W_SCALE = 256.0f, F16_X_SCALE = 16.0f, F16_ACC_UNSCALE = 1.0f / 256.0f, F16_HEAD_SCALE = 1.0f / 16.0f;
__device__ __forceinline__ unsigned f16_pair(float e1, float e0) {            // v_cvt_pk_f16_f32, round to nearest even
    const f16x2_t v = __builtin_convertvector(f32x2_t{e0, e1}, f16x2_t);
    return __builtin_bit_cast(unsigned, v);
}
// x minus half HI of an f16 pair, exact: one v_fma_mix_f32 (the f16 operand is widened inside the instruction)
template <int HI>
__device__ __forceinline__ float f16_rest(float x, unsigned pair) {
    float r;
    if (HI) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pair), "v"(x));
    else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pair), "v"(x));
    return r;
}
// EXPERIMENT (round 5, -DF16_MIX_SPLIT=1; not the product): the LOW limb of a pair written straight into its half of the packed word --
// v_fma_mixlo_f16 / v_fma_mixhi_f16 compute fma(f16 half of `pair`, -1.0, x) = x - hi in f32 (exact, see above) and round it to f16 into the
// low / high half of the destination: 1 + 1 instructions per pair where f16_rest + f16_pair take 1 + 1 + 1, i.e. 1.5 instead of 2 split
// instructions per element (of ~5 element instructions per layer), same bits.  Measured same-box, three alternations each
// (profiles/r05_f16_mix_split_ab.txt): fine render pass 77.56-77.71 ms with it, 77.29-77.50 ms without; planes-only training step 1.80 / 1.80-1.83 ms;
// SR stage 48.90-49.13 / 48.78-48.93 ms -- a tenth fewer element instructions move nothing: these kernels are not bound by vector issue
// (the matrix pipe runs at 2.0 GHz of 2.4 under load: a power-managed clock).  Left off.
#ifndef F16_MIX_SPLIT
#define F16_MIX_SPLIT 0
#endif
__device__ __forceinline__ void f16_low_limb_lo(unsigned& lo, float x0, unsigned pair) {
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(pair), "v"(x0));
}
__device__ __forceinline__ void f16_low_limb_hi(unsigned& lo, float x1, unsigned pair) {
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(pair), "v"(x1));
}
template <int LIMBS>
__device__ __forceinline__ f32x16 mfma_limb(u32x4 a, u32x4 b, f32x16 c) {
    if constexpr (LIMBS == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return mfma_bf16(a, b, c);
}

template <int LIMBS> struct Limbs { u32x4 v[LIMBS]; };
struct SplitPend { float r0, r1; };

// The split of a K-block's 8 values get(0..7) into limbs, cut into 4 * NP slices (NP = products per fragment pair = MFMAs of one
// output block): pair j = slice / NP handles values 2j, 2j+1 -> word j of every limb.  <= 3 VALU per slice.
template <int LIMBS, class Get>
__device__ __forceinline__ void split_slice(int slice, Get get, Limbs<LIMBS>& out, SplitPend& p) {
    constexpr int NP = limb_products(LIMBS);
    const int j = slice / NP, st = slice % NP;
    if (j >= 4) return;
#if defined(R3_ABLATE) && (R3_ABLATE & 4096)      // timing experiment: no limb split (the limbs of the previous K-block are reused)
    return;
#endif
    if (LIMBS == 3) {
        if (st == 0) { out.v[0][j] = trunc_pair(get(2 * j + 1), get(2 * j)); p.r0 = limb_rest(get(2 * j)); }
        if (st == 1) { p.r1 = limb_rest(get(2 * j + 1)); }
        if (st == 2) { out.v[1][j] = trunc_pair(p.r1, p.r0); }
        if (st == 3) { p.r0 = limb_rest(p.r0); }
        if (st == 4) { p.r1 = limb_rest(p.r1); }
        if (st == 5) { out.v[2][j] = trunc_pair(p.r1, p.r0); }
    } else {
#if F16_MIX_SPLIT
        if (st == 0) { out.v[0][j] = f16_pair(get(2 * j + 1), get(2 * j)); unsigned w; f16_low_limb_lo(w, get(2 * j), out.v[0][j]); out.v[1][j] = w; }
        if (st == 1) { unsigned w = out.v[1][j]; f16_low_limb_hi(w, get(2 * j + 1), out.v[0][j]); out.v[1][j] = w; }
#else
        if (st == 0) { out.v[0][j] = f16_pair(get(2 * j + 1), get(2 * j)); p.r0 = f16_rest<0>(get(2 * j), out.v[0][j]); }
        if (st == 1) { p.r1 = f16_rest<1>(get(2 * j + 1), out.v[0][j]); }
        if (st == 2) { out.v[1][j] = f16_pair(p.r1, p.r0); }
#endif
    }
}
template <int LIMBS, class Get>
__device__ __forceinline__ void split_all(Get get, Limbs<LIMBS>& out) {
    SplitPend p;
#pragma unroll
    for (int s = 0; s < 4 * limb_products(LIMBS); ++s) split_slice<LIMBS>(s, get, out, p);
}

// straight-line 3-limb split of 8 values (the sliced form above is for MFMA gaps)
__device__ __forceinline__ void split8(const float (&e)[8], Limbs<3>& out) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float r0 = e[2 * j], r1 = e[2 * j + 1];
        out.v[0][j] = trunc_pair(r1, r0);
        r0 = limb_rest(r0); r1 = limb_rest(r1);
        out.v[1][j] = trunc_pair(r1, r0);
        r0 = limb_rest(r0); r1 = limb_rest(r1);
        out.v[2][j] = trunc_pair(r1, r0);
    }
}

// steps [0, NSTEPS) of a piece of side work spread evenly over the slots [S0, S1) of a block
template <int NSTEPS, int S0, int S1, class F>
__device__ __forceinline__ void spread(int slot, F f) {
    constexpr int NS = S1 - S0, PER = (NSTEPS + NS - 1) / NS + 1;
    if (slot < S0 || slot >= S1) return;
    const int a = (slot - S0) * NSTEPS / NS, b = (slot - S0 + 1) * NSTEPS / NS;
#pragma unroll
    for (int k = 0; k < PER; ++k)
        if (a + k < b) f(a + k);
}

struct NoTail {};
struct NoGate {};

// gates of the element pair whose packed f16 high limbs are `hi` (post-ReLU values: +0, positive, or NaN) -> bits (pos, pos + 16) of m:
// min(half, 1) is 1 exactly when the half is not zero -- one v_pk_min_u16 and one v_lshl_or_b32 per PAIR (nvsr_common.h: gate_bit).
// `ones` = 0x00010001 in a scalar register (VOP3P takes no literal).  Inline asm, both instructions in one statement: written in C++ hipcc
// canonicalises min(x, 1) into compare + select per half (5 instructions per pair), and between two asm statements it pads an s_nop.
__device__ __forceinline__ void gate_pair(unsigned hi, int pos, unsigned& m, unsigned ones) {
    unsigned t;
    asm("v_pk_min_u16 %1, %2, %3\n\tv_lshl_or_b32 %0, %1, %4, %0" : "+v"(m), "=&v"(t) : "v"(hi), "s"(ones), "n"(pos));
}
__device__ __forceinline__ unsigned gate_ones() {
    unsigned ones = 0x00010001u;
    asm volatile("" : "+s"(ones));
    return ones;
}

// One block: acc[0..3] (+)= W[chunk K-blocks 0..NKB-1] x (limbs of the source), 4 * NKB * NP MFMAs.
//   cur   : limbs of K-block 0 on entry; on exit the limbs the tail produced (the next block's K-block 0) -- unchanged without a tail
//   fa    : A fragments of (K-block 0, output block 0); LOAD_FIRST reads them here, otherwise they come from the previous block on the
//           same chunk, whose last slots prefetch them (the fragment index wraps around)
//   src(kb, i): value i of K-block kb of this block (read when K-block kb - 1 is being multiplied)
//   side(slot): the other tile's work, slot = ((kb * 4 + ob) * NP + p)
//   tail(slice, nxt): 4 * NP slices during the last K-block, to split the next block's first K-block into nxt
//   gate(kb, j, hi): optional; called once per pair j = 0..3 of every K-block kb >= 1 split inside the block, with the pair's packed high limbs
template <int LIMBS, int NKB, bool ZERO, bool LOAD_FIRST, class Src, class Side, class Tail, class Gate = NoGate>
__device__ __forceinline__ void limb_block(const unsigned* wl, int lane, f32x16 (&acc)[4], Limbs<LIMBS>& cur, Limbs<LIMBS>& fa, Src src,
                                           Side side, Tail tail, Gate gate = Gate{}) {
    constexpr int NP = limb_products(LIMBS), NQ = NKB * 4;
    constexpr bool HAS_TAIL = !std::is_same<Tail, NoTail>::value;
    constexpr bool HAS_GATE = !std::is_same<Gate, NoGate>::value;
    const u32x4* wv = reinterpret_cast<const u32x4*>(wl) + lane;
    if (LOAD_FIRST) {
#pragma unroll
        for (int t = 0; t < LIMBS; ++t) fa.v[t] = wv[t * 64];
    }
    SplitPend sp;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        Limbs<LIMBS> nxt;
#if defined(R3_ABLATE) && (R3_ABLATE & 4096)
        nxt = cur;
#endif
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            Limbs<LIMBS> fn;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int q = kb * 4 + ob;
                if (ZERO && kb == 0 && p == 0) {
                    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                    acc[ob] = mfma_limb<LIMBS>(fa.v[limb_w(LIMBS, p)], cur.v[limb_x(LIMBS, p)], zero);
                } else {
                    acc[ob] = mfma_limb<LIMBS>(fa.v[limb_w(LIMBS, p)], cur.v[limb_x(LIMBS, p)], acc[ob]);
                }
                __builtin_amdgcn_sched_barrier(0);
#if defined(R3_ABLATE) && (R3_ABLATE & 128)       // timing experiment (render3.hip): no A-fragment reads after the block's first
                if (p < LIMBS) { fn.v[p] = fa.v[p]; fn.v[p][0] ^= (unsigned)(q + 1); }     // (distinct per output block: no CSE of the MFMAs)
#else
                // (the fragments of the next group are read LOWEST LIMB FIRST: the next group's first product uses limb 0, so ONE s_waitcnt for
                //  the youngest read covers them all; limb 0 first costs another wait in front of every later limb's first product -- an issue
                //  slot each, 1 767 -> 1 066 waits per step of the 3-limb render pass, 1 401 -> 1 040 of the f16 one)
                if (p < LIMBS) { const int t_ = LIMBS - 1 - p; fn.v[t_] = wv[(((q + 1) % NQ) * LIMBS + t_) * 64]; }
#endif
                if (kb + 1 < NKB) {
                    split_slice<LIMBS>(ob * NP + p, [&](int i) { return src(kb + 1, i); }, nxt, sp);
                    if constexpr (HAS_GATE) { if ((ob * NP + p) % NP == 0 && (ob * NP + p) / NP < 4) gate(kb + 1, (ob * NP + p) / NP, nxt.v[0][(ob * NP + p) / NP]); }
                } else if constexpr (HAS_TAIL) tail(ob * NP + p, nxt);
                side(q * NP + p);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int t = 0; t < LIMBS; ++t) fa.v[t] = fn.v[t];
        }
        if (kb + 1 < NKB || HAS_TAIL) {
#pragma unroll
            for (int t = 0; t < LIMBS; ++t) cur.v[t] = nxt.v[t];
        }
    }
}

}  // namespace nvsr
