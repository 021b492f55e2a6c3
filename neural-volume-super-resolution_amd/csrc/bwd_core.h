// Definitions shared by the backward kernels (render_bwd.hip, render_bwd2.hip) -- not part of the public ABI.
#pragma once
#include "decode_core.h"

namespace nvsr {

// per-wave transposition tile [32 points][48 channels], rows TILE_STRIDE floats apart.  Round 6: 52, not 48 -- the rows are written four channels
// at a time (ds_write_b128: 8 contiguous lanes = 8 points per LDS cycle, bank = word address mod 32); at a stride of 48 words the 8 rows start on
// 2 different bank quads (48 p mod 32 is 0 or 16): a 4-way conflict on every write, 17 % of the kernel's LDS-array cycles (SQ_LDS_BANK_CONFLICT,
// profiles/r05f_train_issue_counters.txt; the forward has no such tile and 0 %).  At 52 words (13 quads, odd) the 8 rows hit 8 different quads.
// The per-point reads (lane = channel: 48 consecutive words) are conflict-free at any stride.
constexpr int TILE_STRIDE = 52;
constexpr int TILE_FLOATS = 31 * TILE_STRIDE + C;     // (the last row ends with its 48th channel: two 3-limb workgroups fill the 160 KB exactly)

// ---- transposed-weight blob ("packed_bwd"), in consumption order ------------------------------------------------------------
//   hidden^T layer (16384 floats): [kb][q][ib][lane][j] = W[32kb + 8q + 4h + j][32ib + (lane&31)]        (W = [out][in])
//   layer-0^T of one plane (8192 floats): [kb][q][ib 2][lane][j] = W0[32kb + 8q + 4h + j][48p + 32ib + (lane&31)], 0 for channel >= 48
constexpr int B_DEN_H = 0;                               // density L3^T, L2^T, L1^T
constexpr int B_DEN0 = B_DEN_H + 3 * P_HID_FLOATS;       // 49152
constexpr int B_RGB_H = B_DEN0 + 8192;                   // 57344: rgb L3^T, L2^T, L1^T
constexpr int B_RGB0 = B_RGB_H + 3 * P_HID_FLOATS;       // 106496: planes 0..3
constexpr int B_TOTAL = B_RGB0 + 4 * 8192;               // 139264

struct Masks { unsigned m[2]; };   // bit gate_bit(ib, r) of m[ib>>1]  <=>  post-ReLU activation acc[ib][r] > 0
struct GradPlanes { float* p[4]; };

// ---- the bf16-limb fragments of the same transposed layers (render_bwd_limb.hip), behind the f32 blob; 816 fragments of 256 words
// ([lane][4 words of 2 bf16], limb_core.h) in consumption order, cut into 34 chunks of 24 fragments (24 KB):
//   hidden^T layer: [K-block 8 (16 output features)][in-feature block 4][limb 3]        96 fragments = 4 chunks
//   layer-0^T of one plane: [K-block 8][channel block 2][limb 3]                         48 fragments = 2 chunks
//   order: density L3^T, L2^T, L1^T, density layer-0^T, rgb L3^T, L2^T, L1^T, rgb layer-0^T of planes 0..3
constexpr int BL_FRAGS = 6 * 96 + 5 * 48;                // 816
constexpr int BL_CHUNK_WORDS = 24 * 256;                 // 6144
constexpr int BL_CHUNKS = BL_FRAGS / 24;                 // 34
constexpr int BL_WORDS = BL_FRAGS * 256;                 // 208896
// ---- and, behind those, the same fragments as 2 f16 limbs (round 3; limb_core.h: round to nearest, UNSCALED -- a transposed weight's low
// limb is a subnormal below |w| = 0.125, absolute error <= 2^-25: about 1e-6 of a typical weight, three orders inside the tolerance of a
// gradient, and the accumulator of one layer is the operand of the next without a rescaling multiply): the same 34 chunks, 16 fragments each
// Round 5: the rgb layer-0^T is ONE [192 x 128] product, not four [48 (padded to 64) x 128] ones -- the four planes' shares of W0 are contiguous rows
// and all multiply the same gradient, so the 192 rows are cut into THREE pairs of 32-row blocks (rows 64 q .. 64 q + 63 = plane boundaries at 48, 96,
// 144) instead of four padded pairs: 144 instead of 192 MFMAs (f16 limbs), 6 instead of 8 chunks; the kernel re-assembles a plane's 48 rows from
// the pair(s) that hold them (static register renames).  The density layer-0^T keeps its padded pair.  RGB0_PAIRS = 4 restores the old layout.
#ifndef BL_RGB0_PAIRS
#define BL_RGB0_PAIRS 3
#endif
template <int LF>
struct BLimb {
    static constexpr int HID_FRAGS = 32 * LF, L0_FRAGS = 16 * LF, CHUNK_FRAGS = 8 * LF;
    static constexpr int FRAGS = 6 * HID_FRAGS + (1 + BL_RGB0_PAIRS) * L0_FRAGS, CHUNK_WORDS = CHUNK_FRAGS * 256, WORDS = FRAGS * 256;
    static constexpr int CHUNKS = FRAGS / CHUNK_FRAGS;                             // 32 (34 with four padded pairs)
    static constexpr int OFFSET = LF == 3 ? 0 : 6 * 96 * 256 + 5 * 48 * 256;      // words behind B_TOTAL (the regions keep the places of the four-pair layout)
};
static_assert(BLimb<3>::WORDS <= BL_WORDS && BLimb<3>::CHUNK_WORDS == BL_CHUNK_WORDS, "3-limb region");
static_assert(B_TOTAL + BL_WORDS + BLimb<2>::WORDS <= NVSR_DECODER_PACKED_BWD_FLOATS, "backward blob size");

__device__ __forceinline__ void apply_mask(const Masks& k, f32x16 (&g)[4]) {
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // gate bit sign-extended to 0 / ~0 (v_bfe_i32), then one v_and: 2 VALU per element instead of extract + compare + select
            const int keep = __builtin_amdgcn_sbfe((int)k.m[ib >> 1], gate_bit(ib, r), 1);
            g[ib][r] = __int_as_float(__float_as_int(g[ib][r]) & keep);
        }
}

// feature gradients of one plane (acc2: rows c = 32b + (r&3) + 8(r>>2) + 4h) -> LDS tile [pt][48] -> atomics into the plane
__device__ __forceinline__ void scatter_plane(const f32x16 (&acc2)[2], float* tile, const Taps& t, float* __restrict__ gplane, int lane,
                                              bool valid) {
    const int h = lane >> 5, pt = lane & 31;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (b == 1 && r >= 8) continue;                        // rows 48..63 are padding
            const int c = 32 * b + (r & 3) + 8 * (r >> 2) + 4 * h;
            tile[pt * TILE_STRIDE + c] = acc2[b][r];
        }
    __builtin_amdgcn_wave_barrier();
    // taps of an invalid (padding) ray carry zero weight
    const float w0 = valid ? t.nw : 0.0f, w1 = valid ? t.ne : 0.0f, w2 = valid ? t.sw : 0.0f, w3 = valid ? t.se : 0.0f;
    // Point p's four texel offsets and weights are wave-uniform once read out of lane p: v_readlane into SGPRs, so that an atomic's
    // address is (scalar texel base) + lane and its operand one v_mul by a scalar.  (With __shfl these were 8 ds_bpermute per point --
    // 256 LDS round trips per plane -- plus 64-bit vector address arithmetic in front of every atomic.)
#pragma unroll 8
    for (int p = 0; p < 32; ++p) {
        float* const b0 = gplane + __builtin_amdgcn_readlane(t.o00, p);
        float* const b1 = gplane + __builtin_amdgcn_readlane(t.o01, p);
        float* const b2 = gplane + __builtin_amdgcn_readlane(t.o10, p);
        float* const b3 = gplane + __builtin_amdgcn_readlane(t.o11, p);
        const float a0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w0), p));
        const float a1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w1), p));
        const float a2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w2), p));
        const float a3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w3), p));
        if (lane < C) {
            const float v = tile[p * TILE_STRIDE + lane];
            unsafeAtomicAdd(b0 + lane, v * a0);
            unsafeAtomicAdd(b1 + lane, v * a1);
            unsafeAtomicAdd(b2 + lane, v * a2);
            unsafeAtomicAdd(b3 + lane, v * a3);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// The same for a tile of 32 CONSECUTIVE SAMPLES OF ONE RAY (render_bwd_limb.hip): neighbouring samples of a ray mostly fall into the same
// texel cell of a plane (a step along the ray is shorter than a texel at 128 samples over a 200^2 plane), so the contributions of a run of
// samples with the same four texels are summed in registers and go out as ONE set of 4 atomics -- the float atomics are what the backward
// pass is bound by.  Offsets and weights are wave-uniform per point (v_readlane), the run test is scalar.
__device__ __forceinline__ void scatter_plane_runs(const f32x16 (&acc2)[2], float* tile, const Taps& t, float* __restrict__ gplane, int lane,
                                                   bool valid) {
    const int h = lane >> 5, pt = lane & 31;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (b == 1 && r >= 8) continue;                        // rows 48..63 are padding
            const int c = 32 * b + (r & 3) + 8 * (r >> 2) + 4 * h;
            tile[pt * TILE_STRIDE + c] = acc2[b][r];
        }
    __builtin_amdgcn_wave_barrier();
    const float w0 = valid ? t.nw : 0.0f, w1 = valid ? t.ne : 0.0f, w2 = valid ? t.sw : 0.0f, w3 = valid ? t.se : 0.0f;
    int c0 = -1, c1 = -1, c2 = -1, c3 = -1;                        // texels of the current run
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;              // this lane's channel, summed over the run
    auto flush = [&]() {
        if (c0 >= 0 && lane < C) {
            unsafeAtomicAdd(gplane + c0 + lane, s0);
            unsafeAtomicAdd(gplane + c1 + lane, s1);
            unsafeAtomicAdd(gplane + c2 + lane, s2);
            unsafeAtomicAdd(gplane + c3 + lane, s3);
        }
    };
    // (Keeping four texel accumulators keyed by offset, so that a texel shared with the NEIGHBOURING cell survives the step as well, was
    //  measured slower: 3.13 vs 2.87 ms per step -- 16 select / fma pairs per point and 25 more spilled registers cost more than the atomics
    //  they save.)
#pragma unroll 4
    for (int p = 0; p < 32; ++p) {
        const int o0 = __builtin_amdgcn_readlane(t.o00, p), o1 = __builtin_amdgcn_readlane(t.o01, p);
        const int o2 = __builtin_amdgcn_readlane(t.o10, p), o3 = __builtin_amdgcn_readlane(t.o11, p);
        const float a0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w0), p));
        const float a1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w1), p));
        const float a2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w2), p));
        const float a3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w3), p));
        if (o0 != c0 || o1 != c1 || o2 != c2 || o3 != c3) {        // (wave-uniform)
            flush();
            c0 = o0; c1 = o1; c2 = o2; c3 = o3;
            s0 = s1 = s2 = s3 = 0.0f;
        }
        const float v = lane < C ? tile[p * TILE_STRIDE + lane] : 0.0f;
        s0 = fmaf(v, a0, s0); s1 = fmaf(v, a1, s1); s2 = fmaf(v, a2, s2); s3 = fmaf(v, a3, s3);
    }
    flush();
    __builtin_amdgcn_wave_barrier();
}

#if defined(BL_ABLATE) && (BL_ABLATE & 16)      // timing experiment: the scatter loop without its atomics (the sums stay live through an empty asm)
#define NVSR_BWD_ATOMIC(P, V) asm volatile("" : : "v"(P), "v"(V))
#else
#define NVSR_BWD_ATOMIC(P, V) unsafeAtomicAdd(P, V)
#endif
// The same tile with EVERY texel written once (round 3).  Along a ray the cell index is monotone in x and in y, so a texel that the path
// has left never comes back: a four-entry cache keyed by the texel's (x & 1, y & 1) holds exactly the texels that the current cell and the
// next one can share -- tap k = (dx, dy) of a cell with parity par lands in slot k ^ par, the four taps of a cell in four different slots.
// The slot -> (texel, weight) assignment of a point is a permutation of wave-uniform values and is done on the scalar unit (two rounds of
// conditional swaps); the vector side is the same 4 fused multiply-adds per point as in scatter_plane_runs, the accumulators are 4 registers.
// A slot is flushed (one 192-byte atomic) when its texel changes.  Atomic payload of a training step's fine pass: 0.35 of the
// one-set-per-point scatter (scatter_plane_runs: 0.54), of its coarse pass 0.62 (0.82) -- the tile's distinct texels, i.e. the minimum
// without merging across waves.  `t.o00` carries the parity of the cell in its two low bits (texel offsets are multiples of 48).
// (Leaving the atomics in flight behind a counted s_waitcnt at the next chunk boundary -- they are younger than the copy it waits for --
//  measured no gain: 0.807 vs 0.800 ms.)
__device__ __forceinline__ void scatter_plane_cached(const f32x16 (&acc2)[2], float* tile, const Taps& t, float* __restrict__ gplane, int lane,
                                                    bool valid) {
    const int h = lane >> 5, pt = lane & 31;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (b == 1 && r >= 8) continue;                        // rows 48..63 are padding
            const int c = 32 * b + (r & 3) + 8 * (r >> 2) + 4 * h;
            tile[pt * TILE_STRIDE + c] = acc2[b][r];
        }
    __builtin_amdgcn_wave_barrier();
    const float w0 = valid ? t.nw : 0.0f, w1 = valid ? t.ne : 0.0f, w2 = valid ? t.sw : 0.0f, w3 = valid ? t.se : 0.0f;
    int k0 = -1, k1 = -1, k2 = -1, k3 = -1;                        // texel held by slot j
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;              // this lane's channel of slot j
    float* const gl = gplane + lane;
#define NVSR_SLOT(K, S, NK, A)                                                                          \
        if (NK != K) {                                             /* (wave-uniform) */                 \
            if (K >= 0 && lane < C) NVSR_BWD_ATOMIC(gl + K, S);                                         \
            K = NK;                                                                                     \
            S = 0.0f;                                                                                   \
        }                                                                                               \
        S = fmaf(v, A, S);
    // (lanes 48..63 run along on the next point's first channels -- their sums are never written; the last point's read is clamped into the tile)
    const int li = lane < C ? lane : lane - 16;
#pragma unroll 2
    for (int p0 = 0; p0 < 32; p0 += 4) {
        float vv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) vv[j] = tile[(p0 + j) * TILE_STRIDE + li];                  // 4 points' rows in flight
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = p0 + j;
            int o0 = __builtin_amdgcn_readlane(t.o00, p), o1 = __builtin_amdgcn_readlane(t.o01, p);
            int o2 = __builtin_amdgcn_readlane(t.o10, p), o3 = __builtin_amdgcn_readlane(t.o11, p);
            float a0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w0), p));
            float a1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w1), p));
            float a2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w2), p));
            float a3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w3), p));
            const int par = o0 & 3;
            o0 &= ~3;
            if (par & 1) { int x = o0; o0 = o1; o1 = x; x = o2; o2 = o3; o3 = x; float y = a0; a0 = a1; a1 = y; y = a2; a2 = a3; a3 = y; }
            if (par & 2) { int x = o0; o0 = o2; o2 = x; x = o1; o1 = o3; o3 = x; float y = a0; a0 = a2; a2 = y; y = a1; a1 = a3; a3 = y; }
            const float v = vv[j];
            NVSR_SLOT(k0, s0, o0, a0)
            NVSR_SLOT(k1, s1, o1, a1)
            NVSR_SLOT(k2, s2, o2, a2)
            NVSR_SLOT(k3, s3, o3, a3)
        }
    }
#undef NVSR_SLOT
    if (lane < C) {
        if (k0 >= 0) NVSR_BWD_ATOMIC(gl + k0, s0);
        if (k1 >= 0) NVSR_BWD_ATOMIC(gl + k1, s1);
        if (k2 >= 0) NVSR_BWD_ATOMIC(gl + k2, s2);
        if (k3 >= 0) NVSR_BWD_ATOMIC(gl + k3, s3);
    }
    __builtin_amdgcn_wave_barrier();
}

// scatter_plane_cached with the per-point bookkeeping done ONCE PER TILE on the vector unit (lane = point) instead of per point on the scalar
// unit: the parity permutation of (texel, weight) is 16 selects per tile instead of 16 s_cselect per point, "slot j changes at this point" is
// one compare against the previous point's lane (ds_bpermute) packed into 4 flag bits.  The loop then reads 1 flag word + 4 weights per point
// (v_readlane; the texel offsets only when a slot is flushed) -- 5 lane reads, 4 FMAs and 4 bit tests per point where the scalar version spent
// 8 lane reads, ~16 conditional moves and 4 compares.  Same sums in the same order: the results differ only through the order of the atomics.
__device__ __forceinline__ void scatter_plane_cached_v(const f32x16 (&acc2)[2], float* tile, const Taps& t, float* __restrict__ gplane, int lane,
                                                      bool valid) {
    const int h = lane >> 5, pt = lane & 31;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (b == 1 && r >= 8) continue;                        // rows 48..63 are padding
            const int c = 32 * b + (r & 3) + 8 * (r >> 2) + 4 * h;
            tile[pt * TILE_STRIDE + c] = acc2[b][r];
        }
    __builtin_amdgcn_wave_barrier();
    // this lane's point: (texel, weight) of slot j = tap j ^ par
    const int par = t.o00 & 3;
    const bool sx = par & 1, sy = par & 2;
    const float w0 = valid ? t.nw : 0.0f, w1 = valid ? t.ne : 0.0f, w2 = valid ? t.sw : 0.0f, w3 = valid ? t.se : 0.0f;
    const int q0 = t.o00 & ~3, q1 = t.o01, q2 = t.o10, q3 = t.o11;
    const int x0 = sx ? q1 : q0, x1 = sx ? q0 : q1, x2 = sx ? q3 : q2, x3 = sx ? q2 : q3;
    const float y0 = sx ? w1 : w0, y1 = sx ? w0 : w1, y2 = sx ? w3 : w2, y3 = sx ? w2 : w3;
    const int o0 = sy ? x2 : x0, o1 = sy ? x3 : x1, o2 = sy ? x0 : x2, o3 = sy ? x1 : x3;
    const float a0 = sy ? y2 : y0, a1 = sy ? y3 : y1, a2 = sy ? y0 : y2, a3 = sy ? y1 : y3;
    // bit j: slot j holds another texel than at the previous point (point 0: every slot starts)
    const int prev = (lane & 32) | (pt > 0 ? pt - 1 : 0);
    const int fl = pt == 0 ? 15 : ((o0 != __shfl(o0, prev)) ? 1 : 0) | ((o1 != __shfl(o1, prev)) ? 2 : 0) | ((o2 != __shfl(o2, prev)) ? 4 : 0) |
                                  ((o3 != __shfl(o3, prev)) ? 8 : 0);
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;              // this lane's channel of slot j
    float* const gl = gplane + lane;
#if defined(BL_SCATTER_FALLTHROUGH) && BL_SCATTER_FALLTHROUGH
    // experiment (round 5, -DBL_SCATTER_FALLTHROUGH=1; off): the flush of a slot is the UNCOMMON case (the fine pass flushes 1.4 of 4 slots per
    // point) -- marked unlikely, so that the common path falls through instead of taking a branch around every flush block, and all four tests
    // sit behind one test of the point's whole flag word.  Same box, three alternations (tools/bwd_time_one.py): S = 64 0.391-0.398 ms product /
    // 0.397-0.407 with it, S = 128 0.593-0.605 / 0.600-0.604: branches are not what the loop waits for
#define NVSR_FLUSH_V(BIT, S, O)                                                                          \
        if (f & BIT) {                                                                                   \
            if (p > 0) NVSR_BWD_ATOMIC(gl + __builtin_amdgcn_readlane(O, p - 1), S);                     \
            S = 0.0f;                                                                                    \
        }
#define NVSR_SLOT_V(BIT, S, O, A)                                                                        \
        S = fmaf(v, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(A), p)), S);
#else
#define NVSR_SLOT_V(BIT, S, O, A)                                                                        \
        if (f & BIT) {                                             /* (wave-uniform) */                  \
            if (p > 0) NVSR_BWD_ATOMIC(gl + __builtin_amdgcn_readlane(O, p - 1), S);                     \
            S = 0.0f;                                                                                    \
        }                                                                                                \
        S = fmaf(v, (SCV_ABLATE & 2) ? A : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(A), p)), S);
#endif
    // lanes 48..63 sit the whole loop out (one exec mask around it instead of one around every atomic; v_readlane ignores exec)
#ifndef SCV_ABLATE
#define SCV_ABLATE 0      // timing experiments only (wrong results): 1 no per-point loop, 2 weights without v_readlane, 4 no flag tests / flushes, 8 no LDS reads
#endif
    if (lane < C) {
#pragma unroll 2
    for (int p0 = 0; p0 < ((SCV_ABLATE & 1) ? 0 : 32); p0 += 4) {
        float vv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) vv[j] = (SCV_ABLATE & 8) ? (float)(p0 + j) : tile[(p0 + j) * TILE_STRIDE + lane];                // 4 points' rows in flight
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = p0 + j;
            const int f = (SCV_ABLATE & 4) ? 0 : __builtin_amdgcn_readlane(fl, p);
            const float v = vv[j];
#if defined(BL_SCATTER_FALLTHROUGH) && BL_SCATTER_FALLTHROUGH
            if (__builtin_expect(f != 0, 0)) {
                NVSR_FLUSH_V(1, s0, o0)
                NVSR_FLUSH_V(2, s1, o1)
                NVSR_FLUSH_V(4, s2, o2)
                NVSR_FLUSH_V(8, s3, o3)
            }
#endif
            NVSR_SLOT_V(1, s0, o0, a0)
            NVSR_SLOT_V(2, s1, o1, a1)
            NVSR_SLOT_V(4, s2, o2, a2)
            NVSR_SLOT_V(8, s3, o3, a3)
        }
    }
#undef NVSR_SLOT_V
#ifdef NVSR_FLUSH_V
#undef NVSR_FLUSH_V
#endif
        NVSR_BWD_ATOMIC(gl + __builtin_amdgcn_readlane(o0, 31), s0);
        NVSR_BWD_ATOMIC(gl + __builtin_amdgcn_readlane(o1, 31), s1);
        NVSR_BWD_ATOMIC(gl + __builtin_amdgcn_readlane(o2, 31), s2);
        NVSR_BWD_ATOMIC(gl + __builtin_amdgcn_readlane(o3, 31), s3);
    }
    __builtin_amdgcn_wave_barrier();
}

// View-direction plane: every sample of a ray hits the SAME 4 texels of a 32 x 32 plane, so direct atomics pile ~2 000 adds on each
// address (measured: +1.2 ms on a 1.2 ms kernel).  Instead the feature gradient of each point is written as a plain 192-byte row
// gview[ray*S + s][48]; view_reduce_scatter_kernel sums a ray's S rows and does the 4 x 48 atomics once per ray.
__device__ __forceinline__ void store_view_rows(const f32x16 (&acc2)[2], float* tile, float* __restrict__ gview, long ray_w0, long N, int S,
                                                int s, int lane) {
    const int h = lane >> 5, pt = lane & 31;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (b == 1 && r >= 8) continue;
            const int c = 32 * b + (r & 3) + 8 * (r >> 2) + 4 * h;
            tile[pt * TILE_STRIDE + c] = acc2[b][r];
        }
    __builtin_amdgcn_wave_barrier();
    if (lane < C) {
        for (int p = 0; p < 32; ++p)
            if (ray_w0 + p < N) gview[((ray_w0 + p) * S + s) * C + lane] = tile[p * TILE_STRIDE + lane];
    }
    __builtin_amdgcn_wave_barrier();
}



}  // namespace nvsr
