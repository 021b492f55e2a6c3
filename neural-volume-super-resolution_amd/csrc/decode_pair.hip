// Training forward on TILE PAIRS (round 4): decode_rays_limb_kernel<MASKS, false, 2> (decode_limb.hip) on the skeleton of the fused render
// pass (render3.hip) -- raw [N,S,4] and the ReLU gates of a training batch in the 2-f16-limb arithmetic of limb_core.h.
//
// The reference runs this as run_network -> TwoDimPlanesModel.forward under autograd (train_utils.py:15-64, models.py:381-421).
//
// decode_limb.hip gives every wave ONE 32-point tile and lets two independent 4-wave workgroups per CU cover each other's side work: each
// workgroup streams its own copy of the 336 KB of weight fragments per step and whatever the hardware arbitration between the two waves of a
// SIMD does not overlap is lost (48.7 % matrix-pipe busy, 6.7 VALU instructions per MFMA).  Here a wave (one per SIMD, 512 registers) owns
// the two tiles X and Y of a pair -- 2 x 32 CONSECUTIVE SAMPLES OF ONE RAY -- and, exactly as in render3.hip, issues one tile's plane
// gathers, limb splits, bias + ReLU and heads in the MFMA gaps of the other tile's block: one weight stream per 256 points instead of per
// 128 (and the first 9 K-blocks resident in LDS), half the ring barriers per MFMA, a hand-placed interleave instead of arbitration.
// What the training forward adds to the render pass's step:
//   * the view-plane features belong to the RAY: gathered once per pair (both tiles share them);
//   * raw goes to memory instead of the compositor;
//   * MASKS: the ReLU gates of all eight layers.  A hidden layer's activations are split into f16 limbs for the next layer anyway, and
//     post-ReLU values are never negative: the two gates of an element pair are min(high limb, 1) of the packed pair -- one v_pk_min_u16 and
//     one v_lshl_or_b32 per PAIR where decode_limb.hip spends 2 instructions per ELEMENT (gate_pair, nvsr_common.h: gate_bit); the two
//     layers in front of the heads are not split and convert their pairs for the gates alone (3 per pair).
#include <type_traits>
#include <utility>

#include "pair_core.h"

#ifndef DP_ABLATE
#define DP_ABLATE 0    // timing experiments (wrong results): 1 no gates, 2 no exposed view gather
#endif

namespace nvsr {

struct LdsP {                                                       // Lds3<2> without the ray cache
    static constexpr int SLOT = Lds3<2>::SLOT;                      // 32 KB ring slot (4 K-blocks)
    static constexpr int SMALL = 2 * SLOT;
    static constexpr int BOUNCE = SMALL + SMALL_FLOATS;              // 1 KB per wave: the accumulators' way to the VALU (relu_bias_step)
    static constexpr int RES = BOUNCE + (R3_BOUNCE ? NW2 * 256 : 0);
    static constexpr int RES_KB = 9;                                // view plane, planes 0 and 1 of rgb layer 0 stay resident
    static constexpr int TOTAL = RES + RES_KB * kb_words(2);
};
static_assert(LdsP::TOTAL * 4 <= 160 * 1024, "LDS budget");

struct TileP {
    f32x16 acc[4];   // layer accumulators (AGPRs), written by MFMAs only
    f32x16 act[4];   // max(acc + bias, 0) of the finished layer
    float D[HALF_C], F[HALF_C];
    float raw[4];
    unsigned g0, g1; // gate words of the layer whose activations are being split
};

// relu_bias_step for a layer that is NOT split afterwards (the layers in front of the heads): every second element also converts its pair
// to f16 for the gates (v_cvt_pk_f16_f32 + gate_pair)
template <bool MASKS>
__device__ __forceinline__ void relu_gate_step(int k, const float* bias, int h, const f32x16 (&acc)[4], f32x16 (&act)[4], BiasPend4& pend, f32x2_t nsc,
                                               unsigned& g0, unsigned& g1, unsigned ones) {
    relu_bias_step<2>(k, bias, h, acc, act, pend, nsc);
    if constexpr (MASKS) {
        const int r = k - 4;
        if (r >= 1 && (r & 1)) {
            const int ib = r >> 4, rr = (r & 15) - 1;
            gate_pair(f16_pair(act[ib][rr + 1], act[ib][rr]), gate_bit(ib, rr), ib < 2 ? g0 : g1, ones);
        }
    }
}

// (the kernels below are thin shells around this body: hipcc's HOST pass cannot resolve the LDS-DMA helpers inside a kernel template and
//  then drops the kernel's launch stub -- the body is compiled in the device pass only, like render3.hip's)
template <bool MASKS>
__device__ __forceinline__ void decode_pair_body(const SceneDev& sc, const float* __restrict__ packed, long N, int S,
                                                 const float* __restrict__ rays, const float* __restrict__ z,
                                                 float* __restrict__ raw_out, unsigned* __restrict__ gates, unsigned* __restrict__ flag) {
    constexpr int LIMBS = 2;
    using L = LdsP;
    constexpr int NP = limb_products(LIMBS);
    constexpr int NSF = 3 * 4 * NP, NSH = 4 * 4 * NP;          // slots of a feature block / of half a hidden layer
    __shared__ __attribute__((aligned(16))) unsigned lds[L::TOTAL];
    NVSR_RACE_PROBE_DELAY(lds);      // (probe builds only, nvsr_common.h)
    Ring3<LIMBS> rs{__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(packed + limb_region(LIMBS)), 0, KB_TOTAL * kb_words(LIMBS) * 4, 0x00020000),
                    lds, 0, __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), (int)(threadIdx.x & 63), (threadIdx.x >> 6) * 1024u + (threadIdx.x & 63) * 16u};
    float* ldsf = reinterpret_cast<float*>(lds);
    // (f16 limbs: biases x 2^SX, head weights x 2^-SX, the packer's poison into the head biases -- render3.hip)
    for (int i = threadIdx.x; i < SMALL_FLOATS; i += TPB2) {
        float v = packed[P_SMALL + i] * (i < S_ALPHA_W ? F16_X_SCALE : i < S_HEAD_B ? F16_HEAD_SCALE : 1.0f);
        if (i >= S_HEAD_B && i < S_HEAD_B + 4) v += packed[P_SMALL + S_F16_POISON];
        ldsf[L::SMALL + i] = v;
    }
    const float* small = ldsf + L::SMALL;

    // work: pairs of 32-sample chunks of one ray; a pair never straddles two rays (the last pair of a ray with an odd chunk count has an empty Y)
    const int nsc = (S + 31) / 32, npr = (nsc + 1) / 2;
    const long npairs = N * npr, nsteps = (npairs + NW2 - 1) / NW2;

    TileP X, Y;
    float V[HALF_C];         // view-plane features of the pair's ray
    RawTaps4 rt;
    Limbs<LIMBS> cur, fa;
    f32x2_t nsc2 = {-F16_ACC_UNSCALE, -F16_ACC_UNSCALE};                  // relu_bias_step
    asm volatile("" : "+s"(nsc2));
    const unsigned ones = gate_ones();
    constexpr int KB_FIRST = KB_RGB0 + 9;                                 // the first chunk of a step that goes through the ring (plane 2 of rgb layer 0)
    unsigned* const res = lds + L::RES;
    ring3_load_resident<LIMBS, L::RES_KB>(rs, res, KB_RGB0);
    unsigned* cw = const_cast<unsigned*>(ring3_issue<LIMBS, 3>(rs, KB_FIRST));     // first ring chunk of the first step; every later one is issued during the previous step

    auto scale_taps = [](Taps& t) NVSR_INL { t.nw *= F16_X_SCALE; t.ne *= F16_X_SCALE; t.sw *= F16_X_SCALE; t.se *= F16_X_SCALE; };

    for (long step = blockIdx.x; step < nsteps; step += gridDim.x) {
        asm volatile("" : "+v"(rs.voff), "+v"(rs.lane));
        const int lane = rs.lane, h = lane >> 5;
        const long pi = step * NW2 + rs.wave;
        const long ray0 = pi / npr;
        const int c0 = (int)(pi - ray0 * npr) * 2;
        const long ray = ray0 < N ? ray0 : N - 1;
        const int sX0 = c0 * 32 + (lane & 31), sY0 = sX0 + 32;
        const bool validX = ray0 < N && sX0 < S, validY = ray0 < N && sY0 < S;
        const int sX = sX0 < S ? sX0 : S - 1, sY = sY0 < S ? sY0 : S - 1;
        const float* r = rays + ray * 11;
        const float zX = z[ray * S + sX], zY = z[ray * S + sY];
        // the ray's view-plane features (project_viewdir, models.py:312-326): every lane of a half reads the same four texels
        {
            GatherJob vj;
            vj.plane = sc.plane[3];
            vj.t = view_taps(sc, r[8], r[9], r[10]);
            scale_taps(vj.t);
#if !(DP_ABLATE & 2)
#pragma unroll
            for (int k = 0; k < 12; ++k) gather4_load(k, vj, h, rt);
#pragma unroll
            for (int c = 0; c < HALF_C; ++c) gather4_blend(c, vj, rt, V);
#else
#pragma unroll
            for (int c = 0; c < HALF_C; ++c) V[c] = vj.t.nw * (float)c;
#endif
        }
        float xn0, xn1, xn2, yn0, yn1, yn2;
        {
            const float ox = r[0], oy = r[1], oz = r[2], dx = r[3], dy = r[4], dz = r[5];
            xn0 = norm_coord(__fadd_rn(ox, __fmul_rn(dx, zX)), sc.lo[0], sc.range[0]);
            xn1 = norm_coord(__fadd_rn(oy, __fmul_rn(dy, zX)), sc.lo[1], sc.range[1]);
            xn2 = norm_coord(__fadd_rn(oz, __fmul_rn(dz, zX)), sc.lo[2], sc.range[2]);
            yn0 = norm_coord(__fadd_rn(ox, __fmul_rn(dx, zY)), sc.lo[0], sc.range[0]);
            yn1 = norm_coord(__fadd_rn(oy, __fmul_rn(dy, zY)), sc.lo[1], sc.range[1]);
            yn2 = norm_coord(__fadd_rn(oz, __fmul_rn(dz, zY)), sc.lo[2], sc.range[2]);
        }
        // gate record of a lane: [point ray*S+s][lane half][16 words]; padding lanes rewrite a valid point's record with the same values
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        u32x2* const glX = MASKS ? reinterpret_cast<u32x2*>(gates + ((ray * S + sX) * 2 + h) * 16) : nullptr;
        u32x2* const glY = MASKS ? reinterpret_cast<u32x2*>(gates + ((ray * S + sY) * 2 + h) * 16) : nullptr;
        BiasPend4 bp;
        bp.slot = bounce_slot(reinterpret_cast<float*>(lds) + LdsP::BOUNCE + rs.wave * 256 + lane * 4);
        HeadPend<3> hp3;
        HeadPend<1> hp1;
        SplitPend tp;

        // side-work pieces
        auto feat = [](const float (&f)[HALF_C]) NVSR_INL { return [&f](int kb, int i) NVSR_INL { return f[8 * kb + i]; }; };
        auto hid = [](const f32x16 (&a)[4], int kb0) NVSR_INL { return [&a, kb0](int kb, int i) NVSR_INL { const int k = kb0 + kb; return a[k >> 1][8 * (k & 1) + i]; }; };
        auto split_feat = [&](const float (&f)[HALF_C]) NVSR_INL { split_all<LIMBS>([&f](int i) NVSR_INL { return f[i]; }, cur); };
        // gates of the pairs of hidden K-block k (0..7) of a tile: pair j -> bits (4 (k & 3) + j, + 16) of word k >> 2
        auto gate_of = [ones](TileP& t, int kb0) NVSR_INL {
            return [&t, kb0, ones](int kb, int j, unsigned hi) NVSR_INL {
                if constexpr (MASKS && !(DP_ABLATE & 1)) { const int k = kb0 + kb; gate_pair(hi, 4 * (k & 3) + j, k < 4 ? t.g0 : t.g1, ones); }
            };
        };
        // tail: split K-block kb of t.act into the limbs the next block starts with (+ its gates)
        auto tail_of = [&](TileP& t, int kb) NVSR_INL {
            return [&t, kb, &tp, ones](int slice, Limbs<LIMBS>& nxt) NVSR_INL {
                split_slice<LIMBS>(slice, [&t, kb](int i) NVSR_INL { return t.act[kb >> 1][8 * (kb & 1) + i]; }, nxt, tp);
                if constexpr (MASKS && !(DP_ABLATE & 1)) { if (slice % NP == 0 && slice / NP < 4) gate_pair(nxt.v[0][slice / NP], 4 * (kb & 3) + slice / NP, kb < 4 ? t.g0 : t.g1, ones); }
            };
        };
        auto none = [](int) NVSR_INL {};

        // ---- rgb layer 0: (view plane, planes 0..2) x (X block, Y block); the gathers roll through the blocks (gather_roll) -------------
        GatherJob ja, jb;
        ring3_sync<0>();                                         // the step's first ring chunk (issued during the previous step); the resident region
        unsigned* nw = nullptr;
#define NVSR_ROLL(TL, JL, TB, JB, LOADS, BLENDS) [&](int slot) NVSR_INL { gather_roll<NSF, LOADS, BLENDS, false>(slot, JL, TL.F, JB, TB.F, h, rt); }
#define NVSR_ROLL_DMA(TL, JL, TB, JB, LOADS, BLENDS, NKB, KB0) \
        [&](int slot) NVSR_INL { gather_roll<NSF, LOADS, BLENDS, false>(slot, JL, TL.F, JB, TB.F, h, rt); dma_side<LIMBS, NKB>(slot, rs, nw, KB0); }
        // X view | loads X plane 0
        ja.plane = sc.plane[0]; ja.t = pos_taps2(sc, 0, xn0, xn1, xn2); scale_taps(ja.t);
        split_feat(V);
        limb_block<LIMBS, 3, true, true>(res, lane, X.acc, cur, fa, feat(V), NVSR_ROLL(X, ja, Y, jb, true, false), NoTail{});
        // Y view | blends X plane 0, loads Y plane 0
        jb.plane = sc.plane[0]; jb.t = pos_taps2(sc, 0, yn0, yn1, yn2); scale_taps(jb.t);
        split_feat(V);
        limb_block<LIMBS, 3, true, false>(res, lane, Y.acc, cur, fa, feat(V), NVSR_ROLL(Y, jb, X, ja, true, true), NoTail{});
        const unsigned* w_p0 = res + 3 * kb_words(LIMBS);
        // X plane 0 | blends Y plane 0, loads X plane 1
#pragma unroll
        for (int c = 0; c < HALF_C; ++c) X.D[c] = X.F[c];
        ja.plane = sc.plane[1]; ja.t = pos_taps2(sc, 1, xn0, xn1, xn2); scale_taps(ja.t);
        split_feat(X.F);
        limb_block<LIMBS, 3, false, true>(w_p0, lane, X.acc, cur, fa, feat(X.F), NVSR_ROLL(X, ja, Y, jb, true, true), NoTail{});
        // Y plane 0 | blends X plane 1, loads Y plane 1
#pragma unroll
        for (int c = 0; c < HALF_C; ++c) Y.D[c] = Y.F[c];
        jb.plane = sc.plane[1]; jb.t = pos_taps2(sc, 1, yn0, yn1, yn2); scale_taps(jb.t);
        split_feat(Y.F);
        limb_block<LIMBS, 3, false, false>(w_p0, lane, Y.acc, cur, fa, feat(Y.F), NVSR_ROLL(Y, jb, X, ja, true, true), NoTail{});
        const unsigned* w_p1 = res + 6 * kb_words(LIMBS);
        // X plane 1 | blends Y plane 1, loads X plane 2
#pragma unroll
        for (int c = 0; c < HALF_C; ++c) X.D[c] = __fadd_rn(X.D[c], X.F[c]);
        ja.plane = sc.plane[2]; ja.t = pos_taps2(sc, 2, xn0, xn1, xn2); scale_taps(ja.t);
        split_feat(X.F);
        limb_block<LIMBS, 3, false, true>(w_p1, lane, X.acc, cur, fa, feat(X.F), NVSR_ROLL(X, ja, Y, jb, true, true), NoTail{});
        // Y plane 1 | blends X plane 2, loads Y plane 2
#pragma unroll
        for (int c = 0; c < HALF_C; ++c) Y.D[c] = __fadd_rn(Y.D[c], Y.F[c]);
        jb.plane = sc.plane[2]; jb.t = pos_taps2(sc, 2, yn0, yn1, yn2); scale_taps(jb.t);
        split_feat(Y.F);
        limb_block<LIMBS, 3, false, false>(w_p1, lane, Y.acc, cur, fa, feat(Y.F), NVSR_ROLL(Y, jb, X, ja, true, true), NoTail{});
        nw = ring3_take(rs);                                     // (cw is plane 2's chunk since the top of the step)
        // X plane 2 | blends Y plane 2;  D = (D + F) / 3   (combine_pos_planes 'avg', models.py:358-359)
#pragma unroll
        for (int c = 0; c < HALF_C; ++c) X.D[c] = div3(__fadd_rn(X.D[c], X.F[c]));
        split_feat(X.F);
        limb_block<LIMBS, 3, false, true>(cw, lane, X.acc, cur, fa, feat(X.F), NVSR_ROLL_DMA(X, ja, Y, jb, false, true, 4, KB_RGB1), NoTail{});
        // Y plane 2 | X: act = max(acc + bias, 0); tail: limbs of X's K-block 0
#pragma unroll
        for (int c = 0; c < HALF_C; ++c) Y.D[c] = div3(__fadd_rn(Y.D[c], Y.F[c]));
        split_feat(Y.F);
        X.g0 = X.g1 = 0u;
        limb_block<LIMBS, 3, false, false>(cw, lane, Y.acc, cur, fa, feat(Y.F),
                                           [&](int slot) NVSR_INL { spread<RELU_STEPS, 0, NSF>(slot, [&](int k) NVSR_INL { relu_bias_step<LIMBS>(k, small + S_BIAS + 4 * HID, h, X.acc, X.act, bp, nsc2); }); },
                                           tail_of(X, 0));
        cw = nw;

        // ---- hidden layers.  Layer l of a decoder = chunks a (K-blocks 0..3), b (4..7):
        //   X a | Y: act of layer l-1; tail Y kb 0        Y a | tail X kb 4        X b | tail Y kb 4        Y b | X: act of layer l; tail X kb 0
        // the gate words of layer l-1 are complete (and stored) when the tile's K-blocks 5..7 have been split: X after X b, Y after Y b
        auto relu_side = [&](TileP& t, int bias_vec) NVSR_INL {
            return [&, bias_vec](int slot) NVSR_INL { spread<RELU_STEPS, 0, NSH>(slot, [&](int k) NVSR_INL { relu_bias_step<LIMBS>(k, small + S_BIAS + bias_vec * HID, h, t.acc, t.act, bp, nsc2); }); };
        };
        // last hidden layer of a decoder (feeds the heads, is not split): bias + ReLU with its gates
        auto relu_gate_side = [&](TileP& t, int bias_vec) NVSR_INL {
            return [&, bias_vec](int slot) NVSR_INL { spread<RELU_STEPS, 0, NSH>(slot, [&](int k) NVSR_INL { relu_gate_step<MASKS && !(DP_ABLATE & 1)>(k, small + S_BIAS + bias_vec * HID, h, t.acc, t.act, bp, nsc2, t.g0, t.g1, ones); }); };
        };
#define NVSR_STORE_GATES(T, GL, VEC) if constexpr (MASKS) { GL[VEC] = u32x2{T.g0, T.g1}; }
        // VPREV: the layer whose activations this layer consumes (its bias vector finishes Y in block X a; its gates are collected here)
#define NVSR_HIDDEN_LAYER(VPREV, KB_NEXT_A, NKB_A, KB_NEXT_B, NKB_B, Y_B_BLOCK)                                                        \
        ring3_sync<0>();                                                                                                            \
        nw = ring3_take(rs);                                                                                                        \
        Y.g0 = Y.g1 = 0u;                                                                                                           \
        limb_block<LIMBS, 4, true, true>(cw, lane, X.acc, cur, fa, hid(X.act, 0),                                                   \
                                         [&](int slot) NVSR_INL { relu_side(Y, VPREV)(slot); dma_side<LIMBS, NKB_A>(slot, rs, nw, KB_NEXT_A); }, tail_of(Y, 0), gate_of(X, 0)); \
        limb_block<LIMBS, 4, true, false>(cw, lane, Y.acc, cur, fa, hid(Y.act, 0), none, tail_of(X, 4), gate_of(Y, 0));             \
        cw = nw;                                                                                                                    \
        ring3_sync<0>();                                                                                                            \
        nw = ring3_take(rs);                                                                                                        \
        limb_block<LIMBS, 4, false, true>(cw, lane, X.acc, cur, fa, hid(X.act, 4),                                                  \
                                          [&](int slot) NVSR_INL { dma_side<LIMBS, NKB_B>(slot, rs, nw, KB_NEXT_B); }, tail_of(Y, 4), gate_of(X, 4)); \
        NVSR_STORE_GATES(X, glX, VPREV)                                                                                             \
        Y_B_BLOCK;                                                                                                                  \
        NVSR_STORE_GATES(Y, glY, VPREV)                                                                                             \
        cw = nw;
        // rgb layers 1, 2 (density 1, 2): Y b | X relu of this layer, tail X kb 0 (first gates of this layer)
#define NVSR_YB_PLAIN(VTHIS) X.g0 = X.g1 = 0u; \
        limb_block<LIMBS, 4, false, false>(cw, lane, Y.acc, cur, fa, hid(Y.act, 4), relu_side(X, VTHIS), tail_of(X, 0), gate_of(Y, 4))
        NVSR_HIDDEN_LAYER(4, KB_RGB1 + 4, 4, KB_RGB1 + 8, 4, NVSR_YB_PLAIN(5))
        NVSR_HIDDEN_LAYER(5, KB_RGB1 + 12, 4, KB_RGB1 + 16, 4, NVSR_YB_PLAIN(6))
        // rgb layer 3: Y b | X relu + gates (no tail: X continues with the density decoder from X.D)
        NVSR_HIDDEN_LAYER(6, KB_RGB1 + 20, 4, KB_DEN0, 3,
                          X.g0 = X.g1 = 0u;
                          (limb_block<LIMBS, 4, false, false>(cw, lane, Y.acc, cur, fa, hid(Y.act, 4), relu_gate_side(X, 7), NoTail{}, gate_of(Y, 4))))
        NVSR_STORE_GATES(X, glX, 7)

        // ---- density layer 0 (from D) -------------------------------------------------------------------------------------------
        ring3_sync<0>();
        nw = ring3_take(rs);
        // X density 0 | Y: act of rgb layer 3 (+ gates); X: rgb heads
        float hx[3] = {0.0f, 0.0f, 0.0f};
        split_feat(X.D);
        Y.g0 = Y.g1 = 0u;
        limb_block<LIMBS, 3, true, true>(cw, lane, X.acc, cur, fa, feat(X.D),
                                         [&](int slot) NVSR_INL {
                                             spread<RELU_STEPS, 0, NSF>(slot, [&](int k) NVSR_INL { relu_gate_step<MASKS && !(DP_ABLATE & 1)>(k, small + S_BIAS + 7 * HID, h, Y.acc, Y.act, bp, nsc2, Y.g0, Y.g1, ones); });
                                             spread<64, 0, NSF>(slot, [&](int k) NVSR_INL { heads_side<3>(k >> 2, k & 3, small + S_RGB_W, h, X.act, hx, hp3); });
                                             dma_side<LIMBS, 4>(slot, rs, nw, KB_DEN1);
                                         },
                                         NoTail{});
        NVSR_STORE_GATES(Y, glY, 7)
#pragma unroll
        for (int c = 0; c < 3; ++c) X.raw[c] = (hx[c] + __shfl_xor(hx[c], 32)) + small[S_HEAD_B + 1 + c];
        // Y density 0 | Y: rgb heads, then X: act of density layer 0; tail X kb 0
        float hy[3] = {0.0f, 0.0f, 0.0f};
        split_feat(Y.D);
        X.g0 = X.g1 = 0u;
        limb_block<LIMBS, 3, true, false>(cw, lane, Y.acc, cur, fa, feat(Y.D),
                                          [&](int slot) NVSR_INL {
                                              spread<64, 0, NSF>(slot, [&](int k) NVSR_INL { heads_side<3>(k >> 2, k & 3, small + S_RGB_W, h, Y.act, hy, hp3); });
                                              spread<RELU_STEPS, 0, NSF>(slot, [&](int k) NVSR_INL { relu_bias_step<LIMBS>(k, small + S_BIAS + 0 * HID, h, X.acc, X.act, bp, nsc2); });
                                          },
                                          tail_of(X, 0));
#pragma unroll
        for (int c = 0; c < 3; ++c) Y.raw[c] = (hy[c] + __shfl_xor(hy[c], 32)) + small[S_HEAD_B + 1 + c];
        cw = nw;

        // ---- density layers 1..3 -------------------------------------------------------------------------------------------------
        NVSR_HIDDEN_LAYER(0, KB_DEN1 + 4, 4, KB_DEN1 + 8, 4, NVSR_YB_PLAIN(1))
        NVSR_HIDDEN_LAYER(1, KB_DEN1 + 12, 4, KB_DEN1 + 16, 4, NVSR_YB_PLAIN(2))
        // density layer 3: the chunk issued last is the first ring chunk of the NEXT step (after the last step: a harmless copy);
        // Y b | X: act + gates, then the sigma head
        float sx[1] = {0.0f};
        NVSR_HIDDEN_LAYER(2, KB_DEN1 + 20, 4, KB_FIRST, 3,
                          X.g0 = X.g1 = 0u;
                          (limb_block<LIMBS, 4, false, false>(cw, lane, Y.acc, cur, fa, hid(Y.act, 4),
                                                              [&](int slot) NVSR_INL {
                                                                  spread<RELU_STEPS, 0, NSH / 2>(slot, [&](int k) NVSR_INL { relu_gate_step<MASKS && !(DP_ABLATE & 1)>(k, small + S_BIAS + 3 * HID, h, X.acc, X.act, bp, nsc2, X.g0, X.g1, ones); });
                                                                  spread<64, NSH / 2, NSH>(slot, [&](int k) NVSR_INL { heads_side<1>(k >> 2, k & 3, small + S_ALPHA_W, h, X.act, sx, hp1); });
                                                              },
                                                              NoTail{}, gate_of(Y, 4))))
        NVSR_STORE_GATES(X, glX, 3)
#undef NVSR_HIDDEN_LAYER
#undef NVSR_ROLL
#undef NVSR_ROLL_DMA
#undef NVSR_YB_PLAIN
        X.raw[3] = (sx[0] + __shfl_xor(sx[0], 32)) + small[S_HEAD_B];

        // ---- epilogue (exposed): Y's last activation (+ gates) + sigma head, both tiles' raw rows ----------------------------------
        Y.g0 = Y.g1 = 0u;
#pragma unroll
        for (int k = 0; k < RELU_STEPS; ++k) relu_gate_step<MASKS && !(DP_ABLATE & 1)>(k, small + S_BIAS + 3 * HID, h, Y.acc, Y.act, bp, nsc2, Y.g0, Y.g1, ones);
        NVSR_STORE_GATES(Y, glY, 3)
#undef NVSR_STORE_GATES
        {
            float hd[1];
            head_dots<1>(small + S_ALPHA_W, h, Y.act, hd);
            Y.raw[3] = hd[0] + small[S_HEAD_B];
        }
        if (lane < 32) {
            if (validX) *reinterpret_cast<f32x4*>(raw_out + (ray * S + sX) * 4) = f32x4{X.raw[0], X.raw[1], X.raw[2], X.raw[3]};
            if (validY) *reinterpret_cast<f32x4*>(raw_out + (ray * S + sY) * 4) = f32x4{Y.raw[0], Y.raw[1], Y.raw[2], Y.raw[3]};
            // range flag of the f16 limbs (nvsr.h: nvsr_set_range_flag)
            const float tx = X.raw[0] + X.raw[1] + X.raw[2] + X.raw[3], ty = Y.raw[0] + Y.raw[1] + Y.raw[2] + Y.raw[3];
            if (flag && ((validX && !(fabsf(tx) <= 3.0e38f)) || (validY && !(fabsf(ty) <= 3.0e38f))))
                __hip_atomic_fetch_or(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the copy issued for a step after the last one must land before the wave ends
}

__global__ __launch_bounds__(TPB2, 1) void decode_rays_pair_gates_kernel(SceneDev sc, const float* __restrict__ packed, long N, int S,
                                                                        const float* __restrict__ rays, const float* __restrict__ z,
                                                                        float* __restrict__ raw_out, unsigned* __restrict__ gates, unsigned* __restrict__ flag) {
#if defined(__HIP_DEVICE_COMPILE__)
    decode_pair_body<true>(sc, packed, N, S, rays, z, raw_out, gates, flag);
#endif
}
__global__ __launch_bounds__(TPB2, 1) void decode_rays_pair_kernel(SceneDev sc, const float* __restrict__ packed, long N, int S,
                                                                  const float* __restrict__ rays, const float* __restrict__ z,
                                                                  float* __restrict__ raw_out, unsigned* __restrict__ flag) {
#if defined(__HIP_DEVICE_COMPILE__)
    decode_pair_body<false>(sc, packed, N, S, rays, z, raw_out, nullptr, flag);
#endif
}

}  // namespace nvsr

using namespace nvsr;

// nvsr_decode_rays_ex (render.hip) in NVSR_ARITH_F16X2 without a weight-gradient record; arguments already validated there
extern "C" int nvsr_decode_rays_pair_launch(const nvsr_scene* scene, const float* packed_decoder, int64_t N, int S, const float* rays, const float* z,
                                            float* raw, uint32_t* gates, nvsr_stream_t stream) {
    const int nsc = (S + 31) / 32, npr = (nsc + 1) / 2;
    const int64_t nsteps = (N * (int64_t)npr + NW2 - 1) / NW2;      // 4 tile pairs per workgroup step
    // one workgroup per CU (159 KB of LDS); every workgroup walks ceil(nsteps / grid) steps
    const int64_t cus = 256;
    const int64_t per = (nsteps + cus - 1) / cus;
    const int grid = (int)((nsteps + per - 1) / per);
    if (gates)
        hipLaunchKernelGGL(decode_rays_pair_gates_kernel, dim3(grid), dim3(TPB2), 0, (hipStream_t)stream, to_dev(scene), packed_decoder, (long)N, S,
                           rays, z, raw, gates, nvsr_get_range_flag());
    else
        hipLaunchKernelGGL(decode_rays_pair_kernel, dim3(grid), dim3(TPB2), 0, (hipStream_t)stream, to_dev(scene), packed_decoder, (long)N, S,
                           rays, z, raw, nvsr_get_range_flag());
    return NVSR_CHECK_LAUNCH();
}
