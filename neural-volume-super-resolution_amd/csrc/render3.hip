// Fused render pass, third generation: the decoder GEMMs on the bf16 matrix pipe with every f32 operand split into bf16 limbs
// (limb_core.h; error bound there) -- products at 2.7x (3 limbs) or 16-bit operands at 5.3x (2 limbs) the rate of v_mfma_f32_32x32x2_f32.
//
// Same skeleton as render2.hip: one wave per SIMD owns two 32-point tiles X and Y; while the matrix pipe multiplies one tile, the other
// tile's gathers / bias + ReLU / heads are issued in the gaps.  What changes with a 32-cycle MFMA: a gap hides ~5 single-issue
// instructions (measured, tools/limb_ubench.hip: 32.5 cycles per MFMA bare, 34.4 with the limb split of the next K-block in the gaps,
// 35.4 with 3 more VALU per gap, 42 with 5 more), so all side work is cut into slices of <= 3 VALU and spread over the slots of a block.
//
// Per sample and tile: 63 K-blocks (16 input channels) x 4 output blocks x NP MFMAs; weights stream through a 2-slot LDS ring in 17
// chunks (a plane's share of a feature layer = 3 K-blocks, half a hidden layer = 4), 12 * LIMBS KB per K-block.
// Biases are not preloaded into the accumulators: the first MFMA of a layer takes C = 0 and act = max(acc + bias, 0).
#include "pair_core.h"

namespace nvsr {

// =====================================================================================================================
// The body of both kernels below (one instantiation per LIMBS; the kernels are thin shells so that the coarse and the fine pass are two
// symbols in a rocprofv3 kernel trace -- a second template parameter on one kernel trips hipcc's host pass over the LDS-DMA builtins).
// ZCOMP: the depths are the un-jittered coarse ones (train_utils.py:95-100) and are computed from the ray's near / far in registers
// (coarse_depth, bit for bit what nvsr_coarse_z writes) instead of being read: `z` is NULL and `lindisp` selects the spacing
// (ONE template parameter, LZ = LIMBS + 8 * ZCOMP: with a second one hipcc's host pass fails to resolve the LDS-DMA helpers inside the body)
template <int LZ>
__device__ __forceinline__ void render_pass3_body(const SceneDev& sc, const float* __restrict__ packed, long N, int S,
                                                  const float* __restrict__ rays, const float* __restrict__ z, int lindisp,
                                                  const float* __restrict__ noise, int white,
                                                  float* __restrict__ rgb, float* __restrict__ disp,
                                                  float* __restrict__ acc, float* __restrict__ weights,
                                                  float* __restrict__ depth, float* __restrict__ raw_out, unsigned* __restrict__ flag) {
    constexpr int LIMBS = LZ & 7;
    constexpr bool ZCOMP = (LZ & 8) != 0;
    using L = Lds3<LIMBS>;
    constexpr int NP = limb_products(LIMBS);
    constexpr int NSF = 3 * 4 * NP, NSH = 4 * 4 * NP;          // slots of a feature block / of half a hidden layer
    __shared__ __attribute__((aligned(16))) unsigned lds[L::TOTAL];
    NVSR_RACE_PROBE_DELAY(lds);      // (probe builds only, nvsr_common.h)
    Ring3<LIMBS> rs{__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(packed + limb_region(LIMBS)), 0, KB_TOTAL * kb_words(LIMBS) * 4, 0x00020000),
                    lds, 0, __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), (int)(threadIdx.x & 63), (threadIdx.x >> 6) * 1024u + (threadIdx.x & 63) * 16u};
    // (wave index as a SCALAR: the LDS destination of every weight-copy piece is then scalar arithmetic into M0 instead of a vector add + v_readfirstlane per piece)
    float* ldsf = reinterpret_cast<float*>(lds);
    // (f16 limbs: activations are held as x 2^F16_SX -- biases scaled up, head weights scaled down, all exact; limb_core.h)
    // A weight beyond the f16 range was packed as inf: the packer then left a NaN in the blob's spare slot S_F16_POISON, which goes into
    // the head biases here -- every output of such a decoder is NaN instead of a wrong number.
    for (int i = threadIdx.x; i < SMALL_FLOATS; i += TPB2) {
        float v = packed[P_SMALL + i] * (LIMBS != 2 ? 1.0f : i < S_ALPHA_W ? F16_X_SCALE : i < S_HEAD_B ? F16_HEAD_SCALE : 1.0f);
        if (LIMBS == 2 && i >= S_HEAD_B && i < S_HEAD_B + 4) v += packed[P_SMALL + S_F16_POISON];
        ldsf[L::SMALL + i] = v;
    }
    const float* small = ldsf + L::SMALL;

    const int lane0 = rs.lane;
    // XCD-aware ray blocks: workgroup b runs on XCD b % 8 (round-robin dispatch), each XCD has its own L2.  Give XCD x the x-th contiguous
    // eighth of the ray blocks, so that the workgroups that share an L2 render neighbouring image rows (their taps share texels).
    const unsigned nblk = gridDim.x, xcd = blockIdx.x & 7u, per = nblk >> 3, rem = nblk & 7u;
    const unsigned blk = xcd * per + (xcd < rem ? xcd : rem) + (blockIdx.x >> 3);
    const long base = (long)blk * RAYS2 + rs.wave * 64 + (lane0 & 31);
    long rayX = base, rayY = base + 32;
    const bool validX = rayX < N, validY = rayY < N;
    if (!validX) rayX = N - 1;
    if (!validY) rayY = N - 1;
    constexpr int RAY3_FLOATS = L::RAY_FLOATS;
    float* rcX = ldsf + L::RAYS + (rs.wave * 64 + (lane0 & 31)) * RAY3_FLOATS;
    float* rcY = rcX + 32 * RAY3_FLOATS;
    float* rtX = LIMBS == 2 ? ldsf + L::VTAPS + (rs.wave * 64 + (lane0 & 31)) * L::TAP_FLOATS : rcX + 8;     // view-plane taps of the ray
    float* rtY = rtX + 32 * (LIMBS == 2 ? L::TAP_FLOATS : RAY3_FLOATS);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float* r = rays + (k ? rayY : rayX) * 11;
        float* rc = k ? rcY : rcX;
        float* rtp = k ? rtY : rtX;
        const float dx = r[3], dy = r[4], dz = r[5];
        const Taps vt = view_taps(sc, r[8], r[9], r[10]);
        const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
        if (lane0 < 32) {
            reinterpret_cast<f32x4*>(rc)[0] = f32x4{r[0], r[1], r[2], dx};
            reinterpret_cast<f32x4*>(rc)[1] = f32x4{dy, dz, nrm, r[6]};
            if constexpr (L::FAR >= 0) ldsf[L::FAR + rs.wave * 64 + (lane0 & 31) + 32 * k] = r[7];
            else rc[16] = r[7];
            reinterpret_cast<f32x4*>(rtp)[0] = f32x4{__int_as_float(vt.o00), __int_as_float(vt.o01), __int_as_float(vt.o10), __int_as_float(vt.o11)};
            reinterpret_cast<f32x4*>(rtp)[1] = f32x4{vt.nw, vt.ne, vt.sw, vt.se};
        }
    }
    const float* zX = ZCOMP ? nullptr : z + rayX * S;
    const float* zY = ZCOMP ? nullptr : z + rayY * S;
    auto depth_of = [&](const float* zp, const float* rc, int k) NVSR_INL {
        if constexpr (ZCOMP) return coarse_depth(rc[7], L::FAR >= 0 ? ldsf[L::FAR + rs.wave * 64 + (rs.lane & 31) + (rc == rcX ? 0 : 32)] : rc[16], k, S, lindisp);
        else return zp[k];
    };

    Tile3 X, Y;
    X.T = Y.T = 1.0f;
    X.cr = X.cg = X.cb = X.dep = X.ac = 0.0f;
    Y.cr = Y.cg = Y.cb = Y.dep = Y.ac = 0.0f;
    if constexpr (ZCOMP) __syncthreads();   // (the ray cache is written by lanes 0..31 and read by all 64, see below)
    X.zc = depth_of(zX, rcX, 0); Y.zc = depth_of(zY, rcY, 0);
    RawTaps4 rt;

    auto point_norm = [&](const float* rc, float zc, float& n0, float& n1, float& n2) NVSR_INL {
        const f32x4 c0 = reinterpret_cast<const f32x4*>(rc)[0], c1 = reinterpret_cast<const f32x4*>(rc)[1];
        n0 = norm_coord(__fadd_rn(c0[0], __fmul_rn(c0[3], zc)), sc.lo[0], sc.range[0]);
        n1 = norm_coord(__fadd_rn(c0[1], __fmul_rn(c1[0], zc)), sc.lo[1], sc.range[1]);
        n2 = norm_coord(__fadd_rn(c0[2], __fmul_rn(c1[1], zc)), sc.lo[2], sc.range[2]);
    };
    // f16 limbs: features carry the activation scale 2^F16_SX, put on the four blend weights (exact; D = (F0 + F1 + F2) / 3 inherits it)
    auto scale_taps = [](Taps& t) NVSR_INL {
        if constexpr (LIMBS == 2) { t.nw *= F16_X_SCALE; t.ne *= F16_X_SCALE; t.sw *= F16_X_SCALE; t.se *= F16_X_SCALE; }
    };
    auto view_job = [&](const float* rtp) NVSR_INL {
        const f32x4 c2 = reinterpret_cast<const f32x4*>(rtp)[0], c3 = reinterpret_cast<const f32x4*>(rtp)[1];
        GatherJob j;
        j.plane = sc.plane[3];
        j.t.o00 = __float_as_int(c2[0]); j.t.o01 = __float_as_int(c2[1]); j.t.o10 = __float_as_int(c2[2]); j.t.o11 = __float_as_int(c2[3]);
        j.t.nw = c3[0]; j.t.ne = c3[1]; j.t.sw = c3[2]; j.t.se = c3[3];
        scale_taps(j.t);
        return j;
    };

    __syncthreads();   // the ray cache is written by lanes 0..31 and read by all 64: without a barrier hipcc moves the reads of the upper
                       // half above the writes (per lane there is no dependence)
    // The view-plane features (project_viewdir, models.py:312-326) depend on the ray only: gathered once.  They open every sample's rgb
    // layer 0, so that the first plane gathers of a sample have two blocks to land in.
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
        Tile3& t = k2 ? Y : X;
        const GatherJob vj = view_job(k2 ? rtY : rtX);
#pragma unroll
        for (int k = 0; k < 12; ++k) gather4_load(k, vj, lane0 >> 5, rt);
#pragma unroll
        for (int c = 0; c < HALF_C; ++c) gather4_blend(c, vj, rt, t.V);
    }
    Limbs<LIMBS> cur, fa;
    f32x2_t nsc = {-F16_ACC_UNSCALE, -F16_ACC_UNSCALE};                  // relu_bias_step
    asm volatile("" : "+s"(nsc));
#if R3_STAMP
    float stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
#endif
    constexpr bool RESIDENT = L::RES_KB > 0;
    constexpr int KB_FIRST = RESIDENT ? KB_RGB0 + 9 : KB_RGB0;      // the first chunk of a step that goes through the ring
    unsigned* const res = lds + L::RES;
    if constexpr (RESIDENT) ring3_load_resident<LIMBS, L::RES_KB>(rs, res, KB_RGB0);
    unsigned* cw = const_cast<unsigned*>(ring3_issue<LIMBS, 3>(rs, KB_FIRST));     // first ring chunk of sample 0; every later one is issued during the previous sample
    for (int s = 0; s < S; ++s) {
        asm volatile("" : "+v"(rs.voff), "+v"(rs.lane));
#if R3_NO_VIEW_HOIST
        // The view features are loop-invariant and so are their limbs: hipcc hoists the two split_feat(V) of a step out of the sample loop
        // (72 registers of limbs), runs out of registers and spills 8 of them plus the two depth-row pointers -- 4 scratch reloads per
        // sample, each behind an s_waitcnt vmcnt(0) that also waits for every gather and weight copy in flight.  Opaque per iteration.
#pragma unroll
        for (int c = 0; c < HALF_C; ++c) asm volatile("" : "+v"(X.V[c]), "+v"(Y.V[c]));
#endif
        const int lane = rs.lane, h = lane >> 5;
        const bool last = (s + 1 == S);
        X.zn = depth_of(zX, rcX, last ? s : s + 1);            // (unconditional loads unless ZCOMP: ring3_sync<2> below counts them)
        Y.zn = depth_of(zY, rcY, last ? s : s + 1);
        const float nzX = noise ? noise[rayX * S + s] : 0.0f;
        const float nzY = noise ? noise[rayY * S + s] : 0.0f;
        float xn0, xn1, xn2, yn0, yn1, yn2;
        point_norm(rcX, X.zc, xn0, xn1, xn2);
        point_norm(rcY, Y.zc, yn0, yn1, yn2);
        BiasPend4 bp;
        bp.slot = bounce_slot(ldsf + L::VTAPS + rs.wave * 64 * L::TAP_FLOATS + lane * 4);     // (f16 limbs: the wave's tap region, dead since the prologue)
        HeadPend<3> hp3;
        HeadPend<1> hp1;
        SplitPend tp;

        // side-work pieces
        auto feat = [](const float (&f)[HALF_C]) NVSR_INL { return [&f](int kb, int i) NVSR_INL { return f[8 * kb + i]; }; };
        auto hid = [](const f32x16 (&a)[4], int kb0) NVSR_INL { return [&a, kb0](int kb, int i) NVSR_INL { const int k = kb0 + kb; return a[k >> 1][8 * (k & 1) + i]; }; };
        auto split_feat = [&](const float (&f)[HALF_C]) NVSR_INL { split_all<LIMBS>([&f](int i) NVSR_INL { return f[i]; }, cur); };
        // tail: split K-block kb of t.act into the limbs the next block starts with
        auto tail_of = [&](const f32x16 (&a)[4], int kb) NVSR_INL {
            return [&a, kb, &tp](int slice, Limbs<LIMBS>& nxt) NVSR_INL { split_slice<LIMBS>(slice, [&a, kb](int i) NVSR_INL { return a[kb >> 1][8 * (kb & 1) + i]; }, nxt, tp); };
        };
        auto none = [](int) NVSR_INL {};

        // ---- rgb layer 0: (view plane, planes 0..2) x (X block, Y block).  The gathers roll through the blocks (gather_roll): the block that
        // multiplies plane p - 1 of a tile loads plane p of the same tile, the next block blends it.
        GatherJob ja, jb;
        R3_MARK(0)      // loop top
        ring3_sync<ZCOMP ? 0 : 2>();                             // the step's first ring chunk (issued during the previous sample) -- younger: the two z loads above
        // RESIDENT (f16 limbs): view plane, planes 0 and 1 multiply out of the resident region; the ring chunk that has just landed is plane 2's,
        // and the blocks that issue gathers (B0 .. B5) issue no weight copy and need no ring wait
        unsigned* nw = RESIDENT ? nullptr : ring3_take(rs);
        const unsigned* const w_view = RESIDENT ? res : cw;
        R3_MARK(1)      // first ring wait
        R3_RESET
#define NVSR_ROLL(TL, JL, TB, JB, LOADS, BLENDS) [&](int slot) NVSR_INL { gather_roll<NSF, LOADS, BLENDS, LIMBS == 2 && R3_BLEND_PK>(slot, JL, TL.F, JB, TB.F, h, rt); }
#define NVSR_ROLL_DMA(TL, JL, TB, JB, LOADS, BLENDS, NKB, KB0) \
        [&](int slot) NVSR_INL { gather_roll<NSF, LOADS, BLENDS, LIMBS == 2 && R3_BLEND_PK>(slot, JL, TL.F, JB, TB.F, h, rt); dma_side<LIMBS, NKB>(slot, rs, nw, KB0); }
        // X view | loads X plane 0
        ja.plane = sc.plane[0]; ja.t = pos_taps2(sc, 0, xn0, xn1, xn2); scale_taps(ja.t);
        split_feat(X.V);
        if constexpr (RESIDENT) limb_block<LIMBS, 3, true, true>(w_view, lane, X.acc, cur, fa, feat(X.V), NVSR_ROLL(X, ja, Y, jb, true, false), NoTail{});
        else limb_block<LIMBS, 3, true, true>(cw, lane, X.acc, cur, fa, feat(X.V), NVSR_ROLL_DMA(X, ja, Y, jb, true, false, 3, KB_RGB0 + 3), NoTail{});
        R3_MARKB(0)
        // Y view | blends X plane 0, loads Y plane 0
        jb.plane = sc.plane[0]; jb.t = pos_taps2(sc, 0, yn0, yn1, yn2); scale_taps(jb.t);
        split_feat(Y.V);
        limb_block<LIMBS, 3, true, false>(w_view, lane, Y.acc, cur, fa, feat(Y.V), NVSR_ROLL(Y, jb, X, ja, true, true), NoTail{});
        R3_MARKB(1)
        const unsigned* w_p0 = res + 3 * kb_words(LIMBS);
        if constexpr (!RESIDENT) {
            cw = nw;
            ring3_sync<YOUNGER_THAN_CHUNK>();
            nw = ring3_take(rs);
            w_p0 = cw;
        }
        // X plane 0 | blends Y plane 0, loads X plane 1
#pragma unroll
        for (int c = 0; c < HALF_C; ++c) X.D[c] = X.F[c];
        ja.plane = sc.plane[1]; ja.t = pos_taps2(sc, 1, xn0, xn1, xn2); scale_taps(ja.t);
        split_feat(X.F);
        if constexpr (RESIDENT) limb_block<LIMBS, 3, false, true>(w_p0, lane, X.acc, cur, fa, feat(X.F), NVSR_ROLL(X, ja, Y, jb, true, true), NoTail{});
        else limb_block<LIMBS, 3, false, true>(cw, lane, X.acc, cur, fa, feat(X.F), NVSR_ROLL_DMA(X, ja, Y, jb, true, true, 3, KB_RGB0 + 6), NoTail{});
        R3_MARKB(2)
        // Y plane 0 | blends X plane 1, loads Y plane 1
#pragma unroll
        for (int c = 0; c < HALF_C; ++c) Y.D[c] = Y.F[c];
        jb.plane = sc.plane[1]; jb.t = pos_taps2(sc, 1, yn0, yn1, yn2); scale_taps(jb.t);
        split_feat(Y.F);
        limb_block<LIMBS, 3, false, false>(w_p0, lane, Y.acc, cur, fa, feat(Y.F), NVSR_ROLL(Y, jb, X, ja, true, true), NoTail{});
        R3_MARKB(3)
        const unsigned* w_p1 = res + 6 * kb_words(LIMBS);
        if constexpr (!RESIDENT) {
            cw = nw;
            ring3_sync<YOUNGER_THAN_CHUNK>();
            nw = ring3_take(rs);
            w_p1 = cw;
        }
        // X plane 1 | blends Y plane 1, loads X plane 2
#pragma unroll
        for (int c = 0; c < HALF_C; ++c) X.D[c] = __fadd_rn(X.D[c], X.F[c]);
        ja.plane = sc.plane[2]; ja.t = pos_taps2(sc, 2, xn0, xn1, xn2); scale_taps(ja.t);
        split_feat(X.F);
        if constexpr (RESIDENT) limb_block<LIMBS, 3, false, true>(w_p1, lane, X.acc, cur, fa, feat(X.F), NVSR_ROLL(X, ja, Y, jb, true, true), NoTail{});
        else limb_block<LIMBS, 3, false, true>(cw, lane, X.acc, cur, fa, feat(X.F), NVSR_ROLL_DMA(X, ja, Y, jb, true, true, 3, KB_RGB0 + 9), NoTail{});
        R3_MARKB(4)
        // Y plane 1 | blends X plane 2, loads Y plane 2
#pragma unroll
        for (int c = 0; c < HALF_C; ++c) Y.D[c] = __fadd_rn(Y.D[c], Y.F[c]);
        jb.plane = sc.plane[2]; jb.t = pos_taps2(sc, 2, yn0, yn1, yn2); scale_taps(jb.t);
        split_feat(Y.F);
        limb_block<LIMBS, 3, false, false>(w_p1, lane, Y.acc, cur, fa, feat(Y.F), NVSR_ROLL(Y, jb, X, ja, true, true), NoTail{});
        R3_MARKB(5)
        if constexpr (!RESIDENT) {
            cw = nw;
            ring3_sync<YOUNGER_THAN_CHUNK>();
        }
        nw = ring3_take(rs);                                     // (RESIDENT: cw is plane 2's chunk since the top of the step)
        // X plane 2 | blends Y plane 2;  D = (D + F) / 3   (combine_pos_planes 'avg', models.py:358-359)
#pragma unroll
        for (int c = 0; c < HALF_C; ++c) X.D[c] = div3(__fadd_rn(X.D[c], X.F[c]));
        split_feat(X.F);
        limb_block<LIMBS, 3, false, true>(cw, lane, X.acc, cur, fa, feat(X.F), NVSR_ROLL_DMA(X, ja, Y, jb, false, true, 4, KB_RGB1), NoTail{});
        R3_MARKB(6)
        // Y plane 2 | X: act = max(acc + bias, 0); tail: limbs of X's K-block 0
#pragma unroll
        for (int c = 0; c < HALF_C; ++c) Y.D[c] = div3(__fadd_rn(Y.D[c], Y.F[c]));
        split_feat(Y.F);
        limb_block<LIMBS, 3, false, false>(cw, lane, Y.acc, cur, fa, feat(Y.F),
                                           [&](int slot) NVSR_INL { spread<RELU_STEPS, 0, NSF>(slot, [&](int k) NVSR_INL { relu_bias_step<LIMBS>(k, small + S_BIAS + 4 * HID, h, X.acc, X.act, bp, nsc); }); },
                                           tail_of(X.act, 0));
        R3_MARKB(7)
        cw = nw;
        R3_MARK(2)      // rgb layer 0

        // ---- hidden layers.  Layer l of a decoder = chunks a (K-blocks 0..3), b (4..7):
        //   X a | Y: act of layer l-1; tail Y kb 0        Y a | tail X kb 4        X b | tail Y kb 4        Y b | X: act of layer l; tail X kb 0
        auto relu_side = [&](Tile3& t, int bias_vec) NVSR_INL {
            return [&, bias_vec](int slot) NVSR_INL { spread<RELU_STEPS, 0, NSH>(slot, [&](int k) NVSR_INL { relu_bias_step<LIMBS>(k, small + S_BIAS + bias_vec * HID, h, t.acc, t.act, bp, nsc); }); };
        };
        // hidden layer with bias vectors: vprev (layer l-1, finishing Y) and vthis (layer l, finishing X); kbn = next chunk to issue (two per layer)
#define NVSR_HIDDEN_LAYER_(SYNC, XA_SIDE, VPREV, VTHIS, KB_NEXT_A, NKB_A, KB_NEXT_B, NKB_B, X_B_SIDE)                                \
        R3_RESETH                                                                                                                   \
        SYNC;                                                                                                                       \
        R3_MARKH(0)                                                                                                                 \
        nw = ring3_take(rs);                                                                                                        \
        R3_MARKH(1)                                                                                                                 \
        limb_block<LIMBS, 4, true, true>(cw, lane, X.acc, cur, fa, hid(X.act, 0),                                                   \
                                         [&](int slot) NVSR_INL { (XA_SIDE)(slot); dma_side<LIMBS, NKB_A>(slot, rs, nw, KB_NEXT_A); }, tail_of(Y.act, 0)); \
        R3_MARKH(2)                                                                                                                 \
        limb_block<LIMBS, 4, true, false>(cw, lane, Y.acc, cur, fa, hid(Y.act, 0), none, tail_of(X.act, 4));                        \
        R3_MARKH(3)                                                                                                                 \
        cw = nw;                                                                                                                    \
        ring3_sync<0>();                                                                                                               \
        nw = ring3_take(rs);                                                                                                        \
        R3_MARKH(4)                                                                                                                 \
        limb_block<LIMBS, 4, false, true>(cw, lane, X.acc, cur, fa, hid(X.act, 4),                                                  \
                                          [&](int slot) NVSR_INL { dma_side<LIMBS, NKB_B>(slot, rs, nw, KB_NEXT_B); }, tail_of(Y.act, 4));   \
        R3_MARKH(5)                                                                                                                 \
        X_B_SIDE;                                                                                                                   \
        R3_MARKH(6)                                                                                                                 \
        cw = nw;

#define NVSR_HIDDEN_LAYER(VPREV, ...) NVSR_HIDDEN_LAYER_(ring3_sync<0>(), relu_side(Y, VPREV), VPREV, __VA_ARGS__)
        // rgb layers 1, 2: Y b | X relu, tail X kb 0
#define NVSR_YB_PLAIN(VTHIS) (limb_block<LIMBS, 4, false, false>(cw, lane, Y.acc, cur, fa, hid(Y.act, 4), relu_side(X, VTHIS), tail_of(X.act, 0)))
        NVSR_HIDDEN_LAYER(4, 5, KB_RGB1 + 4, 4, KB_RGB1 + 8, 4, NVSR_YB_PLAIN(5))
        NVSR_HIDDEN_LAYER(5, 6, KB_RGB1 + 12, 4, KB_RGB1 + 16, 4, NVSR_YB_PLAIN(6))
        // rgb layer 3: Y b | X relu (no tail: X continues with the density decoder from X.D)
        NVSR_HIDDEN_LAYER(6, 7, KB_RGB1 + 20, 4, KB_DEN0, 3,
                          (limb_block<LIMBS, 4, false, false>(cw, lane, Y.acc, cur, fa, hid(Y.act, 4), relu_side(X, 7), NoTail{})))

        R3_MARK(3)      // rgb layers 1..3
        // ---- density layer 0 (from D) -------------------------------------------------------------------------------------------
        ring3_sync<0>();
        nw = ring3_take(rs);
        // X density 0 | Y: act of rgb layer 3; X: rgb heads
        float hx[3] = {0.0f, 0.0f, 0.0f};
        split_feat(X.D);
        limb_block<LIMBS, 3, true, true>(cw, lane, X.acc, cur, fa, feat(X.D),
                                         [&](int slot) NVSR_INL {
                                             spread<RELU_STEPS, 0, NSF>(slot, [&](int k) NVSR_INL { relu_bias_step<LIMBS>(k, small + S_BIAS + 7 * HID, h, Y.acc, Y.act, bp, nsc); });
                                             spread<64, 0, NSF>(slot, [&](int k) NVSR_INL { heads_side<3>(k >> 2, k & 3, small + S_RGB_W, h, X.act, hx, hp3); });
                                             dma_side<LIMBS, 4>(slot, rs, nw, KB_DEN1);
                                         },
                                         NoTail{});
#pragma unroll
        for (int c = 0; c < 3; ++c) X.raw[c] = (hx[c] + __shfl_xor(hx[c], 32)) + small[S_HEAD_B + 1 + c];
        // Y density 0 | Y: rgb heads, then X: act of density layer 0; tail X kb 0
        float hy[3] = {0.0f, 0.0f, 0.0f};
        split_feat(Y.D);
        limb_block<LIMBS, 3, true, false>(cw, lane, Y.acc, cur, fa, feat(Y.D),
                                          [&](int slot) NVSR_INL {
                                              spread<64, 0, NSF>(slot, [&](int k) NVSR_INL { heads_side<3>(k >> 2, k & 3, small + S_RGB_W, h, Y.act, hy, hp3); });
                                              spread<RELU_STEPS, 0, NSF>(slot, [&](int k) NVSR_INL { relu_bias_step<LIMBS>(k, small + S_BIAS + 0 * HID, h, X.acc, X.act, bp, nsc); });
                                          },
                                          tail_of(X.act, 0));
#pragma unroll
        for (int c = 0; c < 3; ++c) Y.raw[c] = (hy[c] + __shfl_xor(hy[c], 32)) + small[S_HEAD_B + 1 + c];
        cw = nw;

        R3_MARK(4)      // density layer 0
        // ---- density layers 1..3 -------------------------------------------------------------------------------------------------
        NVSR_HIDDEN_LAYER(0, 1, KB_DEN1 + 4, 4, KB_DEN1 + 8, 4, NVSR_YB_PLAIN(1))
        NVSR_HIDDEN_LAYER(1, 2, KB_DEN1 + 12, 4, KB_DEN1 + 16, 4, NVSR_YB_PLAIN(2))
        // density layer 3: the chunk issued last is chunk 0 of the NEXT sample (after the last sample: a harmless copy);
        // Y b | X: act, then the sigma head
        float sx[1] = {0.0f};
        NVSR_HIDDEN_LAYER(2, 3, KB_DEN1 + 20, 4, KB_FIRST, 3,
                          (limb_block<LIMBS, 4, false, false>(cw, lane, Y.acc, cur, fa, hid(Y.act, 4),
                                                              [&](int slot) NVSR_INL {
                                                                  spread<RELU_STEPS, 0, NSH / 2>(slot, [&](int k) NVSR_INL { relu_bias_step<LIMBS>(k, small + S_BIAS + 3 * HID, h, X.acc, X.act, bp, nsc); });
                                                                  spread<64, NSH / 2, NSH>(slot, [&](int k) NVSR_INL { heads_side<1>(k >> 2, k & 3, small + S_ALPHA_W, h, X.act, sx, hp1); });
                                                              },
                                                              NoTail{})))
#undef NVSR_HIDDEN_LAYER
#undef NVSR_HIDDEN_LAYER_
#undef NVSR_ROLL
#undef NVSR_ROLL_DMA
#undef NVSR_YB_PLAIN
        X.raw[3] = (sx[0] + __shfl_xor(sx[0], 32)) + small[S_HEAD_B];
        R3_MARK(5)      // density layers 1..3

        // ---- epilogue (exposed): Y's last activation + sigma head, both tiles' compositing -----------------------------------------
#pragma unroll
        for (int k = 0; k < RELU_STEPS; ++k) relu_bias_step<LIMBS>(k, small + S_BIAS + 3 * HID, h, Y.acc, Y.act, bp, nsc);
        {
            float hd[1];
            head_dots<1>(small + S_ALPHA_W, h, Y.act, hd);
            Y.raw[3] = hd[0] + small[S_HEAD_B];
        }
        if (raw_out && lane < 32) {
            if (validX) *reinterpret_cast<f32x4*>(raw_out + (rayX * S + s) * 4) = f32x4{X.raw[0], X.raw[1], X.raw[2], X.raw[3]};
            if (validY) *reinterpret_cast<f32x4*>(raw_out + (rayY * S + s) * 4) = f32x4{Y.raw[0], Y.raw[1], Y.raw[2], Y.raw[3]};
        }
        composite_sample(X, reinterpret_cast<const f32x4*>(rcX)[1][2], nzX, last);
        composite_sample(Y, reinterpret_cast<const f32x4*>(rcY)[1][2], nzY, last);
        if (weights && lane < 32) {
            if (validX) weights[rayX * S + s] = X.raw[3];
            if (validY) weights[rayY * S + s] = Y.raw[3];
        }
        X.zc = X.zn; Y.zc = Y.zn;
        R3_MARK(6)      // epilogue
    }
#if R3_STAMP
    if (raw_out && rs.lane < 32 && validX) {
        for (int i = 0; i < 8; ++i) raw_out[(rayX * S + (i >> 2)) * 4 + (i & 3)] = stamp[i] / (float)S;
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the copy issued for a sample after the last one must land before the wave ends

    if (rs.lane < 32) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const Tile3& t = k ? Y : X;
            const long ray = k ? rayY : rayX;
            if (!(k ? validY : validX)) continue;
            float cr = t.cr, cg = t.cg, cb = t.cb;
            const float q = t.dep / t.ac;                       // NaN when acc == 0, like torch.max(1e-10, nan)
            disp[ray] = 1.0f / ((q != q) ? q : fmaxf(1e-10f, q));
            if (white) { const float bg = 1.0f - t.ac; cr += bg; cg += bg; cb += bg; }
            rgb[ray * 3 + 0] = cr; rgb[ray * 3 + 1] = cg; rgb[ray * 3 + 2] = cb;
            acc[ray] = t.ac;
            // range flag of the f16 limbs (nvsr.h: nvsr_set_range_flag): a non-finite colour / opacity is an operand beyond the static scales
            if (LIMBS == 2 && flag && !(fabsf(cr + cg + cb + t.ac) <= 3.0e38f)) __hip_atomic_fetch_or(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (depth) depth[ray] = t.dep;
        }
    }
}

// coarse pass: also writes the per-sample compositing weights (the input of the importance resampling)
template <int LIMBS>
__global__ __launch_bounds__(TPB2, 1) void render_pass3_coarse_kernel(SceneDev sc, const float* __restrict__ packed, long N, int S,
                                                                     const float* __restrict__ rays, const float* __restrict__ z,
                                                                     const float* __restrict__ noise, int white,
                                                                     float* __restrict__ rgb, float* __restrict__ disp,
                                                                     float* __restrict__ acc, float* __restrict__ weights,
                                                                     float* __restrict__ depth, float* __restrict__ raw_out, unsigned* __restrict__ flag) {
    render_pass3_body<LIMBS>(sc, packed, N, S, rays, z, 0, noise, white, rgb, disp, acc, weights, depth, raw_out, flag);
}
// coarse pass of an inference frame: un-jittered depths computed in registers (no [N,S] depth tensor is written or read)
template <int LIMBS>
__global__ __launch_bounds__(TPB2, 1) void render_pass3_coarse_z_kernel(SceneDev sc, const float* __restrict__ packed, long N, int S,
                                                                       const float* __restrict__ rays, int lindisp,
                                                                       const float* __restrict__ noise, int white,
                                                                       float* __restrict__ rgb, float* __restrict__ disp,
                                                                       float* __restrict__ acc, float* __restrict__ weights,
                                                                       float* __restrict__ depth, float* __restrict__ raw_out, unsigned* __restrict__ flag) {
#if defined(__HIP_DEVICE_COMPILE__)      // (hipcc's host pass cannot resolve the LDS-DMA helpers inside this instantiation; it only needs the stub)
    render_pass3_body<LIMBS + 8>(sc, packed, N, S, rays, nullptr, lindisp, noise, white, rgb, disp, acc, weights, depth, raw_out, flag);
#endif
}
// fine pass (or any pass whose weights are not wanted)
template <int LIMBS>
__global__ __launch_bounds__(TPB2, 1) void render_pass3_kernel(SceneDev sc, const float* __restrict__ packed, long N, int S,
                                                              const float* __restrict__ rays, const float* __restrict__ z,
                                                              const float* __restrict__ noise, int white,
                                                              float* __restrict__ rgb, float* __restrict__ disp,
                                                              float* __restrict__ acc, float* __restrict__ depth,
                                                              float* __restrict__ raw_out, unsigned* __restrict__ flag) {
    render_pass3_body<LIMBS>(sc, packed, N, S, rays, z, 0, noise, white, rgb, disp, acc, nullptr, depth, raw_out, flag);
}

// ---- natural blob -> bf16 limb fragments (the tail of the packed blob) -----------------------------------------------------------
template <int LIMBS>
__global__ void pack_decoder_limbs_kernel(const float* __restrict__ nat, unsigned* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= KB_TOTAL * kb_words(LIMBS)) return;
    const int w = idx & 3, lane = (idx >> 2) & 63, frag = (idx >> 8) % (4 * LIMBS), rec = idx / kb_words(LIMBS);
    const int t = frag % LIMBS, ob = frag / LIMBS;
    const int i = 32 * ob + (lane & 31), h = lane >> 5;
    unsigned word = 0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int e = 2 * w + half;
        int src;
        if (rec < KB_RGB1) {
            const int plane = rec < 3 ? 3 : rec / 3 - 1;          // consumption order: view plane, then planes 0..2
            src = N_RGB_W0 + i * (4 * C) + C * plane + HALF_C * h + 8 * (rec % 3) + e;
        } else if (rec >= KB_DEN0 && rec < KB_DEN1) {
            src = N_DEN_W0 + i * C + HALF_C * h + 8 * (rec - KB_DEN0) + e;
        } else {
            const bool is_rgb = rec < KB_DEN0;
            const int r = rec - (is_rgb ? KB_RGB1 : KB_DEN1), l = r / 8, kb = r % 8;
            const int k = 32 * (kb >> 1) + 16 * (kb & 1) + 8 * (e >> 2) + 4 * h + (e & 3);
            src = (is_rgb ? N_RGB_W1 : N_DEN_W1) + l * N_HID_STRIDE + i * HID + k;
        }
        float v = nat[src];
        unsigned bits = 0;
        if (LIMBS == 3) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (k == t) bits = __float_as_uint(v) >> 16;
                v = limb_rest(v);
            }
        } else {
            v *= F16_W_SCALE;                                   // limb_core.h: W 2^F16_SW as hi + lo, both rounded to nearest
            const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
            if (!(fabsf(v) <= 65504.0f)) reinterpret_cast<float*>(out)[P_SMALL + S_F16_POISON - P_LIMB2] = __builtin_nanf("");   // (out = packed + P_LIMB2)
            bits = __builtin_bit_cast(unsigned short, t == 0 ? hi : lo);
        }
        word |= bits << (16 * half);
    }
    out[idx] = word;
}

// ---- the arithmetic primitive alone: Y[32 x 32] = W[32 x K] X[K x 32] by one wave, operands split exactly as the kernels / the pack
// kernel split them (LIMBS = 0: v_mfma_f32_32x32x2_f32).  Test hook (nvsr_limb_gemm_probe): error bounds on chosen operands.
template <int LIMBS>
__global__ void limb_gemm_probe_kernel(int K, const float* __restrict__ W, const float* __restrict__ X, float* __restrict__ Y) {
    const int lane = threadIdx.x, m = lane & 31, h = lane >> 5;
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float unscale = 1.0f;
    if constexpr (LIMBS == 0) {
        for (int k = 0; k < K; k += 2) acc = mfma32(W[m * K + k + h], X[(k + h) * 32 + m], acc);
    } else {
        const float sw = LIMBS == 2 ? F16_W_SCALE : 1.0f, sx = LIMBS == 2 ? F16_X_SCALE : 1.0f;
        unscale = 1.0f / (sw * sx);
        for (int kb = 0; kb < K / 16; ++kb) {
            Limbs<LIMBS> a, b;
            float wa[8], xb[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { wa[i] = W[m * K + 16 * kb + 8 * h + i] * sw; xb[i] = X[(16 * kb + 8 * h + i) * 32 + m] * sx; }
            split_all<LIMBS>([&](int i) NVSR_INL { return wa[i]; }, a);
            split_all<LIMBS>([&](int i) NVSR_INL { return xb[i]; }, b);
#pragma unroll
            for (int p = 0; p < limb_products(LIMBS); ++p) acc = mfma_limb<LIMBS>(a.v[limb_w(LIMBS, p)], b.v[limb_x(LIMBS, p)], acc);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) Y[(8 * (r >> 2) + 4 * h + (r & 3)) * 32 + m] = acc[r] * unscale;
}

}  // namespace nvsr

using namespace nvsr;

extern "C" int nvsr_limb_gemm_probe(int arithmetic, int K, const float* W, const float* X, float* Y, nvsr_stream_t stream) {
    if (!W || !X || !Y) return NVSR_ERR_NULL;
    if (K < 16 || K % 16) return NVSR_ERR_SHAPE;
    if (arithmetic == NVSR_ARITH_F32) hipLaunchKernelGGL(limb_gemm_probe_kernel<0>, dim3(1), dim3(64), 0, (hipStream_t)stream, K, W, X, Y);
    else if (arithmetic == NVSR_ARITH_BF16X3) hipLaunchKernelGGL(limb_gemm_probe_kernel<3>, dim3(1), dim3(64), 0, (hipStream_t)stream, K, W, X, Y);
    else if (arithmetic == NVSR_ARITH_F16X2) hipLaunchKernelGGL(limb_gemm_probe_kernel<2>, dim3(1), dim3(64), 0, (hipStream_t)stream, K, W, X, Y);
    else return NVSR_ERR_SHAPE;
    return NVSR_CHECK_LAUNCH();
}

extern "C" int nvsr_pack_decoder_limbs_launch(const float* natural, float* packed, nvsr_stream_t stream) {
    unsigned* out = reinterpret_cast<unsigned*>(packed);
    const int n3 = KB_TOTAL * kb_words(3), n2 = KB_TOTAL * kb_words(2);
    hipLaunchKernelGGL(pack_decoder_limbs_kernel<3>, dim3((n3 + 255) / 256), dim3(256), 0, (hipStream_t)stream, natural, out + P_LIMB3);
    hipLaunchKernelGGL(pack_decoder_limbs_kernel<2>, dim3((n2 + 255) / 256), dim3(256), 0, (hipStream_t)stream, natural, out + P_LIMB2);
    return NVSR_CHECK_LAUNCH();
}

extern "C" int nvsr_render_pass3_launch(int limbs, const nvsr_scene* scene, const float* packed_decoder, int64_t N, int S, const float* rays,
                                        const float* z, const float* noise, int white_bkgd, float* rgb, float* disp, float* acc,
                                        float* weights, float* depth, float* raw_out, nvsr_stream_t stream) {
    const int64_t grid = (N + RAYS2 - 1) / RAYS2;
    if (grid > 0x7fffffff || (limbs != 2 && limbs != 3)) return NVSR_ERR_SHAPE;
#define NVSR_LAUNCH3(LIMBS_)                                                                                                               \
    if (weights)                                                                                                                           \
        hipLaunchKernelGGL(render_pass3_coarse_kernel<LIMBS_>, dim3((unsigned)grid), dim3(TPB2), 0, (hipStream_t)stream, to_dev(scene),    \
                           packed_decoder, (long)N, S, rays, z, noise, white_bkgd, rgb, disp, acc, weights, depth, raw_out, nvsr_get_range_flag());               \
    else                                                                                                                                   \
        hipLaunchKernelGGL(render_pass3_kernel<LIMBS_>, dim3((unsigned)grid), dim3(TPB2), 0, (hipStream_t)stream, to_dev(scene),           \
                           packed_decoder, (long)N, S, rays, z, noise, white_bkgd, rgb, disp, acc, depth, raw_out, nvsr_get_range_flag())
    if (limbs == 3) { NVSR_LAUNCH3(3); } else { NVSR_LAUNCH3(2); }
#undef NVSR_LAUNCH3
    return NVSR_CHECK_LAUNCH();
}

// the coarse pass with its depths computed in the kernel (z = coarse_depth(near, far, s, S, lindisp)); weights are always written
extern "C" int nvsr_render_pass3_coarse_z_launch(int limbs, const nvsr_scene* scene, const float* packed_decoder, int64_t N, int S, const float* rays,
                                                 int lindisp, const float* noise, int white_bkgd, float* rgb, float* disp, float* acc,
                                                 float* weights, float* depth, float* raw_out, nvsr_stream_t stream) {
    const int64_t grid = (N + RAYS2 - 1) / RAYS2;
    if (grid > 0x7fffffff || (limbs != 2 && limbs != 3) || !weights) return NVSR_ERR_SHAPE;
    if (limbs == 3)
        hipLaunchKernelGGL(render_pass3_coarse_z_kernel<3>, dim3((unsigned)grid), dim3(TPB2), 0, (hipStream_t)stream, to_dev(scene), packed_decoder,
                           (long)N, S, rays, lindisp, noise, white_bkgd, rgb, disp, acc, weights, depth, raw_out, nvsr_get_range_flag());
    else
        hipLaunchKernelGGL(render_pass3_coarse_z_kernel<2>, dim3((unsigned)grid), dim3(TPB2), 0, (hipStream_t)stream, to_dev(scene), packed_decoder,
                           (long)N, S, rays, lindisp, noise, white_bkgd, rgb, disp, acc, weights, depth, raw_out, nvsr_get_range_flag());
    return NVSR_CHECK_LAUNCH();
}
