// Training forward on the bf16 matrix pipe: decode_rays_kernel<MASKS, RECORD> (render.hip) with the decoder GEMMs in the exact 3-limb
// arithmetic of limb_core.h -- raw [N,S,4] for a batch too small for one workgroup per ray block (4 096 training rays), the ReLU gates
// the gate-driven backward consumes, and optionally the layer-input half of the weight-gradient record.
//
// The reference runs this as run_network -> TwoDimPlanesModel.forward under autograd (train_utils.py:15-64, models.py:381-421).
//
// Skeleton of the first-generation kernel, not of render3.hip: a wave owns ONE 32-point tile (32 consecutive samples of one ray), and
// the latency of its gathers, limb splits, bias + ReLU, gate words and record stores is covered by the wave of the OTHER workgroup on the
// same SIMD -- two independent 4-wave workgroups per CU (own barriers, start-up stagger), which is why the weight ring is cut into
// 36-KB slots (3 K-blocks of 16 input channels x 3 limbs x 4 output blocks): 2 x (2 x 36 KB + 6 KB) = 157 KB of LDS.  A plane's share
// of a feature layer is one chunk, a hidden layer three (3 + 3 + 2 K-blocks): 23 chunks per step, each staged by MUBUF LDS-DMA one
// chunk ahead.  Per step and tile 1 512 MFMAs (v_mfma_f32_32x32x16_bf16) instead of 2 016 v_mfma_f32_32x32x2_f32 of twice the length.
#include <type_traits>

#include "limb_core.h"

#ifndef L3_REC_LATE
#define L3_REC_LATE 0   // recording forward on f16 limbs, experiment (round 6): 1 = a layer's record rows are stored inside the NEXT layer's first block, two
                        // blocks of MFMAs ahead of the wait that needs them done, instead of where the layer ends (one block).  Same box, three
                        // alternations: 0.731 / 0.698 / 0.722 ms against 0.724 / 0.731 / 0.709 at S = 128, 0.43 against 0.42 at S = 64 -- the stores
                        // are bound by bytes and requests, not by the latency a later wait sees.  Off.
#endif
#ifndef L3_ABLATE
#define L3_ABLATE 0   // timing experiments only (wrong results): 1 no plane gathers, 2 no gate words, 4 no wait for the weight copies,
#endif                // 8 no bias + ReLU; 16 (correct results) the old vmcnt(0) behind a layer's record stores          (tools/ab_flags.sh)

namespace nvsr {

constexpr int L3_TPB = 256, L3_WAVES = L3_TPB / 64, L3_PTS = L3_WAVES * 32;
// LF = limbs of the forward: 3 bf16 limbs, or 2 f16 limbs (limb_core.h: round to nearest, static scales; round 3) when no weight-gradient
// record is wanted -- the gates a forward publishes are signs of pre-activations and feed the 3-limb backward whatever arithmetic found them
template <int LF>
struct L3 {
    static constexpr int SLOT = 3 * kb_words(LF);             // words: 36 KB (3 limbs) / 24 KB
    static constexpr int SMALL = 2 * SLOT;
    static constexpr int LDS = SMALL + SMALL_FLOATS;
};
static_assert(2 * L3<3>::LDS * 4 <= 160 * 1024, "two workgroups per CU");
// the recording forward on f16 limbs stages its record rows through LDS (record128_staged, decode_core.h): 4.5 KB per wave behind the ring; the
// 3-limb kernel has no room for it (2 x 80 KB) and stores its rows directly.  (Sample-major record rows -- an A/B switch -- are not consecutive.)
#ifdef NVSR_RECORD_SAMPLE_MAJOR
template <int LF> constexpr bool L3_STAGED = false;
#else
template <int LF> constexpr bool L3_STAGED = (LF == 2);
#endif
static_assert(2 * (L3<2>::LDS + L3_WAVES * RSTG_FLOATS) * 4 <= 160 * 1024, "two recording workgroups per CU");

struct RingL {
    __amdgpu_buffer_rsrc_t rsrc;   // 3-limb fragment region of the packed blob
    unsigned* lds;
    int slot;
    int wave, lane;
    unsigned voff;                 // wave * 1024 + lane * 16
};

// chunk = K-blocks kb0 .. kb0 + NKB - 1 -> the free slot, in 1-KiB pieces round-robin over the waves
template <int NKB, int LF = 3>
__device__ __forceinline__ const unsigned* ringl_issue(RingL& rs, int kb0) {
    unsigned* dst = rs.lds + rs.slot * L3<LF>::SLOT;
    constexpr int PIECES = NKB * 4 * LF;
    static_assert(PIECES % L3_WAVES == 0, "chunk must split evenly over the waves");
#pragma unroll
    for (int i = 0; i < PIECES / L3_WAVES; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs.rsrc, (__attribute__((address_space(3))) void*)(dst + (i * L3_WAVES + rs.wave) * 256), 16,
                                                 (int)rs.voff, kb0 * kb_words(LF) * 4 + i * (L3_WAVES * 1024), 0, 0);
    rs.slot ^= 1;
    return dst;
}
// YOUNGER: vector-memory operations this wave has issued AFTER the copy it waits for and that may stay in flight (vmcnt counts loads, stores and
// LDS-DMA together, in issue order: MI355X_MICROARCH.md) -- the record / gate stores of a finished layer, which come behind the next chunk's copy
template <int YOUNGER = 0>
__device__ __forceinline__ void ringl_sync() {
#if !(L3_ABLATE & 4)
    if constexpr (YOUNGER == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" : : "n"(YOUNGER) : "memory");
#endif
    __syncthreads();
}

// act = max(acc + bias, 0)   (bias packed in accumulator-register order: [group of 4 registers][lane half][4])
// LF = 2: act 2^SX = relu(acc 2^-SW + bias 2^SX) with the NaN-propagating ReLU of render3.hip (negated-source FMA + integer max)
template <int LF>
__device__ __forceinline__ void bias_relu(const f32x16 (&acc)[4], const float* bias, int h, f32x16 (&act)[4], float nsc) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bias + g * 8 + h * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (LF == 2) {
                float v;        // (one statement: hipcc pads an s_nop between an asm statement and a dependent instruction)
                asm("v_fma_f32 %0, -%1, %2, %3\n\tv_max_i32 %0, 0, %0" : "=v"(v) : "v"(acc[g >> 2][4 * (g & 3) + j]), "s"(nsc), "v"(b[j]));
                act[g >> 2][4 * (g & 3) + j] = v;
            } else act[g >> 2][4 * (g & 3) + j] = fmaxf(acc[g >> 2][4 * (g & 3) + j] + b[j], 0.0f);
        }
    }
}
// the layer's ReLU gate in relu_publish's format: bit gate_bit(ib, r) of word ib >> 1  <=>  pre-activation > 0
__device__ __forceinline__ void publish_gates(const f32x16 (&act)[4], unsigned* __restrict__ rec, int layer) {
    // act = max(., 0) is +0 or a positive number (never -0: v_max_f32 of (x, +0) returns +0 for x = -0): act > 0  <=>  its bits, read as
    // a signed integer, are >= 1.  clamp(bits, 0, 1) is one v_med3_i32 and the insertion one v_lshl_or_b32: 2 instructions per element, no
    // VCC.  (The compare + select form cost 3.5 issue slots per element -- v_cmp, 1-2 wait-state s_nops, v_cndmask from one of 32 constant
    // registers, half a v_or3 -- 1 800 of the 9 700 VALU issue slots of a tile, none of them in an MFMA gap.  Inline asm, four elements per
    // statement: written in C++ hipcc canonicalises clamp(bits, 0, 1) << k back into compare + select, and between two asm statements it
    // pads an s_nop.)
    unsigned m0 = 0u, m1 = 0u;
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int r = 0; r < 16; r += 4) {
            unsigned t0, t1, t2, t3;
            unsigned& m = ib < 2 ? m0 : m1;
            asm("v_med3_i32 %1, %5, 0, 1\n\tv_med3_i32 %2, %6, 0, 1\n\tv_med3_i32 %3, %7, 0, 1\n\tv_med3_i32 %4, %8, 0, 1\n\t"
                "v_lshl_or_b32 %0, %1, %9, %0\n\tv_lshl_or_b32 %0, %2, %10, %0\n\tv_lshl_or_b32 %0, %3, %11, %0\n\tv_lshl_or_b32 %0, %4, %12, %0"
                : "+v"(m), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                : "v"(act[ib][r]), "v"(act[ib][r + 1]), "v"(act[ib][r + 2]), "v"(act[ib][r + 3]), "n"(gate_bit(ib, r)), "n"(gate_bit(ib, r + 1)),
                  "n"(gate_bit(ib, r + 2)), "n"(gate_bit(ib, r + 3)));
        }
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    *reinterpret_cast<u32x2*>(rec + 2 * layer) = u32x2{m0, m1};
}

template <bool MASKS, bool RECORD, int LF>
__device__ __forceinline__ void decode_step_limb(const SceneDev& sc, RingL& rs, const float* small, float px, float py, float pz, const Taps& vt_,
                                                 float (&raw)[4], unsigned* __restrict__ gates, const DecRecord& rec, long q, bool rec_ok, float nsc,
                                                 float* stage = nullptr, long q0 = 0, int nvalid = 0) {
    auto scaled = [](Taps t) {          // f16 limbs: features carry 2^F16_SX, put on the four blend weights (exact)
        if constexpr (LF == 2) { t.nw *= F16_X_SCALE; t.ne *= F16_X_SCALE; t.sw *= F16_X_SCALE; t.se *= F16_X_SCALE; }
        return t;
    };
    const Taps vt = scaled(vt_);
    asm volatile("" : "+v"(rs.voff), "+v"(rs.lane));       // (see decode_step: keeps hipcc from hoisting per-lane addresses out of the tile loop)
    const int lane = rs.lane, h = lane >> 5;
    const int& lane_h = h;
    // (f16 limbs: features and activations carry 2^F16_SX in registers; the weight-gradient record holds UNSCALED f32 layer inputs: every
    //  recorded value is multiplied by 2^-F16_SX on its way out -- exact)
    constexpr float REC_UNSCALE = LF == 2 ? 1.0f / F16_X_SCALE : 1.0f;
    auto rec24 = [&](float* row, const float (&f)[HALF_C]) {
        if (L3_ABLATE & 32) return;                         // (timing experiment: no feature rows in the record)
        if constexpr (LF == 2) record24_scaled(row, lane_h, f, REC_UNSCALE); else record24(row, lane_h, f);
    };
    const float n0 = norm_coord(px, sc.lo[0], sc.range[0]);
    const float n1 = norm_coord(py, sc.lo[1], sc.range[1]);
    const float n2 = norm_coord(pz, sc.lo[2], sc.range[2]);
    f32x16 acc[4], act[4];
    float D[HALF_C], F[HALF_C];
    Limbs<LF> cur, fa;
    auto feat = [](const float (&f)[HALF_C]) { return [&f](int kb, int i) { return f[8 * kb + i]; }; };
    auto hid = [](const f32x16 (&a)[4], int kb0) { return [&a, kb0](int kb, int i) { const int k = kb0 + kb; return a[k >> 1][8 * (k & 1) + i]; }; };
    auto none = [](int) {};
    SplitPend tp;
    auto tail_of = [&tp](const f32x16 (&a)[4], int kb) {
        return [&a, kb, &tp](int slice, Limbs<LF>& nxt) { split_slice<LF>(slice, [&a, kb](int i) { return a[kb >> 1][8 * (kb & 1) + i]; }, nxt, tp); };
    };
    auto pos_taps = [&](int d) {
        const float* M = sc.proj + 6 * d;
        return scaled(make_taps(sc, d, n0 * M[0] + n1 * M[2] + n2 * M[4], n0 * M[1] + n1 * M[3] + n2 * M[5]));
    };
    // Nothing moves across the end of a block: left alone hipcc hoists the next plane's 24 gather loads (96 registers) above the block's
    // MFMAs -- good for latency, but with the accumulators, D, F and the limbs live it spills 120 registers; the other workgroup's wave on
    // the SIMD covers the gather instead.
    // (the asm takes the accumulators as operands: a bare memory clobber orders the loads but lets the MFMAs sink below them)
#define L3_FENCE                                                                                    \
    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : : "memory");         \
    __builtin_amdgcn_sched_barrier(0);
#if L3_ABLATE & 1
#define L3_GATHER(PLANE, TAPS, H_, F_) { const Taps t_ = TAPS; for (int c_ = 0; c_ < HALF_C; ++c_) F_[c_] = t_.nw * (float)(c_ + H_); }
#else
#define L3_GATHER(PLANE, TAPS, H_, F_) gather24(PLANE, TAPS, H_, F_);
#endif
    // one chunk: wait for it, start the copy of the next one, split the block's first K-block, multiply
    // (FIRST: the block's first K-block is split here, exposed; otherwise the previous block's tail produced it in its MFMA gaps)
#define L3_BLOCK_(NKB, ZERO, FIRST, SRC, NEXT, TAIL) L3_BLOCK_Y(0, NKB, ZERO, FIRST, SRC, NEXT, TAIL)
#define L3_BLOCK_Y(YOUNGER, NKB, ZERO, FIRST, SRC, NEXT, TAIL) L3_BLOCK_YS(YOUNGER, (void)0, NKB, ZERO, FIRST, SRC, NEXT, TAIL)
    // (STMT: issued behind the next chunk's copy -- the pending record rows of the layer below, L3_REC_LATE)
#define L3_BLOCK_YS(YOUNGER, STMT, NKB, ZERO, FIRST, SRC, NEXT, TAIL)                          \
    {                                                                                          \
        ringl_sync<YOUNGER>();                                                                 \
        const unsigned* nw = NEXT;                                                             \
        STMT;                                                                                  \
        if (FIRST) { auto s_ = SRC; split_all<LF>([&](int i) { return s_(0, i); }, cur); }      \
        limb_block<LF, NKB, ZERO, true>(cw, lane, acc, cur, fa, SRC, none, TAIL);               \
        cw = nw;                                                                               \
        L3_FENCE                                                                               \
    }
#define L3_ISSUE(NKB_, KB_) (ringl_issue<NKB_, LF>(rs, KB_))
#define L3_BLOCK(NKB, ZERO, SRC, NEXT) L3_BLOCK_(NKB, ZERO, true, SRC, NEXT, NoTail{})
    // a hidden layer = 3 + 3 + 2 K-blocks of the previous activation; NEXT = the chunk that follows the layer
#ifdef NVSR_NO_TAILS      // A/B switch (tools/): every block splits its first K-block itself
#define L3_HIDDEN(KB0, NEXT) L3_HIDDEN_Y(YA_LAYER, KB0, NEXT)
#define L3_HIDDEN_Y(YA, KB0, NEXT)                                                             \
    L3_BLOCK_YS(YA, flush_pending(), 3, true, true, hid(act, 0), L3_ISSUE(3, (KB0) + 3), NoTail{})  \
    L3_BLOCK_YS(YB_LAYER, (void)0, 3, false, true, hid(act, 3), L3_ISSUE(2, (KB0) + 6), NoTail{})   \
    L3_BLOCK(2, false, hid(act, 6), NEXT)
#else
#define L3_HIDDEN(KB0, NEXT) L3_HIDDEN_Y(YA_LAYER, KB0, NEXT)
#define L3_HIDDEN_Y(YA, KB0, NEXT)                                                             \
    L3_BLOCK_YS(YA, flush_pending(), 3, true, true, hid(act, 0), L3_ISSUE(3, (KB0) + 3), tail_of(act, 3))      \
    L3_BLOCK_YS(YB_LAYER, (void)0, 3, false, false, hid(act, 3), L3_ISSUE(2, (KB0) + 6), tail_of(act, 6))    \
    L3_BLOCK_(2, false, false, hid(act, 6), NEXT, NoTail{})
#endif
    // Round 6: the 16 record stores of a finished layer (512 B per point) are the YOUNGEST vector-memory operations when the next layer's first
    // block waits for its weight chunk -- the chunk's copy was issued a block earlier.  That wait used to be vmcnt(0): every layer's record went
    // out to HBM with the wave standing still behind it (the recording forward took compute + stores, 0.41 + 0.57 ms at S = 128).  It now leaves
    // those 16 in flight (ringl_sync<16>); they have the whole first block to land before the second block's vmcnt(0).  For the count to hold on
    // every wave the record stores are issued unconditionally: padding lanes rewrite the record row of the valid point they mirror with the same
    // values (rec_ok is true for every lane of a recording launch, see the kernel).
    constexpr int FIN_YOUNG = (RECORD && !(L3_ABLATE & 16)) ? 16 : 0;
    // L3_REC_LATE (staged rows only): a finished layer's rows are not stored where the layer ends but INSIDE the next layer's first block, behind the
    // copy of that layer's second chunk (`act` lives on as the next layer's operand): the second block's wait then leaves them in flight too
    // (vmcnt(16)) and only the third block's wait needs them done -- two blocks of MFMAs to land in instead of one.
    constexpr bool LATE = RECORD && L3_STAGED<LF> && (L3_REC_LATE != 0) && !(L3_ABLATE & 16);
    constexpr int YA_LAYER = LATE ? 0 : FIN_YOUNG, YB_LAYER = LATE ? 16 : 0;
    float* pend_row = nullptr;
    auto flush_pending = [&]() {
        if constexpr (LATE) record128_staged<true>(stage, pend_row, q0, nvalid, rec.dump, lane, act, REC_UNSCALE);
    };
    auto finish = [&](int vec, float* hrow) {             // bias + ReLU of the finished layer, its gate words, its record row
        if (L3_ABLATE & 8) {
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) act[ib] = acc[ib];
        } else {
            bias_relu<LF>(acc, small + S_BIAS + vec * HID, h, act, nsc);
        }
        if (MASKS && !(L3_ABLATE & 2)) publish_gates(act, gates, vec);
        if constexpr (LATE) pend_row = hrow;                                                                                             // (stored by the next block)
        else if constexpr (RECORD && L3_STAGED<LF>) record128_staged<true>(stage, hrow, q0, nvalid, rec.dump, lane, act, REC_UNSCALE);     // (whole cache lines per store)
        else if (RECORD && rec_ok) { if constexpr (LF == 2) record128_scaled(hrow, q, h, act, REC_UNSCALE); else record128(hrow, q, h, act); }
    };
    const long LP = (long)HID * rec.Pp;

    // ---- rgb layer 0: K = 192 in the limb blob's order [f_view | f0 | f1 | f2], one plane = one chunk ---------------------------------
    // Round 6: a plane's feature rows of the record go out INSIDE the block that multiplies them, behind the next chunk's copy -- F is intact
    // until the next gather, and the stores then have the block's 72 MFMAs to land (they used to sit between the gather and the block's
    // vmcnt(0): a timing build without them ran the recording forward 0.74 -> 0.65 ms) -- and, on f16 limbs, as whole segments through the
    // LDS stage (record24_staged).
    auto rec24x = [&](float* colbase, int rowstride, const float (&f)[HALF_C]) {
        if constexpr (!RECORD) return;
        if (L3_ABLATE & 32) return;                         // (timing experiment: no feature rows in the record)
        if constexpr (L3_STAGED<LF>) record24_staged<true>(stage, colbase, rowstride, q0, nvalid, rec.dump, lane, f, REC_UNSCALE);
        else if (rec_ok) rec24(colbase + q * rowstride, f);
    };
#define L3_BLOCK_R(STMT, NKB, ZERO, SRC, NEXT)                                                 \
    {                                                                                          \
        ringl_sync<0>();                                                                       \
        const unsigned* nw = NEXT;                                                             \
        STMT;                                                                                  \
        { auto s_ = SRC; split_all<LF>([&](int i) { return s_(0, i); }, cur); }                 \
        limb_block<LF, NKB, ZERO, true>(cw, lane, acc, cur, fa, SRC, none, NoTail{});           \
        cw = nw;                                                                               \
        L3_FENCE                                                                               \
    }
    const unsigned* cw = L3_ISSUE(3, KB_RGB0);
    L3_GATHER(sc.plane[3], vt, h, F);
    L3_BLOCK_R(rec24x(rec.Xr + 3 * C, 4 * C, F), 3, true, feat(F), L3_ISSUE(3, KB_RGB0 + 3))
    L3_GATHER(sc.plane[0], pos_taps(0), h, F);
#pragma unroll
    for (int c = 0; c < HALF_C; ++c) D[c] = F[c];
    L3_BLOCK_R(rec24x(rec.Xr, 4 * C, F), 3, false, feat(F), L3_ISSUE(3, KB_RGB0 + 6))
    L3_GATHER(sc.plane[1], pos_taps(1), h, F);
#pragma unroll
    for (int c = 0; c < HALF_C; ++c) D[c] = __fadd_rn(D[c], F[c]);
    L3_BLOCK_R(rec24x(rec.Xr + C, 4 * C, F), 3, false, feat(F), L3_ISSUE(3, KB_RGB0 + 9))
    L3_GATHER(sc.plane[2], pos_taps(2), h, F);
    // combine_pos_planes 'avg' = stack(...).mean(0)  (models.py:358-359)
#pragma unroll
    for (int c = 0; c < HALF_C; ++c) D[c] = div3(__fadd_rn(D[c], F[c]));
    auto rec_last = [&]() {
        rec24x(rec.Xr + 2 * C, 4 * C, F);
        rec24x(rec.Xd, 64, D);
        if (RECORD && rec_ok && !(L3_ABLATE & 32)) {          // (columns 48..63 of the density input: the contraction's padded block reads them)
            *reinterpret_cast<f32x4*>(rec.Xd + q * 64 + C + 8 * h) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            *reinterpret_cast<f32x4*>(rec.Xd + q * 64 + C + 8 * h + 4) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
    };
    L3_BLOCK_R(rec_last(), 3, false, feat(F), L3_ISSUE(3, KB_RGB1))
#undef L3_BLOCK_R
    finish(4, rec.Hr);
    // ---- rgb layers 1..3, rgb head ---------------------------------------------------------------------------------------------------
    L3_HIDDEN(KB_RGB1, L3_ISSUE(3, KB_RGB1 + 8))
    finish(5, rec.Hr + LP);
    L3_HIDDEN(KB_RGB1 + 8, L3_ISSUE(3, KB_RGB1 + 16))
    finish(6, rec.Hr + 2 * LP);
    L3_HIDDEN(KB_RGB1 + 16, L3_ISSUE(3, KB_DEN0))
    finish(7, rec.Hr + 3 * LP);
    {
        float hd[3];
        head_dots<3>(small + S_RGB_W, h, act, hd);
#pragma unroll
        for (int c = 0; c < 3; ++c) raw[c] = hd[c] + small[S_HEAD_B + 1 + c];
    }
    // ---- density decoder: 48 -> 128 x 4 -> 1 -----------------------------------------------------------------------------------------
    L3_BLOCK_YS(YA_LAYER, flush_pending(), 3, true, true, feat(D), L3_ISSUE(3, KB_DEN1), NoTail{})        // (behind / carrying finish(7)'s record stores)
    finish(0, rec.Hd);
    // (LATE: rgb layer 3's rows went out inside the density layer-0 block, behind the copy this layer's first block waits for: 16 young stores)
    L3_HIDDEN_Y(FIN_YOUNG, KB_DEN1, L3_ISSUE(3, KB_DEN1 + 8))
    finish(1, rec.Hd + LP);
    L3_HIDDEN(KB_DEN1 + 8, L3_ISSUE(3, KB_DEN1 + 16))
    finish(2, rec.Hd + 2 * LP);
    L3_HIDDEN(KB_DEN1 + 16, (const unsigned*)nullptr)
    finish(3, rec.Hd + 3 * LP);
    flush_pending();                                       // (the tile's last layer: nothing follows to carry its rows)
    {
        float hd[1];
        head_dots<1>(small + S_ALPHA_W, h, act, hd);
        raw[3] = hd[0] + small[S_HEAD_B];
    }
#undef L3_GATHER
#undef L3_ISSUE
#undef L3_HIDDEN
#undef L3_BLOCK
#undef L3_BLOCK_
#undef L3_FENCE
}

template <bool MASKS, bool RECORD, int LF = 3>
__global__ __launch_bounds__(L3_TPB, 2) void decode_rays_limb_kernel(SceneDev sc, const float* __restrict__ packed, long N, int S,
                                                                    const float* __restrict__ rays, const float* __restrict__ z,
                                                                    float* __restrict__ raw_out, unsigned* __restrict__ gates, DecRecord rec,
                                                                    unsigned* __restrict__ flag) {
    constexpr int L3_SMALL = L3<LF>::SMALL;
    __shared__ __attribute__((aligned(16))) unsigned lds[L3<LF>::LDS + ((RECORD && L3_STAGED<LF>) ? L3_WAVES * RSTG_FLOATS : 0)];
    NVSR_RACE_PROBE_DELAY(lds);      // (probe builds only, nvsr_common.h)
    // (wave index as a SCALAR: the LDS destination of every weight-copy piece then is scalar arithmetic into M0 instead of a vector add + v_readfirstlane per piece)
    RingL rs{__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(packed + limb_region(LF)), 0, KB_TOTAL * kb_words(LF) * 4, 0x00020000), lds, 0,
             __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), (int)(threadIdx.x & 63), (threadIdx.x >> 6) * 1024u + (threadIdx.x & 63) * 16u};
    float* ldsf = reinterpret_cast<float*>(lds);
    for (int i = threadIdx.x; i < SMALL_FLOATS; i += L3_TPB) {   // published by the first ring barrier.  f16 limbs: biases x 2^SX, head weights x 2^-SX, the packer's poison (render3.hip)
        float v = packed[P_SMALL + i] * (LF != 2 ? 1.0f : i < S_ALPHA_W ? F16_X_SCALE : i < S_HEAD_B ? F16_HEAD_SCALE : 1.0f);
        if (LF == 2 && i >= S_HEAD_B && i < S_HEAD_B + 4) v += packed[P_SMALL + S_F16_POISON];
        ldsf[L3_SMALL + i] = v;
    }
    float nscale = -F16_ACC_UNSCALE;                 // bias_relu<2>: opaque, or the compiler folds the two negations away
    asm volatile("" : "+s"(nscale));
    const float* small = ldsf + L3_SMALL;
    // start-up stagger of the two workgroups of a CU (decode_prologue): the one in an odd wave slot starts half a phase late
    unsigned hw_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    if (__builtin_amdgcn_readfirstlane(hw_id) & 1u) __builtin_amdgcn_s_sleep(64);
    // A wave's tile = 32 CONSECUTIVE SAMPLES OF ONE RAY (like render_bwd_limb.hip): their bilinear taps fall into the same or neighbouring
    // texel cells, so the 64 lanes of a gather load share cache lines (tools/gather_ubench.hip: 90 vs 30 GB/s per CU against 32 unrelated
    // rays); the ray, its gates and raw rows are contiguous per tile.
    // S = 1 -- a list of POINTS handed over as one-sample rays (TwoDimPlanesModel.forward stand-alone, models.py:381-421): a tile is 32
    // consecutive points, not the one sample of one ray
    const int nsc = (S + 31) / 32;                                     // sample chunks per ray
    const long nwt = S == 1 ? (N + 31) / 32 : N * nsc;                 // wave tiles
    const long ntiles = (nwt + L3_WAVES - 1) / L3_WAVES;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {   // uniform trip count per workgroup
        const long wt = tile * L3_WAVES + rs.wave;
        const long ray0 = S == 1 ? wt * 32 + (rs.lane & 31) : wt / nsc;
        const int s0 = S == 1 ? 0 : (int)(wt - ray0 * nsc) * 32 + (rs.lane & 31);
        const bool valid = ray0 < N && s0 < S;
        const long ray = ray0 < N ? ray0 : N - 1;
        const int s = s0 < S ? s0 : S - 1;
        // z = NULL: `rays` is a list of points [N,6] = [xyz, viewdir] (S = 1; nvsr_triplane_decode)
        const float* r = rays + ray * (z ? 11 : 6);
        const float zc = z ? z[ray * S + s] : 0.0f;
        const Taps vt = z ? view_taps(sc, r[8], r[9], r[10]) : view_taps(sc, r[3], r[4], r[5]);
        const float ppx = z ? __fadd_rn(r[0], __fmul_rn(r[3], zc)) : r[0];
        const float ppy = z ? __fadd_rn(r[1], __fmul_rn(r[4], zc)) : r[1];
        const float ppz = z ? __fadd_rn(r[2], __fmul_rn(r[5], zc)) : r[2];
        float raw[4];
        // gate record of this lane: [point ray*S+s][lane half][16 words]; padding lanes rewrite a valid point's record with the same values
        unsigned* gl = MASKS ? gates + ((ray * S + s) * 2 + (rs.lane >> 5)) * 16 : nullptr;
        // (rec_ok = true on every lane of a recording launch: padding lanes hold the clamped point's values and rewrite its row -- FIN_YOUNG)
        // staged record rows: the tile's 32 points are the consecutive rows q0 .. q0 + 31; points behind the ray's last sample / the last ray -> dump rows
        // (S = 1: a tile is 32 consecutive points = rows 32 wt ..)
        const int chunk0 = S == 1 ? 0 : (int)(wt - (wt / nsc) * nsc) * 32;
        const long tq0 = S == 1 ? wt * 32 : (wt / nsc) * S + chunk0;
        const long tleft = S == 1 ? N - wt * 32 : ((wt / nsc) < N ? (long)(S - chunk0) : 0L);
        decode_step_limb<MASKS, RECORD, LF>(sc, rs, small, ppx, ppy, ppz, vt, raw, gl, rec, record_row(ray, s, N, S), RECORD ? true : valid, nscale,
                                            ldsf + L3<LF>::LDS + rs.wave * RSTG_FLOATS, tleft > 0 ? tq0 : 0, (int)(tleft < 0 ? 0 : tleft > 32 ? 32 : tleft));
        if (valid && rs.lane < 32) {
            *reinterpret_cast<f32x4*>(raw_out + (ray * S + s) * 4) = f32x4{raw[0], raw[1], raw[2], raw[3]};
            // range flag of the f16 limbs (nvsr.h: nvsr_set_range_flag)
            if (LF == 2 && flag && !(fabsf(raw[0] + raw[1] + raw[2] + raw[3]) <= 3.0e38f)) __hip_atomic_fetch_or(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace nvsr

using namespace nvsr;

// nvsr_decode_rays_ex (render.hip) with the decoder arithmetic set to bf16 limbs; arguments already validated there
extern "C" int nvsr_decode_rays_limb_launch(int limbs, const nvsr_scene* scene, const float* packed_decoder, int64_t N, int S, const float* rays,
                                            const float* z, float* raw, uint32_t* gates, float* record, nvsr_stream_t stream) {
    const int64_t nwt = S == 1 ? (N + 31) / 32 : N * (int64_t)((S + 31) / 32);            // wave tiles: (ray, 32 samples), or 32 points when S = 1
    const int64_t ntiles = (nwt + L3_WAVES - 1) / L3_WAVES;                                // 4 wave tiles per workgroup step
    const int grid = (int)(ntiles < 2048 ? ntiles : 2048);
    if (limbs == 2 && record) {         // f16 limbs with the layer-input half of the weight-gradient record (written unscaled)
        hipLaunchKernelGGL((decode_rays_limb_kernel<true, true, 2>), dim3(grid), dim3(L3_TPB), 0, (hipStream_t)stream, to_dev(scene), packed_decoder,
                           (long)N, S, rays, z, raw, gates, make_record(record, (long)N, S), nvsr_get_range_flag());
        return NVSR_CHECK_LAUNCH();
    }
    if (limbs == 2 && !record) {        // f16 limbs
        if (gates)
            hipLaunchKernelGGL((decode_rays_limb_kernel<true, false, 2>), dim3(grid), dim3(L3_TPB), 0, (hipStream_t)stream, to_dev(scene), packed_decoder,
                               (long)N, S, rays, z, raw, gates, DecRecord{}, nvsr_get_range_flag());
        else
            hipLaunchKernelGGL((decode_rays_limb_kernel<false, false, 2>), dim3(grid), dim3(L3_TPB), 0, (hipStream_t)stream, to_dev(scene),
                               packed_decoder, (long)N, S, rays, z, raw, (unsigned*)nullptr, DecRecord{}, nvsr_get_range_flag());
        return NVSR_CHECK_LAUNCH();
    }
    if (record)
        hipLaunchKernelGGL((decode_rays_limb_kernel<true, true>), dim3(grid), dim3(L3_TPB), 0, (hipStream_t)stream, to_dev(scene), packed_decoder,
                           (long)N, S, rays, z, raw, gates, make_record(record, (long)N, S), (unsigned*)nullptr);
    else if (gates)
        hipLaunchKernelGGL((decode_rays_limb_kernel<true, false>), dim3(grid), dim3(L3_TPB), 0, (hipStream_t)stream, to_dev(scene), packed_decoder,
                           (long)N, S, rays, z, raw, gates, DecRecord{}, (unsigned*)nullptr);
    else
        hipLaunchKernelGGL((decode_rays_limb_kernel<false, false>), dim3(grid), dim3(L3_TPB), 0, (hipStream_t)stream, to_dev(scene),
                           packed_decoder, (long)N, S, rays, z, raw, (unsigned*)nullptr, DecRecord{}, (unsigned*)nullptr);
    return NVSR_CHECK_LAUNCH();
}
