// Gate-driven backward of the render pass on the bf16 matrix pipe: render_pass_backward_gates_kernel (render_bwd.hip) with the transposed
// layers in the exact 3-limb arithmetic of limb_core.h.
//
// The reference obtains these gradients from torch.autograd through run_network / TwoDimPlanesModel.forward / grid_sample
// (train_utils.py:185-282, models.py:381-421).
//
// D[in-feature 32-block][point] += W^T[in][out] * G[out][point]: A = limb fragments of W^T (the region behind the f32 blob of
// nvsr_pack_decoder_bwd, bwd_core.h), B = the gradient of the layer above, split into limbs on the fly.  The C/D layout of one layer is the
// B layout of the next one below, exactly as in the forward, so the chain runs through registers; the ReLU gate of the layer below is
// applied to the finished accumulators.  Per tile 6 x 192 + 5 x 96 = 1 632 v_mfma_f32_32x32x16_bf16 instead of 2 176 f32 MFMAs of twice
// the length.  Skeleton as in decode_limb.hip, except that a wave's 32-point tile is 32 CONSECUTIVE SAMPLES OF ONE RAY (their texel cells
// repeat, so the scatter merges runs of samples into one set of atomics); two independent 4-wave workgroups per CU cover each other's
// splits, masks, transposes and atomics; the weights stream through a ring of two 24-KB slots
// (2 K-blocks of a hidden layer / 4 of a layer-0 block pair), 32 chunks per step (round 5: the rgb layer-0^T as three 64-row pairs, bwd_core.h).  Scatter into the planes, view-plane rows and the record of
// the pre-activation gradients are those of the f32 kernel (bwd_core.h).
#include <type_traits>

#ifndef BL_ABLATE
#define BL_ABLATE 0   // timing experiments only (wrong results): 1 no wait for the weight copies, 2 no gate masks, 4 no plane scatter / view rows,
#endif                // 8 no splits outside the MFMA gaps (first K-block of a chunk), 16 scatter loop without its atomics, 64 no workgroup barriers,
                      // 128 every tile of a workgroup loads the inputs (depth, dL/draw, gates) of the workgroup's FIRST tile (cache hits: the exposed load latency at a tile's top)
                      // (tools/bwd_limb_ablate.sh, tools/bwd_limb_kernel_ablate.sh)

#ifndef BL_SCATTER
#define BL_SCATTER 3  // plane scatter of a tile: 0 one set of 4 atomics per point, 1 per run of points in one cell, 2 every texel of the tile once (bookkeeping per point on the scalar unit), 3 the same with the bookkeeping per tile on the vector unit
#endif                // (0 / 1: A/B builds for the WRITE_SIZE comparison, tools/bwd_scatter_pmc.sh)

#ifndef BL_PROLOGUE_BARRIER
#define BL_PROLOGUE_BARRIER 1   // workgroup barrier behind the head weights' copy into LDS (0 with -DNVSR_RACE_PROBE=1: a build that demonstrates the race it closes)
#endif
#ifndef BL_GATE_PREFETCH
#define BL_GATE_PREFETCH 1   // the tile's gate words loaded once at its top (A/B: 0 = re-read per layer)
#endif
#ifndef BL_F16_UP
#define BL_F16_UP 3   // f16-limb backward: power of two put on top of a point's normalised gradient (see pow2_scales)
#endif

#include "limb_core.h"
#include "bwd_core.h"

namespace nvsr {

#ifndef BL_WAVES_N
#define BL_WAVES_N 4    // waves per workgroup: 4 = two independent workgroups per CU; 8 (experiment) = one workgroup, twice the points per weight chunk
#endif
constexpr int BL_TPB = 64 * BL_WAVES_N, BL_WAVES = BL_TPB / 64, BL_PTS = BL_WAVES * 32;
constexpr int BL_WG_PER_CU = BL_WAVES_N == 4 ? 2 : 1;
// LF = limbs of the transposed-layer products: 3 bf16 limbs, or (round 3) 2 f16 limbs with the gradient of every POINT
// scaled by a power of two so that its largest magnitude is in [1, 2) -- gradients span many decades, a point's 128 features do not
template <int LF>
struct BLds {
    static constexpr int SMALL = 2 * BLimb<LF>::CHUNK_WORDS;
    static constexpr int TILES = SMALL + SMALL_FLOATS;
    static constexpr int LDS = TILES + BL_WAVES * TILE_FLOATS;
};
static_assert(BL_WG_PER_CU * BLds<3>::LDS * 4 <= 160 * 1024, "workgroups per CU");

// natural blob -> limb fragments of the transposed layers (LF = 3: bf16 limbs by truncation; LF = 2: f16 limbs, round to nearest, unscaled)
template <int LF>
__global__ void pack_decoder_bwd_limbs_kernel(const float* __restrict__ nat, unsigned* __restrict__ out) {
    constexpr int FH = BLimb<LF>::HID_FRAGS, F0 = BLimb<LF>::L0_FRAGS;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= BLimb<LF>::WORDS) return;
    const int w = idx & 3, lane = (idx >> 2) & 63, h = lane >> 5;
    int f = idx >> 8;                                   // fragment
    // region: 0 density hidden (3 x FH), 1 density layer 0 (F0), 2 rgb hidden (3 x FH), 3 rgb layer 0 (4 x F0)
    const bool rgb = f >= 3 * FH + F0;
    if (rgb) f -= 3 * FH + F0;
    const bool hidden = f < 3 * FH;
    unsigned word = 0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int e = 2 * w + half;
        float v = 0.0f;
        if (hidden) {
            const int li = f / FH, r = f % FH;          // li 0 -> layer 3, 1 -> layer 2, 2 -> layer 1
            const int t_ob = r % (4 * LF), kb = r / (4 * LF), ob = t_ob / LF;
            const int k = 32 * (kb >> 1) + 16 * (kb & 1) + 8 * (e >> 2) + 4 * h + (e & 3);       // output feature = the MFMA's K
            const int m = 32 * ob + (lane & 31);                                                   // input feature = the row of W^T
            v = nat[(rgb ? N_RGB_W1 : N_DEN_W1) + (2 - li) * N_HID_STRIDE + k * HID + m];
        } else {
            const int g = f - 3 * FH, p = g / F0, r = g % F0;
            const int kb = r / (2 * LF), ob = (r % (2 * LF)) / LF;
            const int k = 32 * (kb >> 1) + 16 * (kb & 1) + 8 * (e >> 2) + 4 * h + (e & 3);
            const int c = 32 * ob + (lane & 31);
#if BL_RGB0_PAIRS == 3
            // rgb: pair p holds rows 64 p + c of the [192 x 128] layer-0^T (the four planes' 48 rows each, contiguous in W0's input index)
            if (rgb) v = nat[N_RGB_W0 + k * (4 * C) + 64 * p + c];
            else if (c < C) v = nat[N_DEN_W0 + k * C + c];
#else
            if (c < C) v = rgb ? nat[N_RGB_W0 + k * (4 * C) + C * p + c] : nat[N_DEN_W0 + k * C + c];
#endif
        }
        const int t = hidden ? (f % FH) % LF : ((f - 3 * FH) % F0) % LF;
        unsigned bits = 0;
        if constexpr (LF == 2) {
            const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
            bits = __builtin_bit_cast(unsigned short, t == 0 ? hi : lo);
        } else {
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                if (l == t) bits = __float_as_uint(v) >> 16;
                v = limb_rest(v);
            }
        }
        word |= bits << (16 * half);
    }
    out[idx] = word;
}

struct RingB {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned* lds;
    int slot;
    int wave, lane;
    unsigned voff;
};
template <int LF = 3>
__device__ __forceinline__ const unsigned* ringb_issue(RingB& rs, int chunk) {
    constexpr int BL_CHUNK_WORDS = BLimb<LF>::CHUNK_WORDS;
    // The chunk's byte offset and the slot go through an opaque asm: as compile-time constants of a call site they are loop-invariant,
    // hipcc hoists the 204 scalar offsets / LDS addresses of a step out of the tile loop and spills them (one scratch reload per DMA).
    int base = chunk * (BL_CHUNK_WORDS * 4), slot = rs.slot;
    asm volatile("" : "+s"(base), "+s"(slot));
    unsigned* dst = rs.lds + slot * BL_CHUNK_WORDS;
#if !(BL_ABLATE & 32)         // (32: no weight copies at all -- the matrix work runs on whatever the slots hold)
#pragma unroll
    for (int i = 0; i < BLimb<LF>::CHUNK_FRAGS / BL_WAVES; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs.rsrc, (__attribute__((address_space(3))) void*)(dst + (i * BL_WAVES + rs.wave) * 256), 16,
                                                 (int)rs.voff, base + i * (BL_WAVES * 1024), 0, 0);
#endif
    rs.slot = slot ^ 1;
    return dst;
}
// YOUNGER: vector-memory operations issued AFTER the copy being waited for that may stay in flight (the 16 record stores of the layer just finished:
// vmcnt counts loads, stores and LDS-DMA together, in issue order); `counted` (wave-uniform): the wave really issued them
template <int YOUNGER = 0>
__device__ __forceinline__ void ringb_sync(bool counted = false) {
#if !(BL_ABLATE & 1)
    if (YOUNGER > 0 && counted) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(YOUNGER) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#if !(BL_ABLATE & 64)
    __syncthreads();
#endif
}

// acc2[ob] += W0^T fragments [K-block 4][channel block 2][limb 3] x limbs of src(kb, 0..7): 4 x 2 x 6 MFMAs
template <int LF, class Src>
__device__ __forceinline__ void limb_mm2(const unsigned* wl, int lane, f32x16 (&acc2)[2], Src src) {
    constexpr int NP = limb_products(LF);
    const u32x4* wv = reinterpret_cast<const u32x4*>(wl) + lane;
    Limbs<LF> cur, fa;
    split_all<LF>([&](int i) { return src(0, i); }, cur);
#pragma unroll
    for (int t = 0; t < LF; ++t) fa.v[t] = wv[t * 64];
    SplitPend sp;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        Limbs<LF> nxt;
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
            Limbs<LF> fn;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int q = kb * 2 + ob;
                acc2[ob] = mfma_limb<LF>(fa.v[limb_w(LF, p)], cur.v[limb_x(LF, p)], acc2[ob]);
                __builtin_amdgcn_sched_barrier(0);
                if (p < LF) fn.v[p] = wv[(((q + 1) % 8) * LF + p) * 64];
                if (kb + 1 < 4) {      // the 4 NP slices of the next K-block's split over the 2 NP slots of this K-block
                    split_slice<LF>(2 * (ob * NP + p), [&](int i) { return src(kb + 1, i); }, nxt, sp);
                    split_slice<LF>(2 * (ob * NP + p) + 1, [&](int i) { return src(kb + 1, i); }, nxt, sp);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int t = 0; t < LF; ++t) fa.v[t] = fn.v[t];
        }
        if (kb + 1 < 4) {
#pragma unroll
            for (int t = 0; t < LF; ++t) cur.v[t] = nxt.v[t];
        }
    }
}

template <bool RECORD, int LF = 3>
__global__ __launch_bounds__(BL_TPB, BL_WG_PER_CU) void render_pass_backward_gates_limb_kernel(SceneDev sc, const float* __restrict__ packed,
                                                                                   const float* __restrict__ packed_bwd, long N, int S,
                                                                                   const float* __restrict__ rays, const float* __restrict__ z,
                                                                                   const float* __restrict__ g_raw,
                                                                                   const unsigned* __restrict__ gates, GradPlanes gp,
                                                                                   float* __restrict__ gview, DecRecord rec) {
    constexpr int BL_SMALL = BLds<LF>::SMALL, BL_TILES = BLds<LF>::TILES;
    __shared__ __attribute__((aligned(16))) unsigned lds[BLds<LF>::LDS];
    NVSR_RACE_PROBE_DELAY(lds);      // (probe builds only, nvsr_common.h)
    RingB rs{__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(packed_bwd + B_TOTAL + BLimb<LF>::OFFSET), 0, BLimb<LF>::WORDS * 4, 0x00020000), lds, 0,
             __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), (int)(threadIdx.x & 63), (threadIdx.x >> 6) * 1024u + (threadIdx.x & 63) * 16u};
    float* ldsf = reinterpret_cast<float*>(lds);
    for (int i = threadIdx.x; i < SMALL_FLOATS; i += BL_TPB) ldsf[BL_SMALL + i] = packed[P_SMALL + i];   // head weights of the FORWARD blob
    const float* small = ldsf + BL_SMALL;
#if BL_PROLOGUE_BARRIER
    // The transposed alpha head is the FIRST thing a tile computes, out of these LDS words and in front of the first ring barrier: without a barrier
    // here a wave could read head weights another wave had not written yet (stale LDS of the previous workgroup -> a non-finite density chain: gD, the
    // position planes' and the density decoder's gradients).  Never seen in a process that has the GPU to itself (the waves of a workgroup start
    // together and the reads sit behind two global loads); two processes time-slicing one GPU hit it in ~1.5 % of their refine iterations
    // (round 5: 2 of 133 pairs of independent processes, 0 of 133 alone).  The forward kernels use these words behind their
    // first ring barrier.
    __syncthreads();
#endif
    float* tile = ldsf + BL_TILES + rs.wave * TILE_FLOATS;
    unsigned hw_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    if (__builtin_amdgcn_readfirstlane(hw_id) & 1u) __builtin_amdgcn_s_sleep(64);
    // A wave's tile = 32 CONSECUTIVE SAMPLES OF ONE RAY (not one sample of 32 rays as in the forward): neighbouring samples share texels,
    // which lets the scatter merge their atomics (scatter_plane_runs), and the ray, its gates and dL/draw are contiguous per tile.
    const int nsc = (S + 31) / 32;                                   // sample chunks per ray
    const long nwt = N * nsc;                                        // wave tiles
    const long ntiles = (nwt + BL_WAVES - 1) / BL_WAVES;
    auto hid = [](const f32x16 (&a)[4], int kb0) { return [&a, kb0](int kb, int i) { const int k = kb0 + kb; return a[k >> 1][8 * (k & 1) + i]; }; };
    auto none = [](int) {};

    for (long tix = blockIdx.x; tix < ntiles; tix += gridDim.x) {
        asm volatile("" : "+v"(rs.voff), "+v"(rs.lane));
        const int lane = rs.lane, h = lane >> 5;
        const long wt = ((BL_ABLATE & 128) ? (long)blockIdx.x : tix) * BL_WAVES + rs.wave;
        const long ray0 = wt / nsc;
        const int s0 = (int)(wt - ray0 * nsc) * 32 + (lane & 31);
        const bool valid = ray0 < N && s0 < S;
        const long ray = ray0 < N ? ray0 : N - 1;
        const int s = s0 < S ? s0 : S - 1;
        const unsigned* cw = ringb_issue<LF>(rs, 0);
        const float* r = rays + ray * 11;
        const float zc = z[ray * S + s];
        f32x4 graw = *reinterpret_cast<const f32x4*>(g_raw + (ray * S + s) * 4);
        if (!valid) graw = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        // f16 limbs: every POINT's gradient times 2^-floor(log2(max |dL/draw| of the point)) (exact), undone on the feature gradients below.
        // A point is a column of every product of the chain (D[:, pt] = W^T G[:, pt]): a scale per column is exact, and both lane halves of
        // a column hold the same dL/draw, so no reduction is needed.  Round 3 scaled a whole wave tile (32 samples of one ray) by the tile's
        // largest magnitude: behind a surface dL/draw falls with the transmittance by more than 2^24 INSIDE a tile, and the samples far below
        // the largest lost their low limb, then their high one (relative error of a texel up to 1.3 % where the 3-bf16-limb backward has 0.1 %:
        // tests/test_hip_round4.py::test_f16_backward_with_the_dynamic_range_of_an_opaque_ray).  Through the four transposed layers
        // a point's gradient changes by a few binades at most: it stays inside the f16 limbs' 30; if not, the overflow reaches the planes as
        // NaN.  The exponent is clamped to [1, 253]: both factors stay normal numbers (a largest magnitude of 2^127 would give a scale of 0).
        // The two chains of a point -- density (from dL/dsigma) and rgb (from the three dL/drgb) -- carry a scale each: behind a surface
        // dL/dsigma and dL/drgb of ONE sample differ by many binades too; they meet as true f32 magnitudes (gD is unscaled when the density
        // chain ends, a plane's rgb part by the FMA that adds gD).
        float gs_d = 1.0f, gu_d = 1.0f, gs_r = 1.0f, gu_r = 1.0f;
        const f32x4 graw_true = graw;
        if constexpr (LF == 2) {
            auto pow2_scales = [](float m, float& sc_, float& un_) {
                const int e = (int)((__float_as_uint(m) >> 23) & 0xffu);          // biased exponent of the largest magnitude (0: all zero)
                // BL_F16_UP: the largest magnitude goes to [2^UP, 2^(UP+1)) instead of [1, 2) -- the chain shrinks a gradient by ~0.4 per
                // layer on typical weights and nothing rescales between layers, so from [1, 2) the low limbs of the last layers would be
                // subnormal; f16 overflows at 2^16: 2^(15 - UP) of growth over head + 4 layers remain
                const int eu = ((e == 0 || e == 255) ? 127 : (e > 253 ? 253 : (e < 1 + BL_F16_UP ? 1 + BL_F16_UP : e))) - BL_F16_UP;
                sc_ = __uint_as_float((unsigned)(254 - eu) << 23);                 // 2^-(e - UP - 127)
                un_ = __uint_as_float((unsigned)eu << 23);
            };
            pow2_scales(fabsf(graw[3]), gs_d, gu_d);
            pow2_scales(fmaxf(fabsf(graw[0]), fmaxf(fabsf(graw[1]), fabsf(graw[2]))), gs_r, gu_r);
        }
        const long q = record_row(ray, s, N, S);                             // record row (the forward wrote X / H of the same row)
        const bool rok = RECORD && valid;
        // Round 6: a layer's 16 record stores are the youngest vector-memory operations when the next layer's first block waits for its weight chunk
        // (the chunk's copy was issued a block earlier): that wait leaves them in flight instead of draining them (decode_limb.hip FIN_YOUNG).  The
        // stores are lane-masked (padding lanes hold zero gradients and must not touch a valid row), so the count holds only on a wave with a valid lane.
        constexpr int REC_YOUNG = RECORD ? 16 : 0;
#ifdef NVSR_RECORD_SAMPLE_MAJOR
        const bool rec_any = RECORD && __builtin_amdgcn_ballot_w64(valid) != 0ull;
#else
        const bool rec_any = RECORD;              // (staged rows: issued by every wave)
#endif
        // (f16 limbs: the accumulators carry the tile's power-of-two scale; the weight-gradient contraction reads unscaled f32 rows)
        float gunscale = gu_d;        // unscale of the chain being walked (record rows)
        // (round 6: whole cache lines per store through the wave's transposition tile, free until the planes are emitted -- record128_staged,
        //  decode_core.h; the tile's points are the consecutive record rows rq0 .. rq0 + 31, padding points go to the record's dump rows, so every
        //  wave issues all 16 stores of a layer: REC_YOUNG)
#ifdef NVSR_RECORD_SAMPLE_MAJOR
        auto record_grad = [&](float* base, long q_, int h_, const f32x16 (&a)[4]) {
            if (!rok) return;
            if constexpr (LF == 2) record128_scaled(base, q_, h_, a, gunscale);
            else record128(base, q_, h_, a);
        };
#else
        const int rchunk0 = (int)(wt - ray0 * nsc) * 32;
        const long rq0 = ray0 < N ? ray0 * S + rchunk0 : 0;
        const int rnvalid = ray0 < N ? (S - rchunk0 < 32 ? S - rchunk0 : 32) : 0;
        auto record_grad = [&](float* base, long, int, const f32x16 (&a)[4]) {
            record128_staged<LF == 2>(tile, base, rq0, rnvalid, rec.dump, lane, a, gunscale);
        };
#endif
        if (rok && h == 0) *reinterpret_cast<f32x4*>(rec.g4 + 4 * q) = graw_true;     // (the record holds UNSCALED gradients)
        // Everything below is re-read where it is used instead of being kept across the step (two accumulator sets, gD and the limbs
        // already fill the 256 registers of a wave at two waves per SIMD): a layer's two gate words (slot 0..3 density, 4..7 rgb) ...
        typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
        const u32x2_* gk = reinterpret_cast<const u32x2_*>(gates + ((ray * S + s) * 2 + h) * 16);
#if BL_GATE_PREFETCH
        // the eight gate word pairs of the tile, loaded up front (16 registers): re-read where they are used each load sat behind a block's
        // fence with its full latency exposed
        u32x2_ gw[8];
#pragma unroll
        for (int l = 0; l < 8; ++l) gw[l] = gk[l];
        auto gate = [&](int slot) { return Masks{{gw[slot][0], gw[slot][1]}}; };
#else
        auto gate = [&](int slot) { const u32x2_ v = gk[slot]; return Masks{{v[0], v[1]}}; };
#endif
        // ... and the taps of a plane, from the ray
        auto pos_taps = [&](int d) {
            const float n0 = norm_coord(__fadd_rn(r[0], __fmul_rn(r[3], zc)), sc.lo[0], sc.range[0]);
            const float n1 = norm_coord(__fadd_rn(r[1], __fmul_rn(r[4], zc)), sc.lo[1], sc.range[1]);
            const float n2 = norm_coord(__fadd_rn(r[2], __fmul_rn(r[5], zc)), sc.lo[2], sc.range[2]);
            const float* M = sc.proj + 6 * d;
            int ix, iy;
            Taps t = make_taps_cell(sc, d, n0 * M[0] + n1 * M[2] + n2 * M[4], n0 * M[1] + n1 * M[3] + n2 * M[5], ix, iy);
#if BL_SCATTER >= 2
            t.o00 |= (ix & 1) | ((iy & 1) << 1);             // the cell's parity rides in the low bits (scatter_plane_cached)
#endif
            return t;
        };
        f32x16 accA[4], accB[4];
        Limbs<LF> cur, fa;
        SplitPend tp;
        // tail of a block: split K-block kb of the same gradient into the limbs the next block starts with
        auto tail_of = [&tp](const f32x16 (&a)[4], int kb) {
            return [&a, kb, &tp](int slice, Limbs<LF>& nxt) { split_slice<LF>(slice, [&a, kb](int i) { return a[kb >> 1][8 * (kb & 1) + i]; }, nxt, tp); };
        };
        // one chunk of a hidden^T layer (2 K-blocks): wait, start the next copy, split the first K-block of G, multiply
#define BL_FENCE(ACC)                                                                                   \
        asm volatile("" : "+v"(ACC[0]), "+v"(ACC[1]), "+v"(ACC[2]), "+v"(ACC[3]) : : "memory");         \
        __builtin_amdgcn_sched_barrier(0);
        // (FIRST: the chunk's first K-block is split here, exposed; otherwise the previous block's tail produced it in its MFMA gaps)
#define BL_HBLOCK(ZERO, FIRST, G, KB0, GN, NEXT_CHUNK, TAIL) BL_HBLOCK_Y(0, ZERO, FIRST, G, KB0, GN, NEXT_CHUNK, TAIL)
#define BL_HBLOCK_Y(YOUNGER, ZERO, FIRST, G, KB0, GN, NEXT_CHUNK, TAIL)                                 \
        {                                                                                               \
            ringb_sync<YOUNGER>(rec_any);                                                               \
            const unsigned* nw = ringb_issue<LF>(rs, NEXT_CHUNK);                                           \
            if (FIRST && !(BL_ABLATE & 8)) { auto s_ = hid(G, KB0); split_all<LF>([&](int i) { return s_(0, i); }, cur); } \
            limb_block<LF, 2, ZERO, true>(cw, lane, GN, cur, fa, hid(G, KB0), none, TAIL);               \
            cw = nw;                                                                                    \
            BL_FENCE(GN)                                                                                \
        }
        // gn = mask .* (W^T g): chunks C0 .. C0 + 3
#ifdef NVSR_NO_TAILS      // A/B switch (tools/): every block splits its first K-block itself
#define BL_HIDDEN_T(G, MK, GN, C0)                                                                      \
        BL_HBLOCK(true, true, G, 0, GN, (C0) + 1, NoTail{})                                             \
        BL_HBLOCK(false, true, G, 2, GN, (C0) + 2, NoTail{})                                            \
        BL_HBLOCK(false, true, G, 4, GN, (C0) + 3, NoTail{})                                            \
        BL_HBLOCK(false, true, G, 6, GN, (C0) + 4, NoTail{})                                            \
        if (!(BL_ABLATE & 2)) { apply_mask(gate(MK), GN); BL_FENCE(GN) }
#else
#define BL_HIDDEN_T(G, MK, GN, C0)                                                                      \
        BL_HBLOCK_Y(REC_YOUNG, true, true, G, 0, GN, (C0) + 1, tail_of(G, 2))                           \
        BL_HBLOCK(false, false, G, 2, GN, (C0) + 2, tail_of(G, 4))                                      \
        BL_HBLOCK(false, false, G, 4, GN, (C0) + 3, tail_of(G, 6))                                      \
        BL_HBLOCK(false, false, G, 6, GN, (C0) + 4, NoTail{})                                           \
        if (!(BL_ABLATE & 2)) { apply_mask(gate(MK), GN); BL_FENCE(GN) }
#endif
        // acc2 += W0^T g: chunks C0, C0 + 1; LAST: no chunk follows in this step
#define BL_LAYER0_T(G, ACC2, C0, LAST)                                                                  \
        {                                                                                               \
            ringb_sync();                                                                               \
            const unsigned* nw = ringb_issue<LF>(rs, (C0) + 1);                                             \
            limb_mm2<LF>(cw, lane, ACC2, hid(G, 0));                                                        \
            cw = nw;                                                                                    \
            asm volatile("" : "+v"(ACC2[0]), "+v"(ACC2[1]) : : "memory");                               \
            __builtin_amdgcn_sched_barrier(0);                                                          \
            ringb_sync();                                                                               \
            if (!(LAST)) nw = ringb_issue<LF>(rs, (C0) + 2);                                                \
            limb_mm2<LF>(cw, lane, ACC2, hid(G, 4));                                                        \
            cw = nw;                                                                                    \
            asm volatile("" : "+v"(ACC2[0]), "+v"(ACC2[1]) : : "memory");                               \
            __builtin_amdgcn_sched_barrier(0);                                                          \
        }
        // ---- density branch -> gD
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(small + S_ALPHA_W + (ib * 4 + qq) * 8 + h * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) accA[ib][4 * qq + j] = wv[j] * (graw[3] * gs_d);
            }
        apply_mask(gate(3), accA);
        BL_FENCE(accA)      // the masked gradient is a value of its own: without the fence hipcc keeps mask AND unmasked value alive to fold
                            // `(g & keep) & 0xffff0000` of the limb split into one v_bitop3 -- 128 more live registers, 165 spills
        if (RECORD) record_grad(rec.Gd + 3L * HID * rec.Pp, q, h, accA);
        BL_HIDDEN_T(accA, 2, accB, 0)
        if (RECORD) record_grad(rec.Gd + 2L * HID * rec.Pp, q, h, accB);
        BL_HIDDEN_T(accB, 1, accA, 4)
        if (RECORD) record_grad(rec.Gd + 1L * HID * rec.Pp, q, h, accA);
        BL_HIDDEN_T(accA, 0, accB, 8)
        if (RECORD) record_grad(rec.Gd, q, h, accB);
        f32x16 gD[2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) gD[b][rr] = 0.0f;
        BL_LAYER0_T(accB, gD, 12, false)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) gD[b][rr] = LF == 2 ? div3(gD[b][rr]) * gu_d : div3(gD[b][rr]);     // (f16 limbs: back to its true magnitude)
        // ---- rgb branch
        asm volatile("" ::: "memory");
        graw = *reinterpret_cast<const f32x4*>(g_raw + (ray * S + s) * 4);
        if (!valid) graw = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (LF == 2) graw = graw * gs_r;
        gunscale = gu_r;
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int o = (ib * 4 + qq) * 8 + h * 4;
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(small + S_RGB_W + o);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(small + S_RGB_W + HID + o);
                const f32x4 w2 = *reinterpret_cast<const f32x4*>(small + S_RGB_W + 2 * HID + o);
#pragma unroll
                for (int j = 0; j < 4; ++j) accA[ib][4 * qq + j] = fmaf(w2[j], graw[2], fmaf(w1[j], graw[1], w0[j] * graw[0]));
            }
        apply_mask(gate(7), accA);
        BL_FENCE(accA)
        if (RECORD) record_grad(rec.Gr + 3L * HID * rec.Pp, q, h, accA);
        BL_HIDDEN_T(accA, 6, accB, 14)
        if (RECORD) record_grad(rec.Gr + 2L * HID * rec.Pp, q, h, accB);
        BL_HIDDEN_T(accB, 5, accA, 18)
        if (RECORD) record_grad(rec.Gr + 1L * HID * rec.Pp, q, h, accA);
        BL_HIDDEN_T(accA, 4, accB, 22)
        if (RECORD) record_grad(rec.Gr, q, h, accB);
        // what happens to a plane's finished feature gradient gF (rows c = 32 b + (r & 3) + 8 (r >> 2) + 4 h, TRUE magnitudes): view rows or scatter
        auto emit_plane = [&](int d, f32x16 (&gF)[2]) {
            if (gp.p[d] && !(BL_ABLATE & 4)) {
                if (d == 3 && gview) {
                    // every sample of the ray taps the same four view-plane texels: the tile's 32 gradient rows are summed here, ONE row
                    // [ray * nsc + chunk][48] goes to the workspace and view_reduce_scatter_kernel adds a ray's nsc rows (padding samples
                    // carry zero gradient)
                    const int hh = lane >> 5, pt = lane & 31;
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int rr = 0; rr < 16; ++rr) {
                            if (b == 1 && rr >= 8) continue;
                            tile[pt * TILE_STRIDE + 32 * b + (rr & 3) + 8 * (rr >> 2) + 4 * hh] = gF[b][rr];
                        }
                    __builtin_amdgcn_wave_barrier();
                    if (lane < C && ray0 < N) {
                        float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, v3 = 0.0f;
#pragma unroll
                        for (int p_ = 0; p_ < 32; p_ += 4) {
                            v0 += tile[p_ * TILE_STRIDE + lane]; v1 += tile[(p_ + 1) * TILE_STRIDE + lane]; v2 += tile[(p_ + 2) * TILE_STRIDE + lane]; v3 += tile[(p_ + 3) * TILE_STRIDE + lane];
                        }
                        gview[wt * C + lane] = (v0 + v1) + (v2 + v3);            // wt = ray * nsc + chunk
                    }
                    __builtin_amdgcn_wave_barrier();
                } else {
                    const Taps t = (d < 3) ? pos_taps(d) : view_taps(sc, r[8], r[9], r[10]);
#if BL_SCATTER == 3
                    if (d < 3) scatter_plane_cached_v(gF, tile, t, gp.p[d], lane, valid);
                    else scatter_plane_runs(gF, tile, t, gp.p[d], lane, valid);
#elif BL_SCATTER == 2
                    if (d < 3) scatter_plane_cached(gF, tile, t, gp.p[d], lane, valid);
                    else scatter_plane_runs(gF, tile, t, gp.p[d], lane, valid);          // (view plane without a row workspace: one cell per ray)
#elif BL_SCATTER == 1
                    scatter_plane_runs(gF, tile, t, gp.p[d], lane, valid);
#else
                    scatter_plane(gF, tile, t, gp.p[d], lane, valid);
#endif
                }
            }
        };
#if BL_RGB0_PAIRS == 3
        // Three pairs of 32-row blocks cover the 192 rows of the rgb layer-0^T; pair q holds rows 64 q .. 64 q + 63:
        //   P0 = plane 0 (rows 0..47) + plane 1's channels 0..15;  P1 = plane 1's channels 16..47 + plane 2's channels 0..31;
        //   P2 = plane 2's channels 32..47 + plane 3 (the view plane).
        // Register r of block b is row 32 b + (r & 3) + 8 (r >> 2) + 4 h: registers 0..7 are rows 0..15 of the block, 8..15 rows 16..31, for either
        // lane half -- so a 16-row shift is a shift by 8 registers and the planes' rows are re-assembled by renaming registers.
        const float ur = LF == 2 ? gu_r : 1.0f;                    // (f16 limbs: the rgb chain back to its true magnitude)
        auto val = [&](float a) { return LF == 2 ? a * ur : a; };
        auto vald = [&](float a, float d_) { return LF == 2 ? fmaf(a, ur, d_) : a + d_; };        // + the density part (position planes)
        f32x16 P[2], gF[2];
        float carry[16];
        auto zeroP = [&]() {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) P[b][rr] = 0.0f;
        };
        zeroP();
        BL_LAYER0_T(accB, P, 26, false)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {                          // plane 0 = P0 rows 0..47
            gF[0][rr] = vald(P[0][rr], gD[0][rr]);
            gF[1][rr] = rr < 8 ? vald(P[1][rr], gD[1][rr]) : 0.0f;
        }
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) carry[rr] = vald(P[1][rr + 8], gD[0][rr]);             // plane 1's channels 0..15 (P0 rows 48..63)
        emit_plane(0, gF);
        zeroP();
        BL_LAYER0_T(accB, P, 28, false)
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {                           // plane 1 = [carry | P1 rows 0..31]
            gF[0][rr] = carry[rr];
            gF[0][rr + 8] = vald(P[0][rr], gD[0][rr + 8]);         // channels 16..31
            gF[1][rr] = vald(P[0][rr + 8], gD[1][rr]);             // channels 32..47
            gF[1][rr + 8] = 0.0f;
        }
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) carry[rr] = vald(P[1][rr], gD[0][rr]);                // plane 2's channels 0..31 (P1 rows 32..63)
        emit_plane(1, gF);
        zeroP();
        BL_LAYER0_T(accB, P, 30, true)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {                          // plane 2 = [carry | P2 rows 0..15]
            gF[0][rr] = carry[rr];
            gF[1][rr] = rr < 8 ? vald(P[0][rr], gD[1][rr]) : 0.0f;
        }
        emit_plane(2, gF);
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {                           // plane 3 (view) = P2 rows 16..63: no density part
            gF[0][rr] = val(P[0][rr + 8]);
            gF[0][rr + 8] = val(P[1][rr]);
            gF[1][rr] = val(P[1][rr + 8]);
            gF[1][rr + 8] = 0.0f;
        }
        emit_plane(3, gF);
#else
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            f32x16 gF[2];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) gF[b][rr] = (d < 3 && LF != 2) ? gD[b][rr] : 0.0f;
            BL_LAYER0_T(accB, gF, 26 + 2 * d, d == 3)
            if constexpr (LF == 2) {        // the rgb part back to its true magnitude, plus the density part (position planes)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int rr = 0; rr < 16; ++rr) gF[b][rr] = (d < 3) ? fmaf(gF[b][rr], gu_r, gD[b][rr]) : gF[b][rr] * gu_r;
            }
            emit_plane(d, gF);
        }
#endif
#undef BL_LAYER0_T
#undef BL_HIDDEN_T
#undef BL_HBLOCK
#undef BL_FENCE
    }
    ringb_sync();
}

}  // namespace nvsr

using namespace nvsr;

extern "C" int nvsr_pack_decoder_bwd_limbs_launch(const float* natural, float* packed_bwd, nvsr_stream_t stream) {
    hipLaunchKernelGGL(pack_decoder_bwd_limbs_kernel<3>, dim3((BL_WORDS + 255) / 256), dim3(256), 0, (hipStream_t)stream, natural,
                       reinterpret_cast<unsigned*>(packed_bwd) + B_TOTAL);
    hipLaunchKernelGGL(pack_decoder_bwd_limbs_kernel<2>, dim3((BLimb<2>::WORDS + 255) / 256), dim3(256), 0, (hipStream_t)stream, natural,
                       reinterpret_cast<unsigned*>(packed_bwd) + B_TOTAL + BLimb<2>::OFFSET);
    return NVSR_CHECK_LAUNCH();
}

// nvsr_render_pass_backward_gates (render_bwd.hip) with the decoder arithmetic set to bf16 limbs; arguments already validated there
extern "C" int nvsr_render_pass_backward_gates_limb_launch(int limbs, const nvsr_scene* scene, const float* packed_decoder, const float* packed_bwd,
                                                           int64_t N, int S, const float* rays, const float* z, const float* g_raw,
                                                           const uint32_t* gates, float* const* grad_planes, float* view_ws, float* record,
                                                           nvsr_stream_t stream) {
    GradPlanes gp;
    for (int d = 0; d < 4; ++d) gp.p[d] = grad_planes ? grad_planes[d] : nullptr;
    const int64_t ntiles = (N * (int64_t)((S + 31) / 32) + BL_WAVES - 1) / BL_WAVES;       // 4 wave tiles (ray, 32 samples) per workgroup step
    const int64_t grid = ntiles < 2048 ? ntiles : 2048;
    if (limbs == 2 && !record)      // f16 limbs
        hipLaunchKernelGGL((render_pass_backward_gates_limb_kernel<false, 2>), dim3((unsigned)grid), dim3(BL_TPB), 0, (hipStream_t)stream, to_dev(scene),
                           packed_decoder, packed_bwd, (long)N, S, rays, z, g_raw, gates, gp, view_ws, DecRecord{});
    else if (limbs == 2)            // f16 limbs + the gradient half of the weight-gradient record (written unscaled)
        hipLaunchKernelGGL((render_pass_backward_gates_limb_kernel<true, 2>), dim3((unsigned)grid), dim3(BL_TPB), 0, (hipStream_t)stream, to_dev(scene),
                           packed_decoder, packed_bwd, (long)N, S, rays, z, g_raw, gates, gp, view_ws, make_record(record, (long)N, S));
    else if (record)
        hipLaunchKernelGGL(render_pass_backward_gates_limb_kernel<true>, dim3((unsigned)grid), dim3(BL_TPB), 0, (hipStream_t)stream, to_dev(scene),
                           packed_decoder, packed_bwd, (long)N, S, rays, z, g_raw, gates, gp, view_ws, make_record(record, (long)N, S));
    else
        hipLaunchKernelGGL(render_pass_backward_gates_limb_kernel<false>, dim3((unsigned)grid), dim3(BL_TPB), 0, (hipStream_t)stream, to_dev(scene),
                           packed_decoder, packed_bwd, (long)N, S, rays, z, g_raw, gates, gp, view_ws, DecRecord{});
    return NVSR_CHECK_LAUNCH();
}
