// Fused render pass, second generation: TWO point tiles per wave, software-pipelined against each other.
//
// render.hip's kernel keeps the matrix pipe ~78 % busy: per step and wave, 2016 MFMAs (129k cycles) are followed / interleaved by
// ~45k cycles in which the wave issues no MFMA (plane gathers, ring barriers, bias + ReLU, the VALU heads), and a second wave on
// the same SIMD hides only about a third of that (MFMA issue is arbitrated oldest-first, so the partner mostly starves).
// Here one wave per SIMD (512 registers) owns two independent tiles X and Y of 32 points each and alternates
//        [ MFMAs of X on weight chunk c  |  everything non-MFMA of Y, sliced into the 64-cycle shadows of those MFMAs ]
//        [ MFMAs of Y on weight chunk c  |  everything non-MFMA of X ]
// so gathers, bilinear blends, bias loads, ReLUs, heads and compositing of one tile are issued while the matrix pipe works for
// the other.  A block's side work is a functor called once per group of 4 MFMAs (<= ~12 VALU / LDS / VMEM instructions).
// Weight chunks are 64 KB (9 per step instead of 17), one barrier per chunk serves both tiles.
//
// Register plan (512 per wave): per tile ONE accumulator set `acc` that only MFMAs (and the bias initialisation) write -- hipcc keeps such
// tuples in AGPRs -- and one activation set `act` = ReLU(acc) in VGPRs, the B operands of the next layer.  (A ping-pong pair of
// accumulator sets that the VALU ReLUs in place is forced into architectural VGPRs: 226 spills in the first version of this file.)
#include <type_traits>
#include <utility>

#include "side_work.h"

#ifndef R2_STAMP
#define R2_STAMP 0    // debug builds: raw_out[..., 0:2] of tile X = s_memtime cycles of the first hidden block pair
#endif
namespace nvsr {

constexpr int RAY2_FLOATS = 16;
constexpr int LDS2_RAYS = LDS2_SMALL + SMALL_FLOATS;
constexpr int LDS2_FLOATS = LDS2_RAYS + RAYS2 * RAY2_FLOATS;
static_assert(LDS2_FLOATS * 4 <= 160 * 1024, "LDS budget");

// per-tile registers
struct Tile {
    f32x16 acc[4];   // layer accumulators (AGPRs): bias, then += W * input
    f32x16 act[4];   // ReLU(acc) of the finished layer (VGPRs)
    float D[HALF_C], F[HALF_C];
    float T, cr, cg, cb, dep, ac, zc, zn;
    float raw[4];
};

// act = ReLU(acc) (64 elements) over the first NG groups of a block, spread over the 4 slots of each group.
// What hipcc makes of it: it copies each 16-register accumulator tuple to VGPRs right behind the tuple's last MFMA (s_nop 17 + 16
// v_accvgpr_read, exposed) and the slots only do the v_max.  Forcing the read into the slot (inline-asm v_accvgpr_read with an "a"
// operand) interleaves exactly as written -- and is SLOWER (257.2 vs 253.9 ms): a VALU read of the AGPR file between two MFMAs costs
// the matrix pipe more than the exposed copy does.
template <int NG>
__device__ __forceinline__ void relu_side(int g, int j, const f32x16 (&acc)[4], f32x16 (&act)[4]) {
    constexpr int PER = (64 + 4 * NG - 1) / (4 * NG);      // elements per slot
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int r = (g * 4 + j) * PER + k;
        if (g < NG && r < 64) {
            // one v_max: written as fmaxf, hipcc first canonicalises the operand (v_max x, x, x) -- and a VALU instruction in an MFMA
            // shadow is not free: measured (tools/mfma_ubench.hip), every one adds its ~4 issue cycles to the MFMA stream
            float o;
            asm("v_max_f32 %0, 0, %1" : "=v"(o) : "v"(acc[r >> 4][r & 15]));
            act[r >> 4][r & 15] = o;
        }
    }
}
// bias -> accumulator init: one ds_read_b128 per group from group G0 on (PER per group in short blocks), read in slot 0, written to the
// accumulators in slot 3
struct BiasPend { f32x4 v[2]; };
template <int G0, int PER = 1>
__device__ __forceinline__ void bias_side(int g, int j, const float* bias, int h, f32x16 (&acc)[4], BiasPend& pend) {
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int v = (g - G0) * PER + k;
        if (g >= G0 && v < 16) {
            if (j == 0) pend.v[k] = *reinterpret_cast<const f32x4*>(bias + v * 8 + h * 4);
            if (j == 3) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[v >> 2][4 * (v & 3) + e] = pend.v[k][e];
            }
        }
    }
}
__device__ __forceinline__ void load_bias2(const float* bias, int h, f32x16 (&acc)[4]) {
    BiasPend p;
#pragma unroll
    for (int g = 0; g < 16; ++g) { bias_side<0>(g, 0, bias, h, acc, p); bias_side<0>(g, 3, bias, h, acc, p); }
}

// =====================================================================================================================
__global__ __launch_bounds__(TPB2, 1) void render_pass2_kernel(SceneDev sc, const float* __restrict__ packed, long N, int S,
                                                              const float* __restrict__ rays, const float* __restrict__ z,
                                                              const float* __restrict__ noise, int white,
                                                              float* __restrict__ rgb, float* __restrict__ disp,
                                                              float* __restrict__ acc, float* __restrict__ weights,
                                                              float* __restrict__ depth, float* __restrict__ raw_out) {
    __shared__ __attribute__((aligned(16))) float lds[LDS2_FLOATS];
    NVSR_RACE_PROBE_DELAY(lds);      // (probe builds only, nvsr_common.h)
    Ring2 rs{packed, lds, 0, (int)(threadIdx.x >> 6), (int)(threadIdx.x & 63), (threadIdx.x >> 6) * 1024u + (threadIdx.x & 63) * 16u};
    for (int i = threadIdx.x; i < SMALL_FLOATS; i += TPB2) lds[LDS2_SMALL + i] = packed[P_SMALL + i];
    const float* small = lds + LDS2_SMALL;

    // rays: tile X = rays [base, base+32), tile Y = [base+32, base+64) of this wave
    const int lane0 = rs.lane;
    const long base = (long)blockIdx.x * RAYS2 + rs.wave * 64 + (lane0 & 31);
    long rayX = base, rayY = base + 32;
    const bool validX = rayX < N, validY = rayY < N;
    if (!validX) rayX = N - 1;
    if (!validY) rayY = N - 1;
    float* rcX = lds + LDS2_RAYS + (rs.wave * 64 + (lane0 & 31)) * RAY2_FLOATS;
    float* rcY = rcX + 32 * RAY2_FLOATS;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float* r = rays + (k ? rayY : rayX) * 11;
        float* rc = k ? rcY : rcX;
        const float dx = r[3], dy = r[4], dz = r[5];
        const Taps vt = view_taps(sc, r[8], r[9], r[10]);
        const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
        if (lane0 < 32) {
            reinterpret_cast<f32x4*>(rc)[0] = f32x4{r[0], r[1], r[2], dx};
            reinterpret_cast<f32x4*>(rc)[1] = f32x4{dy, dz, nrm, 0.0f};
            reinterpret_cast<f32x4*>(rc)[2] = f32x4{__int_as_float(vt.o00), __int_as_float(vt.o01), __int_as_float(vt.o10), __int_as_float(vt.o11)};
            reinterpret_cast<f32x4*>(rc)[3] = f32x4{vt.nw, vt.ne, vt.sw, vt.se};
        }
    }
    const float* zX = z + rayX * S;
    const float* zY = z + rayY * S;

    Tile X, Y;
    X.T = Y.T = 1.0f;
    X.cr = X.cg = X.cb = X.dep = X.ac = 0.0f;
    Y.cr = Y.cg = Y.cb = Y.dep = Y.ac = 0.0f;
    X.zc = zX[0]; Y.zc = zY[0];
    RawTaps2 rt;

    auto point_norm = [&](const float* rc, float zc, float& n0, float& n1, float& n2) {
        const f32x4 c0 = reinterpret_cast<const f32x4*>(rc)[0], c1 = reinterpret_cast<const f32x4*>(rc)[1];
        n0 = norm_coord(__fadd_rn(c0[0], __fmul_rn(c0[3], zc)), sc.lo[0], sc.range[0]);
        n1 = norm_coord(__fadd_rn(c0[1], __fmul_rn(c1[0], zc)), sc.lo[1], sc.range[1]);
        n2 = norm_coord(__fadd_rn(c0[2], __fmul_rn(c1[1], zc)), sc.lo[2], sc.range[2]);
    };
    auto view_job = [&](const float* rc) {
        const f32x4 c2 = reinterpret_cast<const f32x4*>(rc)[2], c3 = reinterpret_cast<const f32x4*>(rc)[3];
        GatherJob j;
        j.plane = sc.plane[3];
        j.t.o00 = __float_as_int(c2[0]); j.t.o01 = __float_as_int(c2[1]); j.t.o10 = __float_as_int(c2[2]); j.t.o11 = __float_as_int(c2[3]);
        j.t.nw = c3[0]; j.t.ne = c3[1]; j.t.sw = c3[2]; j.t.se = c3[3];
        return j;
    };

    // Round 6 (found by the race probe, tools/race_probe.sh): the first sample's bias preload (load_bias2 below) reads `small` IN FRONT OF the
    // first ring barrier -- words that every wave of the workgroup has just copied into LDS.  A wave that starts late leaves the others reading
    // whatever the previous workgroup left there: the right bytes after a launch of this kernel with the same decoder (why every test passed), the
    // wrong ones otherwise (probe: 3-5 % of a frame's pixels off by 1e-5 .. 1e-2).  Same class as round 5's backward-prologue race.  The barrier also
    // publishes the ray cache (written by lanes 0..31, read by all 64).
    __syncthreads();
    const float* cur = ring2_issue<48>(rs, P_RGB0);        // chunk C0 of sample 0; every later C0 is issued during the previous sample
    for (int s = 0; s < S; ++s) {
        asm volatile("" : "+v"(rs.voff), "+v"(rs.lane));
        const int lane = rs.lane, h = lane >> 5;
        const bool last = (s + 1 == S);
        X.zn = last ? 0.0f : zX[s + 1];
        Y.zn = last ? 0.0f : zY[s + 1];
        const float nzX = noise ? noise[rayX * S + s] : 0.0f;
        const float nzY = noise ? noise[rayY * S + s] : 0.0f;
        float xn0, xn1, xn2, yn0, yn1, yn2;
        point_norm(rcX, X.zc, xn0, xn1, xn2);
        point_norm(rcY, Y.zc, yn0, yn1, yn2);

        BiasPend bp;
        HeadPend<3> hp3;
        HeadPend<1> hp1;
        // ---- prologue (exposed): plane 0 of X gathered and blended; chunk C0 has been in flight since the previous sample's last chunk.
        // (Producing the gather during the previous sample's last block as well was tried: with both activation sets live there the
        // VGPR file overflows -- 125 spills, 268.7 vs 256.1 ms.)
        GatherJob job;
        job.plane = sc.plane[0]; job.t = pos_taps2(sc, 0, xn0, xn1, xn2);
#if R2_ABLATE & 4
#pragma unroll
        for (int c = 0; c < HALF_C; ++c) X.F[c] = job.t.nw * (float)c;
#else
#pragma unroll
        for (int g = 0; g < 24; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) gather_side(g, j, job, h, rt, X.F);
#endif
#pragma unroll
        for (int c = 0; c < HALF_C; ++c) X.D[c] = X.F[c];
        load_bias2(small + S_BIAS + 4 * HID, h, X.acc);
        ring2_sync();
        const float* nxt = ring2_issue<48>(rs, P_RGB0 + 2 * P_PLANE_FLOATS);

        // ---- rgb layer 0 ---------------------------------------------------------------------------------------------------
        auto fb = [](const float (&f)[HALF_C]) { return [&f](int g, int j) { return f[4 * (g >> 2) + j]; }; };
        // B1: X plane 0 | Y: gather plane 0, bias
        job.plane = sc.plane[0]; job.t = pos_taps2(sc, 0, yn0, yn1, yn2);
        mfma_block<24>(cur, lane, X.acc, fb(X.F), [&](int g, int j) {
            gather_side(g, j, job, h, rt, Y.F);
            bias_side<0>(g, j, small + S_BIAS + 4 * HID, h, Y.acc, bp);
        });
#pragma unroll
        for (int c = 0; c < HALF_C; ++c) Y.D[c] = Y.F[c];
        // B2: Y plane 0 | X: gather plane 1, D += F
        job.plane = sc.plane[1]; job.t = pos_taps2(sc, 1, xn0, xn1, xn2);
        mfma_block<24>(cur, lane, Y.acc, fb(Y.F), [&](int g, int j) { gather_side(g, j, job, h, rt, X.F); });
#pragma unroll
        for (int c = 0; c < HALF_C; ++c) X.D[c] = __fadd_rn(X.D[c], X.F[c]);
        // B3: X plane 1 | Y: gather plane 1
        job.plane = sc.plane[1]; job.t = pos_taps2(sc, 1, yn0, yn1, yn2);
        mfma_block<24>(cur + P_PLANE_FLOATS, lane, X.acc, fb(X.F), [&](int g, int j) { gather_side(g, j, job, h, rt, Y.F); });
#pragma unroll
        for (int c = 0; c < HALF_C; ++c) Y.D[c] = __fadd_rn(Y.D[c], Y.F[c]);
        // B4: Y plane 1 | X: gather plane 2, D = (D + F) / 3   (combine_pos_planes 'avg', models.py:358-359)
        job.plane = sc.plane[2]; job.t = pos_taps2(sc, 2, xn0, xn1, xn2);
        mfma_block<24>(cur + P_PLANE_FLOATS, lane, Y.acc, fb(Y.F), [&](int g, int j) { gather_side(g, j, job, h, rt, X.F); });
#pragma unroll
        for (int c = 0; c < HALF_C; ++c) X.D[c] = div3(__fadd_rn(X.D[c], X.F[c]));
        cur = nxt;
        ring2_sync();
        nxt = ring2_issue<64>(rs, P_RGB1);
        // B5: X plane 2 | Y: gather plane 2
        job.plane = sc.plane[2]; job.t = pos_taps2(sc, 2, yn0, yn1, yn2);
        mfma_block<24>(cur, lane, X.acc, fb(X.F), [&](int g, int j) { gather_side(g, j, job, h, rt, Y.F); });
#pragma unroll
        for (int c = 0; c < HALF_C; ++c) Y.D[c] = div3(__fadd_rn(Y.D[c], Y.F[c]));
        // B6: Y plane 2 | X: gather view plane
        job = view_job(rcX);
        mfma_block<24>(cur, lane, Y.acc, fb(Y.F), [&](int g, int j) { gather_side(g, j, job, h, rt, X.F); });
        // B7: X view plane | Y: gather view plane
        job = view_job(rcY);
        mfma_block<24>(cur + P_PLANE_FLOATS, lane, X.acc, fb(X.F), [&](int g, int j) { gather_side(g, j, job, h, rt, Y.F); });
        // B8: Y view plane | X: act = ReLU(acc), acc = bias of rgb layer 1
        mfma_block<24>(cur + P_PLANE_FLOATS, lane, Y.acc, fb(Y.F), [&](int g, int j) {
            relu_side<12>(g, j, X.acc, X.act);
            bias_side<12, 2>(g, j, small + S_BIAS + 5 * HID, h, X.acc, bp);
        });
        cur = nxt;

        // ---- hidden layers: rgb 1..3 (a chunk = a whole 128 x 128 layer) ------------------------------------------------------
        auto hb = [](const f32x16 (&in)[4]) { return [&in](int g, int j) { return in[g >> 4][4 * ((g >> 2) & 3) + j]; }; };
        // block pair of layer l: [X layer l | Y finishes layer l-1 and arms layer l]  [Y layer l | X finishes layer l, arms layer l+1]
        ring2_sync();
        nxt = ring2_issue<64>(rs, P_RGB1 + P_HID_FLOATS);
#if R2_STAMP
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long st0 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
#endif
        mfma_block<64>(cur, lane, X.acc, hb(X.act), [&](int g, int j) { relu_side<32>(g, j, Y.acc, Y.act); bias_side<32>(g, j, small + S_BIAS + 5 * HID, h, Y.acc, bp); });
#if R2_STAMP
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long st1 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
#endif
        mfma_block<64>(cur, lane, Y.acc, hb(Y.act), [&](int g, int j) { relu_side<32>(g, j, X.acc, X.act); bias_side<32>(g, j, small + S_BIAS + 6 * HID, h, X.acc, bp); });
#if R2_STAMP
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long st2 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
#endif
        cur = nxt;
        ring2_sync();
        nxt = ring2_issue<64>(rs, P_RGB1 + 2 * P_HID_FLOATS);
        mfma_block<64>(cur, lane, X.acc, hb(X.act), [&](int g, int j) { relu_side<32>(g, j, Y.acc, Y.act); bias_side<32>(g, j, small + S_BIAS + 6 * HID, h, Y.acc, bp); });
        mfma_block<64>(cur, lane, Y.acc, hb(Y.act), [&](int g, int j) { relu_side<32>(g, j, X.acc, X.act); bias_side<32>(g, j, small + S_BIAS + 7 * HID, h, X.acc, bp); });
        cur = nxt;
        ring2_sync();
        nxt = ring2_issue<24>(rs, P_DEN0);
        mfma_block<64>(cur, lane, X.acc, hb(X.act), [&](int g, int j) { relu_side<32>(g, j, Y.acc, Y.act); bias_side<32>(g, j, small + S_BIAS + 7 * HID, h, Y.acc, bp); });
        // Y rgb layer 3 | X: ReLU, rgb heads, bias of density layer 0
        float hx[3] = {0.0f, 0.0f, 0.0f};
        mfma_block<64>(cur, lane, Y.acc, hb(Y.act), [&](int g, int j) {
            relu_side<32>(g, j, X.acc, X.act);
            if (g >= 32) heads_side<3>(g - 32, j, small + S_RGB_W, h, X.act, hx, hp3);
            bias_side<48>(g, j, small + S_BIAS + 0 * HID, h, X.acc, bp);
        });
#pragma unroll
        for (int c = 0; c < 3; ++c) X.raw[c] = (hx[c] + __shfl_xor(hx[c], 32)) + small[S_HEAD_B + 1 + c];
#if R2_STAMP
        X.raw[0] = (float)(st1 - st0); X.raw[1] = (float)(st2 - st1);
#endif
        cur = nxt;

        // ---- density decoder -----------------------------------------------------------------------------------------------
        ring2_sync();
        nxt = ring2_issue<64>(rs, P_DEN1);
        // X density layer 0 | Y: ReLU, rgb heads
        float hy[3] = {0.0f, 0.0f, 0.0f};
        mfma_block<24>(cur, lane, X.acc, fb(X.D), [&](int g, int j) {
            relu_side<8>(g, j, Y.acc, Y.act);
            if (g >= 8) heads_side<3>(g - 8, j, small + S_RGB_W, h, Y.act, hy, hp3);
        });
        load_bias2(small + S_BIAS + 0 * HID, h, Y.acc);
#pragma unroll
        for (int c = 0; c < 3; ++c) Y.raw[c] = (hy[c] + __shfl_xor(hy[c], 32)) + small[S_HEAD_B + 1 + c];
        mfma_block<24>(cur, lane, Y.acc, fb(Y.D), [&](int g, int j) {
            relu_side<12>(g, j, X.acc, X.act);
            bias_side<12, 2>(g, j, small + S_BIAS + 1 * HID, h, X.acc, bp);
        });
        cur = nxt;
        ring2_sync();
        nxt = ring2_issue<64>(rs, P_DEN1 + P_HID_FLOATS);
        mfma_block<64>(cur, lane, X.acc, hb(X.act), [&](int g, int j) { relu_side<32>(g, j, Y.acc, Y.act); bias_side<32>(g, j, small + S_BIAS + 1 * HID, h, Y.acc, bp); });
        mfma_block<64>(cur, lane, Y.acc, hb(Y.act), [&](int g, int j) { relu_side<32>(g, j, X.acc, X.act); bias_side<32>(g, j, small + S_BIAS + 2 * HID, h, X.acc, bp); });
        cur = nxt;
        ring2_sync();
        nxt = ring2_issue<64>(rs, P_DEN1 + 2 * P_HID_FLOATS);
        mfma_block<64>(cur, lane, X.acc, hb(X.act), [&](int g, int j) { relu_side<32>(g, j, Y.acc, Y.act); bias_side<32>(g, j, small + S_BIAS + 2 * HID, h, Y.acc, bp); });
        mfma_block<64>(cur, lane, Y.acc, hb(Y.act), [&](int g, int j) { relu_side<32>(g, j, X.acc, X.act); bias_side<32>(g, j, small + S_BIAS + 3 * HID, h, X.acc, bp); });
        cur = nxt;
        ring2_sync();
        nxt = ring2_issue<48>(rs, P_RGB0);     // C0 of the NEXT sample into the slot every wave has just left (after the last sample: a harmless copy)
        mfma_block<64>(cur, lane, X.acc, hb(X.act), [&](int g, int j) { relu_side<32>(g, j, Y.acc, Y.act); bias_side<32>(g, j, small + S_BIAS + 3 * HID, h, Y.acc, bp); });
        // Y density layer 3 | X: ReLU, sigma head
        float sx[1] = {0.0f};
        mfma_block<64>(cur, lane, Y.acc, hb(Y.act), [&](int g, int j) {
            relu_side<32>(g, j, X.acc, X.act);
            if (g >= 32) heads_side<1>(g - 32, j, small + S_ALPHA_W, h, X.act, sx, hp1);
        });
        X.raw[3] = (sx[0] + __shfl_xor(sx[0], 32)) + small[S_HEAD_B];
        cur = nxt;
        // ---- epilogue (exposed): Y's ReLU + sigma head, both tiles' compositing -----------------------------------------------
#if R2_ABLATE & 2
        Y.raw[3] = Y.acc[0][0];
#else
#pragma unroll
        for (int g = 0; g < 32; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) relu_side<32>(g, j, Y.acc, Y.act);
        {
            float hd[1];
            head_dots<1>(small + S_ALPHA_W, h, Y.act, hd);      // (weights streamed with a pinned prefetch: nothing hides an LDS round trip here)
            Y.raw[3] = hd[0] + small[S_HEAD_B];
        }
#endif
        if (raw_out && lane < 32) {
            if (validX) *reinterpret_cast<f32x4*>(raw_out + (rayX * S + s) * 4) = f32x4{X.raw[0], X.raw[1], X.raw[2], X.raw[3]};
            if (validY) *reinterpret_cast<f32x4*>(raw_out + (rayY * S + s) * 4) = f32x4{Y.raw[0], Y.raw[1], Y.raw[2], Y.raw[3]};
        }
#if R2_ABLATE & 1
        X.cr += X.raw[0] + X.raw[3]; Y.cr += Y.raw[0] + Y.raw[3];
#else
        composite_sample(X, reinterpret_cast<const f32x4*>(rcX)[1][2], nzX, last);
        composite_sample(Y, reinterpret_cast<const f32x4*>(rcY)[1][2], nzY, last);
#endif
        if (weights && lane < 32) {
            if (validX) weights[rayX * S + s] = X.raw[3];
            if (validY) weights[rayY * S + s] = Y.raw[3];
        }
        X.zc = X.zn; Y.zc = Y.zn;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the copy issued for a sample after the last one must land before the wave ends

    if (rs.lane < 32) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const Tile& t = k ? Y : X;
            const long ray = k ? rayY : rayX;
            if (!(k ? validY : validX)) continue;
            float cr = t.cr, cg = t.cg, cb = t.cb;
            const float q = t.dep / t.ac;                       // NaN when acc == 0, like torch.max(1e-10, nan)
            disp[ray] = 1.0f / ((q != q) ? q : fmaxf(1e-10f, q));
            if (white) { const float bg = 1.0f - t.ac; cr += bg; cg += bg; cb += bg; }
            rgb[ray * 3 + 0] = cr; rgb[ray * 3 + 1] = cg; rgb[ray * 3 + 2] = cb;
            acc[ray] = t.ac;
            if (depth) depth[ray] = t.dep;
        }
    }
}

}  // namespace nvsr

using namespace nvsr;

extern "C" int nvsr_render_pass2_launch(const nvsr_scene* scene, const float* packed_decoder, int64_t N, int S, const float* rays,
                                        const float* z, const float* noise, int white_bkgd, float* rgb, float* disp, float* acc,
                                        float* weights, float* depth, float* raw_out, nvsr_stream_t stream) {
    const int64_t grid = (N + RAYS2 - 1) / RAYS2;
    if (grid > 0x7fffffff) return NVSR_ERR_SHAPE;
    hipLaunchKernelGGL(render_pass2_kernel, dim3((unsigned)grid), dim3(TPB2), 0, (hipStream_t)stream, to_dev(scene), packed_decoder, (long)N,
                       S, rays, z, noise, white_bkgd, rgb, disp, acc, weights, depth, raw_out);
    return NVSR_CHECK_LAUNCH();
}
