// Backward of the fused render pass with respect to the feature planes (gfx950).
//
// The reference obtains it from torch.autograd through run_network / TwoDimPlanesModel.forward / grid_sample
// (train_utils.py:185-282, models.py:381-421; `grid_sampler_2d_backward` scatter-adds into the planes).  With RECORD the kernel
// also writes every layer's input and pre-activation gradient to a workspace; decoder_wgrad.hip contracts those over the points
// into the weight / bias gradients (`what: ['decoder']`, train_nerf.py:75-77).
//
// Per step a wave takes 32 points (one sample of its 32 rays):
//   1. recomputes the decoder forward on the MFMA path of decode_core.h, keeping only the ReLU masks (2 VGPRs per layer);
//   2. chains dL/d(raw) back through the transposed layers -- the same register-chained v_mfma_f32_32x32x2_f32 scheme, fed with
//      W^T fragments (second packed blob): D[in-feature][point] += W^T[in][out] * G[out][point];
//   3. transposes the feature gradients of one plane at a time through a per-wave LDS tile ([point][48 channels]) and adds
//      them into the channel-last gradient plane with one global_atomic_add_f32 wave-instruction per (point, tap): 48 lanes =
//      192 contiguous bytes, the fast shape of float atomics (one-lane-per-row scatter is ~17x slower, MI355X_MICROARCH.md).
#include "bwd_core.h"
#include "wave_scan.h"

namespace nvsr {

// 4 waves = ONE per SIMD: the step keeps ~430 registers live (two accumulator sets, features, 8 ReLU masks, the density branch's
// input gradient while the rgb branch runs); at two waves per SIMD (256 registers) 160-190 of them spilled, and every spill reload is
// a vector-memory access that queues behind the weight DMA in flight
constexpr int BTPB = 256;
constexpr int BNW = BTPB / 64;
constexpr int BPTS = BNW * 32;
constexpr int RAYB_FLOATS = 16;
constexpr int BWD_LDS_FLOATS = LDS_FLOATS + BPTS * RAYB_FLOATS + BNW * TILE_FLOATS;   // (the RAYB region is unused padding now)
static_assert(BWD_LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");

__global__ void pack_decoder_bwd_kernel(const float* __restrict__ nat, float* __restrict__ packed) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B_TOTAL) return;
    const int j = idx & 3, lane = (idx >> 2) & 63, h = lane >> 5;
    float v = 0.0f;
    if (idx < B_DEN0 || (idx >= B_RGB_H && idx < B_RGB0)) {          // hidden^T
        const bool rgb = idx >= B_RGB_H;
        const int rem0 = idx - (rgb ? B_RGB_H : B_DEN_H);
        const int li = rem0 / P_HID_FLOATS, rem = rem0 % P_HID_FLOATS;   // li 0 -> layer 3, 1 -> layer 2, 2 -> layer 1
        const int ib = (rem >> 8) & 3, q = (rem >> 10) & 3, kb = rem >> 12;
        const int out = 32 * kb + 8 * q + 4 * h + j, in = 32 * ib + (lane & 31);
        const int layer = 3 - li;                                          // forward layer index 1..3
        v = nat[(rgb ? N_RGB_W1 : N_DEN_W1) + (layer - 1) * N_HID_STRIDE + out * HID + in];
    } else {                                                               // layer-0^T
        const bool rgb = idx >= B_RGB0;
        const int rem0 = idx - (rgb ? B_RGB0 : B_DEN0);
        const int p = rem0 / 8192, rem = rem0 % 8192;
        const int ib = (rem >> 8) & 1, q = (rem >> 9) & 3, kb = rem >> 11;
        const int out = 32 * kb + 8 * q + 4 * h + j, c = 32 * ib + (lane & 31);
        if (c < C) v = rgb ? nat[N_RGB_W0 + out * (4 * C) + C * p + c] : nat[N_DEN_W0 + out * C + c];
    }
    packed[idx] = v;
}

// ---- small helpers ---------------------------------------------------------------------------------------------------------

__device__ __forceinline__ Masks relu_masks(f32x16 (&acc)[4]) {
    Masks k{{0u, 0u}};
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool on = acc[ib][r] > 0.0f;
            acc[ib][r] = on ? acc[ib][r] : 0.0f;
            k.m[ib >> 1] |= on ? (1u << gate_bit(ib, r)) : 0u;
        }
    return k;
}
__device__ __forceinline__ void zero_acc(f32x16 (&a)[4]) {
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int r = 0; r < 16; ++r) a[ib][r] = 0.0f;
}

// layer-0^T of one plane: acc2[ib] += W0p^T[32ib.., :] * g     (chunk [kb][q][ib 2][lane][j], 128 MFMAs)
__device__ __forceinline__ void layer0_T(const float* wl, const f32x16 (&g)[4], int lane, f32x16 (&acc2)[2]) {
    const f32x4* wv = reinterpret_cast<const f32x4*>(wl) + lane;
    f32x4 a = wv[0];
#pragma unroll
    for (int gi = 0; gi < 32; ++gi) {
        const int ib = gi & 1, q = (gi >> 1) & 3, kb = gi >> 3;
        acc2[ib] = mfma32(a[0], g[kb][4 * q + 0], acc2[ib]);
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 an = wv[(gi + 1 < 32 ? gi + 1 : gi) * 64];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 1; j < 4; ++j) acc2[ib] = mfma32(a[j], g[kb][4 * q + j], acc2[ib]);
        a = an;
        __builtin_amdgcn_sched_barrier(0);
    }
}

// one full hidden^T layer (two ring chunks): gn = mask .* (W^T g)
template <int NW>
__device__ __forceinline__ void hidden_T(RingState& rs, const float*& cur, int next_off, int next_blocks24, const f32x16 (&g)[4],
                                         const Masks& mk, f32x16 (&gn)[4], int this_off) {
    // chunk h0 is `cur` (already issued); issue h1, compute h0, then issue the caller's next chunk, compute h1
    ring_sync();
    const float* nxt = ring_issue<NW, 32>(rs, this_off + P_HID_FLOATS / 2);
    zero_acc(gn);
    hidden_half<0>(cur, g, rs.lane, gn);
    cur = nxt;
    ring_sync();
    nxt = next_blocks24 ? ring_issue<NW, 24>(rs, next_off) : ring_issue<NW, 32>(rs, next_off);
    hidden_half<1>(cur, g, rs.lane, gn);
    apply_mask(mk, gn);
    cur = nxt;
}

// =====================================================================================================================
template <bool RECORD>
__global__ __launch_bounds__(BTPB, 1) void render_pass_backward_kernel(SceneDev sc, const float* __restrict__ packed,
                                                                      const float* __restrict__ packed_bwd, long N, int S,
                                                                      const float* __restrict__ rays, const float* __restrict__ z,
                                                                      const float* __restrict__ g_raw, GradPlanes gp, DecRecord rec,
                                                                      float* __restrict__ gview) {
    __shared__ __attribute__((aligned(16))) float lds[BWD_LDS_FLOATS];
    NVSR_RACE_PROBE_DELAY(lds);      // (probe builds only, nvsr_common.h)
    RingState rs{packed, lds, 0, (int)(threadIdx.x >> 6), (int)(threadIdx.x & 63), (threadIdx.x >> 6) * 1024u + (threadIdx.x & 63) * 16u};
    decode_prologue<BNW>(rs);
    __syncthreads();          // (the transposed alpha head reads these LDS words in front of the first ring barrier: render_bwd_limb.hip)
    float* tile = lds + LDS_FLOATS + BPTS * RAYB_FLOATS + rs.wave * TILE_FLOATS;
    const float* small = lds + 2 * SLOT_FLOATS;
    constexpr int HH = P_HID_FLOATS / 2;
    const long nrb = (N + BPTS - 1) / BPTS;
    const long ntiles = nrb * S;            // tiles = (ray block, sample): every sample of every ray is independent here

    for (long tix = blockIdx.x; tix < ntiles; tix += gridDim.x) {
        asm volatile("" : "+v"(rs.voff), "+v"(rs.lane));
        const int lane = rs.lane, h = lane >> 5;
        const long rb = tix / S;
        const int s = (int)(tix - rb * S);
        const long ray0 = rb * BPTS + rs.wave * 32 + (lane & 31);
        const bool valid = ray0 < N;
        const long ray = valid ? ray0 : N - 1;
        const float* r = rays + ray * 11;
        const Taps vt = view_taps(sc, r[8], r[9], r[10]);
        const float zc = z[ray * S + s];
        f32x4 graw = *reinterpret_cast<const f32x4*>(g_raw + (ray * S + s) * 4);
        if (!valid) graw = f32x4{0.0f, 0.0f, 0.0f, 0.0f};             // padding rays: every gradient below becomes an exact zero
        const long q = record_row(ray, s, N, S);                             // record row of this point
        const bool rok = RECORD && valid;                             // padding rays write nothing
        if (rok && h == 0) *reinterpret_cast<f32x4*>(rec.g4 + 4 * q) = graw;
        const f32x4 c0 = f32x4{r[0], r[1], r[2], r[3]}, c1 = f32x4{r[4], r[5], 0.0f, 0.0f};
        const float n0 = norm_coord(__fadd_rn(c0[0], __fmul_rn(c0[3], zc)), sc.lo[0], sc.range[0]);
        const float n1 = norm_coord(__fadd_rn(c0[1], __fmul_rn(c1[0], zc)), sc.lo[1], sc.range[1]);
        const float n2 = norm_coord(__fadd_rn(c0[2], __fmul_rn(c1[1], zc)), sc.lo[2], sc.range[2]);
        auto pos_taps = [&](int d) {
            const float* M = sc.proj + 6 * d;
            return make_taps(sc, d, n0 * M[0] + n1 * M[2] + n2 * M[4], n0 * M[1] + n1 * M[3] + n2 * M[5]);
        };

        // ================= forward recompute, ReLU masks only =================
        f32x16 accA[4], accB[4];
        float D[HALF_C], F[HALF_C];
        Masks mr[4], md[4];
        const float* cur = ring_issue<BNW, 24>(rs, P_RGB0);
        gather24(sc.plane[0], pos_taps(0), h, F);
        ring_sync();
        const float* nxt = ring_issue<BNW, 24>(rs, P_RGB0 + P_PLANE_FLOATS);
        load_bias(small + S_BIAS + 4 * HID, h, accA);
#pragma unroll
        for (int c = 0; c < HALF_C; ++c) D[c] = F[c];
        if (rok) record24(rec.Xr + q * (4 * C), h, F);
        feat_layer(cur, F, lane, accA);
        cur = nxt;
        ring_sync();
        nxt = ring_issue<BNW, 24>(rs, P_RGB0 + 2 * P_PLANE_FLOATS);
        gather24(sc.plane[1], pos_taps(1), h, F);
#pragma unroll
        for (int c = 0; c < HALF_C; ++c) D[c] = __fadd_rn(D[c], F[c]);
        if (rok) record24(rec.Xr + q * (4 * C) + C, h, F);
        feat_layer(cur, F, lane, accA);
        cur = nxt;
        ring_sync();
        nxt = ring_issue<BNW, 24>(rs, P_RGB0 + 3 * P_PLANE_FLOATS);
        gather24(sc.plane[2], pos_taps(2), h, F);
#pragma unroll
        for (int c = 0; c < HALF_C; ++c) D[c] = div3(__fadd_rn(D[c], F[c]));
        if (rok) {
            record24(rec.Xr + q * (4 * C) + 2 * C, h, F);
            record24(rec.Xd + q * 64, h, D);
            *reinterpret_cast<f32x4*>(rec.Xd + q * 64 + C + 8 * h) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            *reinterpret_cast<f32x4*>(rec.Xd + q * 64 + C + 8 * h + 4) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        feat_layer(cur, F, lane, accA);
        cur = nxt;
        ring_sync();
        nxt = ring_issue<BNW, 32>(rs, P_RGB1);
        gather24(sc.plane[3], vt, h, F);
        if (rok) record24(rec.Xr + q * (4 * C) + 3 * C, h, F);
        feat_layer(cur, F, lane, accA);
        mr[0] = relu_masks(accA);
        if (rok) record128(rec.Hr, q, h, accA);
        cur = nxt;
#pragma unroll
        for (int l = 1; l <= 3; ++l) {                 // rgb layers 1..3 (ping-pong A -> B -> A -> B)
            f32x16 (&in)[4] = (l & 1) ? accA : accB;
            f32x16 (&out)[4] = (l & 1) ? accB : accA;
            ring_sync();
            nxt = ring_issue<BNW, 32>(rs, P_RGB1 + (l - 1) * P_HID_FLOATS + HH);
            load_bias(small + S_BIAS + (4 + l) * HID, h, out);
            hidden_half<0>(cur, in, lane, out);
            cur = nxt;
            ring_sync();
            nxt = (l < 3) ? ring_issue<BNW, 32>(rs, P_RGB1 + l * P_HID_FLOATS) : ring_issue<BNW, 24>(rs, P_DEN0);
            hidden_half<1>(cur, in, lane, out);
            mr[l] = relu_masks(out);
            if (rok) record128(rec.Hr + (long)l * HID * rec.Pp, q, h, out);
            cur = nxt;
        }
        ring_sync();
        nxt = ring_issue<BNW, 32>(rs, P_DEN1);
        load_bias(small + S_BIAS + 0 * HID, h, accA);
        feat_layer(cur, D, lane, accA);
        md[0] = relu_masks(accA);
        if (rok) record128(rec.Hd, q, h, accA);
        cur = nxt;
#pragma unroll
        for (int l = 1; l <= 3; ++l) {                 // density layers 1..3
            f32x16 (&in)[4] = (l & 1) ? accA : accB;
            f32x16 (&out)[4] = (l & 1) ? accB : accA;
            ring_sync();
            nxt = ring_issue<BNW, 32>(rs, P_DEN1 + (l - 1) * P_HID_FLOATS + HH);
            load_bias(small + S_BIAS + l * HID, h, out);
            hidden_half<0>(cur, in, lane, out);
            cur = nxt;
            ring_sync();
            if (l < 3) nxt = ring_issue<BNW, 32>(rs, P_DEN1 + l * P_HID_FLOATS);
            hidden_half<1>(cur, in, lane, out);
            md[l] = relu_masks(out);
            if (rok) record128(rec.Hd + (long)l * HID * rec.Pp, q, h, out);
            cur = nxt;
        }

        // ================= backward: density branch -> gD (rows = 48 channels, 2 blocks) =================
        rs.packed = packed_bwd;
        cur = ring_issue<BNW, 32>(rs, B_DEN_H);
        // d raw_sigma / d h3 = fc_alpha weight, masked by ReLU(h3)
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(small + S_ALPHA_W + (ib * 4 + q) * 8 + h * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) accA[ib][4 * q + j] = wv[j] * graw[3];
            }
        apply_mask(md[3], accA);
        if (rok) record128(rec.Gd + 3L * HID * rec.Pp, q, h, accA);
        hidden_T<BNW>(rs, cur, B_DEN_H + P_HID_FLOATS, 0, accA, md[2], accB, B_DEN_H);
        if (rok) record128(rec.Gd + 2L * HID * rec.Pp, q, h, accB);
        hidden_T<BNW>(rs, cur, B_DEN_H + 2 * P_HID_FLOATS, 0, accB, md[1], accA, B_DEN_H + P_HID_FLOATS);
        if (rok) record128(rec.Gd + 1L * HID * rec.Pp, q, h, accA);
        hidden_T<BNW>(rs, cur, B_DEN0, 0, accA, md[0], accB, B_DEN_H + 2 * P_HID_FLOATS);
        if (rok) record128(rec.Gd, q, h, accB);
        f32x16 gD[2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) gD[b][r] = 0.0f;
        ring_sync();
        nxt = ring_issue<BNW, 32>(rs, B_RGB_H);
        layer0_T(cur, accB, lane, gD);
        cur = nxt;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) gD[b][r] = div3(gD[b][r]);      // 'avg' combination: each position plane gets gD / 3

        // ================= backward: rgb branch =================
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int o = (ib * 4 + q) * 8 + h * 4;
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(small + S_RGB_W + o);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(small + S_RGB_W + HID + o);
                const f32x4 w2 = *reinterpret_cast<const f32x4*>(small + S_RGB_W + 2 * HID + o);
#pragma unroll
                for (int j = 0; j < 4; ++j) accA[ib][4 * q + j] = fmaf(w2[j], graw[2], fmaf(w1[j], graw[1], w0[j] * graw[0]));
            }
        apply_mask(mr[3], accA);
        if (rok) record128(rec.Gr + 3L * HID * rec.Pp, q, h, accA);
        hidden_T<BNW>(rs, cur, B_RGB_H + P_HID_FLOATS, 0, accA, mr[2], accB, B_RGB_H);
        if (rok) record128(rec.Gr + 2L * HID * rec.Pp, q, h, accB);
        hidden_T<BNW>(rs, cur, B_RGB_H + 2 * P_HID_FLOATS, 0, accB, mr[1], accA, B_RGB_H + P_HID_FLOATS);
        if (rok) record128(rec.Gr + 1L * HID * rec.Pp, q, h, accA);
        hidden_T<BNW>(rs, cur, B_RGB0, 0, accA, mr[0], accB, B_RGB_H + 2 * P_HID_FLOATS);
        if (rok) record128(rec.Gr, q, h, accB);
        // layer 0^T, one plane at a time, + gD/3 on the position planes, then scatter
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            f32x16 gF[2];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) gF[b][r] = (d < 3) ? gD[b][r] : 0.0f;
            ring_sync();
            if (d < 3) nxt = ring_issue<BNW, 32>(rs, B_RGB0 + (d + 1) * 8192);
            layer0_T(cur, accB, lane, gF);
            cur = nxt;
            if (gp.p[d]) {                                            // (wave-uniform) planes frozen: nothing to scatter
                if (d == 3 && gview) {
                    store_view_rows(gF, tile, gview, rb * BPTS + rs.wave * 32, N, S, s, lane);
                } else {
                    const Taps t = (d < 3) ? pos_taps(d) : vt;
                    scatter_plane(gF, tile, t, gp.p[d], lane, valid);
                }
            }
        }
        rs.packed = packed;
    }
    ring_sync();
}

// =====================================================================================================================
// Mask-driven variant: the training forward (decode_rays_kernel<true>) has already published every layer's ReLU gate, so this
// kernel runs the transposed layers only (2 176 MFMAs per tile instead of 4 192) and its live state -- two accumulator sets, 16
// gate words, the density branch's input gradient -- fits 256 registers: 8-wave workgroups, two waves per SIMD.
// Used when the decoder is frozen (what: ['LR_planes'], Feature_Planes_Only.yml); decoder gradients need the record above.
// =====================================================================================================================
// Waves per workgroup: 8 without the record (two waves per SIMD, 256 registers each); with the record the kernel also carries the record's row
// addresses and spilled 23 registers at 256 (rounds 1-4) -- that instantiation runs 4 waves (one per SIMD, up to 512 registers, no scratch).
template <bool RECORD> struct MBwd {
    static constexpr int NW = RECORD ? 4 : 8, TPB = NW * 64, PTS = NW * 32;
    static constexpr int LDS = LDS_FLOATS + NW * TILE_FLOATS;
};
static_assert(MBwd<false>::LDS * 4 <= 160 * 1024, "LDS budget");

template <bool RECORD>
__global__ __launch_bounds__(MBwd<RECORD>::TPB, 1) void render_pass_backward_gates_kernel(SceneDev sc, const float* __restrict__ packed,
                                                                            const float* __restrict__ packed_bwd, long N, int S,
                                                                            const float* __restrict__ rays, const float* __restrict__ z,
                                                                            const float* __restrict__ g_raw,
                                                                            const unsigned* __restrict__ gates, GradPlanes gp,
                                                                            float* __restrict__ gview, DecRecord rec) {
    constexpr int MNW = MBwd<RECORD>::NW, MPTS = MBwd<RECORD>::PTS;
    __shared__ __attribute__((aligned(16))) float lds[MBwd<RECORD>::LDS];
    NVSR_RACE_PROBE_DELAY(lds);      // (probe builds only, nvsr_common.h)
    RingState rs{packed, lds, 0, (int)(threadIdx.x >> 6), (int)(threadIdx.x & 63), (threadIdx.x >> 6) * 1024u + (threadIdx.x & 63) * 16u};
    decode_prologue<MNW>(rs);                    // head weights / biases of the FORWARD blob -> LDS
    __syncthreads();          // (the transposed alpha head reads these LDS words in front of the first ring barrier: render_bwd_limb.hip)
    rs.packed = packed_bwd;
    float* tile = lds + LDS_FLOATS + rs.wave * TILE_FLOATS;
    const float* small = lds + 2 * SLOT_FLOATS;
    const long nrb = (N + MPTS - 1) / MPTS;
    const long ntiles = nrb * S;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

    for (long tix = blockIdx.x; tix < ntiles; tix += gridDim.x) {
        asm volatile("" : "+v"(rs.voff), "+v"(rs.lane));
        const int lane = rs.lane, h = lane >> 5;
        const long rb = tix / S;
        const int s = (int)(tix - rb * S);
        const long ray0 = rb * MPTS + rs.wave * 32 + (lane & 31);
        const bool valid = ray0 < N;
        const long ray = valid ? ray0 : N - 1;
        const float* cur = ring_issue<MNW, 32>(rs, B_DEN_H);
        const float* r = rays + ray * 11;
        const float zc = z[ray * S + s];
        f32x4 graw = *reinterpret_cast<const f32x4*>(g_raw + (ray * S + s) * 4);
        if (!valid) graw = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        const long q = record_row(ray, s, N, S);                             // record row (the forward wrote X / H of the same row)
        const bool rok = RECORD && valid;
        if (rok && h == 0) *reinterpret_cast<f32x4*>(rec.g4 + 4 * q) = graw;
        // (the row index goes through an opaque asm at every use: computed once, hipcc keeps the eight 64-bit row addresses of the record live
        //  across all transposed layers -- 16 registers this kernel does not have)
        auto rec_row = [&](float* base, const f32x16 (&a)[4]) {
            int ql = (int)q;
            asm volatile("" : "+v"(ql));
            record128(base, (long)ql, h, a);
        };
        // the eight gate word pairs of the point (slot 0..3 density layers, 4..7 rgb layers).  Without the record they are loaded up front (16
        // registers); WITH it the kernel also holds the record's row pointers and those 16 registers were what spilled (23 VGPRs, 96 B of
        // scratch in rounds 1-4): a layer's pair is then re-read where the layer uses it (an L1 / L2 hit)
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const u32x2* gk2 = reinterpret_cast<const u32x2*>(gates + ((ray * S + s) * 2 + h) * 16);
        u32x2 gw[RECORD ? 1 : 8];
        if constexpr (!RECORD) {
            const u32x4* gk = reinterpret_cast<const u32x4*>(gk2);
            const u32x4 k0 = gk[0], k1 = gk[1], k2 = gk[2], k3 = gk[3];
            gw[0] = u32x2{k0[0], k0[1]}; gw[1] = u32x2{k0[2], k0[3]}; gw[2] = u32x2{k1[0], k1[1]}; gw[3] = u32x2{k1[2], k1[3]};
            gw[4] = u32x2{k2[0], k2[1]}; gw[5] = u32x2{k2[2], k2[3]}; gw[6] = u32x2{k3[0], k3[1]}; gw[7] = u32x2{k3[2], k3[3]};
        }
        auto gate = [&](int slot) {
            if constexpr (RECORD) { const u32x2 v = gk2[slot]; return Masks{{v[0], v[1]}}; }
            else return Masks{{gw[slot][0], gw[slot][1]}};
        };
        // (taps are computed where a plane is scattered, from the ray re-read there: nothing of them stays live across the transposed layers)
        auto pos_taps = [&](int d) {
            const float n0 = norm_coord(__fadd_rn(r[0], __fmul_rn(r[3], zc)), sc.lo[0], sc.range[0]);
            const float n1 = norm_coord(__fadd_rn(r[1], __fmul_rn(r[4], zc)), sc.lo[1], sc.range[1]);
            const float n2 = norm_coord(__fadd_rn(r[2], __fmul_rn(r[5], zc)), sc.lo[2], sc.range[2]);
            const float* M = sc.proj + 6 * d;
            return make_taps(sc, d, n0 * M[0] + n1 * M[2] + n2 * M[4], n0 * M[1] + n1 * M[3] + n2 * M[5]);
        };
        f32x16 accA[4], accB[4];
        const float* nxt;
        // ---- density branch -> gD
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(small + S_ALPHA_W + (ib * 4 + q) * 8 + h * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) accA[ib][4 * q + j] = wv[j] * graw[3];
            }
        apply_mask(gate(3), accA);
        if (rok) rec_row(rec.Gd + 3L * HID * rec.Pp, accA);
        hidden_T<MNW>(rs, cur, B_DEN_H + P_HID_FLOATS, 0, accA, gate(2), accB, B_DEN_H);
        if (rok) rec_row(rec.Gd + 2L * HID * rec.Pp, accB);
        hidden_T<MNW>(rs, cur, B_DEN_H + 2 * P_HID_FLOATS, 0, accB, gate(1), accA, B_DEN_H + P_HID_FLOATS);
        if (rok) rec_row(rec.Gd + 1L * HID * rec.Pp, accA);
        hidden_T<MNW>(rs, cur, B_DEN0, 0, accA, gate(0), accB, B_DEN_H + 2 * P_HID_FLOATS);
        if (rok) rec_row(rec.Gd, accB);
        f32x16 gD[2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) gD[b][rr] = 0.0f;
        ring_sync();
        nxt = ring_issue<MNW, 32>(rs, B_RGB_H);
        layer0_T(cur, accB, lane, gD);
        cur = nxt;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) gD[b][rr] = div3(gD[b][rr]);
        // ---- rgb branch
        if constexpr (RECORD) {       // (dL/draw re-read instead of kept live across the density branch: see the gate words above)
            asm volatile("" ::: "memory");
            graw = *reinterpret_cast<const f32x4*>(g_raw + (ray * S + s) * 4);
            if (!valid) graw = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int o = (ib * 4 + q) * 8 + h * 4;
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(small + S_RGB_W + o);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(small + S_RGB_W + HID + o);
                const f32x4 w2 = *reinterpret_cast<const f32x4*>(small + S_RGB_W + 2 * HID + o);
#pragma unroll
                for (int j = 0; j < 4; ++j) accA[ib][4 * q + j] = fmaf(w2[j], graw[2], fmaf(w1[j], graw[1], w0[j] * graw[0]));
            }
        apply_mask(gate(7), accA);
        if (rok) rec_row(rec.Gr + 3L * HID * rec.Pp, accA);
        hidden_T<MNW>(rs, cur, B_RGB_H + P_HID_FLOATS, 0, accA, gate(6), accB, B_RGB_H);
        if (rok) rec_row(rec.Gr + 2L * HID * rec.Pp, accB);
        hidden_T<MNW>(rs, cur, B_RGB_H + 2 * P_HID_FLOATS, 0, accB, gate(5), accA, B_RGB_H + P_HID_FLOATS);
        if (rok) rec_row(rec.Gr + 1L * HID * rec.Pp, accA);
        hidden_T<MNW>(rs, cur, B_RGB0, 0, accA, gate(4), accB, B_RGB_H + 2 * P_HID_FLOATS);
        if (rok) rec_row(rec.Gr, accB);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            f32x16 gF[2];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) gF[b][rr] = (d < 3) ? gD[b][rr] : 0.0f;
            ring_sync();
            if (d < 3) nxt = ring_issue<MNW, 32>(rs, B_RGB0 + (d + 1) * 8192);
            layer0_T(cur, accB, lane, gF);
            cur = nxt;
            if (gp.p[d]) {
                if (d == 3 && gview) {
                    store_view_rows(gF, tile, gview, rb * MPTS + rs.wave * 32, N, S, s, lane);
                } else {
                    const Taps t = (d < 3) ? pos_taps(d) : view_taps(sc, r[8], r[9], r[10]);
                    scatter_plane(gF, tile, t, gp.p[d], lane, valid);
                }
            }
        }
    }
    ring_sync();
}

// one wave per ray: sum the ray's S gradient rows, then 4 taps x 48 channels of atomics into the view-direction plane
__global__ __launch_bounds__(256) void view_reduce_scatter_kernel(SceneDev sc, long N, int S, const float* __restrict__ rays,
                                                                  const float* __restrict__ gview, float* __restrict__ gplane) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long ray = (long)blockIdx.x * 4 + wave;
    if (ray >= N) return;
    const float* r = rays + ray * 11;
    const Taps t = view_taps(sc, r[8], r[9], r[10]);
    if (lane >= C) return;
    const float* row = gview + ray * S * C + lane;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    int s = 0;
    for (; s + 4 <= S; s += 4) { a0 += row[s * C]; a1 += row[(s + 1) * C]; a2 += row[(s + 2) * C]; a3 += row[(s + 3) * C]; }
    for (; s < S; ++s) a0 += row[s * C];
    const float v = (a0 + a1) + (a2 + a3);
    unsafeAtomicAdd(gplane + t.o00 + lane, v * t.nw);
    unsafeAtomicAdd(gplane + t.o01 + lane, v * t.ne);
    unsafeAtomicAdd(gplane + t.o10 + lane, v * t.sw);
    unsafeAtomicAdd(gplane + t.o11 + lane, v * t.se);
}

// =====================================================================================================================
// volume_render_radiance_field backward: (g_rgb [N,3], g_acc [N] or NULL, g_dep [N] or NULL) -> g_raw [N,S,4]; one wave per ray, S <= 512
// (g_dep = gradient of depth_map = sum_s w_s z_s, volume_rendering_utils.py:42-43; disp_map's gradient reaches this kernel through g_dep / g_acc)
// =====================================================================================================================
constexpr int CB_WPB = 4;
__global__ __launch_bounds__(CB_WPB * 64) void composite_backward_kernel(long N, int S, const float* __restrict__ raw, const float* __restrict__ z,
                                                                        const float* __restrict__ rd, const float* __restrict__ noise, int white,
                                                                        const float* __restrict__ g_rgb, const float* __restrict__ g_acc,
                                                                        const float* __restrict__ g_dep, float* __restrict__ g_raw, int mip,
                                                                        int rd_stride, int rd_off) {
    // (rd_stride, rd_off): 3, 0 = ray directions [N,3]; 11, 3 = the directions inside packed rays [N,11] (nvsr_composite_backward_rays)
    __shared__ float sT[CB_WPB][512], sA[CB_WPB][512], sG[CB_WPB][512];   // T_s, alpha_s, dL/dw_s
    const int zp = S + (mip ? 1 : 0);                                      // mip: z holds S + 1 interval edges, no 1e10 tail
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long ray = (long)blockIdx.x * CB_WPB + wave;
    if (ray >= N) return;
    const float* rdr = rd + ray * rd_stride + rd_off;
    const float d0 = rdr[0], d1 = rdr[1], d2 = rdr[2];
    const float nrm = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
    const float g0 = g_rgb[ray * 3], g1 = g_rgb[ray * 3 + 1], g2 = g_rgb[ray * 3 + 2];
    const float ga = (g_acc ? g_acc[ray] : 0.0f) - (white ? (g0 + g1 + g2) : 0.0f);
    const float gd = g_dep ? g_dep[ray] : 0.0f;
    const f32x4* rr = reinterpret_cast<const f32x4*>(raw) + ray * S;
    f32x4* gout = reinterpret_cast<f32x4*>(g_raw) + ray * S;
    // forward sweep: T_s (exclusive running product), alpha_s, dL/dw_s
    float Tcarry = 1.0f;
    for (int base = 0; base < S; base += 64) {
        const int s = base + lane;
        float fac = 1.0f, alpha = 0.0f, gw = 0.0f;
        if (s < S) {
            const f32x4 rv = rr[s];
            const float dist = ((mip || s + 1 < S) ? (z[ray * zp + s + 1] - z[ray * zp + s]) : 1e10f) * nrm;
            const float sig = fmaxf(rv[3] + (noise ? noise[ray * S + s] : 0.0f), 0.0f);
            alpha = 1.0f - expf(-sig * dist);
            fac = (1.0f - alpha) + 1e-10f;
            gw = g0 / (1.0f + expf(-rv[0])) + g1 / (1.0f + expf(-rv[1])) + g2 / (1.0f + expf(-rv[2])) + ga;
            if (g_dep) gw += gd * (mip ? 0.5f * (z[ray * zp + s] + z[ray * zp + s + 1]) : z[ray * zp + s]);      // d depth_map / d w_s = the sample's depth
        }
        // (the forward compositor's own scan and carry, aux.hip composite_kernel: the transmittances recomputed here are the forward's bit for bit
        //  -- rounds 1-5 ran a Hillis-Steele order over __shfl_up here, which since round 5's DPP scans differed from the forward's in the last ulp)
        const float incl = wave_scan_mul(fac, lane);
        float excl = __shfl_up(incl, 1);
        if (lane == 0) excl = 1.0f;
        const float T = __fmul_rn(excl, Tcarry);
        Tcarry = __fmul_rn(__shfl(incl, 63), Tcarry);
        if (s < S) { sT[wave][s] = T; sA[wave][s] = alpha; sG[wave][s] = gw; }
    }
    __builtin_amdgcn_wave_barrier();
    // second sweep, back to front: suffix_s = sum_{k>s} w_k dL/dw_k by a REVERSE scan (total - prefix would cancel, and the
    // suffix is divided by (1 - alpha + 1e-10), which reaches 1e-10 behind an opaque sample)
    float scarry = 0.0f;
    const int nchunks = (S + 63) / 64;
    for (int ch = nchunks - 1; ch >= 0; --ch) {
        const int s = ch * 64 + lane;
        float T = 0.0f, alpha = 0.0f, gw = 0.0f;
        f32x4 rv = {0.0f, 0.0f, 0.0f, 0.0f};
        if (s < S) { T = sT[wave][s]; alpha = sA[wave][s]; gw = sG[wave][s]; rv = rr[s]; }
        const float w = alpha * T;
        const float own = w * gw;
        float suf = own;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_down(suf, o); if (lane + o < 64) suf += t; }
        const float suffix = (suf - own) + scarry;
        scarry += __shfl(suf, 0);
        if (s < S) {
            const float g_alpha = T * gw - suffix / ((1.0f - alpha) + 1e-10f);
            const float dist = ((mip || s + 1 < S) ? (z[ray * zp + s + 1] - z[ray * zp + s]) : 1e10f) * nrm;
            const float pre_sig = rv[3] + (noise ? noise[ray * S + s] : 0.0f);
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float c = 1.0f / (1.0f + expf(-rv[k]));
                const float gk = (k == 0) ? g0 : (k == 1 ? g1 : g2);
                o[k] = w * gk * c * (1.0f - c);
            }
            o[3] = (pre_sig > 0.0f) ? g_alpha * dist * (1.0f - alpha) : 0.0f;
            gout[s] = o;
        }
    }
}

}  // namespace nvsr

using namespace nvsr;

// render_bwd_limb.hip
extern "C" int nvsr_pack_decoder_bwd_limbs_launch(const float* natural, float* packed_bwd, nvsr_stream_t stream);
extern "C" int nvsr_render_pass_backward_gates_limb_launch(int limbs, const nvsr_scene* scene, const float* packed_decoder, const float* packed_bwd,
                                                           int64_t N, int S, const float* rays, const float* z, const float* g_raw,
                                                           const uint32_t* gates, float* const* grad_planes, float* view_ws, float* record,
                                                           nvsr_stream_t stream);
extern "C" int nvsr_internal_resolve_decoder_arith(int arithmetic);      // render.hip

extern "C" {

int nvsr_pack_decoder_bwd(const float* natural, float* packed_bwd, nvsr_stream_t stream) {
    if (!natural || !packed_bwd) return NVSR_ERR_NULL;
    if (!aligned16(packed_bwd)) return NVSR_ERR_ALIGN;
    hipLaunchKernelGGL(pack_decoder_bwd_kernel, dim3((B_TOTAL + 255) / 256), dim3(256), 0, (hipStream_t)stream, natural, packed_bwd);
    if (int e = NVSR_CHECK_LAUNCH()) return e;
    return nvsr_pack_decoder_bwd_limbs_launch(natural, packed_bwd, stream);      // the bf16-limb fragments behind the f32 ones
}

static int composite_backward_launch(int64_t N, int S, const float* raw, const float* z, const float* rd, int rd_stride, int rd_off, const float* noise,
                                     int white_bkgd, const float* g_rgb, const float* g_acc, const float* g_depth, int mip_nerf, float* g_raw,
                                     nvsr_stream_t stream) {
    if (!raw || !z || !rd || !g_rgb || !g_raw) return NVSR_ERR_NULL;
    if (!aligned16(raw) || !aligned16(g_raw)) return NVSR_ERR_ALIGN;
    if (N < 0 || S < 1 || S > 512) return NVSR_ERR_SHAPE;
    if (N == 0) return NVSR_OK;
    hipLaunchKernelGGL(composite_backward_kernel, dim3((unsigned)((N + CB_WPB - 1) / CB_WPB)), dim3(CB_WPB * 64), 0, (hipStream_t)stream, (long)N, S,
                       raw, z, rd, noise, white_bkgd, g_rgb, g_acc, g_depth, g_raw, mip_nerf ? 1 : 0, rd_stride, rd_off);
    return NVSR_CHECK_LAUNCH();
}
int nvsr_composite_backward_depth(int64_t N, int S, const float* raw, const float* z, const float* rd, const float* noise, int white_bkgd,
                                  const float* g_rgb, const float* g_acc, const float* g_depth, int mip_nerf, float* g_raw, nvsr_stream_t stream) {
    return composite_backward_launch(N, S, raw, z, rd, 3, 0, noise, white_bkgd, g_rgb, g_acc, g_depth, mip_nerf, g_raw, stream);
}
/* the same with the ray directions read out of packed rays [N,11] (columns 3..5, nvsr_pack_rays): no [N,3] copy per backward pass */
int nvsr_composite_backward_rays(int64_t N, int S, const float* raw, const float* z, const float* rays, const float* noise, int white_bkgd,
                                 const float* g_rgb, const float* g_acc, const float* g_depth, int mip_nerf, float* g_raw, nvsr_stream_t stream) {
    return composite_backward_launch(N, S, raw, z, rays, 11, 3, noise, white_bkgd, g_rgb, g_acc, g_depth, mip_nerf, g_raw, stream);
}

int nvsr_composite_backward(int64_t N, int S, const float* raw, const float* z, const float* rd, const float* noise, int white_bkgd,
                            const float* g_rgb, const float* g_acc, float* g_raw, nvsr_stream_t stream) {
    return nvsr_composite_backward_depth(N, S, raw, z, rd, noise, white_bkgd, g_rgb, g_acc, nullptr, 0, g_raw, stream);
}

int nvsr_composite_backward_mip(int64_t N, int S, const float* raw, const float* z, const float* rd, const float* noise, int white_bkgd,
                                const float* g_rgb, const float* g_acc, float* g_raw, nvsr_stream_t stream) {
    return nvsr_composite_backward_depth(N, S, raw, z, rd, noise, white_bkgd, g_rgb, g_acc, nullptr, 1, g_raw, stream);
}

int64_t nvsr_decoder_record_floats(int64_t N, int S) {
    if (N < 0 || S < 1) return 0;
    return record_alloc_rows((long)N, S) * DEC_RECORD_FLOATS_PER_SLOT;
}

static int launch_view_reduce(const nvsr_scene* scene, int64_t N, int S, const float* rays, const float* view_ws, float* gplane,
                              hipStream_t stream) {
    hipLaunchKernelGGL(view_reduce_scatter_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, stream, to_dev(scene), (long)N, S, rays, view_ws,
                       gplane);
    return NVSR_CHECK_LAUNCH();
}

int64_t nvsr_view_grad_workspace_floats(int64_t N, int S) { return (N < 0 || S < 1) ? 0 : N * (int64_t)S * C; }

int nvsr_render_pass_backward_ex(const nvsr_scene* scene, const float* packed_decoder, const float* packed_bwd, int64_t N, int S,
                                 const float* rays, const float* z, const float* g_raw, float* const* grad_planes, float* record,
                                 float* view_ws, nvsr_stream_t stream) {
    if (!scene || !packed_decoder || !packed_bwd || !rays || !z || !g_raw) return NVSR_ERR_NULL;
    if (!grad_planes && !record) return NVSR_ERR_NULL;
    GradPlanes gp;
    for (int d = 0; d < 4; ++d) {
        if (!scene->planes[d]) return NVSR_ERR_NULL;
        if (!aligned16(scene->planes[d])) return NVSR_ERR_ALIGN;
        if (scene->ph[d] < 1 || scene->pw[d] < 1) return NVSR_ERR_SHAPE;
        gp.p[d] = grad_planes ? grad_planes[d] : nullptr;         // a NULL plane pointer = that plane is frozen
    }
    if (!aligned16(packed_decoder) || !aligned16(packed_bwd) || !aligned16(g_raw) || !aligned16(record)) return NVSR_ERR_ALIGN;
    if (N < 0 || S < 1 || S > 4096) return NVSR_ERR_SHAPE;
    if (N == 0) return NVSR_OK;
    const int64_t ntiles = ((N + BPTS - 1) / BPTS) * S;
    const int64_t grid = ntiles < 1024 ? ntiles : 1024;
    if (record) {
        const DecRecord rec = make_record(record, (long)N, S);
        hipLaunchKernelGGL(render_pass_backward_kernel<true>, dim3((unsigned)grid), dim3(BTPB), 0, (hipStream_t)stream, to_dev(scene),
                           packed_decoder, packed_bwd, (long)N, S, rays, z, g_raw, gp, rec, view_ws);
    } else {
        hipLaunchKernelGGL(render_pass_backward_kernel<false>, dim3((unsigned)grid), dim3(BTPB), 0, (hipStream_t)stream, to_dev(scene),
                           packed_decoder, packed_bwd, (long)N, S, rays, z, g_raw, gp, DecRecord{}, view_ws);
    }
    if (int e = NVSR_CHECK_LAUNCH()) return e;
    if (view_ws && gp.p[3]) return launch_view_reduce(scene, N, S, rays, view_ws, gp.p[3], (hipStream_t)stream);
    return NVSR_OK;
}

int nvsr_render_pass_backward_gates(const nvsr_scene* scene, const float* packed_decoder, const float* packed_bwd, int64_t N, int S,
                                    const float* rays, const float* z, const float* g_raw, const uint32_t* gates, float* const* grad_planes,
                                    float* view_ws, float* record, nvsr_stream_t stream) {
    return nvsr_render_pass_backward_gates_arith(scene, packed_decoder, packed_bwd, N, S, rays, z, g_raw, gates, grad_planes, view_ws, record,
                                                 NVSR_ARITH_INHERIT, stream);
}

int nvsr_render_pass_backward_gates_arith(const nvsr_scene* scene, const float* packed_decoder, const float* packed_bwd, int64_t N, int S,
                                          const float* rays, const float* z, const float* g_raw, const uint32_t* gates,
                                          float* const* grad_planes, float* view_ws, float* record, int arithmetic, nvsr_stream_t stream) {
    const int arith = nvsr_internal_resolve_decoder_arith(arithmetic);
    if (arith < 0) return NVSR_ERR_SHAPE;
    if (!scene || !packed_decoder || !packed_bwd || !rays || !z || !g_raw || !gates) return NVSR_ERR_NULL;
    if (!grad_planes && !record) return NVSR_ERR_NULL;
    GradPlanes gp;
    for (int d = 0; d < 4; ++d) {
        if (!scene->planes[d]) return NVSR_ERR_NULL;
        if (scene->ph[d] < 1 || scene->pw[d] < 1) return NVSR_ERR_SHAPE;
        gp.p[d] = grad_planes ? grad_planes[d] : nullptr;
    }
    if (!aligned16(packed_decoder) || !aligned16(packed_bwd) || !aligned16(g_raw) || !aligned16(gates) || !aligned16(record)) return NVSR_ERR_ALIGN;
    if (N < 0 || S < 1 || S > 4096) return NVSR_ERR_SHAPE;
    if (N == 0) return NVSR_OK;
    if (arith != NVSR_ARITH_F32) {       // limb matrix pipe (render_bwd_limb.hip): 3 bf16 limbs; NVSR_ARITH_F16X2: 2 f16 limbs, per-tile scale
        if (int e = nvsr_render_pass_backward_gates_limb_launch(arith == NVSR_ARITH_F16X2 ? 2 : 3, scene, packed_decoder, packed_bwd, N, S, rays, z, g_raw,
                                                                gates, grad_planes, view_ws, record, stream))
            return e;
        // the limb kernel leaves one pre-summed row per (ray, 32-sample chunk) in view_ws
        if (view_ws && gp.p[3]) return launch_view_reduce(scene, N, (S + 31) / 32, rays, view_ws, gp.p[3], (hipStream_t)stream);
        return NVSR_OK;
    }
    const int pts = record ? MBwd<true>::PTS : MBwd<false>::PTS;
    const int64_t ntiles = ((N + pts - 1) / pts) * S;
    const int64_t grid = ntiles < 1024 ? ntiles : 1024;
    if (record)
        hipLaunchKernelGGL(render_pass_backward_gates_kernel<true>, dim3((unsigned)grid), dim3(MBwd<true>::TPB), 0, (hipStream_t)stream, to_dev(scene),
                           packed_decoder, packed_bwd, (long)N, S, rays, z, g_raw, gates, gp, view_ws, make_record(record, (long)N, S));
    else
        hipLaunchKernelGGL(render_pass_backward_gates_kernel<false>, dim3((unsigned)grid), dim3(MBwd<false>::TPB), 0, (hipStream_t)stream, to_dev(scene),
                           packed_decoder, packed_bwd, (long)N, S, rays, z, g_raw, gates, gp, view_ws, DecRecord{});
    if (int e = NVSR_CHECK_LAUNCH()) return e;
    if (view_ws && gp.p[3]) return launch_view_reduce(scene, N, S, rays, view_ws, gp.p[3], (hipStream_t)stream);
    return NVSR_OK;
}

int nvsr_render_pass_backward(const nvsr_scene* scene, const float* packed_decoder, const float* packed_bwd, int64_t N, int S,
                              const float* rays, const float* z, const float* g_raw, float* const* grad_planes, nvsr_stream_t stream) {
    if (!grad_planes) return NVSR_ERR_NULL;
    for (int d = 0; d < 4; ++d)
        if (!grad_planes[d]) return NVSR_ERR_NULL;
    return nvsr_render_pass_backward_ex(scene, packed_decoder, packed_bwd, N, S, rays, z, g_raw, grad_planes, nullptr, nullptr, stream);
}

}  // extern "C"
