// Device-side core of the tri-plane decoder shared by the forward (render.hip) and backward (render_bwd.hip) kernels:
// LDS-DMA weight ring, bilinear taps + feature gather, MFMA layer blocks, one decode step.  NWAVES = waves per workgroup.
#pragma once
#include "nvsr_common.h"

#ifndef NVSR_ABLATE
#define NVSR_ABLATE 0   // timing experiments only (tools/README.md): 1 no gather, 2 no bias/ReLU, 4 no ring barrier, 8 no heads
#endif

namespace nvsr {

constexpr int SLOT_FLOATS = 8192;                     // 32 KB ring slot: one plane of a feature layer, or half a hidden layer
constexpr int LDS_FLOATS = 2 * SLOT_FLOATS + SMALL_FLOATS;

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// LDS-DMA: the 8 waves copy `BLOCKS` 1-KiB blocks (lane-linear, 16 B per lane); wave w takes blocks w, w+8, ...
// gsrc is wave-uniform (SGPR base), voff = wave*1024 + lane*16 bytes is the only per-lane address register.
template <int NWAVES, int BLOCKS>
__device__ __forceinline__ void stage_chunk(const float* __restrict__ gsrc, float* lds_dst, unsigned voff, int wave) {
    static_assert(BLOCKS % NWAVES == 0, "chunk must split evenly over the waves");
    const char* g = reinterpret_cast<const char*>(gsrc);
#pragma unroll
    for (int i = 0; i < BLOCKS / NWAVES; ++i) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + i * (NWAVES * 1024) + voff),
                                         (__attribute__((address_space(3))) void*)(lds_dst + (i * NWAVES + wave) * 256), 16, 0, 0);
    }
}

// Wait for this wave's DMA, then meet the other waves: afterwards the chunk issued one phase ago is readable by everyone
// and the slot read one phase ago is free.
__device__ __forceinline__ void ring_sync() {
#if !(NVSR_ABLATE & 4)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#endif
}

// ---- bilinear taps (grid_sample, align_corners=True, padding_mode='border') ------------------------------------------
struct Taps {
    int o00, o01, o10, o11;   // float offsets of the 4 texels (channel 0)
    float nw, ne, sw, se;
};

// ix, iy: the cell (north-west texel) the point falls into
__device__ __forceinline__ Taps make_taps_cell(const SceneDev& sc, int d, float gx, float gy, int& ix, int& iy) {
    const int H = sc.ph[d], W = sc.pw[d];
    const float mx = sc.mx[d], my = sc.my[d];
    float x = (gx + 1.0f) * sc.hx[d];
    float y = (gy + 1.0f) * sc.hy[d];
    x = fminf(mx, fmaxf(x, 0.0f));
    y = fminf(my, fmaxf(y, 0.0f));
    const float xw = floorf(x), yn = floorf(y);
    const float w = x - xw, e = 1.0f - w, n = y - yn, s = 1.0f - n;
    Taps t;
    t.nw = s * e; t.ne = s * w; t.sw = n * e; t.se = n * w;
    ix = (int)xw; iy = (int)yn;
    const int ix1 = min(ix + 1, W - 1), iy1 = min(iy + 1, H - 1);   // a clamped neighbour always carries weight 0
    t.o00 = (iy * W + ix) * C;  t.o01 = (iy * W + ix1) * C;
    t.o10 = (iy1 * W + ix) * C; t.o11 = (iy1 * W + ix1) * C;
    return t;
}
__device__ __forceinline__ Taps make_taps(const SceneDev& sc, int d, float gx, float gy) {
    int ix, iy;
    return make_taps_cell(sc, d, gx, gy, ix, iy);
}

// 24 channels (half h of the texel) of the bilinear blend -> f[0..23].  Two taps are in flight at a time (48 registers).
__device__ __forceinline__ void gather24(const float* __restrict__ plane, const Taps& t, int h, float (&f)[HALF_C]) {
#if NVSR_ABLATE & 1
#pragma unroll
    for (int i = 0; i < HALF_C; ++i) f[i] = t.nw * (float)(i + h);
    return;
#endif
    const f32x4* p00 = reinterpret_cast<const f32x4*>(plane + t.o00 + HALF_C * h);
    const f32x4* p01 = reinterpret_cast<const f32x4*>(plane + t.o01 + HALF_C * h);
    const f32x4* p10 = reinterpret_cast<const f32x4*>(plane + t.o10 + HALF_C * h);
    const f32x4* p11 = reinterpret_cast<const f32x4*>(plane + t.o11 + HALF_C * h);
#pragma unroll
    for (int i = 0; i < HALF_C / 4; ++i) {
        const f32x4 a = p00[i], b = p01[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) f[4 * i + j] = fmaf(b[j], t.ne, a[j] * t.nw);
    }
#pragma unroll
    for (int i = 0; i < HALF_C / 4; ++i) {
        const f32x4 c = p10[i], d = p11[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) f[4 * i + j] = fmaf(d[j], t.se, fmaf(c[j], t.sw, f[4 * i + j]));
    }
}

// x / 3 correctly rounded in 3 instructions (Markstein: q = RN(x*c), r = x - 3q exactly by FMA, q + r*c)
__device__ __forceinline__ float div3(float x) {
    const float c = 0x1.555556p-2f;
    const float q = x * c;
    return fmaf(fmaf(-3.0f, q, x), c, q);
}

__device__ __forceinline__ float norm_coord(float v, float lo, float range) {
    return __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, __fsub_rn(v, lo)), range), 1.0f);
}

// ---- MFMA layers ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_bias(const float* bias /*LDS, packed [ib][q][h][j]*/, int h, f32x16 (&acc)[4]) {
#if NVSR_ABLATE & 2
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ib][r] = 0.0f;
    return;
#endif
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(bias + (ib * 4 + q) * 8 + h * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[ib][4 * q + j] = b[j];
        }
}

__device__ __forceinline__ void relu_inplace(f32x16 (&acc)[4]) {
#if NVSR_ABLATE & 2
    return;
#endif
#if NVSR_RELU_PLAIN
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ib][r] = fmaxf(acc[ib][r], 0.0f);
#elif NVSR_RELU_GATEFORM
    // Compare + select with the gate bits folded into two running words (exactly relu_publish minus its store): the serial or-chain makes
    // hipcc process the 64 elements in order, and the step spills ~20 VGPRs instead of ~80 -- at the price of 3 extra VALU ops per element.
    unsigned m0 = 0u, m1 = 0u;
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool on = acc[ib][r] > 0.0f;
            acc[ib][r] = on ? acc[ib][r] : 0.0f;
            if (ib < 2) m0 |= on ? (1u << gate_bit(ib, r)) : 0u;
            else m1 |= on ? (1u << gate_bit(ib, r)) : 0u;
        }
    asm volatile("" :: "v"(m0), "v"(m1));
#else
    // One v_max per element, in place, as volatile inline asm: no operand canonicalisation (fmaxf costs a second v_max) and a fixed order,
    // which is what keeps hipcc from interleaving the four accumulator tuples with the MFMA operands around them.  Every VALU instruction
    // costs its ~4 issue cycles even next to MFMAs (tools/mfma_ubench.hip), so the instruction count matters here.
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("v_max_f32 %0, 0, %0" : "+v"(acc[ib][r]));
#endif
}

// ReLU that also publishes its gate: bit (ib&1)*16 + r of word ib>>1  <=>  acc[ib][r] > 0.  The two words go straight to global
// memory (slot `layer` of the lane's 16-word record), so nothing stays live for them.
__device__ __forceinline__ void relu_publish(f32x16 (&acc)[4], unsigned* __restrict__ rec, int layer) {
    unsigned m0 = 0u, m1 = 0u;
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool on = acc[ib][r] > 0.0f;
            acc[ib][r] = on ? acc[ib][r] : 0.0f;
            if (ib < 2) m0 |= on ? (1u << gate_bit(ib, r)) : 0u;
            else m1 |= on ? (1u << gate_bit(ib, r)) : 0u;
        }
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    *reinterpret_cast<u32x2*>(rec + 2 * layer) = u32x2{m0, m1};
}

// record helpers: one accumulator set (128 features x 32 points, C/D layout) -> rows [row][128]; 32 B per lane pair and store, the
// wave's 16 stores fill 32 complete 512-byte rows.
// NVSR_RECORD_NT (round 6 experiment, OFF): the record is written once and read once, by another kernel, after 7 GB more of it have gone by, so a
// non-temporal store hint looked right -- measured it is 3-5 x SLOWER (recording forward 0.82 -> 4.0 ms, recording backward 0.79 -> 2.35 ms, step 4.3
// -> 10.4-10.9 ms, same box): a record row is 512 B assembled from 32-byte stores of 16 lanes pairs, and `nt` stores leave L2's write combining, so
// HBM sees partial-line writes
#ifndef NVSR_RECORD_NT
#define NVSR_RECORD_NT 0
#endif
__device__ __forceinline__ void rec_store(float* p, f32x4 v) {
#if NVSR_RECORD_NT
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
#else
    *reinterpret_cast<f32x4*>(p) = v;
#endif
}
// NVSR_RECORD_TIMING_COALESCED (timing experiment, WRONG layout): the same 16 stores with every instruction writing 1 KB of consecutive addresses
// inside the tile's 16-KB block of rows -- what the record would cost if a store instruction covered whole cache lines instead of 64 pieces of 16 B
#ifndef NVSR_RECORD_TIMING_COALESCED
#define NVSR_RECORD_TIMING_COALESCED 0
#endif
__device__ __forceinline__ float* rec_piece(float* base, long q, int h, int ib, int qq) {
#if NVSR_RECORD_TIMING_COALESCED
    return base + (q - (q & 31)) * HID + ((ib * 4 + qq) * 64 + h * 32 + (int)(q & 31)) * 4;
#else
    return base + q * HID + 4 * h + 32 * ib + 8 * qq;
#endif
}
__device__ __forceinline__ void record128(float* __restrict__ base, long q, int h, const f32x16 (&a)[4]) {
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
            rec_store(rec_piece(base, q, h, ib, qq), f32x4{a[ib][4 * qq], a[ib][4 * qq + 1], a[ib][4 * qq + 2], a[ib][4 * qq + 3]});
}
// the same with every value multiplied by a (wave-uniform) factor on its way out: accumulators that carry a power-of-two scale
__device__ __forceinline__ void record128_scaled(float* __restrict__ base, long q, int h, const f32x16 (&a)[4], float f) {
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
            rec_store(rec_piece(base, q, h, ib, qq), f32x4{a[ib][4 * qq] * f, a[ib][4 * qq + 1] * f, a[ib][4 * qq + 2] * f, a[ib][4 * qq + 3] * f});
}
// ---- the same rows through a per-wave LDS stage (round 6) ---------------------------------------------------------------------------------
// record128 hands the memory system 64 pieces of 16 B per store instruction (lane (pt, h) owns features 8 qq + 4 h .. + 3 of its point: the two
// lanes of a point sit in different quarter-waves, and neighbouring lanes are neighbouring ROWS, 512 B apart).  Timing build with the same bytes in
// whole lines per instruction: recording forward 0.82 -> 0.74 ms, recording backward 0.80 -> 0.69 ms at S = 128 (NVSR_RECORD_TIMING_COALESCED).
// Here a 32-feature block of the tile goes through LDS -- [32 points][RSTG_STRIDE floats], 4 ds_write_b128 in, 4 ds_read_b128 out -- so that the 8
// lanes 8 k' .. 8 k' + 7 hold the 8 consecutive 16-byte pieces of ONE row's 128-byte segment: a store instruction writes 8 whole cache lines.
// The tile's points must be consecutive record rows q0 .. q0 + 31 (32 consecutive samples of one ray: the limb kernels' tiles); points >= nvalid
// (a partial chunk, a tile behind the last ray) go to the record's dump rows: all 16 stores are issued by every wave.
constexpr int RSTG_STRIDE = 36;                    // 32 + 4: the 8 lanes of a ds_write_b128 group land on 8 different bank quads
constexpr int RSTG_FLOATS = 32 * RSTG_STRIDE;      // per wave
template <bool SCALED>
__device__ __forceinline__ void record128_staged(float* stage, float* __restrict__ base, long q0, int nvalid, long dump, int lane, const f32x16 (&a)[4], float f) {
    const int pt = lane & 31, h = lane >> 5;
    const int p_ = lane >> 3, c = lane & 7;        // read side: point 8 k + p_, piece c of its 128-byte segment
#pragma unroll
    for (int ib = 0; ib < 4; ++ib) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            f32x4 v = f32x4{a[ib][4 * qq], a[ib][4 * qq + 1], a[ib][4 * qq + 2], a[ib][4 * qq + 3]};
            if (SCALED) v = v * f;
            *reinterpret_cast<f32x4*>(stage + pt * RSTG_STRIDE + 8 * qq + 4 * h) = v;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = 8 * k + p_;
            const f32x4 v = *reinterpret_cast<const f32x4*>(stage + p * RSTG_STRIDE + 4 * c);
            const long row = p < nvalid ? q0 + p : dump + p;
            rec_store(base + row * HID + 32 * ib + 4 * c, v);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// A plane's 48 feature columns of the tile's 32 rows the same way (lane (pt, h) holds channels 24 h .. 24 h + 23 of its point): the two lane
// halves go through the stage one after the other -- [32 points][RSTG24_STRIDE floats], 6 quads in per lane, then 192 pieces of 16 B out over
// 3 store instructions in which 6 neighbouring lanes cover 96 consecutive bytes of one row.  6 stores per call like record24.
constexpr int RSTG24_STRIDE = 28;                  // 24 + 4: conflict-free ds_write_b128 (8 lanes on 8 bank quads); 32 x 28 <= RSTG_FLOATS
static_assert(32 * RSTG24_STRIDE <= RSTG_FLOATS, "feature stage fits the hidden-row stage");
template <bool SCALED>
__device__ __forceinline__ void record24_staged(float* stage, float* __restrict__ base /* array + first column */, int rowstride, long q0, int nvalid,
                                                long dump, int lane, const float (&f)[HALF_C], float k) {
    const int pt = lane & 31, h = lane >> 5;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        __builtin_amdgcn_wave_barrier();
        if (h == hh) {
#pragma unroll
            for (int i = 0; i < HALF_C / 4; ++i) {
                f32x4 v = f32x4{f[4 * i], f[4 * i + 1], f[4 * i + 2], f[4 * i + 3]};
                if (SCALED) v = v * k;
                *reinterpret_cast<f32x4*>(stage + pt * RSTG24_STRIDE + 4 * i) = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int g = 64 * j + lane, p = g / 6, c = g - 6 * p;          // point 0..31, piece 0..5 of its 96 bytes
            const f32x4 v = *reinterpret_cast<const f32x4*>(stage + p * RSTG24_STRIDE + 4 * c);
            const long row = p < nvalid ? q0 + p : dump + p;
            rec_store(base + row * rowstride + HALF_C * hh + 4 * c, v);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// a lane's 24 channels of one plane's feature (channels 24h .. 24h+23) -> row[24h ..]
__device__ __forceinline__ void record24(float* __restrict__ row, int h, const float (&f)[HALF_C]) {
#pragma unroll
    for (int i = 0; i < HALF_C / 4; ++i)
        rec_store(row + HALF_C * h + 4 * i, f32x4{f[4 * i], f[4 * i + 1], f[4 * i + 2], f[4 * i + 3]});
}

__device__ __forceinline__ void record24_scaled(float* __restrict__ row, int h, const float (&f)[HALF_C], float k) {
#pragma unroll
    for (int i = 0; i < HALF_C / 4; ++i)
        rec_store(row + HALF_C * h + 4 * i, f32x4{f[4 * i] * k, f[4 * i + 1] * k, f[4 * i + 2] * k, f[4 * i + 3] * k});
}

// MFMA block shared by the feature and hidden layers: NG groups of 4 MFMAs on one accumulator; group g uses the A fragment
// wl[g] (256 floats [lane][j], one conflict-free ds_read_b128) and the 4 B registers b(g, j).  Pinned order per group: MFMA,
// ds_read of the NEXT fragment, 3 MFMAs -- hipcc waits with lgkmcnt(0) in front of a group's first MFMA, i.e. for every
// outstanding read, so the next fragment gets 3 MFMAs (192 cycles) to land; left alone hipcc sinks the read to its use.
// (Measured: a v_mfma_f32_32x32x2_f32 stream fed this way sustains 64.2 cycles per MFMA, tools/mfma_ubench.hip.)
template <int NG, class BFn>
__device__ __forceinline__ void mfma_groups(const float* wl, int lane, f32x16 (&acc)[4], BFn b) {
    const f32x4* wv = reinterpret_cast<const f32x4*>(wl) + lane;
    f32x4 a = wv[0];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int ib = g & 3;
        acc[ib] = mfma32(a[0], b(g, 0), acc[ib]);
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 an = wv[(g + 1 < NG ? g + 1 : g) * 64];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 1; j < 4; ++j) acc[ib] = mfma32(a[j], b(g, j), acc[ib]);
        a = an;
        __builtin_amdgcn_sched_barrier(0);
    }
}

// one plane's 48 channels of a feature layer: acc += W[:, 48p .. 48p+47] * f      (chunk layout [q 6][ib][lane][j])
__device__ __forceinline__ void feat_layer(const float* wl, const float (&f)[HALF_C], int lane, f32x16 (&acc)[4]) {
    mfma_groups<(HALF_C / 4) * 4>(wl, lane, acc, [&](int g, int j) { return f[4 * (g >> 2) + j]; });
}

// half of a hidden layer 128 -> 128: acc += W[:, 64*HALF .. 64*HALF+63] * in[2*HALF .. 2*HALF+1]   (in = previous accumulators,
// already ReLU'd).  wl = the 32 KB chunk [kb 2][q 4][ib][lane][j].
template <int HALF>
__device__ __forceinline__ void hidden_half(const float* wl, const f32x16 (&in)[4], int lane, f32x16 (&acc)[4]) {
    mfma_groups<32>(wl, lane, acc, [&](int g, int j) { return in[2 * HALF + (g >> 4)][4 * ((g >> 2) & 3) + j]; });
}

// 128 -> NH heads on the VALU: each lane owns 64 of the 128 features of its point, the partner lane (l ^ 32) the rest.  The weights
// stream from LDS one (ib, q) group at a time with the next group prefetched and the order pinned: left to itself hipcc hoists all
// 16 (x NH) ds_read_b128 of a head above the FMAs -- 64 live registers per head next to the 64 accumulators, which is what spilled
// ~55 VGPRs per step in the fused kernels.  The per-head fmaf chain (ib, q, j ascending) is unchanged.
template <int NH>
__device__ __forceinline__ void head_dots(const float* w /*LDS packed [head][ib][q][h][j]*/, int h, const f32x16 (&in)[4], float (&out)[NH]) {
#if NVSR_ABLATE & 8
#pragma unroll
    for (int k = 0; k < NH; ++k) out[k] = in[0][0] + in[1][1] + in[2][2] + in[3][3];
    return;
#endif
    const float* wl = w + h * 4;
    f32x4 cur[NH], nxt[NH];
#pragma unroll
    for (int k = 0; k < NH; ++k) { cur[k] = *reinterpret_cast<const f32x4*>(wl + k * HID); out[k] = 0.0f; }
#pragma unroll
    for (int g = 0; g < 16; ++g) {                 // g = ib * 4 + q
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1 < 16) {
#pragma unroll
            for (int k = 0; k < NH; ++k) nxt[k] = *reinterpret_cast<const f32x4*>(wl + k * HID + (g + 1) * 8);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < NH; ++k) out[k] = fmaf(in[g >> 2][4 * (g & 3) + j], cur[k][j], out[k]);
#pragma unroll
        for (int k = 0; k < NH; ++k) cur[k] = nxt[k];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < NH; ++k) out[k] = out[k] + __shfl_xor(out[k], 32);
}

struct RingState {
    const float* packed;   // global packed decoder blob
    float* lds;            // LDS base
    int slot;              // slot that receives the NEXT issued chunk
    int wave, lane;
    unsigned voff;         // wave*1024 + lane*16: per-lane byte offset of the LDS-DMA source
};

template <int NWAVES, int BLOCKS>
__device__ __forceinline__ const float* ring_issue(RingState& rs, int packed_off) {
    float* dst = rs.lds + rs.slot * SLOT_FLOATS;
    stage_chunk<NWAVES, BLOCKS>(rs.packed + packed_off, dst, rs.voff, rs.wave);
    rs.slot ^= 1;
    return dst;
}

// Decode the wave's 32 points (px,py,pz given per lane, both lane halves hold the same point) -> raw rgb + sigma.
// vt: bilinear taps on the view-direction plane.  Must be called by all waves of the workgroup together (ring barriers).
// No DMA is outstanding on entry or on exit.
// Chunk order: RGB0.p0..p3, RGB1..3 (two halves each), DEN0, DEN1..3 (two halves each).
// MASKS: additionally write the 8 layers' ReLU gates to `gates` (this lane's 16 words: density layers 0..3, rgb layers 0..3) for
// the mask-driven backward kernel, which then does not have to recompute the forward.
// RECORD: additionally write the inputs of every layer (plane features, post-ReLU activations) to row q of `rec` (lanes with
// rec_ok == false, i.e. padding rays, skip the stores) for the decoder weight gradient.
template <int NWAVES, bool MASKS = false, bool RECORD = false>
__device__ __forceinline__ void decode_step(const SceneDev& sc, RingState& rs, float px, float py, float pz, const Taps& vt,
                                            float (&raw)[4], unsigned* __restrict__ gates = nullptr, const DecRecord* rec = nullptr,
                                            long q = 0, bool rec_ok = false) {
    // Re-derive the per-lane address registers every step: left loop-invariant, hipcc hoists one 64-bit address per
    // DMA / LDS read out of the sample loop and spills them all.
    asm volatile("" : "+v"(rs.voff), "+v"(rs.lane));
    const int lane = rs.lane, h = lane >> 5;
    const float* small = rs.lds + 2 * SLOT_FLOATS;
    const float n0 = norm_coord(px, sc.lo[0], sc.range[0]);
    const float n1 = norm_coord(py, sc.lo[1], sc.range[1]);
    const float n2 = norm_coord(pz, sc.lo[2], sc.range[2]);

    // Order chosen for register pressure (256 VGPRs at 2 waves/SIMD): the rgb decoder runs first and consumes the plane
    // features as they are gathered (only 24 + 24 feature registers live), the density decoder then runs from the 24
    // registers of the averaged position features.  17 ring chunks per step (<= 32 KB each).
    f32x16 accA[4], accB[4];
    float D[HALF_C], F[HALF_C];
    constexpr int HH = P_HID_FLOATS / 2;   // half hidden layer

    // ---- rgb layer 0: K = 192 = [f0 | f1 | f2 | f_view], one plane (= one chunk) at a time ----------------------------
    // The step's first chunk is issued here, next to the first gather (their latencies overlap), NOT at the end of the previous
    // step: vector-memory waits are in order, so anything the caller loads between two steps would otherwise wait for the DMA.
    const float* cur = ring_issue<NWAVES, 24>(rs, P_RGB0);
    {
        const float* M = sc.proj;
        const Taps t = make_taps(sc, 0, n0 * M[0] + n1 * M[2] + n2 * M[4], n0 * M[1] + n1 * M[3] + n2 * M[5]);
        gather24(sc.plane[0], t, h, F);
    }
    if (RECORD && rec_ok) record24(rec->Xr + q * (4 * C), h, F);
    ring_sync();
    const float* nxt = ring_issue<NWAVES, 24>(rs, P_RGB0 + P_PLANE_FLOATS);
    load_bias(small + S_BIAS + 4 * HID, h, accA);
#pragma unroll
    for (int c = 0; c < HALF_C; ++c) D[c] = F[c];
    feat_layer(cur, F, lane, accA);
    cur = nxt;
    ring_sync();
    nxt = ring_issue<NWAVES, 24>(rs, P_RGB0 + 2 * P_PLANE_FLOATS);
    {
        const float* M = sc.proj + 6;
        const Taps t = make_taps(sc, 1, n0 * M[0] + n1 * M[2] + n2 * M[4], n0 * M[1] + n1 * M[3] + n2 * M[5]);
        gather24(sc.plane[1], t, h, F);
    }
    if (RECORD && rec_ok) record24(rec->Xr + q * (4 * C) + C, h, F);
#pragma unroll
    for (int c = 0; c < HALF_C; ++c) D[c] = __fadd_rn(D[c], F[c]);
    feat_layer(cur, F, lane, accA);
    cur = nxt;
    ring_sync();
    nxt = ring_issue<NWAVES, 24>(rs, P_RGB0 + 3 * P_PLANE_FLOATS);
    {
        const float* M = sc.proj + 12;
        const Taps t = make_taps(sc, 2, n0 * M[0] + n1 * M[2] + n2 * M[4], n0 * M[1] + n1 * M[3] + n2 * M[5]);
        gather24(sc.plane[2], t, h, F);
    }
    if (RECORD && rec_ok) record24(rec->Xr + q * (4 * C) + 2 * C, h, F);
    // combine_pos_planes 'avg' = stack(...).mean(0)  (models.py:358-359)
#pragma unroll
    for (int c = 0; c < HALF_C; ++c) D[c] = div3(__fadd_rn(D[c], F[c]));
    if (RECORD && rec_ok) {
        record24(rec->Xd + q * 64, h, D);
        *reinterpret_cast<f32x4*>(rec->Xd + q * 64 + C + 8 * h) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        *reinterpret_cast<f32x4*>(rec->Xd + q * 64 + C + 8 * h + 4) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    feat_layer(cur, F, lane, accA);
    cur = nxt;
    ring_sync();
    nxt = ring_issue<NWAVES, 32>(rs, P_RGB1);
    gather24(sc.plane[3], vt, h, F);
    if (RECORD && rec_ok) record24(rec->Xr + q * (4 * C) + 3 * C, h, F);
    feat_layer(cur, F, lane, accA);
    if (MASKS) relu_publish(accA, gates, 4); else relu_inplace(accA);
    if (RECORD && rec_ok) record128(rec->Hr + 0L * HID * rec->Pp, q, h, accA);
    cur = nxt;
    // ---- rgb decoder layers 1..3 -> 3 ---------------------------------------------------------------------------------
    ring_sync();
    nxt = ring_issue<NWAVES, 32>(rs, P_RGB1 + HH);
    load_bias(small + S_BIAS + 5 * HID, h, accB);
#if NVSR_ABLATE & 64
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long st0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
#endif
    hidden_half<0>(cur, accA, lane, accB);
#if NVSR_ABLATE & 64
    __builtin_amdgcn_sched_barrier(0);
    raw[0] = (float)(__builtin_amdgcn_s_memtime() - st0);
    __builtin_amdgcn_sched_barrier(0);
#endif
    cur = nxt;
    ring_sync();
    nxt = ring_issue<NWAVES, 32>(rs, P_RGB1 + 2 * HH);
    hidden_half<1>(cur, accA, lane, accB);
    if (MASKS) relu_publish(accB, gates, 5); else relu_inplace(accB);
    if (RECORD && rec_ok) record128(rec->Hr + 1L * HID * rec->Pp, q, h, accB);
    cur = nxt;
    ring_sync();
    nxt = ring_issue<NWAVES, 32>(rs, P_RGB1 + 3 * HH);
    load_bias(small + S_BIAS + 6 * HID, h, accA);
    hidden_half<0>(cur, accB, lane, accA);
    cur = nxt;
    ring_sync();
    nxt = ring_issue<NWAVES, 32>(rs, P_RGB1 + 4 * HH);
    hidden_half<1>(cur, accB, lane, accA);
    if (MASKS) relu_publish(accA, gates, 6); else relu_inplace(accA);
    if (RECORD && rec_ok) record128(rec->Hr + 2L * HID * rec->Pp, q, h, accA);
    cur = nxt;
    ring_sync();
    nxt = ring_issue<NWAVES, 32>(rs, P_RGB1 + 5 * HH);
    load_bias(small + S_BIAS + 7 * HID, h, accB);
    hidden_half<0>(cur, accA, lane, accB);
    cur = nxt;
    ring_sync();
    nxt = ring_issue<NWAVES, 24>(rs, P_DEN0);
    hidden_half<1>(cur, accA, lane, accB);
    if (MASKS) relu_publish(accB, gates, 7); else relu_inplace(accB);
    if (RECORD && rec_ok) record128(rec->Hr + 3L * HID * rec->Pp, q, h, accB);
#if NVSR_ABLATE & 64
    const float stamp = raw[0];
#endif
#pragma unroll
    for (int c = 0; c < 3; ++c) raw[c] = 0.0f;
    {
        float hd[3];
        head_dots<3>(small + S_RGB_W, h, accB, hd);
#pragma unroll
        for (int c = 0; c < 3; ++c) raw[c] = hd[c] + small[S_HEAD_B + 1 + c];
    }
#if NVSR_ABLATE & 64
    raw[1] = stamp;
#endif
    cur = nxt;
    // ---- density decoder: 48 -> 128 x4 -> 1 --------------------------------------------------------------------------
    ring_sync();
    nxt = ring_issue<NWAVES, 32>(rs, P_DEN1);
    load_bias(small + S_BIAS + 0 * HID, h, accA);
    feat_layer(cur, D, lane, accA);
    if (MASKS) relu_publish(accA, gates, 0); else relu_inplace(accA);
    if (RECORD && rec_ok) record128(rec->Hd + 0L * HID * rec->Pp, q, h, accA);
    cur = nxt;
    ring_sync();
    nxt = ring_issue<NWAVES, 32>(rs, P_DEN1 + HH);
    load_bias(small + S_BIAS + 1 * HID, h, accB);
    hidden_half<0>(cur, accA, lane, accB);
    cur = nxt;
    ring_sync();
    nxt = ring_issue<NWAVES, 32>(rs, P_DEN1 + 2 * HH);
    hidden_half<1>(cur, accA, lane, accB);
    if (MASKS) relu_publish(accB, gates, 1); else relu_inplace(accB);
    if (RECORD && rec_ok) record128(rec->Hd + 1L * HID * rec->Pp, q, h, accB);
    cur = nxt;
    ring_sync();
    nxt = ring_issue<NWAVES, 32>(rs, P_DEN1 + 3 * HH);
    load_bias(small + S_BIAS + 2 * HID, h, accA);
    hidden_half<0>(cur, accB, lane, accA);
    cur = nxt;
    ring_sync();
    nxt = ring_issue<NWAVES, 32>(rs, P_DEN1 + 4 * HH);
    hidden_half<1>(cur, accB, lane, accA);
    if (MASKS) relu_publish(accA, gates, 2); else relu_inplace(accA);
    if (RECORD && rec_ok) record128(rec->Hd + 2L * HID * rec->Pp, q, h, accA);
    cur = nxt;
    ring_sync();
    nxt = ring_issue<NWAVES, 32>(rs, P_DEN1 + 5 * HH);
    load_bias(small + S_BIAS + 3 * HID, h, accB);
    hidden_half<0>(cur, accA, lane, accB);
    cur = nxt;
    ring_sync();
    hidden_half<1>(cur, accA, lane, accB);
    if (MASKS) relu_publish(accB, gates, 3); else relu_inplace(accB);
    if (RECORD && rec_ok) record128(rec->Hd + 3L * HID * rec->Pp, q, h, accB);
    {
        float hd[1];
        head_dots<1>(small + S_ALPHA_W, h, accB, hd);
        raw[3] = hd[0] + small[S_HEAD_B];
    }
}

// workgroup prologue: heads/biases -> LDS (plain copy; the first ring barrier publishes them), then the STAGGER: the two
// workgroups of a CU start together and run the same program at the same rate, so left alone they stay phase-locked and
// their gathers, barriers and VALU epilogues coincide (measured: those costs add up instead of hiding behind the partner's
// MFMAs).  The workgroup whose wave 0 sits in an odd wave slot of its SIMD (HW_ID.wave_id) starts half a chunk-phase late.
template <int NWAVES>
__device__ __forceinline__ void decode_prologue(RingState& rs) {
    for (int i = threadIdx.x; i < SMALL_FLOATS; i += NWAVES * 64) rs.lds[2 * SLOT_FLOATS + i] = rs.packed[P_SMALL + i];
#if !(NVSR_ABLATE & 32)
    unsigned hw_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    if (__builtin_amdgcn_readfirstlane(hw_id) & 1u) __builtin_amdgcn_s_sleep(127);   // ~8k cycles = half of a shared 128-MFMA phase
#endif
}

__device__ __forceinline__ Taps view_taps(const SceneDev& sc, float vx, float vy, float vz) {
    // cart2az_el (nerf_helpers.py:492-496) + normalize_coords
    const float az = atan2f(vy, vx);
    const float el = atan2f(vz, sqrtf(__fadd_rn(__fmul_rn(vx, vx), __fmul_rn(vy, vy))));
    return make_taps(sc, 3, norm_coord(az, sc.lo[3], sc.range[3]), norm_coord(el, sc.lo[4], sc.range[4]));
}


}  // namespace nvsr
