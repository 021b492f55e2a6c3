// Decoder weight / bias gradients from the record written by render_pass_backward_kernel<true> (gfx950).
//
// The reference gets them from torch.autograd (addmm backward: dW = delta^T @ input, db = sum(delta)) when 'decoder' is in
// `nerf.train.what` (train_nerf.py:75-77, models.py:169-195).  Here every layer is one contraction over the record's slots:
//
//     dW_l[out][in] = sum_q  G_l[q][out] * X_l[q][in]            (X_0 = plane features, X_l = H_{l-1}),   db_l[out] = sum_q G_l[q][out]
//
// on v_mfma_f32_32x32x2_f32 with the slot index as K.  Both operands are read straight from HBM in operand layout: a lane takes one
// f32x4 of G (4 output rows of 4 MFMAs) and one f32x2 of X (2 input columns) for slot pair (q, q+1), so a half-wave reads whole
// 512-byte / 256-byte rows.  A workgroup owns a [128 out x 64 in] block of one layer for a slab of slots; its 4 waves split the
// slab, reduce through LDS and add the block into the gradient blob (state-dict order) with float atomics.
#include <cstdlib>

#include "limb_core.h"

namespace nvsr {

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct WJob {
    const float* G;     // [Pp][128]
    const float* X;     // [Pp][xstride]
    int xstride, col0;  // this block covers input columns col0 .. col0+63 of X
    int w_off;          // weight [128][in_total] in the gradient blob
    int in_total;       // also the number of valid input columns
    int b_off;          // bias offset, or -1 when another block of the same layer owns it
    int nb;             // limb kernel: 32-column blocks of this job (4: columns col0 .. col0+127, 2: col0 .. col0+63)
};
constexpr int WJOBS = 16;
struct WJobs { WJob j[WJOBS]; };

constexpr int WG_TPB = 256;

__device__ __forceinline__ f32x16 mfma32w(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// P = valid rows (may be odd: the second row of the last pair is then read as zero)
__global__ __launch_bounds__(WG_TPB, 2) void decoder_wgrad_kernel(WJobs jobs, long P, int slab, float* __restrict__ grad) {
    __shared__ __attribute__((aligned(16))) float tile[128 * 64];
    NVSR_RACE_PROBE_DELAY(tile);      // (probe builds only, nvsr_common.h)
    const WJob jb = jobs.j[blockIdx.y];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, kh = lane >> 5;
    const int per_wave = slab / 4;                                   // even (host guarantees slab % 8 == 0)
    long q0 = (long)blockIdx.x * slab + (long)wave * per_wave;
    long q1 = q0 + per_wave;
    const bool odd_tail = q1 >= P && q0 < P && (P & 1);               // this wave owns the unpaired last row
    if (q1 > (P & ~1L)) q1 = P & ~1L;
    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    float bs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (q0 < q1) {
        const float* gp_ = jb.G + (q0 + kh) * HID + 4 * i;
        const float* xp = jb.X + (q0 + kh) * (long)jb.xstride + jb.col0 + 2 * i;
        const long xs2 = 2L * jb.xstride;
        constexpr int UN = 4;
        long q = q0;
        for (; q + 2 * UN <= q1; q += 2 * UN) {
            f32x4 g[UN];
            f32x2 x[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                g[u] = *reinterpret_cast<const f32x4*>(gp_ + (long)u * 2 * HID);
                x[u] = *reinterpret_cast<const f32x2*>(xp + u * xs2);
            }
            gp_ += UN * 2 * HID;
            xp += UN * xs2;
#pragma unroll
            for (int u = 0; u < UN; ++u)
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    acc[a][0] = mfma32w(g[u][a], x[u][0], acc[a][0]);
                    acc[a][1] = mfma32w(g[u][a], x[u][1], acc[a][1]);
                    bs[a] += g[u][a];
                }
        }
        for (; q < q1; q += 2) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gp_);
            const f32x2 x = *reinterpret_cast<const f32x2*>(xp);
            gp_ += 2 * HID;
            xp += xs2;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                acc[a][0] = mfma32w(g[a], x[0], acc[a][0]);
                acc[a][1] = mfma32w(g[a], x[1], acc[a][1]);
                bs[a] += g[a];
            }
        }
    }
    if (odd_tail) {                                                   // row P-1 pairs with a zero row
        f32x4 g = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        f32x2 x = f32x2{0.0f, 0.0f};
        if (kh == 0) {
            g = *reinterpret_cast<const f32x4*>(jb.G + (P - 1) * HID + 4 * i);
            x = *reinterpret_cast<const f32x2*>(jb.X + (P - 1) * (long)jb.xstride + jb.col0 + 2 * i);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            acc[a][0] = mfma32w(g[a], x[0], acc[a][0]);
            acc[a][1] = mfma32w(g[a], x[1], acc[a][1]);
            bs[a] += g[a];
        }
    }
    // acc[a][b][r]: out = 4 * ((r&3) + 8(r>>2) + 4kh) + a,  in = 2i + b.   Sum the 4 waves in LDS (same element per lane in every wave).
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int out = 4 * ((r & 3) + 8 * (r >> 2) + 4 * kh) + a;
                    f32x2* t = reinterpret_cast<f32x2*>(tile + out * 64 + 2 * i);
                    f32x2 v = f32x2{acc[a][0][r], acc[a][1][r]};
                    if (w > 0) { const f32x2 o = *t; v[0] += o[0]; v[1] += o[1]; }
                    *t = v;
                }
        }
        __syncthreads();
    }
    for (int idx = threadIdx.x; idx < 128 * 64; idx += WG_TPB) {
        const int out = idx >> 6, col = jb.col0 + (idx & 63);
        if (col < jb.in_total) unsafeAtomicAdd(grad + jb.w_off + out * jb.in_total + col, tile[idx]);
    }
    if (jb.b_off >= 0) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float v = bs[a] + __shfl_xor(bs[a], 32);
            if (kh == 0) unsafeAtomicAdd(grad + jb.b_off + 4 * i + a, v);
        }
    }
}

// ---- the same contraction on the bf16 matrix pipe (limb_core.h: both operands split exactly into 3 bf16 limbs, 6 products per block) ----
// v_mfma_f32_32x32x16_bf16 takes 16 rows of the record per instruction: lane (i, h) holds rows q + 8h .. q + 8h + 7 of its output row /
// input column, so it reads 8 f32x4 of G (outputs 4i .. 4i+3 -> the A operands of 4 output tiles) and 8 x NB floats of X (columns
// NB i .. NB i + NB - 1 -> the B operands of NB column tiles) -- every row still a contiguous 512-byte read per half wave -- and splits
// each operand once: 4 + NB splits (5.5 VALU per value) feed 4 * NB * 6 MFMAs.  With NB = 4 a wave owns the whole [128 x 128] block of a
// hidden layer (256 accumulators, AGPRs; one wave per SIMD) and G is read once per layer; the two 64-column leftovers (density layer 0,
// columns 128..191 of rgb layer 0) are padded to whole blocks (see the launcher).  The next 16 rows are loaded while the current ones are
// multiplied.
template <int NB>
__device__ __forceinline__ void wgrad_limb_block(const WJob& jb, long P, int slab, float* tile, float* __restrict__ grad) {
    typedef float xvec __attribute__((ext_vector_type(NB)));
    // (wave index in a scalar register, lane from the hardware counter: with 256 accumulators + two row sets in flight every vector register
    //  counts, and threadIdx.x kept alive across the row loop for the flush was one of 5 that spilled)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)), i = lane & 31, kh = lane >> 5;
    const int per_wave = slab / 4;                                   // a multiple of 64 (the host makes slab a multiple of 256)
    const long q0 = (long)blockIdx.x * slab + (long)wave * per_wave;
    long q1 = q0 + per_wave;
    if (q1 > P) q1 = P;
    f32x16 acc[4][NB];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    float bs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const float* gbase = jb.G + 4 * i;
    const float* xbase = jb.X + jb.col0 + NB * i;
    const long xs = jb.xstride;
    f32x4 g[8], gn[8];
    xvec x[8], xn[8];
    // rows q + 8 kh + j, j = 0..7; rows >= P read as zero (the last, partial step of the pass)
    auto load = [&](long q, f32x4 (&gg)[8], xvec (&xx)[8]) {
        if (q + 16 <= P) {
            const float* gp_ = gbase + (q + 8 * kh) * HID;
            const float* xp = xbase + (q + 8 * kh) * xs;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                gg[j] = *reinterpret_cast<const f32x4*>(gp_ + j * HID);
                xx[j] = *reinterpret_cast<const xvec*>(xp + j * xs);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const long row = q + 8 * kh + j;
                const bool ok = row < P;
                const long rr = ok ? row : P - 1;
                gg[j] = *reinterpret_cast<const f32x4*>(gbase + rr * HID);
                xx[j] = *reinterpret_cast<const xvec*>(xbase + rr * xs);
                if (!ok) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) gg[j][c] = 0.0f;
#pragma unroll
                    for (int c = 0; c < NB; ++c) xx[j][c] = 0.0f;
                }
            }
        }
    };
    auto multiply = [&](const f32x4 (&gg)[8], const xvec (&xx)[8]) {
        Limbs<3> xb[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            float e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = xx[j][b];
            split8(e, xb[b]);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { e[j] = gg[j][a]; bs[a] += e[j]; }
            Limbs<3> ga;
            split8(e, ga);
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int p_ = 0; p_ < 6; ++p_) acc[a][b] = mfma_bf16(ga.v[limb_w(3, p_)], xb[b].v[limb_x(3, p_)], acc[a][b]);
        }
    };
    if (q0 < q1) {
        load(q0, g, x);
        for (long q = q0; q < q1; q += 16) {
            const bool more = q + 16 < q1;
            if (more) load(q + 16, gn, xn);          // in flight while the current 16 rows are multiplied
            multiply(g, x);
            if (more) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { g[j] = gn[j]; x[j] = xn[j]; }
            }
        }
    }
    // the bias sums first (they die here: the flush below needs every vector register); lane indices recomputed, not kept across the loop
    const int lane2 = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)), i2 = lane2 & 31, kh2 = lane2 >> 5;
    if (jb.b_off >= 0) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float v = bs[a] + __shfl_xor(bs[a], 32);
            if (kh2 == 0) unsafeAtomicAdd(grad + jb.b_off + 4 * i2 + a, v);
        }
    }
    // acc[a][b][r]: out = 4 * ((r&3) + 8(r>>2) + 4kh) + a,  column = NB * i + b.   Sum the 4 waves in LDS (same element per lane in every wave).
    constexpr int TW = 32 * NB;
    // (ONE copy of the flush: unrolled four times hipcc moved all 256 accumulators out of their AGPRs ahead of the four copies and spilled 33)
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
            // (an LDS pointer with compile-time offsets: as a generic pointer the 64 addresses were computed as 64-bit values ahead of the
            //  stores and 35 of them spilled -- the kernel's only scratch traffic)
            typedef __attribute__((address_space(3))) xvec* lds_xvec_p;
            const lds_xvec_p tb = (lds_xvec_p)(tile + 16 * kh2 * TW + NB * i2);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int out0 = 4 * ((r & 3) + 8 * (r >> 2)) + a;          // out = out0 + 16 kh
                    lds_xvec_p t = tb + out0 * (TW / NB);
                    xvec v;
#pragma unroll
                    for (int b = 0; b < NB; ++b) v[b] = acc[a][b][r];
                    if (w > 0) v += *t;
                    *t = v;
                }
        }
        __syncthreads();
    }
    for (int idx = wave * 64 + lane2; idx < 128 * TW; idx += WG_TPB) {
        const int out = idx / TW, col = jb.col0 + (idx % TW);
        if (col < jb.in_total) unsafeAtomicAdd(grad + jb.w_off + out * jb.in_total + col, tile[idx]);
    }
}

template <int NB>
__global__ __launch_bounds__(WG_TPB, 1) void decoder_wgrad_limb_kernel(WJobs jobs, int job0, long P, int slab, float* __restrict__ grad) {
    __shared__ __attribute__((aligned(16))) float tile[128 * 32 * NB];
    NVSR_RACE_PROBE_DELAY(tile);      // (probe builds only, nvsr_common.h)
    wgrad_limb_block<NB>(jobs.j[job0 + blockIdx.y], P, slab, tile, grad);
}

// fc_alpha / fc_rgb: dW[k][f] = sum_q g4[q][k] * H3[q][f], db[k] = sum_q g4[q][k].  HBM-bound: 1 040 bytes of record per point.  A wave reads
// ONE row of both branches per step as 16-byte loads -- lanes 0..31 the density row, lanes 32..63 the rgb row, 4 features per lane -- and the
// workgroup's 8 waves take rows q0 + wave, q0 + wave + 8, ...; their sums meet in LDS and the workgroup adds its 516 partial sums once
// (every workgroup adds into the same 516 addresses).  (Round 3: one feature per thread with 4-byte loads ran at 2.3 TB/s.)
constexpr int HW_WAVES = 8;
__global__ __launch_bounds__(64 * HW_WAVES) void head_wgrad_kernel(const float* __restrict__ Hd3, const float* __restrict__ Hr3,
                                                                   const float* __restrict__ g4, long Pp, int rows, float* __restrict__ grad) {
    __shared__ float part[HW_WAVES][12][64];
    __shared__ float bsum[HW_WAVES][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, rgb = lane >> 5, f4 = (lane & 31) * 4;
    const float* H = (rgb ? Hr3 : Hd3) + f4;
    const long q0 = (long)blockIdx.x * rows;
    const long q1 = (q0 + rows < Pp) ? q0 + rows : Pp;
    float a[3][4] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
    float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 4
    for (long q = q0 + wave; q < q1; q += HW_WAVES) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(g4 + 4 * q);          // (wave-uniform address)
        const f32x4 hv = *reinterpret_cast<const f32x4*>(H + q * HID);
        // density lanes: the alpha head (g[3]) in slot 0; rgb lanes: the three colour heads
        const float c0 = rgb ? g[0] : g[3];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[0][i] = fmaf(c0, hv[i], a[0][i]);
            a[1][i] = fmaf(g[1], hv[i], a[1][i]);
            a[2][i] = fmaf(g[2], hv[i], a[2][i]);
        }
        s[0] += g[0]; s[1] += g[1]; s[2] += g[2]; s[3] += g[3];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) part[wave][k * 4 + i][lane] = a[k][i];
    if (lane == 0) { bsum[wave][0] = s[0]; bsum[wave][1] = s[1]; bsum[wave][2] = s[2]; bsum[wave][3] = s[3]; }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 1; w < HW_WAVES; ++w) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) a[k][i] += part[w][k * 4 + i][lane];
        if (lane == 0) { s[0] += bsum[w][0]; s[1] += bsum[w][1]; s[2] += bsum[w][2]; s[3] += bsum[w][3]; }
    }
    if (rgb) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) unsafeAtomicAdd(grad + N_FCRGB_W + k * HID + f4 + i, a[k][i]);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) unsafeAtomicAdd(grad + N_ALPHA_W + f4 + i, a[0][i]);
    }
    if (lane == 0) {
        unsafeAtomicAdd(grad + N_FCRGB_B, s[0]); unsafeAtomicAdd(grad + N_FCRGB_B + 1, s[1]); unsafeAtomicAdd(grad + N_FCRGB_B + 2, s[2]);
        unsafeAtomicAdd(grad + N_ALPHA_B, s[3]);
    }
}

}  // namespace nvsr

using namespace nvsr;

extern "C" int nvsr_internal_resolve_decoder_arith(int arithmetic);      // render.hip
extern "C" int nvsr_decoder_weight_grad(int64_t N, int S, const float* record, float* grad_natural, nvsr_stream_t stream) {
    return nvsr_decoder_weight_grad_arith(N, S, record, grad_natural, NVSR_ARITH_INHERIT, stream);
}
extern "C" int nvsr_decoder_weight_grad_arith(int64_t N, int S, const float* record, float* grad_natural, int arithmetic, nvsr_stream_t stream) {
    const int arith = nvsr_internal_resolve_decoder_arith(arithmetic);
    if (arith < 0) return NVSR_ERR_SHAPE;
    if (!record || !grad_natural) return NVSR_ERR_NULL;
    if (!aligned16(record)) return NVSR_ERR_ALIGN;
    if (N < 0 || S < 1 || S > 4096) return NVSR_ERR_SHAPE;
    if (N == 0) return NVSR_OK;
    const DecRecord rec = make_record(const_cast<float*>(record), (long)N, S);
    const long Pp = rec.Pp, P = rec.P;
    const long LP = (long)HID * Pp;
    WJobs jobs;
    int n = 0;
    auto add = [&](const float* G, const float* X, int xstride, int col0, int w_off, int in_total, int b_off, int nb) {
        jobs.j[n++] = WJob{G, X, xstride, col0, w_off, in_total, b_off, nb};
    };
    if (arith != NVSR_ARITH_F32) {
        // limb kernel: 9 [128 x 128] blocks
        for (int l = 1; l <= 3; ++l) {
            const int wd = N_DEN_W1 + (l - 1) * N_HID_STRIDE, wr = N_RGB_W1 + (l - 1) * N_HID_STRIDE;
            add(rec.Gd + l * LP, rec.Hd + (l - 1) * LP, HID, 0, wd, HID, wd + HID * HID, 4);
            add(rec.Gr + l * LP, rec.Hr + (l - 1) * LP, HID, 0, wr, HID, wr + HID * HID, 4);
        }
        add(rec.Gr, rec.Xr, 4 * C, 0, N_RGB_W0, 4 * C, N_RGB_B0, 4);
        // The two 64-column leftovers (columns 128..191 of rgb layer 0, density layer 0) as whole [128 x 128] blocks too: their upper 64
        // columns read past the row into the next row of the record (always inside the allocation: Xd is followed by Hd, Xr by Hr) and are
        // dropped at the flush (col >= in_total) -- twice the MFMAs for those two blocks, but the contraction is bound by reading G, which
        // each of them reads once either way, and 9 x 28 = 252 workgroups are one round of the chip.  (As a second, short launch of
        // NB = 2 blocks they cost 0.19 ms per pass: 1.27 -> 1.14 ms for contraction + heads of a fine pass on the same box.)
        add(rec.Gr, rec.Xr, 4 * C, 128, N_RGB_W0, 4 * C, -1, 4);
        add(rec.Gd, rec.Xd, 64, 0, N_DEN_W0, C, N_DEN_B0, 4);
        long s4 = (P + 27) / 28;
        s4 = ((s4 + 255) / 256) * 256;
        hipLaunchKernelGGL(decoder_wgrad_limb_kernel<4>, dim3((unsigned)((P + s4 - 1) / s4), 9), dim3(WG_TPB), 0, (hipStream_t)stream, jobs, 0, P,
                           (int)s4, grad_natural);
    } else {
        add(rec.Gd, rec.Xd, 64, 0, N_DEN_W0, C, N_DEN_B0, 2);
        for (int l = 1; l <= 3; ++l)
            for (int c = 0; c < 2; ++c) {
                const int w = N_DEN_W1 + (l - 1) * N_HID_STRIDE;
                add(rec.Gd + l * LP, rec.Hd + (l - 1) * LP, HID, 64 * c, w, HID, c == 0 ? w + HID * HID : -1, 2);
            }
        for (int c = 0; c < 3; ++c) add(rec.Gr, rec.Xr, 4 * C, 64 * c, N_RGB_W0, 4 * C, c == 0 ? N_RGB_B0 : -1, 2);
        for (int l = 1; l <= 3; ++l)
            for (int c = 0; c < 2; ++c) {
                const int w = N_RGB_W1 + (l - 1) * N_HID_STRIDE;
                add(rec.Gr + l * LP, rec.Hr + (l - 1) * LP, HID, 64 * c, w, HID, c == 0 ? w + HID * HID : -1, 2);
            }
        if (n != WJOBS) return NVSR_ERR_SHAPE;
        // slabs: ~32 per layer block keeps 512 workgroups in flight (2 per CU) while each block is flushed only 32 times
        long slab = (P + 31) / 32;
        slab = ((slab + 255) / 256) * 256;
        if (slab < 256) slab = 256;
        const unsigned nslabs = (unsigned)((P + slab - 1) / slab);
        hipLaunchKernelGGL(decoder_wgrad_kernel, dim3(nslabs, WJOBS), dim3(WG_TPB), 0, (hipStream_t)stream, jobs, P, (int)slab, grad_natural);
    }
    const int hrows = 1024;       // rows per workgroup (8 waves x 128 rows): 512 workgroups at 524k rows
    hipLaunchKernelGGL(head_wgrad_kernel, dim3((unsigned)((P + hrows - 1) / hrows)), dim3(64 * HW_WAVES), 0, (hipStream_t)stream, rec.Hd + 3 * LP,
                       rec.Hr + 3 * LP, rec.g4, P, hrows, grad_natural);
    return NVSR_CHECK_LAUNCH();
}
