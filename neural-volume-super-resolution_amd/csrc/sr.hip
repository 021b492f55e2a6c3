// Feature-plane super-resolution CNN (EDSR wrapped by PlanesSR) for gfx950.
//
// Reference functions replaced (upstream paths): _Residual_Block.forward models.py:777-786, EDSR.forward models.py:818-822,
// PlanesSR.forward models.py:884-926 (interpolate_LR :858-859).
//
// conv3x3 (valid, stride 1, no bias) is an implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32):
//     D[co 32-block][32 pixels of one output row] += W[co][k] * X[k][pixel],   k = (ci, ky, kx)
// A = weights (pre-packed fragments), B = input pixels read from an LDS patch (NCHW keeps the 32 pixels of a row
// contiguous, so B reads are conflict-free ds_read_b32 and every input row is reused for the 3 ky taps from registers).
// Input channels stream through LDS 4 at a time (36 k = 18 MFMA k-steps) by LDS-DMA, double-buffered, one barrier per
// chunk.  A wave owns 2 co-blocks x 4 pixel-blocks (128 accumulator registers); a workgroup = 8 waves arranged
// CO_WAVES x PX_WAVES: (4,2) = 256 channels x (8 rows x 32 cols) for the wide layers, (1,8) = 64 channels x (32 x 32) for the
// narrow heads.  Epilogues fuse ReLU, the residual block's x0.1 + centre-cropped identity, and PixelShuffle(2).
#include <atomic>
#include <type_traits>

#include <cstdlib>
#include <cstring>

#include "limb_core.h"
#include "sr_core.h"

namespace nvsr {

#ifndef NVSR_CONV_WIDE_4x1
#define NVSR_CONV_WIDE_4x1 1     // 256-channel layers: 1 = two independent 4-wave workgroups per CU (4 rows each), 0 = one 8-wave workgroup (8 rows)
#endif


struct ConvParams {
    const float* in;      // [Cin][H-2pad][W-2pad]
    const float* wpk;     // packed weights [chunk][cb][t][lane]
    const unsigned* wpk_limb;   // bf16-limb fragments behind them (conv_limb_eligible layers), else NULL
    const unsigned* wpk_limb16; // 16x16x32 fragments behind those (conv_limb16_eligible layers), else NULL
    const unsigned* wpk_f16;    // 16x16x32 fragments of the 2-f16-limb arithmetic behind those (same layers), else NULL
    const unsigned* absmax;     // f16 limbs, data gradient: bits of max |in| over the whole input (absmax_kernel) -> the input's power-of-two scale; else NULL
    float* out;           // [Cout][H-2][W-2]  (pixel shuffle: [Cout/4][2(H-2)][2(W-2)])
    const float* skip;    // EPI_RESIDUAL: identity [Cout][H+2][W+2] (block input); EPI_MASK_SCALE: forward activation
                          // [Cout][H-2][W-2] whose sign gates the result; EPI_ADD_CENTER: [Cout][H-6][W-6] added to the centre
    int Cin, Cout, H, W;  // H, W = logical input size, INCLUDING the virtual zero border of `pad` pixels
    int ncb_total;        // padded co-blocks in wpk (multiple of 2)
    int nchunks;          // ceil(Cin/4)
    int epilogue;
    int pad;              // 0, or 2 for the data gradient of a valid conv ("full" correlation with the flipped kernel)
    int ncg;              // co-groups of the grid; blockIdx.z = batch * ncg + co-group
    long in_bs, out_bs, skip_bs;   // floats between consecutive planes of a batch (the 3 planes of a scene share the weights)
    unsigned* out_absmax;          // NULL, or a word that receives (atomicMax) the bits of max |out| over everything this launch writes: the
                                   // power-of-two scale of the NEXT f16-limb gradient launches that read `out` (no separate reduction pass)
};

__device__ float g_zero_word[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // source of the virtual border

// Epilogue shared by the conv kernels: acc[cb][pb] = 32 output channels (co0 + 32 cb + C/D row) x the 32 pixels of output row y0 + pb.
// (one fully unrolled copy per epilogue kind: with the kind tested inside, the unroller gives up and the accumulators are
//  indexed dynamically, i.e. go through scratch)
template <int PB, int NCB = 2>
__device__ __forceinline__ float conv_write_out(const ConvParams& p, const f32x16 (&acc)[NCB][PB], int x, int y0, int co0, int h, int Ho, int Wo) {
    float amax = 0.0f;          // largest magnitude this lane writes (-> p.out_absmax, conv_publish_absmax)
    // Round 3: per (co-block, row) the 16 channel rows of a lane are handled as a GROUP -- all 16 skip values are loaded first (independent
    // loads, one wait), then 16 stores -- through restrict-qualified local pointers with 32-bit element offsets (a plane is < 2^31 floats;
    // the batch offset is already in the pointers).  Before, every element was its own load -> s_waitcnt vmcnt(0) -> store behind 64-bit
    // multiply-adds (hipcc could not prove that out and skip do not alias): 128 serial memory round trips per lane in the residual blocks.
    float* __restrict__ const out = p.out;
    const float* __restrict__ const skip = p.skip;
    const int Cout = p.Cout;
    auto write_out_a = [&](auto kind, auto aligned) {
        constexpr int EPI = decltype(kind)::value;
        constexpr int ALIGN = decltype(aligned)::value;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                const int y = y0 + pb;
                if (!(y < Ho && x < Wo)) continue;
                const int cbase = co0 + cb * 32 + 4 * h;                    // channel of register r: cbase + (r & 3) + 8 (r >> 2)
                // channel in range?  Cout a multiple of 32 (every layer of EDSR but the last): the whole co-block is or is not, no
                // per-element test; a multiple of 8 (the 48-channel output layer): the same answer for the 8 channels
                // 8 (r >> 2) + {0..3} + 4h of a register quad in both lane halves, a wave-uniform test; otherwise per lane
                if (ALIGN == 32 && __builtin_amdgcn_readfirstlane(co0) + cb * 32 >= Cout) continue;
                auto chan_ok = [&](int r) {
                    if (ALIGN == 32) return true;
                    if (ALIGN == 8) return __builtin_amdgcn_readfirstlane(co0) + cb * 32 + 8 * (r >> 2) < Cout;
                    return cbase + (r & 3) + 8 * (r >> 2) < Cout;
                };
                float v[16];
                float sk[16];
                if (EPI == EPI_RESIDUAL || EPI == EPI_MASK_SCALE || EPI == EPI_ADD_CENTER) {
                    // skip layouts: RESIDUAL [Cout][Ho+4][Wo+4] at (y+2, x+2); MASK_SCALE [Cout][Ho][Wo]; ADD_CENTER [Cout][Ho-4][Wo-4] at (y-2, x-2)
                    const int sH = EPI == EPI_RESIDUAL ? Ho + 4 : EPI == EPI_MASK_SCALE ? Ho : Ho - 4;
                    const int sW = EPI == EPI_RESIDUAL ? Wo + 4 : EPI == EPI_MASK_SCALE ? Wo : Wo - 4;
                    const int sy = EPI == EPI_RESIDUAL ? y + 2 : EPI == EPI_MASK_SCALE ? y : y - 2;
                    const int sx = EPI == EPI_RESIDUAL ? x + 2 : EPI == EPI_MASK_SCALE ? x : x - 2;
                    const bool sin = sy >= 0 && sy < sH && sx >= 0 && sx < sW;     // (only ADD_CENTER can be outside)
                    const int splane = sH * sW, s0 = cbase * splane + sy * sW + sx;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dc = (r & 3) + 8 * (r >> 2);
                        // (unconditional load from a clamped address + select: no branch per element)
                        const bool ok = sin && chan_ok(r);
                        const float ld = skip[ok ? s0 + dc * splane : 0];
                        sk[r] = ok ? ld : 0.0f;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float t = acc[cb][pb][r];
                    if (EPI == EPI_RELU) t = fmaxf(t, 0.0f);
                    if (EPI == EPI_RESIDUAL) t = t * 0.1f + sk[r];     // output *= 0.1; output = output + identity[..., 2:-2, 2:-2]  (models.py:781-785)
                    if (EPI == EPI_MASK_SCALE) t = (sk[r] > 0.0f) ? t * 0.1f : 0.0f;   // backward of (x0.1) o conv2 o ReLU: gate by the forward activation
                    if (EPI == EPI_ADD_CENTER) t += sk[r];             // backward of the cropped identity: the block's output gradient lands in the centre
                    v[r] = t;
                    if (chan_ok(r)) amax = fmaxf(amax, fabsf(t));
                }
                if (EPI == EPI_PIXEL_SHUFFLE) {
                    // co -> (co >> 2, 2y + ((co >> 1) & 1), 2x + (co & 1)); cbase is a multiple of 4: r & 3 = co & 3
                    const int oplane = 4 * Ho * Wo, o0 = (cbase >> 2) * oplane + (2 * y) * (2 * Wo) + 2 * x;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dc = (r & 3) + 8 * (r >> 2);
                        if (chan_ok(r)) out[o0 + (dc >> 2) * oplane + ((r >> 1) & 1) * (2 * Wo) + (r & 1)] = v[r];
                    }
                } else {
                    const int oplane = Ho * Wo, o0 = cbase * oplane + y * Wo + x;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dc = (r & 3) + 8 * (r >> 2);
                        if (chan_ok(r)) out[o0 + dc * oplane] = v[r];
                    }
                }
            }
    };
    auto write_out = [&](auto kind) {
        if ((Cout & 31) == 0) write_out_a(kind, std::integral_constant<int, 32>{});
        else if ((Cout & 7) == 0) write_out_a(kind, std::integral_constant<int, 8>{});
        else write_out_a(kind, std::integral_constant<int, 1>{});       // (no layer of EDSR: the per-lane channel test)
    };
    switch (p.epilogue) {
        case EPI_RELU: write_out(std::integral_constant<int, EPI_RELU>{}); break;
        case EPI_RESIDUAL: write_out(std::integral_constant<int, EPI_RESIDUAL>{}); break;
        case EPI_PIXEL_SHUFFLE: write_out(std::integral_constant<int, EPI_PIXEL_SHUFFLE>{}); break;
        case EPI_MASK_SCALE: write_out(std::integral_constant<int, EPI_MASK_SCALE>{}); break;
        case EPI_ADD_CENTER: write_out(std::integral_constant<int, EPI_ADD_CENTER>{}); break;
        default: write_out(std::integral_constant<int, EPI_NONE>{}); break;
    }
    return amax;
}
// p.out_absmax <- max over the wave of the lanes' largest written magnitude (one atomic per wave; fmaxf drops a NaN like absmax_kernel does)
__device__ __forceinline__ void conv_publish_absmax(const ConvParams& p, float m) {
    if (!p.out_absmax) return;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(p.out_absmax, __float_as_uint(m));
}

// Which tile a workgroup takes.  Workgroup b (linear over the grid) runs on XCD b % 8 (round-robin dispatch) and each XCD has its own
// L2: XCD x gets the x-th contiguous eighth of the tile list (column chunk fastest, then row tile, then plane / co-group), so that tiles
// that share halo rows -- a PB = 4 tile reads 6 input rows, two of them its neighbour's -- and the layer's weight fragments meet in one
// L2.  Measured on the SR stage of a scene (all 70 layers): FETCH_SIZE 45.6 -> 25.7 (raw units; x 2 on gfx950: 91 -> 51 GB), 31.4 -> 31.6
// planes/s.  -DCV_XCD_ORDER=0 restores the plain order.
#ifndef CV_XCD_ORDER
#define CV_XCD_ORDER 1
#endif
// the same mapping as a linear index into the tile list (the limb kernel decodes it itself: co-group fastest)
__device__ __forceinline__ unsigned conv_tile_index() {
    const unsigned nb = gridDim.x * gridDim.y * gridDim.z, b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
#if CV_XCD_ORDER
    const unsigned xcd = b & 7u, per = nb >> 3, rem = nb & 7u;
    return xcd * per + (xcd < rem ? xcd : rem) + (b >> 3);
#else
    return b;
#endif
}
__device__ __forceinline__ void conv_tile_of_block(unsigned& bx, unsigned& by, unsigned& bz) {
#if CV_XCD_ORDER
    const unsigned nb = gridDim.x * gridDim.y * gridDim.z, b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned xcd = b & 7u, per = nb >> 3, rem = nb & 7u;
    const unsigned blk = xcd * per + (xcd < rem ? xcd : rem) + (b >> 3);
    bx = blk % gridDim.x; by = (blk / gridDim.x) % gridDim.y; bz = blk / (gridDim.x * gridDim.y);
#else
    bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
#endif
}

// The tensors and the logical size of plane `bi` of the launch -> p; false: this workgroup's tile (x0, y0) lies outside the plane (ragged batch:
// the grid covers the largest plane) and the workgroup has nothing to do.  bi, x0, y0 are workgroup-uniform.
// ragged launch: plane of linear tile `blk`, the tile's index inside the plane's share of the list (-> local), the plane's pixel-tile grid
__device__ __forceinline__ unsigned conv_ragged_locate(const ConvRagged& rag, unsigned blk, int rows, unsigned& local, unsigned& gx, unsigned& npt) {
    const unsigned bi = (blk >= rag.tile0[1] ? 1u : 0u) + (blk >= rag.tile0[2] ? 1u : 0u) + (blk >= rag.tile0[3] ? 1u : 0u);   // (tile0[k >= n] = ~0)
#define CONV_RAG_PICK(A) (bi == 3u ? rag.A[3] : bi == 2u ? rag.A[2] : bi == 1u ? rag.A[1] : rag.A[0])
    local = blk - CONV_RAG_PICK(tile0);
    const int H = CONV_RAG_PICK(H), W = CONV_RAG_PICK(W);
#undef CONV_RAG_PICK
    gx = (unsigned)(W - 2 + 31) / 32u;
    npt = gx * ((unsigned)(H - 2 + rows - 1) / (unsigned)rows);
    return bi;
}
__device__ __forceinline__ bool conv_select_plane(ConvParams& p, const ConvRagged& rag, unsigned bi, int x0, int y0) {
    if (rag.n) {
        // (constant indices + selects: a dynamic index into the by-value parameter struct would put the whole struct into scratch)
#define CONV_RAG_PICK(A) (bi == 3u ? rag.A[3] : bi == 2u ? rag.A[2] : bi == 1u ? rag.A[1] : rag.A[0])
        const float* in = CONV_RAG_PICK(in);
        float* out = CONV_RAG_PICK(out);
        const float* skip = CONV_RAG_PICK(skip);
        const int H = CONV_RAG_PICK(H), W = CONV_RAG_PICK(W);
#undef CONV_RAG_PICK
        p.in = in; p.out = out; p.skip = skip; p.H = H; p.W = W;
        return y0 < H - 2 && x0 < W - 2;
    }
    p.in += bi * p.in_bs;
    p.out += bi * p.out_bs;
    if (p.skip) p.skip += bi * p.skip_bs;
    return true;
}

// PB = output rows (32-pixel blocks) per wave: 4 by default; the launcher picks 3 or 2 for a layer whose tile count would otherwise leave most
// of the last workgroup round empty (a ~270^2 plane is 1.2 rounds of 4-row tiles on 512 workgroup slots, but 1.0 rounds of 3-row tiles... )
template <int CO_WAVES, int PX_WAVES, int PB = 4>
__global__ __launch_bounds__(CO_WAVES * PX_WAVES * 64, 2) void conv3x3_kernel(ConvParams p) {
    constexpr int CONV_WAVES = CO_WAVES * PX_WAVES, CONV_TPB = CONV_WAVES * 64;
    constexpr int NCB = CO_WAVES * 2;            // co-blocks per workgroup
    constexpr int ROWS = PX_WAVES * PB;          // output rows per workgroup tile
    constexpr int PR = ROWS + 2, PC = 34;        // input patch rows / cols (halo)
    constexpr int P_FLOATS = 4 * PR * PC;
    constexpr int P_PAD = (P_FLOATS + 63) / 64 * 64;
    constexpr int W_FLOATS = NCB * FRAG_FLOATS;
    constexpr int BUF = W_FLOATS + P_PAD;
    constexpr int P_ITERS = (P_FLOATS + CONV_TPB - 1) / CONV_TPB;
    constexpr int W_BLOCKS = W_FLOATS / 256;     // 1-KiB DMA blocks
    constexpr int W_ITERS = (W_BLOCKS + CONV_WAVES - 1) / CONV_WAVES;
    __shared__ __attribute__((aligned(16))) float lds[2 * BUF];
    NVSR_RACE_PROBE_DELAY(lds);      // (probe builds only, nvsr_common.h)

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int cw = wave / PX_WAVES, rg = wave % PX_WAVES;   // co-wave, pixel-row group
    const int Ho = p.H - 2, Wo = p.W - 2;
    unsigned bx, by, bz;
    conv_tile_of_block(bx, by, bz);
    const int x0 = bx * 32, y0 = by * ROWS, cg = bz % p.ncg, bi = bz / p.ncg;
    p.in += bi * p.in_bs;
    p.out += bi * p.out_bs;
    if (p.skip) p.skip += bi * p.skip_bs;
    const int Hr = p.H - 2 * p.pad, Wr = p.W - 2 * p.pad;   // the tensor in memory
    const long HW = (long)Hr * Wr;

    // per-thread source offsets of the patch elements it DMA-copies (spatial part; the channel part changes per chunk)
    int p_sp[P_ITERS], p_cl[P_ITERS];
#pragma unroll
    for (int i = 0; i < P_ITERS; ++i) {
        const int e = (i * CONV_WAVES + wave) * 64 + lane;
        const int cl = e / (PR * PC), rem = e - cl * (PR * PC), r = rem / PC, c = rem - r * PC;
        p_cl[i] = (e < P_FLOATS) ? cl : -1;
        const int yr = min(y0 + r, p.H - 1) - p.pad, xr = min(x0 + c, p.W - 1) - p.pad;
        p_sp[i] = (yr >= 0 && yr < Hr && xr >= 0 && xr < Wr) ? yr * Wr + xr : -1;     // -1: virtual zero border
    }
    const float* wsrc = p.wpk + (long)cg * NCB * FRAG_FLOATS;
    const long wchunk_stride = (long)p.ncb_total * FRAG_FLOATS;

    auto issue = [&](int chunk, int buf) {
        float* wl = lds + buf * BUF;
        float* pl = wl + W_FLOATS;
        const char* g = reinterpret_cast<const char*>(wsrc + chunk * wchunk_stride);
#pragma unroll
        for (int i = 0; i < W_ITERS; ++i) {
            const int blk = i * CONV_WAVES + wave;
            if (blk < W_BLOCKS)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + blk * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(wl + blk * 256), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < P_ITERS; ++i) {
            if (p_cl[i] >= 0) {
                const int ci = min(chunk * 4 + p_cl[i], p.Cin - 1);   // padded channels carry zero weights
                const float* src = (p_sp[i] >= 0) ? p.in + ci * HW + p_sp[i] : g_zero_word;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(pl + (i * CONV_WAVES + wave) * 64), 4, 0, 0);
            }
        }
    };

    f32x16 acc[2][PB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < PB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    issue(0, 0);
    for (int chunk = 0; chunk < p.nchunks; ++chunk) {
        const int buf = chunk & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                      // chunk landed for everyone; the other buffer is free
        if (chunk + 1 < p.nchunks) issue(chunk + 1, buf ^ 1);
        const float* wl = lds + buf * BUF + (cw * 2) * FRAG_FLOATS + lane;
        const float* pl = lds + buf * BUF + W_FLOATS + (rg * PB) * PC + j;
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {      // lane half h works on channel c2 + 2h of the chunk
            const float* pc = pl + (c2 + 2 * h) * (PR * PC);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                float b[PB + 2];
#pragma unroll
                for (int r = 0; r < PB + 2; ++r) b[r] = pc[r * PC + kx];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int t = c2 * 9 + ky * 3 + kx;
                    const float a0 = wl[t * 64], a1 = wl[FRAG_FLOATS + t * 64];
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) {
                        acc[0][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b[pb + ky], acc[0][pb], 0, 0, 0);
                        acc[1][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b[pb + ky], acc[1][pb], 0, 0, 0);
                    }
                }
            }
        }
    }

    conv_publish_absmax(p, conv_write_out<PB>(p, acc, x0 + j, y0 + rg * PB, (cg * NCB + cw * 2) * 32, h, Ho, Wo));
}

// ---- the same conv on the bf16 matrix pipe: f32 operands as 3 exact bf16 limbs (limb_core.h), 6 MFMAs per product block -------------
// Workgroup = 4 waves = 256 output channels (2 co-blocks per wave) x PB output rows x 32 pixels; input channels stream 16 at a time
// (one K-block per tap).  The input patch is split into limbs ONCE when it is staged -- every staged value feeds 9 taps x 8 co-blocks --
// and kept in LDS as [limb][row][octet][col][8 bf16], so a B operand (8 channels of one pixel) is one ds_read_b128 and the 32 lanes of a
// half wave (32 neighbouring pixels, one octet) read 512 contiguous bytes: conflict-free.  (Rounds 1-2 kept [row][col][octet]: the lanes
// of a half wave then stride by 32 bytes and every 16-lane group of the read hits each bank twice -- SQ_LDS_BANK_CONFLICT was 50 % of the
// LDS-active cycles, profiles/r03_conv_issue_counters.txt.)
// Weight fragments are not staged: each wave streams its own two co-blocks from L2 (coalesced 1-KiB reads, 6 per tap).
// CO_WAVES = 4: the 4 waves take 4 x 2 output blocks of the same PB rows; CO_WAVES = 1 (layers with <= 64 output channels): the 4 waves
// take the same 2 output blocks of 4 consecutive groups of PB rows.
#ifndef CV16_WAVES
#define CV16_WAVES 4
#endif
#ifndef CV16_ROWCOST3
#define CV16_ROWCOST3 1.05   // cost of an output row in a 3-row / 2-row tile of conv3x3_limb16_kernel relative to a 4-row tile (measured: 4.68 / 4.47 ms, 4.93 / 4.47 ms)
#define CV16_ROWCOST2 1.10
#endif
#ifndef CV16_ROWCOST6
#define CV16_ROWCOST6 0.97   // f16 limbs: cost of an output row in a 6-row tile relative to a 4-row tile (measured on a 1024^2 layer: 2.547 / 2.636 ms)
#endif
#ifndef CV_USE_16X16X32
#define CV_USE_16X16X32 1   // limb layers with Cin % 32 == 0 and Cout % 128 == 0 (all of EDSR's trunk and up-sampling convolutions): 1 = conv3x3_limb16_kernel
#endif
#ifndef CV_WIDE_ROWS8
#define CV_WIDE_ROWS8 0     // wide layers: 1 = one output block x 8 rows per wave, 0 = two output blocks x 2..4 rows (equally fast; see conv3x3_limb_kernel)
#endif
#ifndef CV_COST_MODEL
#define CV_COST_MODEL 2     // rows per tile of the limb convolution: 0 round 1's model, 1 always 4 rows, 2 measured round / row costs (SR stage 31.5 / 32.1 / 32.3 planes/s)
#endif
#ifndef CV_ABLATE
#define CV_ABLATE 0     // variant builds of tools/conv_ablate.sh: 1 weight fragments of tap 0 only, 2 no patch loads, 4 no split + LDS writes
#endif
#ifndef CV_LDS_LAYOUT
#define CV_LDS_LAYOUT 1     // patch items in LDS: 1 = [row][octet][col] (conflict-free reads), 0 = rounds 1-2's [row][col][octet] (A/B builds)
#endif
#if CV_LDS_LAYOUT
#define CV_ITEM(R, O, COL) (((R) * 2 + (O)) * PC + (COL))
#else
#define CV_ITEM(R, O, COL) (((R) * PC + (COL)) * 2 + (O))
#endif
// CBW = output blocks per wave: 2 x PB rows (the default: a wave streams 6 KB of weight fragments per tap for 12 PB MFMAs), or 1 x 8 rows
// (round 3, rows_per_tile = 8): the same 128 accumulators and 48 MFMAs per tap on HALF the weight fragments -- the workgroup is 128 output
// channels x 8 rows x 32 pixels, two of them cover the 256 channels of a pixel tile and are neighbours in the tile list.  Built to test whether
// the weight stream (the kernel's largest mover of bytes) holds the kernel back: it does not -- 254.8 vs 257.2 TFLOP/s on a 1024^2 layer
// (16 full rounds), 86.2 vs 85.7 ms for the SR stage, FETCH_SIZE 26.8 vs 26.6 GB, same bits.  The kernel runs the matrix pipe at 81-82 % of its
// cycles and the chip clocks it at 1.76-1.78 GHz (power; profiles/r03_conv_issue_counters.txt): 0.82 x 1.77 / 2.4 = 0.60 is the fraction of the
// 2.4 GHz roof a launch without tile rounding reaches.
template <int PB, int CO_WAVES = 4, int CBW = 2>
__global__ __launch_bounds__(256, 2) void conv3x3_limb_kernel(ConvParams p, ConvRagged rag) {
    constexpr int PX_WAVES = 4 / CO_WAVES, ROWS = PX_WAVES * PB;
    constexpr int PR = ROWS + 2, PC = 34;
    constexpr int ITEMS = 2 * PR * PC;                    // (octet, row, col): 8 channels of one patch pixel
    constexpr int IT = (ITEMS + 255) / 256;
    constexpr int LIMB_WORDS = PR * PC * 2 * 4;           // one limb of the patch
    constexpr int BUF = 3 * LIMB_WORDS;
    __shared__ __attribute__((aligned(16))) unsigned lds[2 * BUF];
    NVSR_RACE_PROBE_DELAY(lds);      // (probe builds only, nvsr_common.h)

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, h = lane >> 5;
    // tile of this workgroup from the XCD-contiguous linear tile index (conv_tile_index): column chunk, row tile, then (plane, co-group) --
    // co-group SLOWEST for the 2-block wave tile: the workgroups that run together on an XCD then stream ONE co-group's 3.5 MB of weight
    // fragments through its 4 MB L2 (with the co-group fastest the four co-groups of a 256 -> 1024 layer keep 14 MB live and FETCH_SIZE of
    // those layers doubles: measured, 2.1 -> 4.1 GB); co-group FASTEST for the 1-block x 8-row wave tile, whose two co-groups share a patch and
    // together stream the same 3.5 MB
    unsigned blk = conv_tile_index();
    const unsigned ncg = (unsigned)p.ncg;
    unsigned npt = gridDim.x * gridDim.y, gx = gridDim.x, pt, bi, bi_r = 0;
    if (rag.n) bi_r = conv_ragged_locate(rag, blk, ROWS, blk, gx, npt);      // (blk becomes the index inside the plane's tiles)
    int cg;
    if (CBW == 1) { cg = (int)(blk % ncg); const unsigned q = blk / ncg; pt = q % npt; bi = q / npt; }
    else { pt = blk % npt; const unsigned q = blk / npt; cg = (int)(q % ncg); bi = q / ncg; }
    if (rag.n) bi = bi_r;
    const unsigned bx = pt % gx, by = pt / gx;
    const int x0 = bx * 32, y0 = by * ROWS;
    if (!conv_select_plane(p, rag, bi, x0, y0)) return;
    const int Ho = p.H - 2, Wo = p.W - 2;
    const int Hr = p.H - 2 * p.pad, Wr = p.W - 2 * p.pad;   // the tensor in memory
    const long HW = (long)Hr * Wr;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int cw = wave_u / PX_WAVES, rg = wave_u % PX_WAVES;   // co-wave, pixel-row group
    const int cb0 = (cg * CO_WAVES + cw) * CBW;                 // first of this wave's output blocks

    // staging items: every load is unconditional (clamped source, uniform channel base + one 32-bit lane offset); border pixels and the
    // threads past the last item are handled by ONE select / ONE store predicate per item
    int voff[IT], sl[IT];
    bool inside[IT], item[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int e = i * 256 + tid;
        item[i] = e < ITEMS;
        const int ee = item[i] ? e : 0;
        const int o = ee / (PR * PC), rem = ee - o * (PR * PC), r = rem / PC, c = rem - r * PC;
        const int yr = min(y0 + r, p.H - 1) - p.pad, xr = min(x0 + c, p.W - 1) - p.pad;
        inside[i] = yr >= 0 && yr < Hr && xr >= 0 && xr < Wr;
        voff[i] = o * 8 * (int)HW + (inside[i] ? yr * Wr + xr : 0);
        sl[i] = CV_ITEM(r, o, c) * 4;                  // [row][octet][col]: the 32 lanes of a half wave read 32 CONSECUTIVE 16-byte items
    }
    float st[IT][8];
    auto gload = [&](int chunk) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float* cbase = p.in + (long)(chunk * 16 + k) * HW;        // wave-uniform
#pragma unroll
            for (int i = 0; i < IT; ++i) st[i][k] = cbase[voff[i]];
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            Limbs<3> L;
            float e[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) e[k] = inside[i] ? st[i][k] : 0.0f;
            split8(e, L);
            if (item[i]) {
#pragma unroll
                for (int t = 0; t < 3; ++t) *reinterpret_cast<u32x4*>(lds + buf * BUF + t * LIMB_WORDS + sl[i]) = L.v[t];
            }
        }
    };

    f32x16 acc[CBW][PB];
#pragma unroll
    for (int a = 0; a < CBW; ++a)
#pragma unroll
        for (int b = 0; b < PB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    const int nchunks = p.Cin / 16;
    const u32x4* wbase = reinterpret_cast<const u32x4*>(p.wpk_limb) + ((long)cb0 * 27) * 64;   // wave-uniform
    const long wchunk = (long)p.ncb_total * 27 * 64;        // u32x4 per chunk
    gload(0);
    sstore(0);
    __syncthreads();
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int buf = chunk & 1;
        if (chunk + 1 < nchunks && !(CV_ABLATE & 2)) gload(chunk + 1);          // in flight during this chunk's MFMAs
        const u32x4* wa = wbase + chunk * wchunk;
        const unsigned* pl = lds + buf * BUF + CV_ITEM(0, h, j) * 4;
        // A fragments one tap ahead (the sched_barriers keep hipcc from hoisting every load of the chunk to its top: 95+ spills)
        u32x4 A[CBW][3], An[CBW][3];
#pragma unroll
        for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
            for (int t = 0; t < 3; ++t) A[cb][t] = wa[((cb * 9 + 0) * 3 + t) * 64 + lane];
        // B fragments of a tap's FIRST row come from the previous tap (round 3): the sched_barriers at the tap boundaries keep hipcc from
        // hoisting them, and every tap used to open with an exposed LDS round trip (9 per chunk)
        u32x4 B0[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) B0[t] = *reinterpret_cast<const u32x4*>(pl + t * LIMB_WORDS + CV_ITEM(rg * PB, 0, 0) * 4);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            __builtin_amdgcn_sched_barrier(0);
            if (tap + 1 < 9 && !(CV_ABLATE & 1)) {
#pragma unroll
                for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
                    for (int t = 0; t < 3; ++t) An[cb][t] = wa[((cb * 9 + tap + 1) * 3 + t) * 64 + lane];
            }
            // (pinned at the top of the tap: hipcc otherwise sinks these six loads to the tap's END, and the next tap's first MFMA waits
            //  out their L2 latency -- 9 times per chunk)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                u32x4 B[3];
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    B[t] = pb == 0 ? B0[t] : *reinterpret_cast<const u32x4*>(pl + t * LIMB_WORDS + CV_ITEM(rg * PB + pb + ky, 0, kx) * 4);
                if (pb == PB - 1 && tap + 1 < 9) {
                    const int ky1 = (tap + 1) / 3, kx1 = (tap + 1) % 3;
#pragma unroll
                    for (int t = 0; t < 3; ++t) B0[t] = *reinterpret_cast<const u32x4*>(pl + t * LIMB_WORDS + CV_ITEM(rg * PB + ky1, 0, kx1) * 4);
                }
#pragma unroll
                for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
                    for (int q = 0; q < 6; ++q) acc[cb][pb] = mfma_bf16(A[cb][limb_w(3, q)], B[limb_x(3, q)], acc[cb][pb]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (tap + 1 < 9 && !(CV_ABLATE & 1)) {
#pragma unroll
                for (int cb = 0; cb < CBW; ++cb)
#pragma unroll
                    for (int t = 0; t < 3; ++t) A[cb][t] = An[cb][t];
            }
        }
        if (chunk + 1 < nchunks && !(CV_ABLATE & 4)) sstore(buf ^ 1);
        __syncthreads();
    }
    conv_publish_absmax(p, conv_write_out<PB, CBW>(p, acc, x0 + j, y0 + rg * PB, cb0 * 32, h, Ho, Wo));
}

// ---- the same conv on v_mfma_f32_16x16x32_bf16 (round 3) -------------------------------------------------------------------------------------
// Why another shape: these kernels run at the chip's power limit, and bare MFMA streams at two waves per SIMD sustain 2 050 TFLOP/s with the
// 16x16x32 instruction against 1 793 with 32x32x16 on random operands (tools/mfma_power_roof.hip, profiles/r03_mfma_power_roof.txt): the
// smaller accumulator tile moves a quarter of the accumulator bytes per FLOP.
// D[16 co][16 pixels] += W[16 co][32 ci] x X[32 ci][16 pixels]: lane (i = l & 15, g = l >> 4) holds A = 8 input channels 8g..8g+7 of output
// channel i, B = the same 8 channels of pixel i, and D rows 4g..4g+3 (output channels) of column i (pixel).  Input channels stream 32 at a
// time; the patch sits in LDS as [limb][octet 0..3][row][col][8 bf16].  Wave = 2 co-blocks (32 channels) x PB rows x 2 half-rows of 16 pixels
// (64 accumulator registers at PB = 4), workgroup = 4 waves = 128 output channels of one PB x 32 pixel tile, two workgroups per CU.
// Per tap a wave reads 6 weight fragments (one tap ahead) and, per (row, half-row), 3 B fragments for 12 MFMAs.  Same limb products in the same
// order per (ci block, tap), but the K dimension of an instruction spans 32 channels instead of 16: the f32 accumulation order differs from
// conv3x3_limb_kernel's, results agree to rounding (both are held to the CPU checker of the test suite at 3e-5).
// F16 (the 2-f16-limb arithmetic): the accumulators carry 2^(F16_SW + F16_SX) and the ReLU lets a NaN through (an operand beyond the f16
// range turns the accumulators into NaNs; fmaxf would return 0)
template <int PB, bool F16 = false, int NCB = 2>
__device__ __forceinline__ float conv_write_out16(const ConvParams& p, const f32x4 (&acc)[NCB][PB][2], int x0, int y0, int co0, int lane, int Ho, int Wo,
                                                  float unscale = 1.0f) {
    float amax = 0.0f;
    float* __restrict__ const out = p.out;
    const float* __restrict__ const skip = p.skip;
    const int i = lane & 15, g = lane >> 4;
    auto write_out = [&](auto kind) {
        constexpr int EPI = decltype(kind)::value;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int pb = 0; pb < PB; ++pb)
#pragma unroll
                for (int hx = 0; hx < 2; ++hx) {
                    const int y = y0 + pb, x = x0 + 16 * hx + i;
                    if (!(y < Ho && x < Wo)) continue;
                    const int cbase = co0 + 16 * cb + 4 * g;                 // channel of register r: cbase + r
                    float sk[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (EPI == EPI_RESIDUAL || EPI == EPI_MASK_SCALE || EPI == EPI_ADD_CENTER) {
                        const int sH = EPI == EPI_RESIDUAL ? Ho + 4 : EPI == EPI_MASK_SCALE ? Ho : Ho - 4;
                        const int sW = EPI == EPI_RESIDUAL ? Wo + 4 : EPI == EPI_MASK_SCALE ? Wo : Wo - 4;
                        const int sy = EPI == EPI_RESIDUAL ? y + 2 : EPI == EPI_MASK_SCALE ? y : y - 2;
                        const int sx = EPI == EPI_RESIDUAL ? x + 2 : EPI == EPI_MASK_SCALE ? x : x - 2;
                        const bool sin = sy >= 0 && sy < sH && sx >= 0 && sx < sW;
                        const int splane = sH * sW, s0 = cbase * splane + sy * sW + sx;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float ld = skip[sin ? s0 + r * splane : 0];
                            sk[r] = sin ? ld : 0.0f;
                        }
                    }
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float t = acc[cb][pb][hx][r];
                        if (F16) t *= unscale;
                        if (EPI == EPI_RELU) t = F16 ? (t < 0.0f ? 0.0f : t) : fmaxf(t, 0.0f);
                        if (EPI == EPI_RESIDUAL) t = t * 0.1f + sk[r];
                        if (EPI == EPI_MASK_SCALE) t = (sk[r] > 0.0f) ? t * 0.1f : 0.0f;
                        if (EPI == EPI_ADD_CENTER) t += sk[r];
                        v[r] = t;
                        amax = fmaxf(amax, fabsf(t));
                    }
                    if (EPI == EPI_PIXEL_SHUFFLE) {        // co -> (co >> 2, 2y + ((co >> 1) & 1), 2x + (co & 1)); cbase is a multiple of 4
                        const int oplane = 4 * Ho * Wo, o0 = (cbase >> 2) * oplane + (2 * y) * (2 * Wo) + 2 * x;
#pragma unroll
                        for (int r = 0; r < 4; ++r) out[o0 + ((r >> 1) & 1) * (2 * Wo) + (r & 1)] = v[r];
                    } else {
                        const int oplane = Ho * Wo, o0 = cbase * oplane + y * Wo + x;
#pragma unroll
                        for (int r = 0; r < 4; ++r) out[o0 + r * oplane] = v[r];
                    }
                }
    };
    switch (p.epilogue) {
        case EPI_RELU: write_out(std::integral_constant<int, EPI_RELU>{}); break;
        case EPI_RESIDUAL: write_out(std::integral_constant<int, EPI_RESIDUAL>{}); break;
        case EPI_PIXEL_SHUFFLE: write_out(std::integral_constant<int, EPI_PIXEL_SHUFFLE>{}); break;
        case EPI_MASK_SCALE: write_out(std::integral_constant<int, EPI_MASK_SCALE>{}); break;
        case EPI_ADD_CENTER: write_out(std::integral_constant<int, EPI_ADD_CENTER>{}); break;
        default: write_out(std::integral_constant<int, EPI_NONE>{}); break;
    }
    return amax;
}

#ifndef CV16_BPF
#define CV16_BPF 2      // 16x16x32 kernel: B fragments read this many pixel blocks ahead of their MFMAs (0 = where the compiler puts them: a read 2-4
                        // MFMAs ahead of its use).  Same-box A/B on the EDSR trunk layer (256 -> 256, 270^2 x 3): 0.702 / 0.690 / 0.685 ms for 0 / 1 / 2;
                        // same bits (the order of the MFMAs of an accumulator does not change)
#endif
template <int LIMBS>
__device__ __forceinline__ f32x4 mfma16_limb(u32x4 a, u32x4 b, f32x4 c) {
    if constexpr (LIMBS == 2) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16_bf16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// WAVES = 4: 128 output channels per workgroup, two workgroups per CU; WAVES = 8 (experiment, -DCV16_WAVES=8): 256 channels per workgroup, one
// workgroup per CU -- the patch is staged once for all 256 channels
// LIMBS = 3: bf16 limbs (6 products); LIMBS = 2: f16 limbs rounded to nearest with the static scales of limb_core.h (3 products; forward
// convolutions only -- gradients span too many decades for 5 exponent bits)
// NCB = 16-channel output blocks per wave: 2 (a workgroup of 4 waves = 128 channels).  4 (256 channels: one workgroup per pixel tile reads the
// patch once, every B fragment read from LDS feeds 12 MFMAs instead of 6, half the LDS and L2 streams per MFMA) was measured in round 4 and is
// not instantiated: with 6 rows it needs 512 registers = one workgroup per CU and the EDSR(256 x 32) plane takes 20.4 instead of 17.0 ms; with
// 3 rows at two workgroups per CU (256 registers, 20 spilled) 18.4-18.8 against 17.3 ms (same box).  The second wave per SIMD is worth more
// than the halved streams: the kernel is bound by latency the other workgroup covers, not by LDS or L2 bandwidth.
template <int PB, int WAVES = 4, int LIMBS = 3, int NCB = 2>
__global__ __launch_bounds__(64 * WAVES, (WAVES == 4 && NCB == 2) ? 2 : 1) void conv3x3_limb16_kernel(ConvParams p, ConvRagged rag) {
    constexpr int TPB = 64 * WAVES;
    constexpr int PR = PB + 2, PC = 34;
    constexpr int ITEMS = 4 * PR * PC;                    // (octet, row, col): 8 channels of one patch pixel
    constexpr int IT = (ITEMS + TPB - 1) / TPB;
    // patch items (16 bytes = 8 channels of a pixel) as [octet][row][col], the octets a multiple of 16 items apart: the hardware serves a
    // ds_read_b128 in four groups of 16 lanes, each of which holds all 16 values of l & 15 (12 from one octet, 4 from its neighbour), so with
    // the bank quad of an item a function of the pixel alone every group is conflict-free ([row][octet][col] with 34-item rows measured
    // SQ_LDS_BANK_CONFLICT = 48 % of the LDS-active cycles)
    constexpr int OSTR = (PR * PC + 15) / 16 * 16;
    constexpr int LIMB_WORDS = 4 * OSTR * 4;              // one limb of the patch
    constexpr int BUF = LIMBS * LIMB_WORDS;
    __shared__ __attribute__((aligned(16))) unsigned lds[2 * BUF];
    NVSR_RACE_PROBE_DELAY(lds);      // (probe builds only, nvsr_common.h)
#define CV16_ITEM(R, O, COL) ((O) * OSTR + (R) * PC + (COL))

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i16 = lane & 15, g = lane >> 4;
    unsigned blk = conv_tile_index();
    // tile order: the two 128-channel co-groups of a 256-channel set FASTEST (they read the same patch: the second one's loads hit the XCD's L2;
    // their weights together are the 3.5 MB per layer that the 2-block kernel streamed), then the pixel tile, then the 256-channel set and the
    // plane slowest (the sets of a 256 -> 1024 layer would otherwise keep 14 MB of weights live per XCD).  Measured FETCH_SIZE of the SR stage:
    // 34.0 GB with all co-groups slowest
    const unsigned ncg = (unsigned)p.ncg;
    unsigned npt = gridDim.x * gridDim.y, gx = gridDim.x, bi_r = 0;
    if (rag.n) bi_r = conv_ragged_locate(rag, blk, PB, blk, gx, npt);        // (blk becomes the index inside the plane's tiles)
    const unsigned G = (WAVES == 8 || (ncg & 1u)) ? 1u : 2u, nset = ncg / G;
    const unsigned cg_lo = blk % G, r1 = blk / G, pt = r1 % npt, q = r1 / npt, bx = pt % gx, by = pt / gx;
    const int cg = (int)((q % nset) * G + cg_lo);
    const unsigned bi = rag.n ? bi_r : q / nset;
    const int x0 = bx * 32, y0 = by * PB;
    if (!conv_select_plane(p, rag, bi, x0, y0)) return;
    const int Ho = p.H - 2, Wo = p.W - 2;
    const int Hr = p.H - 2 * p.pad, Wr = p.W - 2 * p.pad;   // the tensor in memory
    const long HW = (long)Hr * Wr;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int cb0 = (cg * WAVES + wave_u) * NCB;                    // first of this wave's NCB 16-channel output blocks
    const int ncb16 = p.Cout / 16;

    int voff[IT], sl[IT];
    bool inside[IT], item[IT];
#pragma unroll
    for (int k = 0; k < IT; ++k) {
        const int e = k * TPB + tid;
        item[k] = e < ITEMS;
        const int ee = item[k] ? e : 0;
        const int o = ee / (PR * PC), rem = ee - o * (PR * PC), r = rem / PC, c = rem - r * PC;
        const int yr = min(y0 + r, p.H - 1) - p.pad, xr = min(x0 + c, p.W - 1) - p.pad;
        inside[k] = yr >= 0 && yr < Hr && xr >= 0 && xr < Wr;
        voff[k] = o * 8 * (int)HW + (inside[k] ? yr * Wr + xr : 0);
        sl[k] = CV16_ITEM(r, o, c) * 4;
    }
    // f16 limbs: the input's power-of-two scale -- static 2^F16_SX for activations; for a data gradient (p.absmax) the one that puts the largest
    // |dy| of the whole tensor into [2^12, 2^13): gradients span many decades from layer to layer and step to step, a tensor's values a few
    float xscale = F16_X_SCALE;
    if (LIMBS == 2 && p.absmax) {
        xscale = f16_gradient_scale((unsigned)__builtin_amdgcn_readfirstlane((int)*p.absmax));
    }

    float st[IT][8];
    auto gload = [&](int chunk) {
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) {
            const float* cbase = p.in + (long)(chunk * 32 + c8) * HW;        // wave-uniform
#pragma unroll
            for (int k = 0; k < IT; ++k) st[k][c8] = cbase[voff[k]];
        }
    };
    auto sstore_item = [&](int buf, int k) NVSR_INL {
        {
            Limbs<LIMBS> L;
            float e[8];
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) e[c8] = inside[k] ? (LIMBS == 2 ? st[k][c8] * xscale : st[k][c8]) : 0.0f;
            if constexpr (LIMBS == 3) split8(e, L);
            else split_all<2>([&](int i8) { return e[i8]; }, L);
            if (item[k]) {
#pragma unroll
                for (int t = 0; t < LIMBS; ++t) *reinterpret_cast<u32x4*>(lds + buf * BUF + t * LIMB_WORDS + sl[k]) = L.v[t];
            }
        }
    };
    auto sstore = [&](int buf) NVSR_INL {
#pragma unroll
        for (int k = 0; k < IT; ++k) sstore_item(buf, k);
    };

    f32x4 acc[NCB][PB][2];
#pragma unroll
    for (int a = 0; a < NCB; ++a)
#pragma unroll
        for (int b = 0; b < PB; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[a][b][c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const int nchunks = p.Cin / 32;
    const u32x4* wbase = reinterpret_cast<const u32x4*>(LIMBS == 2 ? p.wpk_f16 : p.wpk_limb16) + ((long)cb0 * 9 * LIMBS) * 64;     // wave-uniform
    const long wchunk = (long)ncb16 * 9 * LIMBS * 64;       // u32x4 per chunk
    gload(0);
    sstore(0);
    __syncthreads();
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int buf = chunk & 1;
        if (chunk + 1 < nchunks) gload(chunk + 1);          // in flight during this chunk's MFMAs
        const u32x4* wa = wbase + chunk * wchunk;
        const unsigned* pl = lds + buf * BUF + CV16_ITEM(0, g, i16) * 4;
        u32x4 A[NCB][LIMBS], An[NCB][LIMBS];
#if CV16_BPF
        u32x4 Bq[CV16_BPF + 1][LIMBS];
#endif
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int t = 0; t < LIMBS; ++t) A[cb][t] = wa[((cb * 9 + 0) * LIMBS + t) * 64 + lane];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            __builtin_amdgcn_sched_barrier(0);
            if (tap + 1 < 9) {
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                    for (int t = 0; t < LIMBS; ++t) An[cb][t] = wa[((cb * 9 + tap + 1) * LIMBS + t) * 64 + lane];
            }
            __builtin_amdgcn_sched_barrier(0);
#if CV16_BPF
            // B fragments CV16_BPF pixel blocks ahead of their MFMAs, through a ring of CV16_BPF + 1 register sets that runs across the taps of
            // the chunk (the compiler's own schedule issues a read 2-4 MFMAs ahead of its first use: less than the LDS latency)
            constexpr int NI = PB * 2, NJ = 9 * NI, RING = CV16_BPF + 1;
            auto loadB = [&](int j, u32x4 (&dst)[LIMBS]) NVSR_INL {
                const int tp = j / NI, it = j % NI, pb_ = it / 2, hx_ = it % 2;
#pragma unroll
                for (int t = 0; t < LIMBS; ++t)
                    dst[t] = *reinterpret_cast<const u32x4*>(pl + t * LIMB_WORDS + CV16_ITEM(pb_ + tp / 3, 0, 16 * hx_ + tp % 3) * 4);
            };
            if (tap == 0) {
#pragma unroll
                for (int j = 0; j < CV16_BPF; ++j) loadB(j, Bq[j % RING]);
            }
#pragma unroll
            for (int it = 0; it < NI; ++it) {
                const int j = tap * NI + it, pb = it / 2, hx = it % 2;
                if (j + CV16_BPF < NJ) loadB(j + CV16_BPF, Bq[(j + CV16_BPF) % RING]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                    for (int qq = 0; qq < limb_products(LIMBS); ++qq)
                        acc[cb][pb][hx] = mfma16_limb<LIMBS>(A[cb][limb_w(LIMBS, qq)], Bq[j % RING][limb_x(LIMBS, qq)], acc[cb][pb][hx]);
                __builtin_amdgcn_sched_barrier(0);
            }
#else
#pragma unroll
            for (int pb = 0; pb < PB; ++pb)
#pragma unroll
                for (int hx = 0; hx < 2; ++hx) {
                    u32x4 B[LIMBS];
#pragma unroll
                    for (int t = 0; t < LIMBS; ++t) B[t] = *reinterpret_cast<const u32x4*>(pl + t * LIMB_WORDS + CV16_ITEM(pb + ky, 0, 16 * hx + kx) * 4);
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                        for (int qq = 0; qq < limb_products(LIMBS); ++qq)
                            acc[cb][pb][hx] = mfma16_limb<LIMBS>(A[cb][limb_w(LIMBS, qq)], B[limb_x(LIMBS, qq)], acc[cb][pb][hx]);
                }
#endif
            __builtin_amdgcn_sched_barrier(0);
            if (tap + 1 < 9) {
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                    for (int t = 0; t < LIMBS; ++t) A[cb][t] = An[cb][t];
            }
        }
        // (spreading this store over the last taps, one patch item behind a queued group of MFMAs each, was measured: the live staging registers
        //  beside the fragment ring spill 22-26 VGPRs in the 6-row kernel, 0.662 -> 0.68 ms on the trunk layer)
        if (chunk + 1 < nchunks) sstore(buf ^ 1);
        __syncthreads();
    }
    conv_publish_absmax(p, conv_write_out16<PB, LIMBS == 2, NCB>(p, acc, x0, y0, cb0 * 16, lane, Ho, Wo, 1.0f / (F16_W_SCALE * xscale)));
#undef CV16_ITEM
}

// limb fragments for conv3x3_limb16_kernel: [chunk of 32 ci][cb16][tap][limb][lane][4 words]; lane (co = 16 cb + (l & 15), g = l >> 4) holds the
// 8 input channels 32 chunk + 8 g + 0..7 of tap `tap` as bf16 pairs (even channel in the low half)
template <int LIMBS>
__device__ __forceinline__ void pack_conv_limbs16_word(long idx, const float* __restrict__ w, unsigned* __restrict__ out, int Cin, int Cout, int transposed) {
    constexpr int FRAG = 9 * LIMBS * 256;
    const int ncb = Cout / 16;
    const int wd = idx & 3, lane = (idx >> 2) & 63, t = (int)((idx >> 8) % LIMBS), tap = (int)((idx / (256 * LIMBS)) % 9);
    const long rest = idx / FRAG;
    const int cb = (int)(rest % ncb), chunk = (int)(rest / ncb);
    const int co = 16 * cb + (lane & 15), g = lane >> 4;
    unsigned word = 0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int ci = 32 * chunk + 8 * g + 2 * wd + half;
        float v = transposed ? w[((long)ci * Cout + co) * 9 + (8 - tap)] : w[((long)co * Cin + ci) * 9 + tap];
        unsigned bits = 0;
        if constexpr (LIMBS == 2) {          // f16 limbs of W 2^F16_SW, both rounded to nearest (limb_core.h); a weight >= 255 packs as inf -> NaN outputs
            v *= F16_W_SCALE;
            const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
            bits = __builtin_bit_cast(unsigned short, t == 0 ? hi : lo);
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (k == t) bits = __float_as_uint(v) >> 16;
                v = limb_rest(v);
            }
        }
        word |= bits << (16 * half);
    }
    out[idx] = word;
}
template <int LIMBS = 3>
__global__ void pack_conv_limbs16_kernel(const float* __restrict__ w, unsigned* __restrict__ out, int Cin, int Cout, int transposed) {
    const long n = (long)(Cin / 32) * (Cout / 16) * (9 * LIMBS * 256);
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) pack_conv_limbs16_word<LIMBS>(idx, w, out, Cin, Cout, transposed);
}

// limb fragments of a conv's weights: [chunk of 16 ci][cb][tap][limb][lane][4 words]; lane (co = 32 cb + (l & 31), h = l >> 5) holds
// the 8 input channels 16 chunk + 8 h + 0..7 of tap `tap` as bf16 pairs (even channel in the low half)
__device__ __forceinline__ void pack_conv_limbs_word(long idx, const float* __restrict__ w, unsigned* __restrict__ out, int Cin, int Cout, int ncb, int transposed) {
    const int wd = idx & 3, lane = (idx >> 2) & 63, t = (int)((idx >> 8) % 3), tap = (int)((idx / 768) % 9);
    const long rest = idx / CL_FRAG_WORDS;
    const int cb = (int)(rest % ncb), chunk = (int)(rest / ncb);
    const int co = 32 * cb + (lane & 31), h = lane >> 5;
    unsigned word = 0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int ci = 16 * chunk + 8 * h + 2 * wd + half;
        float v = 0.0f;
        if (co < Cout && ci < Cin) v = transposed ? w[((long)ci * Cout + co) * 9 + (8 - tap)] : w[((long)co * Cin + ci) * 9 + tap];
        unsigned bits = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (k == t) bits = __float_as_uint(v) >> 16;
            v = limb_rest(v);
        }
        word |= bits << (16 * half);
    }
    out[idx] = word;
}
__global__ void pack_conv_limbs_kernel(const float* __restrict__ w, unsigned* __restrict__ out, int Cin, int Cout, int ncb, int transposed) {
    const long n = (long)(Cin / 16) * ncb * CL_FRAG_WORDS;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) pack_conv_limbs_word(idx, w, out, Cin, Cout, ncb, transposed);
}

// [Cout][Cin][3][3] -> [chunk][cb][t][lane].  transposed: the packed conv is the DATA GRADIENT of w's conv, i.e. it maps Cout
// channels to Cin channels with w'[ci][co][ky][kx] = w[co][ci][2-ky][2-kx]; (Cin, Cout) are then those of the packed conv.
__device__ __forceinline__ void pack_conv_word(long idx, const float* __restrict__ w, float* __restrict__ wpk, int Cin, int Cout, int ncb, int transposed) {
    const int lane = idx & 63, t = (int)((idx >> 6) % K_PER_CHUNK);
    const long rest = idx / FRAG_FLOATS;
    const int cb = (int)(rest % ncb), chunk = (int)(rest / ncb);
    const int co = 32 * cb + (lane & 31), ci = 4 * chunk + t / 9 + 2 * (lane >> 5), tap = t % 9;
    float v = 0.0f;
    if (co < Cout && ci < Cin) v = transposed ? w[((long)ci * Cout + co) * 9 + (8 - tap)] : w[((long)co * Cin + ci) * 9 + tap];
    wpk[idx] = v;
}
__global__ void pack_conv_kernel(const float* __restrict__ w, float* __restrict__ wpk, int Cin, int Cout, int ncb, int nchunks, int transposed) {
    const long n = (long)nchunks * ncb * FRAG_FLOATS;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) pack_conv_word(idx, w, wpk, Cin, Cout, ncb, transposed);
}

// All layers of a network in ONE launch per fragment kind (round 5): packing EDSR(256 x 32) for the forward and for the data gradient was 4 launches per
// layer = 552 launches of a few microseconds each per training iteration (the weights change every iteration); a table of up to PACK_TABLE_LAYERS
// layers rides in the kernel arguments, blockIdx.y is the layer, a grid-stride loop covers the layer's words.
// KIND: 0 f32 fragments, 1 bf16-limb fragments (32x32x16), 2 bf16-limb fragments (16x16x32), 3 f16-limb fragments (16x16x32)
constexpr int PACK_TABLE_LAYERS = 36;
struct PackLayer { const float* w; float* packed; int Cin, Cout, ncb, pad_; long n[4], off[4]; };     // (Cin, Cout) as the packing kernels see them
struct PackTable { int transposed, nlayers; PackLayer layer[PACK_TABLE_LAYERS]; };
template <int KIND>
__global__ __launch_bounds__(256) void pack_table_kernel(PackTable t) {
    const PackLayer& L = t.layer[blockIdx.y];
    const long n = L.n[KIND];
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
        if (KIND == 0) pack_conv_word(idx, L.w, L.packed + L.off[0], L.Cin, L.Cout, L.ncb, t.transposed);
        else if (KIND == 1) pack_conv_limbs_word(idx, L.w, reinterpret_cast<unsigned*>(L.packed + L.off[1]), L.Cin, L.Cout, L.ncb, t.transposed);
        else if (KIND == 2) pack_conv_limbs16_word<3>(idx, L.w, reinterpret_cast<unsigned*>(L.packed + L.off[2]), L.Cin, L.Cout, t.transposed);
        else pack_conv_limbs16_word<2>(idx, L.w, reinterpret_cast<unsigned*>(L.packed + L.off[3]), L.Cin, L.Cout, t.transposed);
    }
}
// natural: the layers' [Cout][Cin][3][3] weights one after the other; packed: their fragment blobs one after the other (conv_packed_floats each)
int pack_layers(const float* natural, const ConvLayer* layers, int nl, float* packed, int transposed, hipStream_t stream, int arith) {
    // arith: NVSR_PACK_ALL_ARITHMETICS = every fragment region of every layer; an arithmetic = only the region launch_conv reads for it
    // (conv_kinds_for, sr_core.h): the others keep whatever the blob held
    for (int l0 = 0; l0 < nl; l0 += PACK_TABLE_LAYERS) {
        PackTable t{};
        t.transposed = transposed;
        t.nlayers = nl - l0 < PACK_TABLE_LAYERS ? nl - l0 : PACK_TABLE_LAYERS;
        long mx[4] = {0, 0, 0, 0};
        for (int k = 0; k < t.nlayers; ++k) {
            const ConvLayer& c = layers[l0 + k];
            const int ci = transposed ? c.Cout : c.Cin, co = transposed ? c.Cin : c.Cout;          // the packed conv's own (Cin, Cout)
            PackLayer& L = t.layer[k];
            L.w = natural; L.packed = packed; L.Cin = ci; L.Cout = co; L.ncb = conv_ncb(co);
            L.n[0] = conv_packed_f32_floats(ci, co);
            L.n[1] = conv_packed_limb_words(ci, co);
            L.n[2] = conv_packed_limb16_words(ci, co);
            L.n[3] = conv_packed_f16_words(ci, co);
            L.off[0] = 0; L.off[1] = L.n[0]; L.off[2] = L.n[0] + L.n[1]; L.off[3] = L.n[0] + L.n[1] + L.n[2];
            const unsigned kinds = conv_kinds_for(ci, co, arith, CV_USE_16X16X32 != 0, CV_WIDE_ROWS8 != 0);
            for (int q = 0; q < 4; ++q) {
                if (!(kinds >> q & 1u)) L.n[q] = 0;          // (a region this arithmetic never reads)
                mx[q] = L.n[q] > mx[q] ? L.n[q] : mx[q];
            }
            natural += 9LL * c.Cin * c.Cout;
            packed += conv_packed_floats(ci, co);
        }
        auto grid = [&](long n) { long b = (n + 255) / 256; return dim3((unsigned)(b < 1 ? 1 : b > 2048 ? 2048 : b), t.nlayers); };
        if (mx[0]) hipLaunchKernelGGL(pack_table_kernel<0>, grid(mx[0]), dim3(256), 0, stream, t);
        if (mx[1]) hipLaunchKernelGGL(pack_table_kernel<1>, grid(mx[1]), dim3(256), 0, stream, t);
        if (mx[2]) hipLaunchKernelGGL(pack_table_kernel<2>, grid(mx[2]), dim3(256), 0, stream, t);
        if (mx[3]) hipLaunchKernelGGL(pack_table_kernel<3>, grid(mx[3]), dim3(256), 0, stream, t);
        if (int e = NVSR_CHECK_LAUNCH()) return e;
    }
    return NVSR_OK;
}

// PlanesSR input: crop with as much real context as available + replicate padding == clamped gather (models.py:906-914),
// optional per-channel normalisation (:899-901).  out [C][ch+2pad][cw+2pad]
__global__ void sr_prepare_kernel(const float* __restrict__ lr, int Cc, int R0, int R1, int lo0, int lo1, int Hp, int Wp, int pad,
                                  const float* __restrict__ mean, const float* __restrict__ stdv, float* __restrict__ out) {
    const long n = (long)Cc * Hp * Wp;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int xx = (int)(i % Wp), y = (int)((i / Wp) % Hp), c = (int)(i / ((long)Wp * Hp));
    const int sy = min(max(lo0 - pad + y, 0), R0 - 1), sx = min(max(lo1 - pad + xx, 0), R1 - 1);
    float v = lr[((long)c * R0 + sy) * R1 + sx];
    if (mean) v = (v - mean[c]) / stdv[c];
    out[i] = v;
}

// PlanesSR output (models.py:915-923): canvas = NaN; canvas[roi] = difference[over:-over] + bilinear_x{sf}(LR)[roi]
// (F.interpolate(mode=PlanesSR.plane_interp, align_corners=PlanesSR.align_corners), :858-859)
__global__ void sr_finish_kernel(const float* __restrict__ diff, int Ho, int Wo, int over, const float* __restrict__ lr, int Cc, int R0,
                                 int R1, int sf, int lo0, int lo1, int hi0, int hi1, float* __restrict__ out, unsigned* __restrict__ flag, int align,
                                 int bicubic) {
    const int HR0 = R0 * sf, HR1 = R1 * sf;
    const long n = (long)Cc * HR0 * HR1;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int ox = (int)(i % HR1), oy = (int)((i / HR1) % HR0), c = (int)(i / ((long)HR1 * HR0));
    if (oy < lo0 * sf || oy >= hi0 * sf || ox < lo1 * sf || ox >= hi1 * sf) { out[i] = __builtin_nanf(""); return; }
    float res;
    if (bicubic) {
        const CubicTap ty = cubic_tap(oy, R0, HR0, sf, align), tx = cubic_tap(ox, R1, HR1, sf, align);
        const float* pc = lr + (long)c * R0 * R1;
        res = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* row = pc + (long)ty.i[i] * R1;
            res += (row[tx.i[0]] * tx.w[0] + row[tx.i[1]] * tx.w[1] + row[tx.i[2]] * tx.w[2] + row[tx.i[3]] * tx.w[3]) * ty.w[i];
        }
    } else {
        const BilinearTap ty = bilinear_tap(oy, R0, HR0, sf, align), tx = bilinear_tap(ox, R1, HR1, sf, align);
        const int y0 = ty.i0, x0 = tx.i0, yp = ty.step, xp = tx.step;
        const float ly1 = ty.w1, ly0 = 1.0f - ly1, lx1 = tx.w1, lx0 = 1.0f - lx1;
        const float* q = lr + ((long)c * R0 + y0) * R1 + x0;
        res = ly0 * (lx0 * q[0] + lx1 * q[xp]) + ly1 * (lx0 * q[(long)yp * R1] + lx1 * q[(long)yp * R1 + xp]);
    }
    const int dy = oy - lo0 * sf + over, dx = ox - lo1 * sf + over;
    const float v = diff[((long)c * Ho + dy) * Wo + dx] + res;
    out[i] = v;
    // range flag of the f16 limbs (nvsr.h: nvsr_set_range_flag): a non-finite value inside the region of interest is a weight / activation of
    // the network beyond the static scales (flag = NULL in every other arithmetic)
    if (flag && !(fabsf(v) <= 3.0e38f)) __hip_atomic_fetch_or(flag, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (bit 1: the SR network)
}

// align_corners of the bilinear residual (process-wide, like the conv arithmetic): 1 unless a binding says otherwise
static int g_sr_align_corners = 1;
int sr_align_corners() { return g_sr_align_corners; }
extern "C" int nvsr_set_sr_align_corners(int align_corners) { g_sr_align_corners = align_corners ? 1 : 0; return NVSR_OK; }
extern "C" int nvsr_get_sr_align_corners(void) { return g_sr_align_corners; }
static int g_sr_bicubic = 0;
int sr_bicubic() { return g_sr_bicubic; }
extern "C" int nvsr_set_sr_plane_interp(int plane_interp) {
    if (plane_interp != NVSR_PLANE_INTERP_BILINEAR && plane_interp != NVSR_PLANE_INTERP_BICUBIC) return NVSR_ERR_SHAPE;
    g_sr_bicubic = plane_interp == NVSR_PLANE_INTERP_BICUBIC;
    return NVSR_OK;
}
extern "C" int nvsr_get_sr_plane_interp(void) { return g_sr_bicubic ? NVSR_PLANE_INTERP_BICUBIC : NVSR_PLANE_INTERP_BILINEAR; }

// arithmetic of the eligible conv layers (process-wide): -1 = not yet read from the environment
static int g_conv_arithmetic = -1;
extern "C" int nvsr_internal_parse_arith_env(const char* name, int dflt);      // render.hip
extern "C" int nvsr_get_conv_arithmetic(void) {
    if (g_conv_arithmetic == -1) g_conv_arithmetic = nvsr_internal_parse_arith_env("NVSR_CONV_ARITHMETIC", NVSR_CONV_ARITH_DEFAULT);
    return g_conv_arithmetic;
}
extern "C" int nvsr_set_conv_arithmetic(int mode) {
    if (mode != NVSR_ARITH_F32 && mode != NVSR_ARITH_BF16X3 && mode != NVSR_ARITH_F16X2) return NVSR_ERR_SHAPE;
    g_conv_arithmetic = mode;
    return NVSR_OK;
}

int conv_resolve_arith(int arith) { return arith == NVSR_ARITH_INHERIT ? nvsr_get_conv_arithmetic() : arith; }

// max |x| of a tensor as the bits of a non-negative float (they order like unsigned integers): the scale of an f16-limb data gradient.
// The result word is the caller's (`owned`: the EDSR backward keeps one per gradient tensor in its workspace) or, for the stand-alone entry
// points, one of a ring handed out round-robin; a word is zeroed by a memset queued in front of its reduction.  A ring word is reused after
// ABSMAX_SLOTS launches of this function in the process: 16 384 -- the backward of EDSR(32 blocks) takes ~70 per plane, so a kernel still
// reading a word would have to be ~230 plane-backwards behind the host (on any stream) when the ring comes round; nothing tracks completion.
constexpr int ABSMAX_SLOTS = 16384;
__device__ unsigned g_absmax[ABSMAX_SLOTS];
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long n, unsigned* __restrict__ out) {
    // 16-byte loads over the aligned body, scalars over head / tail; one atomic per workgroup
    float m = 0.0f;
    const long head = (4 - ((reinterpret_cast<uintptr_t>(x) >> 2) & 3)) & 3, body = n > head ? (n - head) / 4 : 0;
    const f32x4* xv = reinterpret_cast<const f32x4*>(x + head);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < body; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = xv[i];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
    if (blockIdx.x == 0 && threadIdx.x < 8) {            // at most 3 scalars in front of the aligned body, 3 behind it
        const long i = threadIdx.x < 4 ? (long)threadIdx.x : head + 4 * body + (threadIdx.x - 4);
        if ((threadIdx.x < 4 ? i < head : i < n) && i < n) m = fmaxf(m, fabsf(x[i]));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    // (a NaN input: fmaxf drops it -- the convolution itself then propagates it through its products)
    if (threadIdx.x == 0) atomicMax(out, __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
}
const unsigned* launch_absmax(const float* x, long n, hipStream_t stream, unsigned* owned) {
    static std::atomic<unsigned> next{0};
    unsigned* slot = owned;          // a word of the caller's workspace (the EDSR backward), or one of the ring
    if (!slot) {
        unsigned* base = nullptr;
        if (hipGetSymbolAddress(reinterpret_cast<void**>(&base), HIP_SYMBOL(g_absmax)) != hipSuccess) return nullptr;
        slot = base + (next.fetch_add(1) % ABSMAX_SLOTS);
    }
    if (hipMemsetAsync(slot, 0, sizeof(unsigned), stream) != hipSuccess) return nullptr;
    const int blocks = (int)((n + 256 * 32 - 1) / (256 * 32) < 1024 ? (n + 256 * 32 - 1) / (256 * 32) : 1024);
    hipLaunchKernelGGL(absmax_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, stream, x, n, slot);
    return slot;
}

struct AbsmaxRagged { const float* x[CONV_RAGGED_MAX]; long n[CONV_RAGGED_MAX]; };
__global__ __launch_bounds__(256) void absmax_ragged_kernel(AbsmaxRagged a, unsigned* __restrict__ out) {
    // blockIdx.y = tensor; scalar loads with a grid stride (the tensors start anywhere): gradient tensors of a few MB each
    const float* __restrict__ x = a.x[blockIdx.y];
    const long n = a.n[blockIdx.y];
    float m = 0.0f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
}
const unsigned* launch_absmax_ragged(int n, const float* const* x, const long* count, hipStream_t stream, unsigned* owned) {
    if (n < 1 || n > CONV_RAGGED_MAX || !owned) return nullptr;
    AbsmaxRagged a{};
    long mx = 0;
    for (int b = 0; b < n; ++b) { a.x[b] = x[b]; a.n[b] = count[b]; mx = count[b] > mx ? count[b] : mx; }
    if (hipMemsetAsync(owned, 0, sizeof(unsigned), stream) != hipSuccess) return nullptr;
    const long blocks = (mx + 256 * 16 - 1) / (256 * 16);
    hipLaunchKernelGGL(absmax_ragged_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks > 512 ? 512 : blocks), n), dim3(256), 0, stream, a, owned);
    return owned;
}

int launch_conv(const float* in, int Cin, int H, int W, const float* wpk, int Cout, int epilogue, const float* skip, float* out,
                hipStream_t stream, int pad, int batch, ConvExec cx, const ConvRagged* rag) {
    const int arith = conv_resolve_arith(cx.arith);
    if (arith != NVSR_ARITH_F32 && arith != NVSR_ARITH_BF16X3 && arith != NVSR_ARITH_F16X2) return NVSR_ERR_SHAPE;
    if (rag) {
        // ragged batch: the grid is laid out for the largest plane (H, W = the maxima), the row count per tile is chosen for the tiles that exist
        if (rag->n < 1 || rag->n > CONV_RAGGED_MAX || arith == NVSR_ARITH_F32) return NVSR_ERR_SHAPE;
        batch = rag->n;
        H = W = 0;
        for (int b = 0; b < rag->n; ++b) {
            if (!rag->in[b] || !rag->out[b] || rag->H[b] + 2 * pad < 3 || rag->W[b] + 2 * pad < 3) return NVSR_ERR_SHAPE;
            if ((epilogue == EPI_RESIDUAL || epilogue == EPI_MASK_SCALE || epilogue == EPI_ADD_CENTER) && !rag->skip[b]) return NVSR_ERR_NULL;
            H = rag->H[b] > H ? rag->H[b] : H;
            W = rag->W[b] > W ? rag->W[b] : W;
        }
        in = rag->in[0]; out = rag->out[0]; skip = rag->skip[0];
    }
    // tiles of a launch with `rows` output rows per tile and gx-pixel column chunks: all planes of a ragged batch, or batch x the one size
    auto count_tiles = [&](int rows, long per_tile_z) -> long {
        if (!rag) return (long)((W - 2 + 31) / 32) * ((H - 2 + rows - 1) / rows) * per_tile_z * batch;      // (called below: H, W include the border by then)
        long t = 0;
        for (int b = 0; b < rag->n; ++b) t += (long)((rag->W[b] + 2 * pad - 2 + 31) / 32) * ((rag->H[b] + 2 * pad - 2 + rows - 1) / rows) * per_tile_z;
        return t;
    };
    // NVSR_ARITH_F16X2: the forward convolutions of the wide layers (16x16x32 kernel) on 2 f16 limbs; everything else that runs limbs -- the
    // narrow input / output layers, every data gradient (pad = 2, backward epilogues) -- stays on 3 bf16 limbs
    // (round 3, end: the data gradients of those layers too -- pad = 2, backward epilogues -- with the scale of the whole dy tensor, absmax_kernel)
    const bool f16 = arith == NVSR_ARITH_F16X2;
    const bool f16_dgrad = f16 && (pad != 0 || epilogue == EPI_MASK_SCALE || epilogue == EPI_ADD_CENTER);
    if (cx.rows != 0 && (cx.rows < 2 || cx.rows > 4) && cx.rows != 8 && cx.rows != 16 && !(cx.rows >= 18 && cx.rows <= 20) && cx.rows != 22) return NVSR_ERR_SHAPE;
    const long in_bs = (long)Cin * H * W;
    H += 2 * pad; W += 2 * pad;
    if (H < 3 || W < 3 || batch < 1 || batch > 1024) return NVSR_ERR_SHAPE;
    const int Ho = H - 2, Wo = W - 2;
    const long out_bs = (long)Cout * Ho * Wo;       // (PixelShuffle only permutes the Cout*Ho*Wo elements)
    const long skip_bs = epilogue == EPI_RESIDUAL ? (long)Cout * (Ho + 4) * (Wo + 4)
                         : epilogue == EPI_ADD_CENTER ? (long)Cout * (Ho - 4) * (Wo - 4) : out_bs;
    const unsigned* wlimb = conv_limb_eligible(Cin, Cout) ? reinterpret_cast<const unsigned*>(wpk + conv_packed_f32_floats(Cin, Cout)) : nullptr;
    const unsigned* wlimb16 = conv_limb16_eligible(Cin, Cout)
                                  ? reinterpret_cast<const unsigned*>(wpk + conv_packed_f32_floats(Cin, Cout)) + conv_packed_limb_words(Cin, Cout) : nullptr;
    const unsigned* wf16 = wlimb16 ? wlimb16 + conv_packed_limb16_words(Cin, Cout) : nullptr;
    ConvParams p{in, wpk, wlimb, wlimb16, wf16, nullptr, out, skip, Cin, Cout, H, W, conv_ncb(Cout), conv_nchunks(Cin), epilogue, pad, 1, in_bs, out_bs, skip_bs};
    p.out_absmax = cx.out_absmax;
    ConvRagged rg;          // (a second by-value kernel argument, read with constant indices: inside ConvParams -- which the kernels modify -- the
                            //  whole parameter struct went to scratch)
    if (rag) {
        rg = *rag;
        for (int b = 0; b < rag->n; ++b) { rg.H[b] += 2 * pad; rg.W[b] += 2 * pad; }
    }
    // ragged launch: a one-dimensional grid over the planes' tiles, plane after plane (ConvRagged::tile0; no empty tiles)
    auto ragged_grid = [&](int rows, int ncg) {
        unsigned t = 0;
        for (int b = 0; b <= CONV_RAGGED_MAX; ++b) {
            rg.tile0[b] = b <= rg.n ? t : 0xffffffffu;
            if (b < rg.n) t += (unsigned)((rg.W[b] - 2 + 31) / 32) * (unsigned)((rg.H[b] - 2 + rows - 1) / rows) * (unsigned)ncg;
        }
        for (int b = rg.n; b < CONV_RAGGED_MAX; ++b) rg.tile0[b] = 0xffffffffu;
        return dim3(t, 1, 1);
    };
    if (wlimb && arith != NVSR_ARITH_F32 && p.ncb_total == 2) {
        // narrow layer: 4 waves x 2 rows each of the same 64 output channels
        p.ncg = 1;
        dim3 grid((Wo + 31) / 32, (Ho + 7) / 8, batch);
        if (rag) grid = ragged_grid(8, 1);
        hipLaunchKernelGGL((conv3x3_limb_kernel<2, 1>), grid, dim3(256), 0, stream, p, rg);
        return NVSR_CHECK_LAUNCH();
    }
    // rows 16 = the 16x16x32 kernel with its own choice of rows per tile, 18 / 19 / 20 = that kernel with 2 / 3 / 4 rows forced
    if (cx.rows >= 16 && !(wlimb16 && arith != NVSR_ARITH_F32)) return NVSR_ERR_SHAPE;            // (eligible limb layers only)
    if (wlimb16 && arith != NVSR_ARITH_F32 && (cx.rows >= 16 || (cx.rows == 0 && CV_USE_16X16X32))) {
        p.ncg = Cout / (32 * CV16_WAVES);
        dim3 grid((Wo + 31) / 32, 1, p.ncg * batch);
        int best_pb = 4;
        double best_cost = 1e300;
        // f16 limbs: a 6-row tile too (96 accumulator registers, 2 x 35 KB of LDS: still two workgroups per CU) -- with half the MFMAs per weight
        // fragment the stream of fragments out of L2 (4 KB per wave and tap) is what a row costs, and 6 rows amortise it over 1.5 x the MFMAs
        for (int pb = f16 ? 6 : 4; pb >= 2; --pb) {
            if (pb == 5) continue;
            // rounds of the 512 workgroup slots x rows x the measured cost of a row in a pb-row tile relative to a 4-row tile (tools/conv_time.py
            // rows 18 / 19 / 20 on a layer of full rounds); a last round of at most 256 tiles has every CU to itself (~0.62 of a round's time)
            const long tiles = count_tiles(pb, p.ncg);
            constexpr long SLOTS = CV16_WAVES == 4 ? 512 : 256;
            const long full = tiles / SLOTS, rest = tiles % SLOTS;
            const double rounds = (double)full + (rest == 0 ? 0.0 : (CV16_WAVES == 4 && rest <= 256) ? 0.62 : 1.0);
            const double cost = rounds * pb * (pb == 6 ? CV16_ROWCOST6 : pb == 4 ? 1.0 : pb == 3 ? CV16_ROWCOST3 : CV16_ROWCOST2);
            if (cost < best_cost) { best_cost = cost; best_pb = pb; }
        }
        if (cx.rows >= 18) best_pb = cx.rows - 16;
        if (best_pb == 6 && !f16) return NVSR_ERR_SHAPE;
        {   // NVSR_CV16_ROWS=2|3|4 (environment, read once): force the row count of every launch (tools: per-layer comparison of the variants)
            static int forced = -1;
            if (forced < 0) { const char* e = getenv("NVSR_CV16_ROWS"); forced = e ? atoi(e) : 0; }
            if (((forced >= 2 && forced <= 4) || (forced == 6 && f16)) && cx.rows < 18) best_pb = forced;
        }
        grid.y = (Ho + best_pb - 1) / best_pb;
        if (rag) grid = ragged_grid(best_pb, p.ncg);
        if (f16) {
            if (f16_dgrad) {
                if (rag && !cx.in_absmax) return NVSR_ERR_NULL;
                p.absmax = cx.in_absmax ? cx.in_absmax : launch_absmax(in, in_bs * batch, stream);
                if (!p.absmax) return NVSR_ERR_LAUNCH;
            }
            if (best_pb == 6) hipLaunchKernelGGL((conv3x3_limb16_kernel<6, CV16_WAVES, 2>), grid, dim3(64 * CV16_WAVES), 0, stream, p, rg);
            else if (best_pb == 4) hipLaunchKernelGGL((conv3x3_limb16_kernel<4, CV16_WAVES, 2>), grid, dim3(64 * CV16_WAVES), 0, stream, p, rg);
            else if (best_pb == 3) hipLaunchKernelGGL((conv3x3_limb16_kernel<3, CV16_WAVES, 2>), grid, dim3(64 * CV16_WAVES), 0, stream, p, rg);
            else hipLaunchKernelGGL((conv3x3_limb16_kernel<2, CV16_WAVES, 2>), grid, dim3(64 * CV16_WAVES), 0, stream, p, rg);
            return NVSR_CHECK_LAUNCH();
        }
        if (best_pb == 4) hipLaunchKernelGGL((conv3x3_limb16_kernel<4, CV16_WAVES>), grid, dim3(64 * CV16_WAVES), 0, stream, p, rg);
        else if (best_pb == 3) hipLaunchKernelGGL((conv3x3_limb16_kernel<3, CV16_WAVES>), grid, dim3(64 * CV16_WAVES), 0, stream, p, rg);
        else hipLaunchKernelGGL((conv3x3_limb16_kernel<2, CV16_WAVES>), grid, dim3(64 * CV16_WAVES), 0, stream, p, rg);
        return NVSR_CHECK_LAUNCH();
    }
    if (cx.rows == 8 && !(wlimb && arith != NVSR_ARITH_F32 && p.ncb_total % 4 == 0 && p.ncb_total > 2)) return NVSR_ERR_SHAPE;   // (only the wide limb kernel has it)
    if (wlimb && arith != NVSR_ARITH_F32 && p.ncb_total % 4 == 0 && p.ncb_total > 2 && (cx.rows == 8 || (cx.rows == 0 && CV_WIDE_ROWS8))) {
        // bf16-limb kernel, 1 output block x 8 rows per wave: 4-wave workgroups of 128 output channels x 8 rows x 32 pixels
        p.ncg = p.ncb_total / 4;
        dim3 grid((Wo + 31) / 32, (Ho + 7) / 8, p.ncg * batch);
        if (rag) grid = ragged_grid(8, p.ncg);
        hipLaunchKernelGGL((conv3x3_limb_kernel<8, 4, 1>), grid, dim3(256), 0, stream, p, rg);
        return NVSR_CHECK_LAUNCH();
    }
    if (wlimb && arith != NVSR_ARITH_F32) {
        // bf16-limb kernel: 4-wave workgroups of 256 output channels x PB rows x 32 pixels; same choice of the row count as below
        p.ncg = p.ncb_total / 8;
        dim3 grid((Wo + 31) / 32, 1, p.ncg * batch);
        int best_pb = 4;
        double best_cost = 1e300;
        for (int pb = 4; pb >= 2; --pb) {
            const long tiles = count_tiles(pb, p.ncg);
#if CV_COST_MODEL == 0
            const double cost = (double)((tiles + 511) / 512) * pb * (1.0 + 0.03 * (4 - pb));
#elif CV_COST_MODEL == 1
            const double cost = pb == 4 ? 0.0 : 1.0;
#else
            // rounds of the 512 workgroup slots; a last round of at most 256 tiles has every CU to itself (~0.62 of a full round's time);
            // per row of a tile, 3-row tiles cost 1.13 x and 2-row tiles 1.39 x a 4-row tile's (their weight stream per MFMA is 4/3, 2 x)
            const long full = tiles / 512, rest = tiles % 512;
            const double rounds = (double)full + (rest == 0 ? 0.0 : rest <= 256 ? 0.62 : 1.0);
            const double cost = rounds * pb * (pb == 4 ? 1.0 : pb == 3 ? 1.13 : 1.39);
#endif
            if (cost < best_cost) { best_cost = cost; best_pb = pb; }
        }
        if (cx.rows) best_pb = cx.rows;
        grid.y = (Ho + best_pb - 1) / best_pb;
        if (rag) grid = ragged_grid(best_pb, p.ncg);
        if (best_pb == 4) hipLaunchKernelGGL((conv3x3_limb_kernel<4>), grid, dim3(256), 0, stream, p, rg);
        else if (best_pb == 3) hipLaunchKernelGGL((conv3x3_limb_kernel<3>), grid, dim3(256), 0, stream, p, rg);
        else hipLaunchKernelGGL((conv3x3_limb_kernel<2>), grid, dim3(256), 0, stream, p, rg);
        return NVSR_CHECK_LAUNCH();
    }
    if (rag) {
        // a layer no limb kernel is eligible for (e.g. 48 -> 128 channels) runs on the exact-f32 kernels below, which take no ragged batch:
        // plane after plane
        for (int b = 0; b < rag->n; ++b)
            if (int e = launch_conv(rag->in[b], Cin, rag->H[b], rag->W[b], wpk, Cout, epilogue, rag->skip[b], rag->out[b], stream, pad, 1, cx, nullptr)) return e;
        return NVSR_OK;
    }
    if (p.ncb_total >= 8 && p.ncb_total % 8 == 0) {
        p.ncg = p.ncb_total / 8;
        dim3 grid((Wo + 31) / 32, (Ho + 7) / 8, p.ncg * batch);
#if NVSR_CONV_WIDE_4x1
        // two independent 4-wave workgroups per CU (2 x 80 KB of LDS): one's chunk barrier is covered by the other's MFMAs.
        // Rows per tile: all tiles of a launch take the same time, so the launch costs ceil(tiles / 512 slots) rounds of `rows` row-times;
        // pick the row count (4, 3 or 2; fewer rows amortise the weight reads a little worse: +3 % per row dropped) with the cheapest total
        int best_pb = 4;
        double best_cost = 1e300;
        for (int pb = 4; pb >= 2; --pb) {
            const long tiles = (long)grid.x * ((Ho + pb - 1) / pb) * grid.z;
            const double cost = (double)((tiles + 511) / 512) * pb * (1.0 + 0.03 * (4 - pb));
            if (cost < best_cost) { best_cost = cost; best_pb = pb; }
        }
        if (cx.rows) best_pb = cx.rows;
        grid.y = (Ho + best_pb - 1) / best_pb;
        if (best_pb == 4) hipLaunchKernelGGL((conv3x3_kernel<4, 1, 4>), grid, dim3(256), 0, stream, p);
        else if (best_pb == 3) hipLaunchKernelGGL((conv3x3_kernel<4, 1, 3>), grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((conv3x3_kernel<4, 1, 2>), grid, dim3(256), 0, stream, p);
#else
        hipLaunchKernelGGL((conv3x3_kernel<4, 2>), grid, dim3(512), 0, stream, p);
#endif
    } else {
        p.ncg = p.ncb_total / 2;
        dim3 grid((Wo + 31) / 32, (Ho + 31) / 32, p.ncg * batch);
        hipLaunchKernelGGL((conv3x3_kernel<1, 8>), grid, dim3(512), 0, stream, p);
    }
    return NVSR_CHECK_LAUNCH();
}

}  // namespace nvsr

using namespace nvsr;

extern "C" {

/* one conv layer: weights [Cout][Cin][3][3] (nn.Conv2d layout) -> packed fragments */
int64_t nvsr_conv3x3_packed_floats(int Cin, int Cout) { return conv_packed_floats(Cin, Cout); }

int nvsr_pack_conv3x3(const float* w, int Cin, int Cout, float* packed, nvsr_stream_t stream) {
    if (!w || !packed) return NVSR_ERR_NULL;
    if (Cin < 1 || Cout < 1) return NVSR_ERR_SHAPE;
    if (!aligned16(packed)) return NVSR_ERR_ALIGN;
    const int64_t n = conv_packed_f32_floats(Cin, Cout);
    hipLaunchKernelGGL(pack_conv_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, packed, Cin, Cout,
                       conv_ncb(Cout), conv_nchunks(Cin), 0);
    if (const int64_t nl = conv_packed_limb_words(Cin, Cout))
        hipLaunchKernelGGL(pack_conv_limbs_kernel, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                           reinterpret_cast<unsigned*>(packed + n), Cin, Cout, conv_ncb(Cout), 0);
    if (const int64_t n16 = conv_packed_limb16_words(Cin, Cout)) {
        unsigned* r16 = reinterpret_cast<unsigned*>(packed + n) + conv_packed_limb_words(Cin, Cout);
        hipLaunchKernelGGL(pack_conv_limbs16_kernel<3>, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, r16, Cin, Cout, 0);
        const int64_t nf = conv_packed_f16_words(Cin, Cout);
        hipLaunchKernelGGL(pack_conv_limbs16_kernel<2>, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, r16 + n16, Cin, Cout, 0);
    }
    return NVSR_CHECK_LAUNCH();
}

/* weights [Cout][Cin][3][3] of a conv -> packed fragments of its data gradient (a Cout -> Cin conv with the flipped, transposed
 * kernel); nvsr_conv3x3_packed_floats(Cout, Cin) floats */
int nvsr_pack_conv3x3_dgrad(const float* w, int Cin, int Cout, float* packed, nvsr_stream_t stream) {
    if (!w || !packed) return NVSR_ERR_NULL;
    if (Cin < 1 || Cout < 1) return NVSR_ERR_SHAPE;
    if (!aligned16(packed)) return NVSR_ERR_ALIGN;
    const int64_t n = conv_packed_f32_floats(Cout, Cin);
    hipLaunchKernelGGL(pack_conv_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, packed, Cout, Cin,
                       conv_ncb(Cin), conv_nchunks(Cout), 1);
    if (const int64_t nl = conv_packed_limb_words(Cout, Cin))
        hipLaunchKernelGGL(pack_conv_limbs_kernel, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                           reinterpret_cast<unsigned*>(packed + n), Cout, Cin, conv_ncb(Cin), 1);
    if (const int64_t n16 = conv_packed_limb16_words(Cout, Cin)) {
        unsigned* r16 = reinterpret_cast<unsigned*>(packed + n) + conv_packed_limb_words(Cout, Cin);
        hipLaunchKernelGGL(pack_conv_limbs16_kernel<3>, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, r16, Cout, Cin, 1);
        const int64_t nf = conv_packed_f16_words(Cout, Cin);
        hipLaunchKernelGGL(pack_conv_limbs16_kernel<2>, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, r16 + n16, Cout, Cin, 1);
    }
    return NVSR_CHECK_LAUNCH();
}

/* data gradient of nvsr_conv3x3 (epilogue 0): dy [Cout][H-2][W-2] -> dx [Cin][H][W]; packed_dgrad from nvsr_pack_conv3x3_dgrad */
int nvsr_conv3x3_dgrad_arith(const float* dy, int Cin, int H, int W, const float* packed_dgrad, int Cout, float* dx, int arithmetic,
                             int rows_per_tile, nvsr_stream_t stream) {
    if (!dy || !packed_dgrad || !dx) return NVSR_ERR_NULL;
    if (!aligned16(packed_dgrad)) return NVSR_ERR_ALIGN;
    if (H < 3 || W < 3) return NVSR_ERR_SHAPE;
    return launch_conv(dy, Cout, H - 2, W - 2, packed_dgrad, Cin, EPI_NONE, nullptr, dx, (hipStream_t)stream, 2, 1, ConvExec{arithmetic, rows_per_tile});
}
int nvsr_conv3x3_dgrad(const float* dy, int Cin, int H, int W, const float* packed_dgrad, int Cout, float* dx, nvsr_stream_t stream) {
    return nvsr_conv3x3_dgrad_arith(dy, Cin, H, W, packed_dgrad, Cout, dx, NVSR_ARITH_INHERIT, 0, stream);
}

/* epilogue: 0 none, 1 ReLU, 2 residual (out = conv*0.1 + skip[..., 2:-2, 2:-2], skip = [Cout][H+2][W+2]), 3 PixelShuffle(2) */
int nvsr_conv3x3_arith(const float* in, int Cin, int H, int W, const float* packed, int Cout, int epilogue, const float* skip, float* out,
                       int arithmetic, int rows_per_tile, nvsr_stream_t stream) {
    if (!in || !packed || !out) return NVSR_ERR_NULL;
    if (epilogue < 0 || epilogue > 3 || (epilogue == EPI_RESIDUAL && !skip) || (epilogue == EPI_PIXEL_SHUFFLE && Cout % 4)) return NVSR_ERR_SHAPE;
    if (!aligned16(packed)) return NVSR_ERR_ALIGN;
    return launch_conv(in, Cin, H, W, packed, Cout, epilogue, skip, out, (hipStream_t)stream, 0, 1, ConvExec{arithmetic, rows_per_tile});
}
int nvsr_conv3x3(const float* in, int Cin, int H, int W, const float* packed, int Cout, int epilogue, const float* skip, float* out,
                 nvsr_stream_t stream) {
    return nvsr_conv3x3_arith(in, Cin, H, W, packed, Cout, epilogue, skip, out, NVSR_ARITH_INHERIT, 0, stream);
}

/* EDSR(in_channels=Cin, out_channels=Cout, hidden_size=hid, n_blocks, scale_factor=2^n_up, padding=0)  (models.py:789-822).
 * natural blob = state-dict order: conv_input, residual.{b}.conv1, residual.{b}.conv2, conv_mid, upscale.{0,2,..}, conv_output. */
int64_t nvsr_edsr_natural_floats(int Cin, int Cout, int hid, int nblocks, int n_up) {
    return 9LL * ((int64_t)hid * Cin + (2LL * nblocks + 1) * hid * hid + (int64_t)n_up * 4 * hid * hid + (int64_t)Cout * hid);
}
int64_t nvsr_edsr_packed_floats(int Cin, int Cout, int hid, int nblocks, int n_up) {
    ConvLayer L[600]; int n;
    if (nblocks < 0 || nblocks > 290 || n_up < 0 || n_up > 8) return -1;
    edsr_layers(Cin, Cout, hid, nblocks, n_up, L, &n);
    int64_t s = 0;
    for (int i = 0; i < n; ++i) s += conv_packed_floats(L[i].Cin, L[i].Cout);
    return s;
}
int nvsr_pack_edsr_arith(const float* natural, int Cin, int Cout, int hid, int nblocks, int n_up, float* packed, int arithmetic, nvsr_stream_t stream) {
    if (!natural || !packed) return NVSR_ERR_NULL;
    if (nvsr_edsr_packed_floats(Cin, Cout, hid, nblocks, n_up) < 0) return NVSR_ERR_SHAPE;
    if (!aligned16(packed)) return NVSR_ERR_ALIGN;
    if (arithmetic != NVSR_PACK_ALL_ARITHMETICS) {
        arithmetic = conv_resolve_arith(arithmetic);
        if (arithmetic != NVSR_ARITH_F32 && arithmetic != NVSR_ARITH_F16X2 && arithmetic != NVSR_ARITH_BF16X3) return NVSR_ERR_SHAPE;
    }
    ConvLayer L[600]; int n;
    edsr_layers(Cin, Cout, hid, nblocks, n_up, L, &n);
    return pack_layers(natural, L, n, packed, 0, (hipStream_t)stream, arithmetic);       // (4 launches per 36 layers instead of 4 per layer)
}
int nvsr_pack_edsr(const float* natural, int Cin, int Cout, int hid, int nblocks, int n_up, float* packed, nvsr_stream_t stream) {
    return nvsr_pack_edsr_arith(natural, Cin, Cout, hid, nblocks, n_up, packed, NVSR_PACK_ALL_ARITHMETICS, stream);
}

/* spatial size of the EDSR output for an [*, H, W] input */
int nvsr_edsr_out_size(int H, int W, int nblocks, int n_up, int* Ho, int* Wo) {
    int64_t h = (int64_t)H - 2 - 4 * nblocks - 2, w = (int64_t)W - 2 - 4 * nblocks - 2;
    for (int u = 0; u < n_up; ++u) { h = (h - 2) * 2; w = (w - 2) * 2; }
    h -= 2; w -= 2;
    if (h < 1 || w < 1) return NVSR_ERR_SHAPE;
    *Ho = (int)h; *Wo = (int)w;
    return NVSR_OK;
}

/* two ping-pong activation buffers sized for the largest intermediate */
int64_t nvsr_edsr_workspace_floats(int hid, int nblocks, int n_up, int H, int W) {
    int64_t h = H - 2, w = W - 2, mx = (int64_t)hid * h * w;
    h -= 4 * nblocks + 2; w -= 4 * nblocks + 2;
    for (int u = 0; u < n_up; ++u) { h = (h - 2) * 2; w = (w - 2) * 2; if ((int64_t)hid * h * w > mx) mx = (int64_t)hid * h * w; }
    return 3 * mx;
}

/* B planes at once ([B][Cin][H][W] -> [B][Cout][Ho][Wo]): one launch per layer, the batch index rides in the grid.  A 256-channel
 * layer of one 200^2 plane is only ~1.2-1.8 workgroup rounds on 256 CUs (28 % of the issue slots idle in the partial last round);
 * the 3 position planes of a scene together run 3.6-5.4 rounds.  workspace: B * nvsr_edsr_workspace_floats floats. */
int nvsr_edsr_forward_batch_arith(const float* x, int B, int Cin, int H, int W, const float* packed, int Cout, int hid, int nblocks, int n_up,
                                  float* out, float* workspace, int arithmetic, nvsr_stream_t stream_) {
    if (!x || !packed || !out || !workspace) return NVSR_ERR_NULL;
    if (B < 1) return NVSR_ERR_SHAPE;
    const ConvExec cx{conv_resolve_arith(arithmetic), 0};
    int Ho, Wo;
    if (int e = nvsr_edsr_out_size(H, W, nblocks, n_up, &Ho, &Wo)) return e;
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t third = nvsr_edsr_workspace_floats(hid, nblocks, n_up, H, W) / 3 * B;
    float* bufs[3] = {workspace, workspace + third, workspace + 2 * third};
    const float* wp = packed;
    int h = H, w = W, e;
    // conv_input
    float* cur = bufs[0];
    if ((e = launch_conv(x, Cin, h, w, wp, hid, EPI_NONE, nullptr, cur, stream, 0, B, cx))) return e;
    wp += conv_packed_floats(Cin, hid); h -= 2; w -= 2;
    int ci = 0;                                   // index of `cur` in bufs
    for (int b = 0; b < nblocks; ++b) {           // _Residual_Block (models.py:777-786)
        float* t1 = bufs[(ci + 1) % 3];
        float* t2 = bufs[(ci + 2) % 3];
        if ((e = launch_conv(cur, hid, h, w, wp, hid, EPI_RELU, nullptr, t1, stream, 0, B, cx))) return e;
        wp += conv_packed_floats(hid, hid);
        if ((e = launch_conv(t1, hid, h - 2, w - 2, wp, hid, EPI_RESIDUAL, cur, t2, stream, 0, B, cx))) return e;
        wp += conv_packed_floats(hid, hid);
        cur = t2; ci = (ci + 2) % 3; h -= 4; w -= 4;
    }
    {   // conv_mid
        float* t = bufs[(ci + 1) % 3];
        if ((e = launch_conv(cur, hid, h, w, wp, hid, EPI_NONE, nullptr, t, stream, 0, B, cx))) return e;
        wp += conv_packed_floats(hid, hid);
        cur = t; ci = (ci + 1) % 3; h -= 2; w -= 2;
    }
    for (int u = 0; u < n_up; ++u) {              // conv hid -> 4 hid + PixelShuffle(2), fused
        float* t = bufs[(ci + 1) % 3];
        if ((e = launch_conv(cur, hid, h, w, wp, 4 * hid, EPI_PIXEL_SHUFFLE, nullptr, t, stream, 0, B, cx))) return e;
        wp += conv_packed_floats(hid, 4 * hid);
        cur = t; ci = (ci + 1) % 3; h = (h - 2) * 2; w = (w - 2) * 2;
    }
    return launch_conv(cur, hid, h, w, wp, Cout, EPI_NONE, nullptr, out, stream, 0, B, cx);
}
int nvsr_edsr_forward_batch(const float* x, int B, int Cin, int H, int W, const float* packed, int Cout, int hid, int nblocks, int n_up,
                            float* out, float* workspace, nvsr_stream_t stream_) {
    return nvsr_edsr_forward_batch_arith(x, B, Cin, H, W, packed, Cout, hid, nblocks, n_up, out, workspace, NVSR_ARITH_INHERIT, stream_);
}

int nvsr_edsr_forward(const float* x, int Cin, int H, int W, const float* packed, int Cout, int hid, int nblocks, int n_up, float* out,
                      float* workspace, nvsr_stream_t stream_) {
    return nvsr_edsr_forward_batch_arith(x, 1, Cin, H, W, packed, Cout, hid, nblocks, n_up, out, workspace, NVSR_ARITH_INHERIT, stream_);
}

/* ---- training forward: every layer's input is kept for the backward pass (sr_bwd.hip) ---------------------------------- */
int64_t nvsr_edsr_acts_floats(int Cin, int Cout, int hid, int nblocks, int n_up, int H, int W) {
    EdsrPlan P;
    if (edsr_plan(Cin, Cout, hid, nblocks, n_up, H, W, &P)) return -1;
    return P.acts_floats;
}

int nvsr_edsr_forward_train(const float* x, int Cin, int H, int W, const float* packed, int Cout, int hid, int nblocks, int n_up, float* out,
                            float* acts, nvsr_stream_t stream_) {
    return nvsr_edsr_forward_train_arith(x, Cin, H, W, packed, Cout, hid, nblocks, n_up, out, acts, NVSR_ARITH_INHERIT, stream_);
}
int nvsr_edsr_forward_train_arith(const float* x, int Cin, int H, int W, const float* packed, int Cout, int hid, int nblocks, int n_up, float* out,
                                  float* acts, int arithmetic, nvsr_stream_t stream_) {
    if (!x || !packed || !out || !acts) return NVSR_ERR_NULL;
    const ConvExec cx{conv_resolve_arith(arithmetic), 0};
    if (!aligned16(packed)) return NVSR_ERR_ALIGN;
    EdsrPlan P;
    if (int e = edsr_plan(Cin, Cout, hid, nblocks, n_up, H, W, &P)) return e;
    const float* wp = packed;
    for (int l = 0; l < P.n; ++l) {
        const float* in = l ? acts + P.act_off[l] : x;
        float* o = (l + 1 < P.n) ? acts + P.act_off[l + 1] : out;
        const float* skip = (P.epi[l] == EPI_RESIDUAL) ? (l >= 2 ? acts + P.act_off[l - 1] : x) : nullptr;   // the block's input
        if (int e = launch_conv(in, P.L[l].Cin, P.ih[l], P.iw[l], wp, P.L[l].Cout, P.epi[l], skip, o, (hipStream_t)stream_, 0, 1, cx)) return e;
        wp += conv_packed_floats(P.L[l].Cin, P.L[l].Cout);
    }
    return NVSR_OK;
}

/* PlanesSR.forward (models.py:884-926).  lr [C][R0][R1]; roi = NULL (full plane) or 4 HOST floats [[ymin,xmin],[ymax,xmax]] in
 * [-1,1]; pad = inner_model.required_padding, over = HR_overpadding; mean/std optional [C] (device).
 * out [C][sf*R0][sf*R1] (NaN outside the ROI).  workspace: nvsr_planes_sr_workspace_floats(...) floats. */
int64_t nvsr_planes_sr_workspace_floats(int Cc, int R0, int R1, int hid, int nblocks, int n_up, int pad, const float* roi) {
    int lo[2], hi[2];
    sr_roi(R0, R1, roi, lo, hi);
    const int Hp = hi[0] - lo[0] + 2 * pad, Wp = hi[1] - lo[1] + 2 * pad;
    int Ho, Wo;
    if (nvsr_edsr_out_size(Hp, Wp, nblocks, n_up, &Ho, &Wo)) return -1;
    return (int64_t)Cc * Hp * Wp + (int64_t)Cc * Ho * Wo + nvsr_edsr_workspace_floats(hid, nblocks, n_up, Hp, Wp);
}

/* B planes of the same size through the SR network in one batch (the 3 position planes of a scene, models.py:289-310).
 * lr / out: HOST arrays of B device pointers.  workspace: B * nvsr_planes_sr_workspace_floats(...) floats. */
int nvsr_planes_sr_batch(const float* const* lr, int B, int Cc, int R0, int R1, const float* packed, int hid, int nblocks, int n_up, int pad,
                         int over, const float* roi, const float* mean, const float* stdv, float* const* out, float* workspace,
                         nvsr_stream_t stream_) {
    return nvsr_planes_sr_batch_arith(lr, B, Cc, R0, R1, packed, hid, nblocks, n_up, pad, over, roi, mean, stdv, out, workspace, NVSR_ARITH_INHERIT,
                                      stream_);
}
int nvsr_planes_sr_batch_arith(const float* const* lr, int B, int Cc, int R0, int R1, const float* packed, int hid, int nblocks, int n_up, int pad,
                               int over, const float* roi, const float* mean, const float* stdv, float* const* out, float* workspace,
                               int arithmetic, nvsr_stream_t stream_) {
    return nvsr_planes_sr_batch_ex(lr, B, Cc, R0, R1, packed, hid, nblocks, n_up, pad, over, roi, mean, stdv, out, workspace, arithmetic,
                                   sr_align_corners(), sr_bicubic() ? NVSR_PLANE_INTERP_BICUBIC : NVSR_PLANE_INTERP_BILINEAR, stream_);
}
/* the same with align_corners / plane_interp of the residual up-sampling as ARGUMENTS (re-entrant: two PlanesSR models with different settings
 * in one process, backward passes on other threads and streams -- ADVICE r4) */
int nvsr_planes_sr_batch_ex(const float* const* lr, int B, int Cc, int R0, int R1, const float* packed, int hid, int nblocks, int n_up, int pad,
                            int over, const float* roi, const float* mean, const float* stdv, float* const* out, float* workspace,
                            int arithmetic, int align_corners, int plane_interp, nvsr_stream_t stream_) {
    if (!lr || !packed || !out || !workspace) return NVSR_ERR_NULL;
    if (plane_interp != NVSR_PLANE_INTERP_BILINEAR && plane_interp != NVSR_PLANE_INTERP_BICUBIC) return NVSR_ERR_SHAPE;
    if ((mean == nullptr) != (stdv == nullptr)) return NVSR_ERR_NULL;
    if (B < 1 || B > 64) return NVSR_ERR_SHAPE;
    for (int b = 0; b < B; ++b)
        if (!lr[b] || !out[b]) return NVSR_ERR_NULL;
    hipStream_t stream = (hipStream_t)stream_;
    const int sf = 1 << n_up;
    int lo[2], hi[2];
    sr_roi(R0, R1, roi, lo, hi);
    const int ch = hi[0] - lo[0], cw = hi[1] - lo[1];
    const int Hp = ch + 2 * pad, Wp = cw + 2 * pad;
    int Ho, Wo;
    if (int e = nvsr_edsr_out_size(Hp, Wp, nblocks, n_up, &Ho, &Wo)) return e;
    if (Ho != ch * sf + 2 * over || Wo != cw * sf + 2 * over) return NVSR_ERR_SHAPE;
    const int64_t n_in = (int64_t)Cc * Hp * Wp, n_diff = (int64_t)Cc * Ho * Wo;
    float* xin = workspace;                       // [B][C][Hp][Wp]
    float* diff = xin + B * n_in;                 // [B][C][Ho][Wo]
    float* ews = diff + B * n_diff;
    for (int b = 0; b < B; ++b) {
        hipLaunchKernelGGL(sr_prepare_kernel, dim3((unsigned)((n_in + 255) / 256)), dim3(256), 0, stream, lr[b], Cc, R0, R1, lo[0], lo[1], Hp,
                           Wp, pad, mean, stdv, xin + b * n_in);
        if (int e = NVSR_CHECK_LAUNCH()) return e;
    }
    if (int e = nvsr_edsr_forward_batch_arith(xin, B, Cc, Hp, Wp, packed, Cc, hid, nblocks, n_up, diff, ews, arithmetic, stream_)) return e;
    const int64_t n_out = (int64_t)Cc * R0 * sf * R1 * sf;
    for (int b = 0; b < B; ++b) {
        hipLaunchKernelGGL(sr_finish_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, stream, diff + b * n_diff, Ho, Wo, over, lr[b],
                           Cc, R0, R1, sf, lo[0], lo[1], hi[0], hi[1], out[b], conv_resolve_arith(arithmetic) == NVSR_ARITH_F16X2 ? nvsr_get_range_flag() : nullptr,
                           align_corners ? 1 : 0, plane_interp == NVSR_PLANE_INTERP_BICUBIC ? 1 : 0);
        if (int e = NVSR_CHECK_LAUNCH()) return e;
    }
    return NVSR_OK;
}

/* floats kept between nvsr_planes_sr_train and nvsr_planes_sr_backward: the prepared network input + the activation record */
int64_t nvsr_planes_sr_keep_floats(int Cc, int R0, int R1, int hid, int nblocks, int n_up, int pad, const float* roi) {
    int lo[2], hi[2];
    sr_roi(R0, R1, roi, lo, hi);
    const int Hp = hi[0] - lo[0] + 2 * pad, Wp = hi[1] - lo[1] + 2 * pad;
    const int64_t a = nvsr_edsr_acts_floats(Cc, Cc, hid, nblocks, n_up, Hp, Wp);
    if (a < 0) return -1;
    return ((int64_t)Cc * Hp * Wp + 3) / 4 * 4 + a;
}

static int planes_sr_impl(const float* lr, int Cc, int R0, int R1, const float* packed, int hid, int nblocks, int n_up, int pad, int over,
                          const float* roi, const float* mean, const float* stdv, float* out, float* workspace, float* keep,
                          int arithmetic, nvsr_stream_t stream_);

int nvsr_planes_sr(const float* lr, int Cc, int R0, int R1, const float* packed, int hid, int nblocks, int n_up, int pad, int over,
                   const float* roi, const float* mean, const float* stdv, float* out, float* workspace, nvsr_stream_t stream_) {
    return planes_sr_impl(lr, Cc, R0, R1, packed, hid, nblocks, n_up, pad, over, roi, mean, stdv, out, workspace, nullptr, NVSR_ARITH_INHERIT, stream_);
}
int nvsr_planes_sr_arith(const float* lr, int Cc, int R0, int R1, const float* packed, int hid, int nblocks, int n_up, int pad, int over,
                         const float* roi, const float* mean, const float* stdv, float* out, float* workspace, int arithmetic,
                         nvsr_stream_t stream_) {
    return planes_sr_impl(lr, Cc, R0, R1, packed, hid, nblocks, n_up, pad, over, roi, mean, stdv, out, workspace, nullptr, arithmetic, stream_);
}

/* nvsr_planes_sr that keeps what the backward needs in `keep` (nvsr_planes_sr_keep_floats floats); same workspace */
int nvsr_planes_sr_train(const float* lr, int Cc, int R0, int R1, const float* packed, int hid, int nblocks, int n_up, int pad, int over,
                         const float* roi, const float* mean, const float* stdv, float* out, float* workspace, float* keep,
                         nvsr_stream_t stream_) {
    if (!keep) return NVSR_ERR_NULL;
    return planes_sr_impl(lr, Cc, R0, R1, packed, hid, nblocks, n_up, pad, over, roi, mean, stdv, out, workspace, keep, NVSR_ARITH_INHERIT, stream_);
}
int nvsr_planes_sr_train_arith(const float* lr, int Cc, int R0, int R1, const float* packed, int hid, int nblocks, int n_up, int pad, int over,
                               const float* roi, const float* mean, const float* stdv, float* out, float* workspace, float* keep,
                               int arithmetic, nvsr_stream_t stream_) {
    if (!keep) return NVSR_ERR_NULL;
    return planes_sr_impl(lr, Cc, R0, R1, packed, hid, nblocks, n_up, pad, over, roi, mean, stdv, out, workspace, keep, arithmetic, stream_);
}

static int planes_sr_impl(const float* lr, int Cc, int R0, int R1, const float* packed, int hid, int nblocks, int n_up, int pad, int over,
                          const float* roi, const float* mean, const float* stdv, float* out, float* workspace, float* keep,
                          int arithmetic, nvsr_stream_t stream_) {
    if (!lr || !packed || !out || !workspace) return NVSR_ERR_NULL;
    if ((mean == nullptr) != (stdv == nullptr)) return NVSR_ERR_NULL;
    hipStream_t stream = (hipStream_t)stream_;
    const int sf = 1 << n_up;
    int lo[2], hi[2];
    sr_roi(R0, R1, roi, lo, hi);
    const int ch = hi[0] - lo[0], cw = hi[1] - lo[1];
    const int Hp = ch + 2 * pad, Wp = cw + 2 * pad;
    int Ho, Wo;
    if (int e = nvsr_edsr_out_size(Hp, Wp, nblocks, n_up, &Ho, &Wo)) return e;
    if (Ho != ch * sf + 2 * over || Wo != cw * sf + 2 * over) return NVSR_ERR_SHAPE;   // pad/over inconsistent with the net
    float* xin = keep ? keep : workspace;
    float* diff = workspace + (int64_t)Cc * Hp * Wp;
    float* ews = diff + (int64_t)Cc * Ho * Wo;
    const int64_t n_in = (int64_t)Cc * Hp * Wp;
    hipLaunchKernelGGL(sr_prepare_kernel, dim3((unsigned)((n_in + 255) / 256)), dim3(256), 0, stream, lr, Cc, R0, R1, lo[0], lo[1], Hp, Wp,
                       pad, mean, stdv, xin);
    if (int e = NVSR_CHECK_LAUNCH()) return e;
    if (keep) {
        if (int e = nvsr_edsr_forward_train_arith(xin, Cc, Hp, Wp, packed, Cc, hid, nblocks, n_up, diff, keep + ((int64_t)Cc * Hp * Wp + 3) / 4 * 4, arithmetic, stream_))
            return e;
    } else if (int e = nvsr_edsr_forward_batch_arith(xin, 1, Cc, Hp, Wp, packed, Cc, hid, nblocks, n_up, diff, ews, arithmetic, stream_)) return e;
    const int64_t n_out = (int64_t)Cc * R0 * sf * R1 * sf;
    hipLaunchKernelGGL(sr_finish_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, stream, diff, Ho, Wo, over, lr, Cc, R0, R1, sf,
                       lo[0], lo[1], hi[0], hi[1], out, conv_resolve_arith(arithmetic) == NVSR_ARITH_F16X2 ? nvsr_get_range_flag() : nullptr, sr_align_corners(), sr_bicubic());
    return NVSR_CHECK_LAUNCH();
}


/* ---- SR training on B regions of interest at once (the position planes of a scene, one crop each: models.py:270-284) ------------------------
 * One launch per layer for all planes (ragged batch, sr_core.h ConvRagged).  rois: B x 4 HOST floats ([[ymin,xmin],[ymax,xmax]] per plane) or
 * NULL (full planes).  keep: ONE buffer of nvsr_planes_sr_batch_keep_floats floats (plane after plane: prepared input + activation record);
 * workspace: nvsr_planes_sr_batch_workspace_floats floats.  align_corners / plane_interp of the residual up-sampling are ARGUMENTS here
 * (the one-plane entry points read the process-wide setting).  B <= 4. */
static int batch_geometry(int B, int Cc, int R0, int R1, int hid, int nblocks, int n_up, int pad, const float* rois, int (*lo)[2], int (*hi)[2],
                          int* Hp, int* Wp, EdsrPlan* P) {
    if (B < 1 || B > CONV_RAGGED_MAX) return NVSR_ERR_SHAPE;
    for (int b = 0; b < B; ++b) {
        sr_roi(R0, R1, rois ? rois + 4 * b : nullptr, lo[b], hi[b]);
        Hp[b] = hi[b][0] - lo[b][0] + 2 * pad; Wp[b] = hi[b][1] - lo[b][1] + 2 * pad;
        if (P) { if (int e = edsr_plan(Cc, Cc, hid, nblocks, n_up, Hp[b], Wp[b], &P[b])) return e; }
    }
    return NVSR_OK;
}
int64_t nvsr_planes_sr_batch_keep_floats(int B, int Cc, int R0, int R1, int hid, int nblocks, int n_up, int pad, const float* rois) {
    if (B < 1 || B > CONV_RAGGED_MAX) return -1;
    int64_t s = 0;
    for (int b = 0; b < B; ++b) {
        const int64_t k = nvsr_planes_sr_keep_floats(Cc, R0, R1, hid, nblocks, n_up, pad, rois ? rois + 4 * b : nullptr);
        if (k < 0) return -1;
        s += k;
    }
    return s;
}
int64_t nvsr_planes_sr_batch_workspace_floats(int B, int Cc, int R0, int R1, int hid, int nblocks, int n_up, int pad, const float* rois) {
    if (B < 1 || B > CONV_RAGGED_MAX) return -1;
    int64_t s = 0;
    for (int b = 0; b < B; ++b) {       // the network's output of every plane (the one-plane workspace also holds its input and ping-pong buffers)
        int lo[2], hi[2], Ho, Wo;
        sr_roi(R0, R1, rois ? rois + 4 * b : nullptr, lo, hi);
        if (nvsr_edsr_out_size(hi[0] - lo[0] + 2 * pad, hi[1] - lo[1] + 2 * pad, nblocks, n_up, &Ho, &Wo)) return -1;
        s += ((int64_t)Cc * Ho * Wo + 3) / 4 * 4;
    }
    return s;
}
int nvsr_planes_sr_train_batch_arith(const float* const* lr, int B, int Cc, int R0, int R1, const float* packed, int hid, int nblocks, int n_up,
                                     int pad, int over, const float* rois, const float* mean, const float* stdv, float* const* out, float* workspace,
                                     float* keep, int arithmetic, int align_corners, int plane_interp, nvsr_stream_t stream_) {
    if (!lr || !packed || !out || !workspace || !keep) return NVSR_ERR_NULL;
    if ((mean == nullptr) != (stdv == nullptr)) return NVSR_ERR_NULL;
    if (plane_interp != NVSR_PLANE_INTERP_BILINEAR && plane_interp != NVSR_PLANE_INTERP_BICUBIC) return NVSR_ERR_SHAPE;
    if (!aligned16(packed) || !aligned16(keep) || !aligned16(workspace)) return NVSR_ERR_ALIGN;
    if (B < 1 || B > CONV_RAGGED_MAX) return NVSR_ERR_SHAPE;
    for (int b = 0; b < B; ++b)
        if (!lr[b] || !out[b]) return NVSR_ERR_NULL;
    hipStream_t stream = (hipStream_t)stream_;
    const int sf = 1 << n_up;
    const int arith = conv_resolve_arith(arithmetic);
    int lo[CONV_RAGGED_MAX][2], hi[CONV_RAGGED_MAX][2], Hp[CONV_RAGGED_MAX], Wp[CONV_RAGGED_MAX];
    static thread_local EdsrPlan P[CONV_RAGGED_MAX];
    if (int e = batch_geometry(B, Cc, R0, R1, hid, nblocks, n_up, pad, rois, lo, hi, Hp, Wp, P)) return e;
    float* xin[CONV_RAGGED_MAX]; float* acts[CONV_RAGGED_MAX]; float* diff[CONV_RAGGED_MAX];
    float* k = keep; float* w = workspace;
    for (int b = 0; b < B; ++b) {
        if (P[b].Ho != (hi[b][0] - lo[b][0]) * sf + 2 * over || P[b].Wo != (hi[b][1] - lo[b][1]) * sf + 2 * over) return NVSR_ERR_SHAPE;
        const int64_t n_in = (int64_t)Cc * Hp[b] * Wp[b];
        xin[b] = k; acts[b] = k + (n_in + 3) / 4 * 4;
        k = acts[b] + P[b].acts_floats;
        diff[b] = w; w += ((int64_t)Cc * P[b].Ho * P[b].Wo + 3) / 4 * 4;
        hipLaunchKernelGGL(sr_prepare_kernel, dim3((unsigned)((n_in + 255) / 256)), dim3(256), 0, stream, lr[b], Cc, R0, R1, lo[b][0], lo[b][1], Hp[b],
                           Wp[b], pad, mean, stdv, xin[b]);
        if (int e = NVSR_CHECK_LAUNCH()) return e;
    }
    if (arith == NVSR_ARITH_F32 || B == 1) {
        // the exact-f32 kernels take no ragged batch (and one plane needs none): plane after plane
        for (int b = 0; b < B; ++b)
            if (int e = nvsr_edsr_forward_train_arith(xin[b], Cc, Hp[b], Wp[b], packed, Cc, hid, nblocks, n_up, diff[b], acts[b], arith, stream_)) return e;
    } else {
        const ConvExec cx{arith, 0};
        const float* wp = packed;
        for (int l = 0; l < P[0].n; ++l) {
            ConvRagged r;
            r.n = B;
            for (int b = 0; b < B; ++b) {
                r.H[b] = P[b].ih[l]; r.W[b] = P[b].iw[l];
                r.in[b] = l ? acts[b] + P[b].act_off[l] : xin[b];
                r.out[b] = (l + 1 < P[b].n) ? acts[b] + P[b].act_off[l + 1] : diff[b];
                r.skip[b] = (P[b].epi[l] == EPI_RESIDUAL) ? (l >= 2 ? acts[b] + P[b].act_off[l - 1] : xin[b]) : nullptr;      // the block's input
            }
            if (int e = launch_conv(nullptr, P[0].L[l].Cin, 0, 0, wp, P[0].L[l].Cout, P[0].epi[l], nullptr, nullptr, stream, 0, B, cx, &r)) return e;
            wp += conv_packed_floats(P[0].L[l].Cin, P[0].L[l].Cout);
        }
    }
    const int64_t n_out = (int64_t)Cc * R0 * sf * R1 * sf;
    for (int b = 0; b < B; ++b) {
        hipLaunchKernelGGL(sr_finish_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, stream, diff[b], P[b].Ho, P[b].Wo, over, lr[b], Cc, R0, R1,
                           sf, lo[b][0], lo[b][1], hi[b][0], hi[b][1], out[b], arith == NVSR_ARITH_F16X2 ? nvsr_get_range_flag() : nullptr,
                           align_corners ? 1 : 0, plane_interp == NVSR_PLANE_INTERP_BICUBIC ? 1 : 0);
        if (int e = NVSR_CHECK_LAUNCH()) return e;
    }
    return NVSR_OK;
}

}  // extern "C"
