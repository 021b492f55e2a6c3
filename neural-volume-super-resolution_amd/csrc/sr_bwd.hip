// Backward of the feature-plane super-resolution CNN for gfx950 ('SR' in nerf.train.what, train_nerf.py:75-77).
//
// The reference differentiates EDSR.forward (models.py:818-822), _Residual_Block.forward (:777-786) and PlanesSR.forward (:884-926)
// with torch.autograd (cuDNN conv backward-data / backward-filter).  Here:
//   * data gradient of a valid 3x3 conv = the forward implicit-GEMM kernel of sr.hip run on dy with a virtual 2-pixel zero border and
//     the flipped, transposed kernel (nvsr_pack_conv3x3_dgrad); the residual block's ReLU gate + x0.1 and its cropped identity are
//     fused epilogues (EPI_MASK_SCALE, EPI_ADD_CENTER);
//   * weight gradient dW[co][ci][ky][kx] = sum_{y,x} dy[co][y][x] * X[ci][y+ky][x+kx] is conv3x3_wgrad_kernel below: an MFMA
//     contraction with the PIXELS as K.  One dy fragment (A) feeds the 9 taps, whose B fragments are the same LDS patch of X read at
//     9 shifts.  Workgroups split the image rows; their partial sums go to a workspace and a second kernel reduces them in a fixed
//     order (deterministic, no float atomics) into the natural [co][ci][3][3] layout;
//   * PixelShuffle^T is an explicit re-layout kernel (0.1 % of the time of the convs around it).
#include "limb_core.h"
#include "sr_core.h"


namespace nvsr {

constexpr int WG_TPB = 256;                              // 4 waves; two workgroups per CU cover each other's barriers
constexpr int WG_CO = 64, WG_CI = 64, WG_PX = 32;       // workgroup tile: 2 co-waves x 2 ci-waves, 32 pixels of one row per step
constexpr int DY_STRIDE = WG_PX + 1;                     // odd strides: lanes (= channels) hit distinct LDS banks
constexpr int X_ROW = WG_PX + 3, X_CI = 3 * X_ROW;       // 35, 105
constexpr int DY_FLOATS = WG_CO * DY_STRIDE;             // 4224
constexpr int X_FLOATS = WG_CI * X_CI;                   // 6720
constexpr int DY_ITERS = WG_CO * WG_PX / WG_TPB;         // 8
static_assert(WG_TPB == 256 && WG_CO == 64 && WG_CI == 64 && WG_PX == 32, "the tile-fill index arithmetic assumes this shape");

struct WgradParams {
    const float* dy;     // [Cout][Ho][Wo]
    const float* x;      // [Cin][Ho+2][Wo+2]
    float* partial;      // [nslab][9][Cout][Cin]
    int Cin, Cout, Ho, Wo;
    int rows_per_slab;
    const unsigned* dy_absmax;   // f16 limbs: bits of max |dy| over the whole tensor (sr.hip absmax_kernel) -> dy's power-of-two scale; else NULL
    int n_wg;            // conv3x3_wgrad_limb_kernel: workgroups = pieces of the linear range of `total` row steps
    long total;
    // conv3x3_wgrad_limb_kernel: the planes whose contributions ONE pass accumulates (1 = dy / x / Ho / Wo above; up to CONV_RAGGED_MAX planes of
    // different sizes = the regions of interest of an SR training iteration: one weight gradient, one set of partial sums, one reduction).  A
    // tile's steps are its planes' (column chunk, row) steps one plane after the other: plane b owns steps [start[b], start[b + 1]) of TS.
    int nplanes, TS;
    int pHo[CONV_RAGGED_MAX], pWo[CONV_RAGGED_MAX], start[CONV_RAGGED_MAX + 1];
    const float* pdy[CONV_RAGGED_MAX];
    const float* px[CONV_RAGGED_MAX];
};
struct WgradPlane { const float* dy; const float* x; int Ho, Wo; };

__global__ __launch_bounds__(WG_TPB, 2) void conv3x3_wgrad_kernel(WgradParams p) {
    __shared__ float lds[DY_FLOATS + X_FLOATS];
    NVSR_RACE_PROBE_DELAY(lds);      // (probe builds only, nvsr_common.h)
    float* dyt = lds;
    float* xt = lds + DY_FLOATS;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, kh = lane >> 5;
    const int cw = wave & 1, iw = wave >> 1;
    const int co0 = blockIdx.x * WG_CO, ci0 = blockIdx.y * WG_CI, slab = blockIdx.z;
    const int W = p.Wo + 2;
    const long HoWo = (long)p.Ho * p.Wo, HW = (long)(p.Ho + 2) * W;
    const int ya = slab * p.rows_per_slab, yb = min(ya + p.rows_per_slab, p.Ho);
    const int nxc = (p.Wo + WG_PX - 1) / WG_PX;
    const int nsteps = (yb - ya) * nxc;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // Tile fill, 34 loads per thread and step.  Thread (t5 = tid>>5, c = tid&31) takes column c of rows t5, t5+8, ...:
    //   dy : row = co (64 rows);  X body: row = (r, ci) r-major (192 rows), columns 0..31;  X halo: columns 32, 33 of the 192 rows.
    // All index arithmetic is shifts and compile-time constants, so nothing per-element has to stay live across the MFMA loop.
    constexpr int XB_ITERS = 3 * WG_CI / 8;   // 24
    float rdy[DY_ITERS], rxb[XB_ITERS], rxh[2];
    const int t5 = tid >> 5, c = tid & 31;
    const int hrow = tid >> 1, hcol = WG_PX + (tid & 1);            // halo element of this thread: rows hrow and hrow + 128
    auto fetch = [&](int step) {
        const int y = ya + step / nxc, x0 = (step % nxc) * WG_PX;
        const bool px_ok = x0 + c < p.Wo;
        const float* dp = p.dy + (long)(co0 + t5) * HoWo + (long)y * p.Wo + x0 + c;
#pragma unroll
        for (int it = 0; it < DY_ITERS; ++it)
            rdy[it] = (px_ok && co0 + t5 + 8 * it < p.Cout) ? dp[(long)(8 * it) * HoWo] : 0.0f;
        // columns past the image edge only ever meet dy == 0: any finite value will do, so clamp the address
        const float* xp = p.x + (long)(ci0 + t5) * HW + (long)y * W + min(x0 + c, W - 1);
#pragma unroll
        for (int it = 0; it < XB_ITERS; ++it) {
            const int r = it / 8, cib = (it % 8) * 8;
            rxb[it] = (ci0 + t5 + cib < p.Cin) ? xp[(long)cib * HW + r * W] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int row = hrow + 128 * k, r = row >> 6, ci = row & 63;
            rxh[k] = (row < 3 * WG_CI && ci0 + ci < p.Cin) ? p.x[(long)(ci0 + ci) * HW + (long)(y + r) * W + min(x0 + hcol, W - 1)] : 0.0f;
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int it = 0; it < DY_ITERS; ++it) dyt[(t5 + 8 * it) * DY_STRIDE + c] = rdy[it];
#pragma unroll
        for (int it = 0; it < XB_ITERS; ++it) xt[(t5 + (it % 8) * 8) * X_CI + (it / 8) * X_ROW + c] = rxb[it];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int row = hrow + 128 * k;
            if (row < 3 * WG_CI) xt[(row & 63) * X_CI + (row >> 6) * X_ROW + hcol] = rxh[k];
        }
    };

    if (nsteps > 0) fetch(0);
    const float* A = dyt + (cw * 32 + i) * DY_STRIDE + kh;
    const float* B = xt + (iw * 32 + i) * X_CI + kh;
    for (int step = 0; step < nsteps; ++step) {
        __syncthreads();                       // everyone is done reading the previous tile
        stage();
        __syncthreads();
        if (step + 1 < nsteps) fetch(step + 1);   // global loads fly under the MFMAs below
#pragma unroll 4
        for (int kk = 0; kk < WG_PX / 2; ++kk) {
            const float a = A[2 * kk];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float b0 = B[ky * X_ROW + 2 * kk], b1 = B[ky * X_ROW + 2 * kk + 1], b2 = B[ky * X_ROW + 2 * kk + 2];
                acc[ky * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[ky * 3 + 0], 0, 0, 0);
                acc[ky * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[ky * 3 + 1], 0, 0, 0);
                acc[ky * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b2, acc[ky * 3 + 2], 0, 0, 0);
            }
        }
    }
    // D[row = co][col = ci]: lanes run along ci -> 128-byte rows of the [tap][co][ci] partial
    const int ci = ci0 + iw * 32 + i;
    if (ci < p.Cin) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + cw * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (co < p.Cout) p.partial[(((long)slab * 9 + t) * p.Cout + co) * p.Cin + ci] = acc[t][r];
            }
    }
}

// ---- the same contraction on the bf16 matrix pipe (3 exact bf16 limbs per f32 operand, limb_core.h) ---------------------------------
// Both operands are activations here, so both are split when they are staged: dy as [limb][co][40 px] and X as [limb][ci][3 rows][40 px]
// bf16 (20-word rows: conflict-free ds_read_b128 for 16 consecutive channels), 8 consecutive pixels = one operand fragment.  A K-block is
// 16 pixels of the row; the three kx taps of an X row come from ONE 5-word read: words 0..3, words 1..4, and their 16-bit funnel shift.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));    // 16-byte global load from a 4-byte aligned address
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split4(const float (&e)[4], u32x2 (&out)[3]) {   // 3-limb split of 4 values -> 2 words per limb
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float r0 = e[2 * j], r1 = e[2 * j + 1];
        out[0][j] = trunc_pair(r1, r0);
        r0 = limb_rest(r0); r1 = limb_rest(r1);
        out[1][j] = trunc_pair(r1, r0);
        r0 = limb_rest(r0); r1 = limb_rest(r1);
        out[2][j] = trunc_pair(r1, r0);
    }
}
// 2 f16 limbs (round to nearest, limb_core.h) of 4 values times a power of two
__device__ __forceinline__ void split4_f16(const float (&e)[4], float scale, u32x2 (&out)[3]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float v0 = e[2 * j] * scale, v1 = e[2 * j + 1] * scale;
        const unsigned hi = f16_pair(v1, v0);
        out[0][j] = hi;
#if F16_MIX_SPLIT
        unsigned lo;
        f16_low_limb_lo(lo, v0, hi);
        f16_low_limb_hi(lo, v1, hi);
        out[1][j] = lo;
#else
        out[1][j] = f16_pair(f16_rest<1>(v1, hi), f16_rest<0>(v0, hi));
#endif
    }
}
constexpr int WL_ROW = 20;                                // words per row of 40 bf16 pixels
constexpr int WL_DY_WORDS = WG_CO * WL_ROW;               // one limb of dy
constexpr int WL_X_WORDS = WG_CI * 3 * WL_ROW;            // one limb of X
#ifndef WG_STAMP
#define WG_STAMP 0      // debug builds: cycles per step in 6 sections, wave 0 of every workgroup -> 8 floats behind the partial slots (tools/conv_wgrad_time.py)
#endif
#if WG_STAMP
#define WG_MARK(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamp[i] += (float)(t_ - tprev); tprev = t_; __builtin_amdgcn_sched_barrier(0); }
#else
#define WG_MARK(i)
#endif
#ifndef WG_ABLATE
#define WG_ABLATE 0     // variant builds of tools/conv_wgrad_ablate.sh: 1 no global fetches, 2 no split + LDS writes, 4 no MFMAs
#endif

// Work split: the row steps of a layer -- (tile of 64 co x 64 ci, 32-pixel column chunk, output row), in that order -- form ONE linear
// range that is cut into n_wg equal pieces, one per workgroup (n_wg = the 512 workgroup slots of the chip when the layer is large enough):
// every workgroup does the same number of steps and all of them are resident from the start, so there is no partly filled last round
// (the former (chunk, row range) slabs gave 720 workgroups for a trunk layer: 1.4 rounds).  A piece that crosses a chunk boundary restarts
// its X-row ring; one that crosses a TILE boundary (at most once: pieces are no longer than a tile) writes its accumulators and starts
// over -- a workgroup owns two partial slots [9][64][64], and wgrad_reduce_pieces_kernel adds, for every tile, the slots of the pieces
// that overlap it in piece order (deterministic).
__device__ __forceinline__ long wgrad_piece_start(long w, long total, int n_wg) { return w * total / n_wg; }

// rows [ya, yb) of column chunk x0 of tile (co0, ci0), accumulated into acc.
template <int LF>
__device__ __forceinline__ void wgrad_limb_rows(const WgradParams& p, const WgradPlane& pl, unsigned* lds, f32x16 (&acc)[9], int co0, int ci0, int x0, int ya, int yb, float dscale
#if WG_STAMP
                                                , float (&stamp)[8]
#endif
                                                ) {
    unsigned* dyl = lds;                                  // [limb][co][WL_ROW]
    unsigned* xl = lds + LF * WL_DY_WORDS;                // [limb][ci][row][WL_ROW]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, kh = lane >> 5;
    const int cw = wave & 1, iw = wave >> 1;
    const int W = pl.Wo + 2;
    const long HoWo = (long)pl.Ho * pl.Wo, HW = (long)(pl.Ho + 2) * W;

    // A workgroup walks DOWN a 32-pixel column chunk: consecutive steps share two of their three X rows, which stay in LDS (ring of 3 row
    // slots, row ya + k in slot k % 3) -- a step stages dy and ONE new X row.
    // Staging unit = a QUAD of 4 consecutive pixels, quads numbered along the row and then down the channels, thread -> quads tid + 256 k:
    // a wave's load instruction covers whole contiguous rows (8 rows of dy, 7 of X) with one 16-byte request per lane (4-byte aligned:
    // rows start anywhere), a quad becomes 2 words of every limb.  dy: 64 co x 8 quads; X: 64 ci x 9 quads (32 + 2 halo pixels, padded).
    const int rows = yb - ya;
    // Thread -> quads.  dy (64 co x 8 quads): quad tid & 7 of rows (tid >> 3) + 32 k, k < 2.  X (64 ci x 9 quads: 32 + 2 halo pixels,
    // padded): lane l < 63 of wave w takes quad l % 9 of rows 7 w + l / 9 + 28 k, k < 3 (rows >= 64 do not exist).  The rows of one
    // thread differ by a constant, so ONE 32-bit byte offset per operand serves all of them (scalar base per k: the saddr form of
    // global_load, no 64-bit address registers), and so does one LDS address.  Channels past Cout / Cin are not loaded: whatever is
    // staged for them only reaches accumulator rows / columns the reduction never reads.
    float rdy[2][4], rx[3][4];
    const int dco = tid >> 3, xr = 7 * wave + lane / 9, xq = lane % 9;
    const unsigned dy_voff = ((unsigned)dco * (unsigned)HoWo + 4 * (tid & 7)) * 4u;      // (the launcher checks that 64 channels fit 2^31 bytes)
    const unsigned x_voff = ((unsigned)xr * (unsigned)HW + 4 * xq) * 4u;
    const int dy_valid = min(max(pl.Wo - (x0 + 4 * (tid & 7)), 0), 4);   // pixels past the row's end contribute nothing: they are zeroed
    unsigned* const dy_st = dyl + dco * WL_ROW + 2 * (tid & 7);
    unsigned* const x_st = xl + xr * 3 * WL_ROW + 2 * xq;
    const int n_co = min(WG_CO, p.Cout - co0), n_ci = min(WG_CI, p.Cin - ci0);
    const bool x_lane = lane < 63;
    const long dy_rest = (long)(p.Cout - co0) * HoWo - x0, x_rest = (long)(p.Cin - ci0) * HW - x0;   // elements from the bases below to the end
    const float* const dy_base = pl.dy + (long)co0 * HoWo + x0;
    const float* const x_base = pl.x + (long)ci0 * HW + x0;
    // 4 consecutive floats; what lies past a row's end is the next row (finite; it only ever meets dy == 0), what would lie past the
    // tensor's end is not touched: a quad of the LAST row of the LAST channel may reach there, and rows that can (a scalar test per row)
    // take the guarded loads -- a guard is a divergent branch, and hipcc answers each with s_waitcnt vmcnt(0), which serialises the row's
    // loads; all other rows take the bare 16-byte loads
    auto load4 = [&](const float* base, unsigned voff, long rest, float (&o)[4], bool guard) {   // `rest` elements are left from base
        if (!guard || (long)(voff >> 2) + 4 <= rest) {
            const f32x4u v = *reinterpret_cast<const f32x4u*>(reinterpret_cast<const char*>(base) + voff);
            o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = base[min((long)(voff >> 2) + j, rest - 1)];
        }
    };
    const long x_last = (long)(n_ci - 1) * HW + 36, dy_last = (long)(n_co - 1) * HoWo + 32;
    auto fetch_row = [&](int row, float (&dst)[3][4]) {        // X row `row` of this chunk
        const float* b = x_base + (long)row * W;
        const long rest = x_rest - (long)row * W;
        const bool guard = x_last > rest;                      // (uniform)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (x_lane && xr + 28 * k < n_ci) {
                if (guard) load4(b + 28L * k * HW, x_voff, rest - 28L * k * HW, dst[k], true);
                else load4(b + 28L * k * HW, x_voff, 0, dst[k], false);
            }
        }
    };
    auto stage_row = [&](int slot, const float (&src)[3][4]) { // -> row slot `slot`
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            u32x2 L[3];
            if constexpr (LF == 2) split4_f16(src[k], F16_X_SCALE, L); else split4(src[k], L);
            if (x_lane && xr + 28 * k < WG_CI) {
#pragma unroll
                for (int t = 0; t < LF; ++t) *reinterpret_cast<u32x2*>(x_st + t * WL_X_WORDS + (28 * k * 3 + slot) * WL_ROW) = L[t];
            }
        }
    };
    auto fetch = [&](int yr) {                                // what step yr adds: dy row ya + yr, X row ya + yr + 2
        const float* b = dy_base + (long)(ya + yr) * pl.Wo;
        const long rest = dy_rest - (long)(ya + yr) * pl.Wo;
        const bool guard = dy_last > rest;                     // (uniform)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (dco + 32 * k < n_co) {
                if (guard) load4(b + 32L * k * HoWo, dy_voff, rest - 32L * k * HoWo, rdy[k], true);
                else load4(b + 32L * k * HoWo, dy_voff, 0, rdy[k], false);
            }
        }
        fetch_row(ya + yr + 2, rx);
    };
    auto stage_dy = [&]() {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
#pragma unroll
            for (int j = 0; j < 4; ++j) rdy[k][j] = j < dy_valid ? rdy[k][j] : 0.0f;
            u32x2 L[3];
            if constexpr (LF == 2) split4_f16(rdy[k], dscale, L); else split4(rdy[k], L);
#pragma unroll
            for (int t = 0; t < LF; ++t) *reinterpret_cast<u32x2*>(dy_st + t * WL_DY_WORDS + 32 * k * WL_ROW) = L[t];
        }
    };

    const unsigned* Ap = dyl + (cw * 32 + i) * WL_ROW + kh * 4;
    const unsigned* Bp = xl + ((iw * 32 + i) * 3) * WL_ROW + kh * 4;
    // one X fragment read = a conflict-free ds_read_b128 (words 0..3) + a ds_read_b32 (word 4); the three kx taps are built from these
    // five REGISTERS: left alone, hipcc fetches words 1..4 again as two ds_read2_b32 (no v_mov needed then) -- 4-way bank conflicts each,
    // 32 LDS cycles instead of 4 (SQ_LDS_BANK_CONFLICT was 64 % of SQ_LDS_IDX_ACTIVE, the LDS 57 % busy)
    struct Raw { u32x4 w; unsigned w4; };
    auto read_x = [&](int t, int off) {
        const unsigned* bp = Bp + t * WL_X_WORDS + off;
        Raw r;
        r.w = *reinterpret_cast<const u32x4*>(bp);
        r.w4 = bp[4];
        return r;
    };
    auto taps = [&](Raw& r, u32x4 (&b)[3]) {
        asm volatile("" : "+v"(r.w), "+v"(r.w4));
        b[0] = r.w;
        b[2] = u32x4{r.w[1], r.w[2], r.w[3], r.w4};
        b[1] = u32x4{__builtin_amdgcn_alignbit(r.w[1], r.w[0], 16), __builtin_amdgcn_alignbit(r.w[2], r.w[1], 16),
                     __builtin_amdgcn_alignbit(r.w[3], r.w[2], 16), __builtin_amdgcn_alignbit(r.w4, r.w[3], 16)};
    };

    __syncthreads();                           // the previous rows' (piece's) fragment reads are done: the ring may be overwritten
    if (rows > 0) {
        // prologue (exposed, once per call): X rows ya, ya + 1; all loads first
        float r0[3][4], r1[3][4];
        fetch_row(ya, r0);
        fetch_row(ya + 1, r1);
        fetch(0);
        stage_row(0, r0);
        stage_row(1, r1);
    }
#if WG_STAMP
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
#endif
    for (int yr = 0; yr < rows; ++yr) {
        WG_MARK(5)
        __syncthreads();                       // everyone is done reading the previous tile
        WG_MARK(0)
        if (!(WG_ABLATE & 2)) {
            stage_dy();
            stage_row((yr + 2) % 3, rx);
        }
        WG_MARK(1)
        __syncthreads();
        WG_MARK(2)
        if (!(WG_ABLATE & 1) && yr + 1 < rows) fetch(yr + 1);      // global loads fly under the MFMAs below
        WG_MARK(3)
        int rowoff[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) rowoff[ky] = ((yr + ky) % 3) * WL_ROW;
        // 6 groups g = (K-block kb of 16 pixels, ky): 18 MFMAs each, products in the order of limb_w / limb_x (small terms first), which
        // starts with X limb 2: that limb's words of group g + 1 are requested before the MFMAs of group g, limbs 1 and 0 at the start of
        // their own group, under its first MFMAs
        if constexpr (LF == 2) {
            // 2 f16 limbs: 9 MFMAs per group -- dy_hi X_lo, dy_hi X_hi, dy_lo X_hi (limb_w / limb_x of 2 limbs); X_lo of the next group is
            // requested before this group's MFMAs
            static_assert(limb_x(2, 0) == 1 && limb_x(2, 1) == 0 && limb_x(2, 2) == 0 && limb_w(2, 0) == 0 && limb_w(2, 1) == 0 && limb_w(2, 2) == 1, "");
            Raw nxt = read_x(1, rowoff[0]);
            u32x4 A[2];
#pragma unroll
            for (int g = 0; g < 6; ++g) {
                const int kb = g / 3, ky = g % 3, off = rowoff[ky] + kb * 8;
                __builtin_amdgcn_sched_barrier(0);
                if (ky == 0) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) A[t] = *reinterpret_cast<const u32x4*>(Ap + t * WL_DY_WORDS + kb * 8);
                }
                Raw r0 = read_x(0, off);
                u32x4 B1[3], B0[3];
                taps(nxt, B1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = mfma_limb<2>(A[0], B1[kx], acc[ky * 3 + kx]);
                if (g < 5) nxt = read_x(1, rowoff[(g + 1) % 3] + ((g + 1) / 3) * 8);
                __builtin_amdgcn_sched_barrier(0);
                taps(r0, B0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = mfma_limb<2>(A[0], B0[kx], acc[ky * 3 + kx]);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = mfma_limb<2>(A[1], B0[kx], acc[ky * 3 + kx]);
            }
        } else {
        Raw nxt = read_x(2, rowoff[0]);
        u32x4 A[3];
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            const int kb = g / 3, ky = g % 3, off = rowoff[ky] + kb * 8;
            __builtin_amdgcn_sched_barrier(0);
            if (ky == 0) {
#pragma unroll
                for (int t = 0; t < 3; ++t) A[t] = *reinterpret_cast<const u32x4*>(Ap + t * WL_DY_WORDS + kb * 8);
            }
            Raw r1 = read_x(1, off), r0 = read_x(0, off);
            u32x4 B2[3], B1[3], B0[3];
            taps(nxt, B2);
            __builtin_amdgcn_sched_barrier(0);
#if !(WG_ABLATE & 4)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = mfma_bf16(A[limb_w(3, 0)], B2[kx], acc[ky * 3 + kx]);
#endif
            if (g < 5) nxt = read_x(2, rowoff[(g + 1) % 3] + ((g + 1) / 3) * 8);
            __builtin_amdgcn_sched_barrier(0);
            taps(r1, B1);
            __builtin_amdgcn_sched_barrier(0);
#if !(WG_ABLATE & 4)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = mfma_bf16(A[limb_w(3, 1)], B1[kx], acc[ky * 3 + kx]);
#endif
            __builtin_amdgcn_sched_barrier(0);
            taps(r0, B0);
            __builtin_amdgcn_sched_barrier(0);
#if !(WG_ABLATE & 4)
#pragma unroll
            for (int q = 2; q < 6; ++q) {
                static_assert(limb_x(3, 0) == 2 && limb_x(3, 1) == 1 && limb_x(3, 2) == 0 && limb_x(3, 3) == 1 && limb_x(3, 4) == 0 && limb_x(3, 5) == 0, "");
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    acc[ky * 3 + kx] = mfma_bf16(A[limb_w(3, q)], limb_x(3, q) == 1 ? B1[kx] : B0[kx], acc[ky * 3 + kx]);
            }
#else
#pragma unroll
            for (int t = 0; t < 3; ++t) asm volatile("" :: "v"(A[t]), "v"(B0[t]), "v"(B1[t]), "v"(B2[t]));
#endif
        }
        }
        WG_MARK(4)
#if WG_STAMP
        stamp[6] += 1.0f;
#endif
    }
}

template <int LF>
__global__ __launch_bounds__(WG_TPB, 2) void conv3x3_wgrad_limb_kernel(WgradParams p) {
    __shared__ __attribute__((aligned(16))) unsigned lds[LF * (WL_DY_WORDS + WL_X_WORDS)];
    NVSR_RACE_PROBE_DELAY(lds);      // (probe builds only, nvsr_common.h)
    float dscale = 1.0f;          // f16 limbs: the power of two that puts the largest |dy| of the tensor into [2^12, 2^13) (sr.hip, the data gradient's rule)
    if (LF == 2 && p.dy_absmax) {
        dscale = f16_gradient_scale((unsigned)__builtin_amdgcn_readfirstlane((int)*p.dy_absmax));
    }
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, kh = lane >> 5;
    const int cw = wave & 1, iw = wave >> 1;
    const int n_ci = (p.Cin + WG_CI - 1) / WG_CI;
    const int TS = p.TS;                                   // steps of a tile, all planes (the launcher checks that `total` fits an int)
    const int wg = blockIdx.x;
    // (integer division runs on the vector ALU: without readfirstlane everything derived from the quotients counts as divergent, the
    //  TAIL branch below becomes a divergent one and the accumulators are spilled around it)
    int s = __builtin_amdgcn_readfirstlane((int)wgrad_piece_start(wg, p.total, p.n_wg));
    const int s1 = __builtin_amdgcn_readfirstlane((int)wgrad_piece_start(wg + 1, p.total, p.n_wg));
    int slot = 0;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

#if WG_STAMP
    float stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    while (s < s1) {
        const int tile = __builtin_amdgcn_readfirstlane(s / TS), rem = s - tile * TS;
        int b = 0;                                         // the plane this step belongs to (workgroup-uniform)
#pragma unroll
        for (int k = 1; k < CONV_RAGGED_MAX; ++k) b = (k < p.nplanes && rem >= p.start[k]) ? k : b;
        b = __builtin_amdgcn_readfirstlane(b);
        // (constant indices + selects: a dynamic index into the by-value parameter struct would put the whole struct into scratch)
        WgradPlane pl{p.pdy[0], p.px[0], p.pHo[0], p.pWo[0]};
        int start_b = 0;
#pragma unroll
        for (int k = 1; k < CONV_RAGGED_MAX; ++k)
            if (b == k) { pl = WgradPlane{p.pdy[k], p.px[k], p.pHo[k], p.pWo[k]}; start_b = p.start[k]; }
        const int r2 = rem - start_b;
        const int chunk = __builtin_amdgcn_readfirstlane(r2 / pl.Ho), ya = r2 - chunk * pl.Ho;
        const int seg_end = min(s1, tile * TS + start_b + (chunk + 1) * pl.Ho);
        const int yb = ya + (seg_end - s);
        const int cob = __builtin_amdgcn_readfirstlane(tile / n_ci);
        const int co0 = cob * WG_CO, ci0 = (tile - cob * n_ci) * WG_CI, x0 = chunk * WG_PX;
#if WG_STAMP
        wgrad_limb_rows<LF>(p, pl, lds, acc, co0, ci0, x0, ya, yb, dscale, stamp);
#else
        wgrad_limb_rows<LF>(p, pl, lds, acc, co0, ci0, x0, ya, yb, dscale);
#endif
        s = seg_end;
        if (s == s1 || s == (tile + 1) * TS) {
            // D[row = co][col = ci]: lanes run along ci -> 128-byte rows of the slot's [tap][co][ci]
            float* dst = p.partial + ((long)wg * 2 + slot) * (9 * WG_CO * WG_CI) + iw * 32 + i;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    dst[(t * WG_CO + cw * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * WG_CI] = acc[t][r];
                    acc[t][r] = 0.0f;
                }
            ++slot;
        }
    }
#if WG_STAMP
    if (tid == 0)
        for (int k = 0; k < 8; ++k) p.partial[(long)p.n_wg * 2 * (9 * WG_CO * WG_CI) + wg * 8 + k] = stamp[k];
#endif
}

// dw[co][ci][tap] += scale * (sum over the pieces that overlap the element's tile, in piece order, of their slot for that tile)
__global__ void wgrad_reduce_pieces_kernel(const float* __restrict__ partial, int n_wg, long total, long TS, int Cout, int Cin, float scale,
                                           float* __restrict__ dw, const unsigned* __restrict__ dy_absmax) {
    if (dy_absmax) {          // f16 limbs: the partial sums carry 2^F16_SX (X) times dy's scale
        const float dscale = f16_gradient_scale(*dy_absmax);
        scale *= 1.0f / (F16_X_SCALE * dscale);
    }
    const int n_ci = (Cin + WG_CI - 1) / WG_CI;
    const int ci = blockIdx.x * 64 + (threadIdx.x & 63), co = blockIdx.y * 4 + (threadIdx.x >> 6), t = blockIdx.z;
    if (ci >= Cin || co >= Cout) return;
    const long tile = (long)(co / WG_CO) * n_ci + ci / WG_CI, lo = tile * TS, hi = lo + TS;
    // pieces [w0, w1) overlap the tile (uniform over the workgroup: 64 ci and 4 co of ONE tile).  w0 = the last piece that starts at or before
    // lo: it alone may have started in the previous tile (its slot 1 belongs to this one); every later piece starts inside the tile (slot 0)
    long w0 = lo * n_wg / total;
    while (w0 + 1 < n_wg && wgrad_piece_start(w0 + 1, total, n_wg) <= lo) ++w0;
    while (w0 > 0 && wgrad_piece_start(w0, total, n_wg) > lo) --w0;
    long w1 = hi * n_wg / total;
    if (w1 > n_wg) w1 = n_wg;
    while (w1 < n_wg && wgrad_piece_start(w1, total, n_wg) < hi) ++w1;
    while (w1 > w0 + 1 && wgrad_piece_start(w1 - 1, total, n_wg) >= hi) --w1;
    constexpr long S = 9L * WG_CO * WG_CI;
    const long e = ((long)t * WG_CO + (co % WG_CO)) * WG_CI + (ci % WG_CI);
    const float* __restrict__ src = partial + e;
    float sum = 0.0f;
    if (wgrad_piece_start(w0, total, n_wg) < hi) sum = src[(w0 * 2 + (wgrad_piece_start(w0, total, n_wg) >= lo ? 0 : 1)) * S];
    // the same fixed order as ever (piece after piece), eight loads in flight: a trunk layer's element is the sum of ~32 slots 147 KB apart, and
    // one load per round trip (the loop used to recompute the piece boundaries -- two 64-bit divisions -- before every load) made the kernel a
    // chain of memory latencies: 36 us for 75 MB
    long w = w0 + 1;
#ifndef WG_REDUCE_UNROLL
#define WG_REDUCE_UNROLL 1     // 0: one load per round trip (A/B builds)
#endif
    for (; WG_REDUCE_UNROLL && w + 8 <= w1; w += 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = src[(w + k) * 2 * S];
#pragma unroll
        for (int k = 0; k < 8; ++k) sum += v[k];
    }
    for (; w < w1; ++w) sum += src[w * 2 * S];
    dw[((long)co * Cin + ci) * 9 + t] += scale * sum;
}

// dw[co][ci][tap] += scale * sum_slab partial[slab][tap][co][ci]
__global__ void wgrad_reduce_kernel(const float* __restrict__ partial, int nslab, int Cout, int Cin, float scale, float* __restrict__ dw) {
    const long n = 9L * Cout * Cin;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    float s = 0.0f;
    for (int k = 0; k < nslab; ++k) s += partial[k * n + idx];
    const int t = (int)(idx / ((long)Cout * Cin));
    const long cc = idx - (long)t * Cout * Cin;       // co * Cin + ci
    dw[cc * 9 + t] += scale * s;
}

// PixelShuffle(2)^T: g [C][2h][2w] -> out [4C][h][w]
__global__ void pixel_unshuffle_kernel(const float* __restrict__ g, int Cc, int h, int w, float* __restrict__ out) {
    const long n = 4L * Cc * h * w;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int x = (int)(idx % w), y = (int)((idx / w) % h), c4 = (int)(idx / ((long)w * h));
    out[idx] = g[((long)(c4 >> 2) * 2 * h + 2 * y + ((c4 >> 1) & 1)) * 2 * w + 2 * x + (c4 & 1)];
}

// PlanesSR backward, output side (models.py:915-923): d_out [C][sf R0][sf R1] -> d_diff [C][Ho][Wo] (zero in the over-padding
// ring) and, when d_lr != NULL, the bilinear residual's share of d_lr (4 float atomics per HR pixel of the ROI)
__global__ void sr_finish_backward_kernel(const float* __restrict__ d_out, int Cc, int R0, int R1, int sf, int lo0, int lo1, int hi0, int hi1,
                                          int Ho, int Wo, int over, float* __restrict__ d_diff, float* __restrict__ d_lr, int align, int bicubic) {
    const long n = (long)Cc * Ho * Wo;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int dx = (int)(i % Wo), dy = (int)((i / Wo) % Ho), c = (int)(i / ((long)Wo * Ho));
    const int ry = dy - over, rx = dx - over;                 // position inside the ROI, HR pixels
    const int ch = (hi0 - lo0) * sf, cw = (hi1 - lo1) * sf;
    if (ry < 0 || ry >= ch || rx < 0 || rx >= cw) { d_diff[i] = 0.0f; return; }
    const int HR0 = R0 * sf, HR1 = R1 * sf;
    const int oy = lo0 * sf + ry, ox = lo1 * sf + rx;
    const float g = d_out[((long)c * HR0 + oy) * HR1 + ox];
    d_diff[i] = g;
    if (!d_lr) return;
    if (bicubic) {
        const CubicTap ty = cubic_tap(oy, R0, HR0, sf, align), tx = cubic_tap(ox, R1, HR1, sf, align);
        float* pc = d_lr + (long)c * R0 * R1;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) unsafeAtomicAdd(pc + (long)ty.i[i] * R1 + tx.i[k], g * (tx.w[k] * ty.w[i]));
        return;
    }
    const BilinearTap ty = bilinear_tap(oy, R0, HR0, sf, align), tx = bilinear_tap(ox, R1, HR1, sf, align);
    const int y0 = ty.i0, x0 = tx.i0, yp = ty.step, xp = tx.step;
    const float ly1 = ty.w1, ly0 = 1.0f - ly1, lx1 = tx.w1, lx0 = 1.0f - lx1;
    float* q = d_lr + ((long)c * R0 + y0) * R1 + x0;
    unsafeAtomicAdd(q, g * ly0 * lx0);
    unsafeAtomicAdd(q + xp, g * ly0 * lx1);
    unsafeAtomicAdd(q + (long)yp * R1, g * ly1 * lx0);
    unsafeAtomicAdd(q + (long)yp * R1 + xp, g * ly1 * lx1);
}

// The bilinear residual's share of d_lr (models.py:858-868 transposed) as a GATHER: one thread per LR texel sums the HR pixels of the region
// of interest whose two taps per axis include it -- candidates oy in [(y - 1) sf, (y + 2) sf] (both align_corners conventions: src(o) lies in
// (o / sf - 1, o / sf + 1)), each tested with bilinear_tap itself, so the weights are the forward's bit for bit.  The scatter this replaces
// (4 float atomics per HR pixel of the region, 16 sf^2 colliding on every texel) took 2.1 ms per plane at sf = 4; this takes ~0.2 ms.
template <int SF>
__global__ __launch_bounds__(256) void sr_residual_backward_gather_kernel(const float* __restrict__ d_out, int Cc, int R0, int R1, int lo0, int lo1,
                                                                          int hi0, int hi1, int ty0, int tx0, int th, int tw,
                                                                          float* __restrict__ d_lr, int align) {
    constexpr int NC = 3 * SF + 1;
    const long n = (long)Cc * th * tw;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = tx0 + (int)(i % tw), y = ty0 + (int)((i / tw) % th), c = (int)(i / ((long)tw * th));
    const int HR0 = R0 * SF, HR1 = R1 * SF;
    float wy[NC], wx[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int oy = (y - 1) * SF + k, ox = (x - 1) * SF + k;
        wy[k] = wx[k] = 0.0f;
        if (oy >= lo0 * SF && oy < hi0 * SF) {
            const BilinearTap t = bilinear_tap(oy, R0, HR0, SF, align);
            wy[k] = (t.i0 == y ? 1.0f - t.w1 : 0.0f) + (t.i0 + t.step == y ? t.w1 : 0.0f);
        }
        if (ox >= lo1 * SF && ox < hi1 * SF) {
            const BilinearTap t = bilinear_tap(ox, R1, HR1, SF, align);
            wx[k] = (t.i0 == x ? 1.0f - t.w1 : 0.0f) + (t.i0 + t.step == x ? t.w1 : 0.0f);
        }
    }
    const float* __restrict__ src = d_out + (long)c * HR0 * HR1;
    float sum = 0.0f;
#pragma unroll
    for (int ky = 0; ky < NC; ++ky) {
        if (wy[ky] == 0.0f) continue;
        const float* row = src + (long)((y - 1) * SF + ky) * HR1 + (x - 1) * SF;
        float rs = 0.0f;
#pragma unroll
        for (int kx = 0; kx < NC; ++kx)
            if (wx[kx] != 0.0f) rs += row[kx] * wx[kx];
        sum += rs * wy[ky];
    }
    d_lr[((long)c * R0 + y) * R1 + x] += sum;
}

// PlanesSR backward, input side: the clamped gather of sr_prepare_kernel transposed (replicate padding sums into the border texels)
__global__ void sr_prepare_backward_kernel(const float* __restrict__ dxin, int Cc, int R0, int R1, int lo0, int lo1, int Hp, int Wp, int pad,
                                           const float* __restrict__ stdv, float* __restrict__ d_lr) {
    const long n = (long)Cc * Hp * Wp;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int xx = (int)(i % Wp), y = (int)((i / Wp) % Hp), c = (int)(i / ((long)Wp * Hp));
    const int sy = min(max(lo0 - pad + y, 0), R0 - 1), sx = min(max(lo1 - pad + xx, 0), R1 - 1);
    float v = dxin[i];
    if (stdv) v /= stdv[c];
    unsafeAtomicAdd(d_lr + ((long)c * R0 + sy) * R1 + sx, v);
}

static int wgrad_slabs(int Cin, int Cout, int Ho) {
    const int tiles = ((Cout + WG_CO - 1) / WG_CO) * ((Cin + WG_CI - 1) / WG_CI);
    int ns = (640 + tiles - 1) / tiles;          // ~2.5 workgroups per CU
    if (ns > Ho) ns = Ho;
    return ns < 1 ? 1 : ns;
}
// limb kernel: equal pieces of the linear range of row steps (see conv3x3_wgrad_limb_kernel)
static long wgrad_total_steps(int Cin, int Cout, int Ho, int Wo) {
    return (long)((Cout + WG_CO - 1) / WG_CO) * ((Cin + WG_CI - 1) / WG_CI) * ((Wo + WG_PX - 1) / WG_PX) * Ho;
}
static int wgrad_pieces_of(long tiles, long total) {
    long n = total / 8;                      // at least ~8 steps per piece (prologue: 2 rows), ...
    if (n > 512) n = 512;                    // ... the chip's 512 workgroup slots (2 per CU) when the layer is large enough, ...
    if (n < tiles) n = tiles;                // ... and never longer than a tile: a piece crosses at most one tile boundary
    return (int)n;
}
static int wgrad_pieces(int Cin, int Cout, int Ho, int Wo) {
    return wgrad_pieces_of((long)((Cout + WG_CO - 1) / WG_CO) * ((Cin + WG_CI - 1) / WG_CI), wgrad_total_steps(Cin, Cout, Ho, Wo));
}
static int64_t wgrad_partial_floats(int Cin, int Cout, int Ho, int Wo) {
    const int64_t a = (int64_t)wgrad_slabs(Cin, Cout, Ho) * 9 * Cout * Cin, b = (int64_t)wgrad_pieces(Cin, Cout, Ho, Wo) * 2 * 9 * WG_CO * WG_CI;
    return a > b ? a : b;                                  // either kernel may run (nvsr_set_conv_arithmetic)
}
// partial sums of a pass over several planes (limb kernel): as many pieces as the pass can have
static int64_t wgrad_partial_floats_ragged(int Cin, int Cout) {
    const int64_t tiles = (int64_t)((Cout + WG_CO - 1) / WG_CO) * ((Cin + WG_CI - 1) / WG_CI);
    return (tiles > 512 ? tiles : 512) * 2 * 9 * WG_CO * WG_CI;
}

// The reductions of a backward pass on a stream of their own (round 5).  wgrad_reduce_pieces_kernel is the one memory-bound kernel of the layer loop
// (it reads the ~75 MB of partial sums a trunk layer's 512 pieces wrote: ~30 us at HBM rate, 2.5 ms of the 54 ms of a refine iteration's SR backward)
// between matrix-bound ones: on the caller's stream it holds the next data-gradient launch back; on a side stream it runs under that launch.  The
// partial sums alternate between two buffers: the reduction of layer l reads buffer l & 1 while layer l - 1's pieces fill the other one; the pieces of
// layer l - 2 wait for the reduction of layer l (an event that has long fired).  join() -- on every exit of the pass, errors included -- makes the
// caller's stream wait for the last reductions: the caller sees ordinary stream semantics.  Inside a stream capture the lane stays closed (everything on
// the caller's stream).
struct ReduceLane {
    hipStream_t side = nullptr;
    hipEvent_t ready[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr};
    int device = -1;
    bool ok = false;
    bool open_on(int dev) {
        if (ok && dev == device) return true;
        if (ok) close();
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) return false;      // (hi = the numerically lowest = greatest priority)
        if (hipStreamCreateWithPriority(&side, hipStreamNonBlocking, hi) != hipSuccess) { side = nullptr; return false; }
        for (int k = 0; k < 2; ++k)
            if (hipEventCreateWithFlags(&ready[k], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&done[k], hipEventDisableTiming) != hipSuccess) {
                close();
                return false;
            }
        device = dev; ok = true;
        return true;
    }
    void close() {
        for (int k = 0; k < 2; ++k) {
            if (ready[k]) (void)hipEventDestroy(ready[k]);
            if (done[k]) (void)hipEventDestroy(done[k]);
            ready[k] = done[k] = nullptr;
        }
        if (side) (void)hipStreamDestroy(side);
        side = nullptr; ok = false; device = -1;
    }
};
// one pass's use of the lane: which buffer is next, which reductions are in flight
struct ReducePass {
    ReduceLane* lane = nullptr;            // NULL: reductions on the caller's stream, one buffer
    hipStream_t stream = nullptr;
    float* partial[2] = {nullptr, nullptr};
    bool busy[2] = {false, false};
    int next = 0;
    bool failed = false;
    void join() {
        if (!lane) return;
        for (int k = 0; k < 2; ++k)
            if (busy[k]) { if (hipStreamWaitEvent(stream, lane->done[k], 0) != hipSuccess) failed = true; busy[k] = false; }
    }
    ~ReducePass() { join(); }
};
// OFF in the product (round 5, second measurement): alone in a process the lane buys 0.3 ms of an 87 ms refine iteration -- but it is one more
// hardware queue.  The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES = 4 hardware queues; a training process already holds the
// default stream, the backward's side stream, the high-priority prologue stream and whatever captured a graph, and with the lane as the fifth two
// streams share a queue: their event waits become full barriers of that queue, and EVERY phase of the iteration slows down (same box, the refine
// workload after the other workloads of bench.py's default line: 100.8 ms with the lane, 87.8 ms without, 87.6 ms with the lane and
// GPU_MAX_HW_QUEUES=8; stand-alone process: 87.1 ms with it; profiles/r05_hw_queue_oversubscription.txt).  A library must not spend a hardware queue
// for 0.4 %: -DWG_REDUCE_LANE=1 builds the lane for experiments.
#ifndef WG_REDUCE_LANE
#define WG_REDUCE_LANE 0     // 0: the reductions on the caller's stream; 1: on the lane (A/B builds)
#endif
// opens the lane for a pass on `stream` with two buffers of `stride` floats at `partial`; leaves it closed (in-stream reductions) inside a capture
static void reduce_pass_begin(ReducePass& rp, float* partial, int64_t stride, hipStream_t stream) {
    static thread_local ReduceLane lane;
    rp.stream = stream;
    rp.partial[0] = partial; rp.partial[1] = partial + stride;
    rp.lane = nullptr;
#if WG_REDUCE_LANE
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    int dev = 0;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return; }
    if (hipGetDevice(&dev) != hipSuccess) return;
    if (lane.open_on(dev)) rp.lane = &lane;
#endif
}

// dw += scale * sum over the planes of dW(dy[b], x[b]) -- ONE pass of the limb kernel and one reduction for all planes (limb arithmetics only).
// H[b], W[b] = size of x[b]; dy_absmax (f16 limbs): the word of launch_absmax[_ragged] over ALL planes' dy, or NULL with one plane
// rp: the pass's reduce lane (its buffers replace `partial`), or NULL
static int launch_wgrad_planes(int nplanes, const float* const* dy, const float* const* x, int Cin, const int* H, const int* W, int Cout, float scale,
                               float* dw, float* partial, hipStream_t stream, int arith, const unsigned* dy_absmax, ReducePass* rp = nullptr) {
    if (nplanes < 1 || nplanes > CONV_RAGGED_MAX) return NVSR_ERR_SHAPE;
    const long tiles = (long)((Cout + WG_CO - 1) / WG_CO) * ((Cin + WG_CI - 1) / WG_CI);
    WgradParams p{dy[0], x[0], partial, Cin, Cout, H[0] - 2, W[0] - 2, 0, nullptr, 0, 0};
    p.nplanes = nplanes;
    long TS = 0;
    for (int b = 0; b < CONV_RAGGED_MAX; ++b) {
        const int k = b < nplanes ? b : 0;
        const int Ho = H[k] - 2, Wo = W[k] - 2;
        if (Ho < 1 || Wo < 1 || 64L * H[k] * W[k] >= (1L << 29)) return NVSR_ERR_SHAPE;     // (the kernel's int offset arithmetic)
        p.pHo[b] = Ho; p.pWo[b] = Wo; p.pdy[b] = dy[k]; p.px[b] = x[k];
        p.start[b] = (int)TS;
        if (b < nplanes) TS += (long)((Wo + WG_PX - 1) / WG_PX) * Ho;
    }
    p.start[CONV_RAGGED_MAX] = (int)TS;
    const long total = tiles * TS;
    if (TS >= (1L << 30) || total >= (1L << 31) - TS) return NVSR_ERR_SHAPE;                   // (the kernel's int step arithmetic)
    int n_wg = wgrad_pieces_of(tiles, total);
#ifdef WG_TUNE     // variant builds only (tools/conv_wgrad_time.py; the tool sizes the workspace itself)
    if (getenv("NVSR_WGRAD_PIECES")) n_wg = atoi(getenv("NVSR_WGRAD_PIECES"));
#endif
    const unsigned* am = nullptr;
    if (arith == NVSR_ARITH_F16X2) {       // 2 f16 limbs: X with the static activation scale, dy with its tensor's own (one reduction)
        am = dy_absmax;
        if (!am && nplanes == 1) am = launch_absmax(dy[0], (long)Cout * (H[0] - 2) * (W[0] - 2), stream);
        if (!am) return nplanes == 1 ? NVSR_ERR_LAUNCH : NVSR_ERR_NULL;
    }
    p.TS = (int)TS; p.n_wg = n_wg; p.total = total; p.dy_absmax = am;
    hipStream_t rstream = stream;
    int k = 0;
    if (rp && rp->lane) {
        k = rp->next & 1;
        rp->next++;
        partial = p.partial = rp->partial[k];
        rstream = rp->lane->side;
        if (rp->busy[k] && hipStreamWaitEvent(stream, rp->lane->done[k], 0) != hipSuccess) return NVSR_ERR_LAUNCH;   // the reduction that last read this buffer
        rp->busy[k] = false;
    }
    if (am) hipLaunchKernelGGL(conv3x3_wgrad_limb_kernel<2>, dim3(n_wg), dim3(WG_TPB), 0, stream, p);
    else hipLaunchKernelGGL(conv3x3_wgrad_limb_kernel<3>, dim3(n_wg), dim3(WG_TPB), 0, stream, p);
    if (rstream != stream) {
        if (hipEventRecord(rp->lane->ready[k], stream) != hipSuccess || hipStreamWaitEvent(rstream, rp->lane->ready[k], 0) != hipSuccess) return NVSR_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(wgrad_reduce_pieces_kernel, dim3((Cin + 63) / 64, (Cout + 3) / 4, 9), dim3(256), 0, rstream, partial, n_wg, total, TS, Cout, Cin,
                       scale, dw, am);
    if (rstream != stream) {
        if (hipEventRecord(rp->lane->done[k], rstream) != hipSuccess) return NVSR_ERR_LAUNCH;
        rp->busy[k] = true;
    }
    return NVSR_CHECK_LAUNCH();
}

// dw += scale * dW(dy, x);  H, W = size of x
static int launch_wgrad(const float* dy, const float* x, int Cin, int H, int W, int Cout, float scale, float* dw, float* partial,
                        hipStream_t stream, int arith, const unsigned* dy_absmax = nullptr, ReducePass* rp = nullptr) {
    const int Ho = H - 2, Wo = W - 2;
    if (Ho < 1 || Wo < 1) return NVSR_ERR_SHAPE;
    arith = conv_resolve_arith(arith);
    if (arith != NVSR_ARITH_F32 && arith != NVSR_ARITH_BF16X3 && arith != NVSR_ARITH_F16X2) return NVSR_ERR_SHAPE;
    if (arith != NVSR_ARITH_F32) return launch_wgrad_planes(1, &dy, &x, Cin, &H, &W, Cout, scale, dw, partial, stream, arith, dy_absmax, rp);
    const long n = 9L * Cout * Cin;
    const int ns = wgrad_slabs(Cin, Cout, Ho);
    WgradParams p{dy, x, partial, Cin, Cout, Ho, Wo, (Ho + ns - 1) / ns, nullptr, 0, 0};
    hipLaunchKernelGGL(conv3x3_wgrad_kernel, dim3((Cout + WG_CO - 1) / WG_CO, (Cin + WG_CI - 1) / WG_CI, ns), dim3(WG_TPB), 0, stream, p);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, partial, ns, Cout, Cin, scale, dw);
    return NVSR_CHECK_LAUNCH();
}

static bool launch_residual_gather(const float* d_out, int Cc, int R0, int R1, int sf, const int* lo, const int* hi, float* d_lr, int align,
                                   hipStream_t stream) {
    const int ty0 = lo[0] > 0 ? lo[0] - 1 : 0, ty1 = hi[0] + 1 < R0 ? hi[0] + 1 : R0, tx0 = lo[1] > 0 ? lo[1] - 1 : 0, tx1 = hi[1] + 1 < R1 ? hi[1] + 1 : R1;
    const int th = ty1 - ty0, tw = tx1 - tx0;
    const long n = (long)Cc * th * tw;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (sf == 4) hipLaunchKernelGGL(sr_residual_backward_gather_kernel<4>, grid, block, 0, stream, d_out, Cc, R0, R1, lo[0], lo[1], hi[0], hi[1], ty0, tx0, th, tw, d_lr, align);
    else if (sf == 2) hipLaunchKernelGGL(sr_residual_backward_gather_kernel<2>, grid, block, 0, stream, d_out, Cc, R0, R1, lo[0], lo[1], hi[0], hi[1], ty0, tx0, th, tw, d_lr, align);
    else if (sf == 8) hipLaunchKernelGGL(sr_residual_backward_gather_kernel<8>, grid, block, 0, stream, d_out, Cc, R0, R1, lo[0], lo[1], hi[0], hi[1], ty0, tx0, th, tw, d_lr, align);
    else return false;                           // (other scale factors keep the scatter of sr_finish_backward_kernel)
    return true;
}

// sr_finish_backward for one plane: d_diff, and the residual's share of d_lr (gather for the bilinear residual, scatter for the bicubic one)
static int finish_backward(const float* d_out, int Cc, int R0, int R1, int sf, const int* lo, const int* hi, int Ho, int Wo, int over, float* d_diff,
                           float* d_lr, int align, int bicubic, hipStream_t stream) {
    const int64_t n_diff = (int64_t)Cc * Ho * Wo;
    bool gathered = false;
    if (d_lr && !bicubic) gathered = launch_residual_gather(d_out, Cc, R0, R1, sf, lo, hi, d_lr, align, stream);
    hipLaunchKernelGGL(sr_finish_backward_kernel, dim3((unsigned)((n_diff + 255) / 256)), dim3(256), 0, stream, d_out, Cc, R0, R1, sf, lo[0], lo[1], hi[0],
                       hi[1], Ho, Wo, over, d_diff, gathered ? nullptr : d_lr, align, bicubic);
    return NVSR_CHECK_LAUNCH();
}

}  // namespace nvsr

using namespace nvsr;

extern "C" {

int64_t nvsr_conv3x3_wgrad_workspace_floats(int Cin, int H, int W, int Cout) {
    if (Cin < 1 || Cout < 1 || H < 3 || W < 3) return -1;
    return wgrad_partial_floats(Cin, Cout, H - 2, W - 2);
}

/* weight gradient of nvsr_conv3x3 (epilogue 0): dw [Cout][Cin][3][3] += scale * sum_{y,x} dy[co][y][x] x[ci][y+ky][x+kx] */
int nvsr_conv3x3_wgrad_arith(const float* dy, const float* x, int Cin, int H, int W, int Cout, float scale, float* dw, float* workspace,
                             int arithmetic, nvsr_stream_t stream) {
    if (!dy || !x || !dw || !workspace) return NVSR_ERR_NULL;
    if (Cin < 1 || Cout < 1 || H < 3 || W < 3) return NVSR_ERR_SHAPE;
    return launch_wgrad(dy, x, Cin, H, W, Cout, scale, dw, workspace, (hipStream_t)stream, arithmetic);
}
int nvsr_conv3x3_wgrad(const float* dy, const float* x, int Cin, int H, int W, int Cout, float scale, float* dw, float* workspace,
                       nvsr_stream_t stream) {
    return nvsr_conv3x3_wgrad_arith(dy, x, Cin, H, W, Cout, scale, dw, workspace, NVSR_ARITH_INHERIT, stream);
}

/* every layer's data-gradient fragments, layer after layer in state-dict order */
int64_t nvsr_edsr_packed_dgrad_floats(int Cin, int Cout, int hid, int nblocks, int n_up) {
    if (!edsr_geometry_ok(nblocks, n_up)) return -1;
    ConvLayer L[EDSR_MAX_LAYERS]; int n;
    edsr_layers(Cin, Cout, hid, nblocks, n_up, L, &n);
    int64_t s = 0;
    for (int i = 0; i < n; ++i) s += conv_packed_floats(L[i].Cout, L[i].Cin);
    return s;
}
int nvsr_pack_edsr_dgrad_arith(const float* natural, int Cin, int Cout, int hid, int nblocks, int n_up, float* packed_dgrad, int arithmetic,
                               nvsr_stream_t stream) {
    if (!natural || !packed_dgrad) return NVSR_ERR_NULL;
    if (!edsr_geometry_ok(nblocks, n_up)) return NVSR_ERR_SHAPE;
    if (!aligned16(packed_dgrad)) return NVSR_ERR_ALIGN;
    if (arithmetic != NVSR_PACK_ALL_ARITHMETICS) {
        arithmetic = conv_resolve_arith(arithmetic);
        if (arithmetic != NVSR_ARITH_F32 && arithmetic != NVSR_ARITH_F16X2 && arithmetic != NVSR_ARITH_BF16X3) return NVSR_ERR_SHAPE;
    }
    ConvLayer L[EDSR_MAX_LAYERS]; int n;
    edsr_layers(Cin, Cout, hid, nblocks, n_up, L, &n);
    return pack_layers(natural, L, n, packed_dgrad, 1, (hipStream_t)stream, arithmetic);
}
int nvsr_pack_edsr_dgrad(const float* natural, int Cin, int Cout, int hid, int nblocks, int n_up, float* packed_dgrad, nvsr_stream_t stream) {
    return nvsr_pack_edsr_dgrad_arith(natural, Cin, Cout, hid, nblocks, n_up, packed_dgrad, NVSR_PACK_ALL_ARITHMETICS, stream);
}

/* 3 gradient tensors + 1 un-shuffled gradient + the weight-gradient partial sums */
int64_t nvsr_edsr_backward_workspace_floats(int Cin, int Cout, int hid, int nblocks, int n_up, int H, int W) {
    EdsrPlan P;
    if (edsr_plan(Cin, Cout, hid, nblocks, n_up, H, W, &P)) return -1;
    int64_t part = 0;
    for (int l = 0; l < P.n; ++l) {
        const int64_t f = wgrad_partial_floats(P.L[l].Cin, P.L[l].Cout, P.ih[l] - 2, P.iw[l] - 2);
        if (f > part) part = f;
    }
    const int64_t t = (P.max_tensor + 3) / 4 * 4;
    // + one word per gradient tensor for its largest magnitude (f16 limbs: absmax_kernel) -- caller-owned, so that backward passes queued on
    // different streams never share a result word
    return 4 * t + 2 * ((part + 3) / 4 * 4) + (P.n + 7) / 4 * 4;       // (two buffers of partial sums: ReduceLane)
}

/* Backward of nvsr_edsr_forward_train.  x, acts: the forward's input and activation record; d_out [Cout][Ho][Wo];
 * grad_natural (state-dict order, nvsr_edsr_natural_floats) += weight gradients; dx [Cin][H][W] or NULL. */
int nvsr_edsr_backward(const float* x, int Cin, int H, int W, const float* acts, const float* packed_dgrad, int Cout, int hid, int nblocks,
                       int n_up, const float* d_out, float* grad_natural, float* dx, float* workspace, nvsr_stream_t stream_) {
    return nvsr_edsr_backward_arith(x, Cin, H, W, acts, packed_dgrad, Cout, hid, nblocks, n_up, d_out, grad_natural, dx, workspace,
                                    NVSR_ARITH_INHERIT, stream_);
}
int nvsr_edsr_backward_arith(const float* x, int Cin, int H, int W, const float* acts, const float* packed_dgrad, int Cout, int hid, int nblocks,
                             int n_up, const float* d_out, float* grad_natural, float* dx, float* workspace, int arithmetic,
                             nvsr_stream_t stream_) {
    if (!x || !acts || !packed_dgrad || !d_out || !grad_natural || !workspace) return NVSR_ERR_NULL;
    const int arith = conv_resolve_arith(arithmetic);
    const ConvExec cx{arith, 0};
    if (!aligned16(packed_dgrad) || !aligned16(workspace)) return NVSR_ERR_ALIGN;
    EdsrPlan P;
    if (int e = edsr_plan(Cin, Cout, hid, nblocks, n_up, H, W, &P)) return e;
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t t = (P.max_tensor + 3) / 4 * 4;
    float* buf[3] = {workspace, workspace + t, workspace + 2 * t};
    float* unsh = workspace + 3 * t;
    float* partial = workspace + 4 * t;
    int64_t part_floats = 0;
    for (int l = 0; l < P.n; ++l) {
        const int64_t f = wgrad_partial_floats(P.L[l].Cin, P.L[l].Cout, P.ih[l] - 2, P.iw[l] - 2);
        if (f > part_floats) part_floats = f;
    }
    part_floats = (part_floats + 3) / 4 * 4;
    unsigned* amax_words = reinterpret_cast<unsigned*>(partial + 2 * part_floats);     // P.n + 4 words, one per gradient tensor
    int amax_next = 0;
    ReducePass rp;                                                                      // (joins the lane on every exit)
    if (arith != NVSR_ARITH_F32) reduce_pass_begin(rp, partial, part_floats, stream);
    // per-layer offsets into the natural gradient blob and the packed data-gradient blob
    int64_t goff[EDSR_MAX_LAYERS], poff[EDSR_MAX_LAYERS], go = 0, po = 0;
    for (int l = 0; l < P.n; ++l) {
        goff[l] = go; poff[l] = po;
        go += 9LL * P.L[l].Cin * P.L[l].Cout;
        po += conv_packed_floats(P.L[l].Cout, P.L[l].Cin);
    }
    auto input_of = [&](int l) { return l ? acts + P.act_off[l] : x; };
    const float* g = d_out;       // gradient with respect to the output of layer l (after its epilogue)
    int gi = -1;                   // index of g in buf (-1: the caller's d_out)
    auto next_buf = [&](int a, int b) { for (int k = 0; k < 3; ++k) if (k != a && k != b) return k; return 0; };
    int e;
    // f16 limbs: the power-of-two scale of a gradient tensor comes from the bits of its largest magnitude, one word per tensor.  The word of the
    // caller's d_out is a reduction pass; every other gradient tensor is the output of a data-gradient launch, whose epilogue leaves the word
    // behind (ConvExec::out_absmax: an atomicMax per wave) -- no pass over the tensor; the un-shuffled gradient is a permutation of one that has it
    const bool f16 = conv_resolve_arith(arith) == NVSR_ARITH_F16X2;
    if (f16 && hipMemsetAsync(amax_words, 0, sizeof(unsigned) * (P.n + 4), stream) != hipSuccess) return NVSR_ERR_LAUNCH;
    auto new_word = [&]() -> unsigned* { return f16 ? amax_words + amax_next++ : nullptr; };
    auto exec = [&](const unsigned* in_am, unsigned* out_am) { ConvExec c = cx; c.in_absmax = in_am; c.out_absmax = out_am; return c; };
    const unsigned* g_am = nullptr;
    if (f16) {
        const int lz = P.n - 1;
        g_am = launch_absmax(d_out, (long)P.L[lz].Cout * (P.ih[lz] - 2) * (P.iw[lz] - 2), stream, new_word());
        if (!g_am) return NVSR_ERR_LAUNCH;
    }
    for (int l = P.n - 1; l >= 0; --l) {
        const int ci = P.L[l].Cin, co = P.L[l].Cout, ih = P.ih[l], iw = P.iw[l];
        const bool need_dx = l > 0 || dx;
        if (P.epi[l] == EPI_PIXEL_SHUFFLE) {           // g is [co/4][2(ih-2)][2(iw-2)]: undo the shuffle first
            const long n = (long)co * (ih - 2) * (iw - 2);
            hipLaunchKernelGGL(pixel_unshuffle_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, g, co / 4, ih - 2, iw - 2, unsh);
            if ((e = NVSR_CHECK_LAUNCH())) return e;
            if ((e = launch_wgrad(unsh, input_of(l), ci, ih, iw, co, 1.0f, grad_natural + goff[l], partial, stream, arith, g_am, &rp))) return e;
            const int o = next_buf(gi, -1);
            unsigned* w = new_word();
            if ((e = launch_conv(unsh, co, ih - 2, iw - 2, packed_dgrad + poff[l], ci, EPI_NONE, nullptr, buf[o], stream, 2, 1, exec(g_am, w)))) return e;
            g = buf[o]; gi = o; g_am = w;
        } else if (P.epi[l] == EPI_RESIDUAL) {         // block: y = 0.1 conv2(relu(conv1(xb))) + crop(xb); layers l-1 (conv1), l (conv2)
            const float* t1 = input_of(l);             // relu(conv1(xb)), [hid][ih][iw]
            const float* xb = input_of(l - 1);         // [hid][ih+2][iw+2]
            if ((e = launch_wgrad(g, t1, ci, ih, iw, co, 0.1f, grad_natural + goff[l], partial, stream, arith, g_am, &rp))) return e;
            const int o1 = next_buf(gi, -1);
            unsigned* w1 = new_word();
            if ((e = launch_conv(g, co, ih - 2, iw - 2, packed_dgrad + poff[l], ci, EPI_MASK_SCALE, t1, buf[o1], stream, 2, 1, exec(g_am, w1)))) return e;
            const int l1 = l - 1;
            if ((e = launch_wgrad(buf[o1], xb, P.L[l1].Cin, P.ih[l1], P.iw[l1], P.L[l1].Cout, 1.0f, grad_natural + goff[l1], partial, stream, arith, w1, &rp))) return e;
            const int o2 = next_buf(gi, o1);
            unsigned* w2 = new_word();
            if ((e = launch_conv(buf[o1], P.L[l1].Cout, ih, iw, packed_dgrad + poff[l1], P.L[l1].Cin, EPI_ADD_CENTER, g, buf[o2], stream, 2, 1, exec(w1, w2)))) return e;
            g = buf[o2]; gi = o2; g_am = w2;
            --l;                                        // conv1 is done too
        } else {                                        // plain conv (conv_input, conv_mid, conv_output)
            if ((e = launch_wgrad(g, input_of(l), ci, ih, iw, co, 1.0f, grad_natural + goff[l], partial, stream, arith, g_am, &rp))) return e;
            if (need_dx) {
                float* o = (l == 0) ? dx : buf[next_buf(gi, -1)];
                unsigned* w = l ? new_word() : nullptr;
                if ((e = launch_conv(g, co, ih - 2, iw - 2, packed_dgrad + poff[l], ci, EPI_NONE, nullptr, o, stream, 2, 1, exec(g_am, w)))) return e;
                if (l) { gi = next_buf(gi, -1); g = buf[gi]; g_am = w; }
            }
        }
    }
    rp.join();
    return rp.failed ? NVSR_ERR_LAUNCH : NVSR_OK;
}

/* 1 d_diff + the prepared-input gradient + the EDSR backward workspace */
int64_t nvsr_planes_sr_backward_workspace_floats(int Cc, int R0, int R1, int hid, int nblocks, int n_up, int pad, const float* roi) {
    int lo[2], hi[2];
    sr_roi(R0, R1, roi, lo, hi);
    const int Hp = hi[0] - lo[0] + 2 * pad, Wp = hi[1] - lo[1] + 2 * pad;
    EdsrPlan P;
    if (edsr_plan(Cc, Cc, hid, nblocks, n_up, Hp, Wp, &P)) return -1;
    const int64_t w = nvsr_edsr_backward_workspace_floats(Cc, Cc, hid, nblocks, n_up, Hp, Wp);
    return ((int64_t)Cc * P.Ho * P.Wo + 3) / 4 * 4 + ((int64_t)Cc * Hp * Wp + 3) / 4 * 4 + w;
}

/* Backward of nvsr_planes_sr_train.  d_out [C][sf R0][sf R1] (entries outside the ROI are ignored); grad_natural += EDSR weight
 * gradients; d_lr [C][R0][R1] += gradient of the LR plane (network input + bilinear residual) or NULL when the LR plane is detached
 * (models.py:272). */
int nvsr_planes_sr_backward(int Cc, int R0, int R1, const float* keep, const float* packed_dgrad, int hid, int nblocks, int n_up, int pad,
                            int over, const float* roi, const float* stdv, const float* d_out, float* grad_natural, float* d_lr,
                            float* workspace, nvsr_stream_t stream_) {
    return nvsr_planes_sr_backward_arith(Cc, R0, R1, keep, packed_dgrad, hid, nblocks, n_up, pad, over, roi, stdv, d_out, grad_natural, d_lr,
                                         workspace, NVSR_ARITH_INHERIT, stream_);
}
int nvsr_planes_sr_backward_arith(int Cc, int R0, int R1, const float* keep, const float* packed_dgrad, int hid, int nblocks, int n_up, int pad,
                                  int over, const float* roi, const float* stdv, const float* d_out, float* grad_natural, float* d_lr,
                                  float* workspace, int arithmetic, nvsr_stream_t stream_) {
    if (!keep || !packed_dgrad || !d_out || !grad_natural || !workspace) return NVSR_ERR_NULL;
    hipStream_t stream = (hipStream_t)stream_;
    const int sf = 1 << n_up;
    int lo[2], hi[2];
    sr_roi(R0, R1, roi, lo, hi);
    const int ch = hi[0] - lo[0], cw = hi[1] - lo[1];
    const int Hp = ch + 2 * pad, Wp = cw + 2 * pad;
    EdsrPlan P;
    if (int e = edsr_plan(Cc, Cc, hid, nblocks, n_up, Hp, Wp, &P)) return e;
    if (P.Ho != ch * sf + 2 * over || P.Wo != cw * sf + 2 * over) return NVSR_ERR_SHAPE;
    const int64_t n_diff = (int64_t)Cc * P.Ho * P.Wo, n_in = (int64_t)Cc * Hp * Wp;
    float* d_diff = workspace;
    float* dxin = d_diff + (n_diff + 3) / 4 * 4;
    float* ews = dxin + (n_in + 3) / 4 * 4;
    if (int e = finish_backward(d_out, Cc, R0, R1, sf, lo, hi, P.Ho, P.Wo, over, d_diff, d_lr, sr_align_corners(), sr_bicubic(), stream)) return e;
    const float* xin = keep;
    const float* acts = keep + (n_in + 3) / 4 * 4;
    if (int e = nvsr_edsr_backward_arith(xin, Cc, Hp, Wp, acts, packed_dgrad, Cc, hid, nblocks, n_up, d_diff, grad_natural, d_lr ? dxin : nullptr, ews,
                                         arithmetic, stream_))
        return e;
    if (d_lr) {
        hipLaunchKernelGGL(sr_prepare_backward_kernel, dim3((unsigned)((n_in + 255) / 256)), dim3(256), 0, stream, dxin, Cc, R0, R1, lo[0], lo[1],
                           Hp, Wp, pad, stdv, d_lr);
        if (int e = NVSR_CHECK_LAUNCH()) return e;
    }
    return NVSR_OK;
}


/* ---- backward of nvsr_planes_sr_train_batch_arith: B regions of interest at once ------------------------------------------------------------
 * Per layer ONE weight-gradient pass over all planes (one set of partial sums, one reduction: launch_wgrad_planes), ONE data-gradient launch
 * (ragged batch) and -- f16 limbs -- ONE magnitude reduction per gradient tensor across the planes.  keep: the forward's buffer;
 * d_out: B HOST pointers to [C][sf R0][sf R1] gradients (entries outside a plane's ROI are ignored); grad_natural += the EDSR weight gradients
 * of all planes; d_lr: NULL or B HOST pointers, each NULL (plane detached, models.py:272) or [C][R0][R1] += that LR plane's gradient. */
}  // extern "C"   (helpers below are C++)

namespace nvsr {

// backward of the EDSR forward on B inputs of different sizes (nvsr_edsr_backward_arith with one launch per layer and operation for all planes)
// marks: events the caller wants recorded on `stream` as soon as the weight gradients of layer mark_layers[i] AND of every layer behind it (the
// backward walks the layers from the last to the first) are final in grad_natural -- a data-parallel caller starts the all-reduce of that part of
// the blob while the rest of the backward is still running (nvsr_planes_sr_backward_batch_marks)
struct LayerMarks { int n = 0; const int32_t* layers = nullptr; void* const* events = nullptr; };
static int record_marks(const LayerMarks& mk, int layer, hipStream_t stream) {      // layer < 0: every mark (the pass is over)
    for (int i = 0; i < mk.n; ++i)
        if ((layer < 0 || mk.layers[i] == layer) && mk.events[i] && hipEventRecord((hipEvent_t)mk.events[i], stream) != hipSuccess) return NVSR_ERR_LAUNCH;
    return NVSR_OK;
}

static int edsr_backward_planes(int B, const EdsrPlan* P, const float* const* x, const float* const* acts, const float* packed_dgrad,
                                const float* const* d_out, float* grad_natural, float* const* dx, float* workspace, int arith, hipStream_t stream,
                                const LayerMarks& marks = LayerMarks{}) {
    const int n = P[0].n;
    // workspace: 3 gradient buffers + 1 un-shuffled gradient, each holding one tensor per plane, + the partial sums + one word per gradient tensor
    int64_t toff[CONV_RAGGED_MAX + 1] = {0};
    for (int b = 0; b < B; ++b) toff[b + 1] = toff[b] + (P[b].max_tensor + 3) / 4 * 4;
    const int64_t T = toff[B];
    auto buf = [&](int k, int b) { return workspace + k * T + toff[b]; };
    float* partial = workspace + 4 * T;
    int64_t part_floats = 0;
    for (int l = 0; l < n; ++l) {
        const int64_t f = wgrad_partial_floats_ragged(P[0].L[l].Cin, P[0].L[l].Cout);
        if (f > part_floats) part_floats = f;
    }
    part_floats = (part_floats + 3) / 4 * 4;
    unsigned* amax_words = reinterpret_cast<unsigned*>(partial + 2 * part_floats);      // (two buffers of partial sums: ReduceLane)
    ReducePass rp;                                                                       // (its destructor joins the lane on every exit)
    reduce_pass_begin(rp, partial, part_floats, stream);
    int amax_next = 0;
    int64_t goff[EDSR_MAX_LAYERS], poff[EDSR_MAX_LAYERS], go = 0, po = 0;
    for (int l = 0; l < n; ++l) {
        goff[l] = go; poff[l] = po;
        go += 9LL * P[0].L[l].Cin * P[0].L[l].Cout;
        po += conv_packed_floats(P[0].L[l].Cout, P[0].L[l].Cin);
    }
    const bool f16 = arith == NVSR_ARITH_F16X2;
    const float* g[CONV_RAGGED_MAX];       // gradient with respect to the output of layer l (after its epilogue), per plane
    int gi = -1;                           // index of g in the buffers (-1: the caller's d_out)
    for (int b = 0; b < B; ++b) g[b] = d_out[b];
    auto next_buf = [&](int a, int c) { for (int k = 0; k < 3; ++k) if (k != a && k != c) return k; return 0; };
    auto input_of = [&](int l, int b) { return l ? acts[b] + P[b].act_off[l] : x[b]; };
    // one magnitude word per gradient tensor, over all planes: a reduction pass for the caller's d_out, the data-gradient launches' epilogues for
    // every other tensor (see nvsr_edsr_backward_arith)
    if (f16 && hipMemsetAsync(amax_words, 0, sizeof(unsigned) * (n + 4), stream) != hipSuccess) return NVSR_ERR_LAUNCH;
    auto new_word = [&]() -> unsigned* { return f16 ? amax_words + amax_next++ : nullptr; };
    const unsigned* g_am = nullptr;
    if (f16) {
        long cnt[CONV_RAGGED_MAX];
        for (int b = 0; b < B; ++b) cnt[b] = (long)P[0].L[n - 1].Cout * (P[b].ih[n - 1] - 2) * (P[b].iw[n - 1] - 2);
        g_am = launch_absmax_ragged(B, d_out, cnt, stream, new_word());
        if (!g_am) return NVSR_ERR_LAUNCH;
    }
    auto wgrad = [&](const float* const* dy, int l, float scale, const unsigned* am) {
        const float* xs[CONV_RAGGED_MAX]; int H[CONV_RAGGED_MAX], W[CONV_RAGGED_MAX];
        for (int b = 0; b < B; ++b) { xs[b] = input_of(l, b); H[b] = P[b].ih[l]; W[b] = P[b].iw[l]; }
        if (int e_ = launch_wgrad_planes(B, dy, xs, P[0].L[l].Cin, H, W, P[0].L[l].Cout, scale, grad_natural + goff[l], partial, stream, arith, am, &rp)) return e_;
        bool wanted = false;
        for (int i = 0; i < marks.n; ++i) wanted = wanted || marks.layers[i] == l;
        if (!wanted) return (int)NVSR_OK;
        rp.join();                                 // (a reduce lane's reductions of this and the earlier layers, before the mark; the product has no lane)
        return rp.failed ? (int)NVSR_ERR_LAUNCH : record_marks(marks, l, stream);
    };
    // data gradient of layer l: dy [Cout][ih-2][iw-2] -> [Cin][ih][iw] with the given backward epilogue
    auto dgrad = [&](const float* const* dy, int l, int epi, const float* const* skip, float* const* out, const unsigned* am, unsigned* out_am) {
        ConvRagged r;
        r.n = B;
        for (int b = 0; b < B; ++b) {
            r.H[b] = P[b].ih[l] - 2; r.W[b] = P[b].iw[l] - 2;
            r.in[b] = dy[b]; r.out[b] = out[b]; r.skip[b] = skip ? skip[b] : nullptr;
        }
        ConvExec cx{arith, 0};
        cx.in_absmax = am;
        cx.out_absmax = out_am;
        return launch_conv(nullptr, P[0].L[l].Cout, 0, 0, packed_dgrad + poff[l], P[0].L[l].Cin, epi, nullptr, nullptr, stream, 2, B, cx, &r);
    };
    int e;
    for (int l = n - 1; l >= 0; --l) {
        const int co = P[0].L[l].Cout;
        const bool need_dx = l > 0 || dx;
        if (P[0].epi[l] == EPI_PIXEL_SHUFFLE) {           // g is [co/4][2(ih-2)][2(iw-2)]: undo the shuffle first
            const float* un[CONV_RAGGED_MAX];
            for (int b = 0; b < B; ++b) {
                const long cnt = (long)co * (P[b].ih[l] - 2) * (P[b].iw[l] - 2);
                hipLaunchKernelGGL(pixel_unshuffle_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, stream, g[b], co / 4, P[b].ih[l] - 2,
                                   P[b].iw[l] - 2, buf(3, b));
                un[b] = buf(3, b);
            }
            if ((e = NVSR_CHECK_LAUNCH())) return e;
            if ((e = wgrad(un, l, 1.0f, g_am))) return e;
            const int o = next_buf(gi, -1);
            float* outs[CONV_RAGGED_MAX];
            for (int b = 0; b < B; ++b) outs[b] = buf(o, b);
            unsigned* w = new_word();
            if ((e = dgrad(un, l, EPI_NONE, nullptr, outs, g_am, w))) return e;
            for (int b = 0; b < B; ++b) g[b] = outs[b];
            gi = o; g_am = w;
        } else if (P[0].epi[l] == EPI_RESIDUAL) {         // block: y = 0.1 conv2(relu(conv1(xb))) + crop(xb); layers l-1 (conv1), l (conv2)
            const int l1 = l - 1;
            const float* t1[CONV_RAGGED_MAX];              // relu(conv1(xb))
            for (int b = 0; b < B; ++b) t1[b] = input_of(l, b);
            if ((e = wgrad(g, l, 0.1f, g_am))) return e;
            const int o1 = next_buf(gi, -1);
            float* d1[CONV_RAGGED_MAX]; const float* d1c[CONV_RAGGED_MAX];
            for (int b = 0; b < B; ++b) d1c[b] = d1[b] = buf(o1, b);
            unsigned* w1 = new_word();
            if ((e = dgrad(g, l, EPI_MASK_SCALE, t1, d1, g_am, w1))) return e;
            if ((e = wgrad(d1c, l1, 1.0f, w1))) return e;
            const int o2 = next_buf(gi, o1);
            float* d2[CONV_RAGGED_MAX];
            for (int b = 0; b < B; ++b) d2[b] = buf(o2, b);
            unsigned* w2 = new_word();
            if ((e = dgrad(d1c, l1, EPI_ADD_CENTER, g, d2, w1, w2))) return e;
            for (int b = 0; b < B; ++b) g[b] = d2[b];
            gi = o2; g_am = w2;
            --l;                                           // conv1 is done too
        } else {                                           // plain conv (conv_input, conv_mid, conv_output)
            if ((e = wgrad(g, l, 1.0f, g_am))) return e;
            if (need_dx) {
                const int o = next_buf(gi, -1);
                float* outs[CONV_RAGGED_MAX];
                for (int b = 0; b < B; ++b) outs[b] = (l == 0) ? dx[b] : buf(o, b);
                unsigned* w = l ? new_word() : nullptr;
                if ((e = dgrad(g, l, EPI_NONE, nullptr, outs, g_am, w))) return e;
                if (l) { for (int b = 0; b < B; ++b) g[b] = outs[b]; gi = o; g_am = w; }
            }
        }
    }
    rp.join();
    return rp.failed ? NVSR_ERR_LAUNCH : NVSR_OK;
}

}  // namespace nvsr

extern "C" {

int64_t nvsr_planes_sr_batch_backward_workspace_floats(int B, int Cc, int R0, int R1, int hid, int nblocks, int n_up, int pad, const float* rois) {
    if (B < 1 || B > CONV_RAGGED_MAX) return -1;
    static thread_local EdsrPlan P;
    int64_t s = 0, T = 0, one = 0;
    for (int b = 0; b < B; ++b) {
        int lo[2], hi[2];
        sr_roi(R0, R1, rois ? rois + 4 * b : nullptr, lo, hi);
        const int Hp = hi[0] - lo[0] + 2 * pad, Wp = hi[1] - lo[1] + 2 * pad;
        if (edsr_plan(Cc, Cc, hid, nblocks, n_up, Hp, Wp, &P)) return -1;
        s += ((int64_t)Cc * P.Ho * P.Wo + 3) / 4 * 4 + ((int64_t)Cc * Hp * Wp + 3) / 4 * 4;      // d_diff + the prepared input's gradient
        T += (P.max_tensor + 3) / 4 * 4;
        const int64_t w1 = nvsr_edsr_backward_workspace_floats(Cc, Cc, hid, nblocks, n_up, Hp, Wp);   // (the plane-by-plane path of exact f32)
        if (w1 > one) one = w1;
    }
    int64_t part = 0;
    for (int l = 0; l < P.n; ++l) {
        const int64_t f = wgrad_partial_floats_ragged(P.L[l].Cin, P.L[l].Cout);
        if (f > part) part = f;
    }
    const int64_t ragged = 4 * T + 2 * ((part + 3) / 4 * 4) + (P.n + 7) / 4 * 4;       // (edsr_backward_planes: 4 tensors, 2 x partial sums, the words)
    return s + (ragged > one ? ragged : one);
}

int nvsr_planes_sr_backward_batch_arith(int B, int Cc, int R0, int R1, const float* keep, const float* packed_dgrad, int hid, int nblocks, int n_up,
                                        int pad, int over, const float* rois, const float* stdv, const float* const* d_out, float* grad_natural,
                                        float* const* d_lr, float* workspace, int arithmetic, int align_corners, int plane_interp,
                                        nvsr_stream_t stream_) {
    return nvsr_planes_sr_backward_batch_marks(B, Cc, R0, R1, keep, packed_dgrad, hid, nblocks, n_up, pad, over, rois, stdv, d_out, grad_natural, d_lr,
                                               workspace, arithmetic, align_corners, plane_interp, 0, nullptr, nullptr, stream_);
}

int nvsr_planes_sr_backward_batch_marks(int B, int Cc, int R0, int R1, const float* keep, const float* packed_dgrad, int hid, int nblocks, int n_up,
                                        int pad, int over, const float* rois, const float* stdv, const float* const* d_out, float* grad_natural,
                                        float* const* d_lr, float* workspace, int arithmetic, int align_corners, int plane_interp,
                                        int n_marks, const int32_t* mark_layers, void* const* mark_events, nvsr_stream_t stream_) {
    if (!keep || !packed_dgrad || !d_out || !grad_natural || !workspace) return NVSR_ERR_NULL;
    if (n_marks < 0 || (n_marks > 0 && (!mark_layers || !mark_events))) return NVSR_ERR_NULL;
    LayerMarks marks;
    marks.n = n_marks; marks.layers = mark_layers; marks.events = mark_events;
    if (B < 1 || B > CONV_RAGGED_MAX) return NVSR_ERR_SHAPE;
    if (plane_interp != NVSR_PLANE_INTERP_BILINEAR && plane_interp != NVSR_PLANE_INTERP_BICUBIC) return NVSR_ERR_SHAPE;
    if (!aligned16(packed_dgrad) || !aligned16(workspace) || !aligned16(keep)) return NVSR_ERR_ALIGN;
    hipStream_t stream = (hipStream_t)stream_;
    const int sf = 1 << n_up, arith = conv_resolve_arith(arithmetic);
    static thread_local EdsrPlan P[CONV_RAGGED_MAX];
    int lo[CONV_RAGGED_MAX][2], hi[CONV_RAGGED_MAX][2], Hp[CONV_RAGGED_MAX], Wp[CONV_RAGGED_MAX];
    const float* xin[CONV_RAGGED_MAX]; const float* acts[CONV_RAGGED_MAX];
    float* d_diff[CONV_RAGGED_MAX]; float* dxin[CONV_RAGGED_MAX];
    const float* k = keep; float* w = workspace;
    bool any_lr = false;
    for (int b = 0; b < B; ++b) {
        if (!d_out[b]) return NVSR_ERR_NULL;
        sr_roi(R0, R1, rois ? rois + 4 * b : nullptr, lo[b], hi[b]);
        Hp[b] = hi[b][0] - lo[b][0] + 2 * pad; Wp[b] = hi[b][1] - lo[b][1] + 2 * pad;
        if (int e = edsr_plan(Cc, Cc, hid, nblocks, n_up, Hp[b], Wp[b], &P[b])) return e;
        if (P[b].Ho != (hi[b][0] - lo[b][0]) * sf + 2 * over || P[b].Wo != (hi[b][1] - lo[b][1]) * sf + 2 * over) return NVSR_ERR_SHAPE;
        const int64_t n_in = (int64_t)Cc * Hp[b] * Wp[b];
        xin[b] = k; acts[b] = k + (n_in + 3) / 4 * 4;
        k = acts[b] + P[b].acts_floats;
        d_diff[b] = w; w += ((int64_t)Cc * P[b].Ho * P[b].Wo + 3) / 4 * 4;
        dxin[b] = w; w += (n_in + 3) / 4 * 4;
        any_lr = any_lr || (d_lr && d_lr[b]);
    }
    for (int b = 0; b < B; ++b)
        if (int e = finish_backward(d_out[b], Cc, R0, R1, sf, lo[b], hi[b], P[b].Ho, P[b].Wo, over, d_diff[b], d_lr ? d_lr[b] : nullptr, align_corners ? 1 : 0,
                                    plane_interp == NVSR_PLANE_INTERP_BICUBIC ? 1 : 0, stream))
            return e;
    if (arith == NVSR_ARITH_F32 || B == 1) {
        for (int b = 0; b < B; ++b)
            if (int e = nvsr_edsr_backward_arith(xin[b], Cc, Hp[b], Wp[b], acts[b], packed_dgrad, Cc, hid, nblocks, n_up, d_diff[b], grad_natural,
                                                 (d_lr && d_lr[b]) ? dxin[b] : nullptr, w, arith, stream_))
                return e;
        // (plane by plane every layer's gradient is final only after the last plane: all marks here)
        if (int e = record_marks(marks, -1, stream)) return e;
    } else {
        // (a plane whose LR gradient nobody wants still gets its first layer's data gradient when another plane needs it: 0.3 % of the pass)
        for (int i = 0; i < n_marks; ++i)
            if (mark_layers[i] < 0 || mark_layers[i] >= P[0].n) return NVSR_ERR_SHAPE;
        if (int e = edsr_backward_planes(B, P, xin, acts, packed_dgrad, d_diff, grad_natural, any_lr ? dxin : nullptr, w, arith, stream, marks)) return e;
    }
    for (int b = 0; b < B; ++b) {
        if (!(d_lr && d_lr[b])) continue;
        const int64_t n_in = (int64_t)Cc * Hp[b] * Wp[b];
        hipLaunchKernelGGL(sr_prepare_backward_kernel, dim3((unsigned)((n_in + 255) / 256)), dim3(256), 0, stream, dxin[b], Cc, R0, R1, lo[b][0], lo[b][1],
                           Hp[b], Wp[b], pad, stdv, d_lr[b]);
        if (int e = NVSR_CHECK_LAUNCH()) return e;
    }
    return NVSR_OK;
}

}  // extern "C"
