// Host-side helpers shared by sr.hip (forward) and sr_bwd.hip (backward) -- not part of the public ABI.
#pragma once
#include <math.h>

#include "nvsr_common.h"

namespace nvsr {

// Source position of output index `o` of F.interpolate(mode='bilinear', scale_factor=sf) along an axis of `in` texels (ATen UpSample.h:
// area_pixel_compute_source_index + guard_index_and_lambda): align_corners=True maps the first / last output onto the first / last input,
// src = o (in - 1) / (out - 1); False maps pixel centres, src = (o + 0.5) / sf - 0.5, clamped at 0.  -> first tap i0, offset of the second
// tap (0 at the last texel), weight of the second tap.
struct BilinearTap { int i0, step; float w1; };
__device__ __forceinline__ BilinearTap bilinear_tap(int o, int in, int out, int sf, int align) {
    float src;
    if (align) src = (out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.0f) * (float)o;
    else {
        src = __fsub_rn(__fmul_rn((float)(1.0 / (double)sf), (float)o + 0.5f), 0.5f);
        if (src < 0.0f) src = 0.0f;
    }
    BilinearTap t;
    t.i0 = min((int)src, in - 1);
    t.step = t.i0 < in - 1 ? 1 : 0;
    t.w1 = fminf(fmaxf(src - (float)t.i0, 0.0f), 1.0f);
    return t;
}
// The same for mode='bicubic' (UpSample.h: area_pixel_compute_source_index with cubic = true -- no clamp at 0 --, the four taps floor - 1 ..
// floor + 2 clamped to the axis, cubic convolution weights with A = -0.75)
struct CubicTap { int i[4]; float w[4]; };
__device__ __forceinline__ CubicTap cubic_tap(int o, int in, int out, int sf, int align) {
    float src;
    if (align) src = (out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.0f) * (float)o;
    else src = __fsub_rn(__fmul_rn((float)(1.0 / (double)sf), (float)o + 0.5f), 0.5f);
    const float fl = floorf(src);
    const float t = fminf(fmaxf(src - fl, 0.0f), 1.0f);
    const float A = -0.75f;
    const float x0 = t + 1.0f, x1 = t, x2 = 1.0f - t, x3 = x2 + 1.0f;
    CubicTap c;
    c.w[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
    c.w[1] = ((A + 2.0f) * x1 - (A + 3.0f)) * x1 * x1 + 1.0f;
    c.w[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
    c.w[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
    const int i0 = (int)fl;
#pragma unroll
    for (int k = 0; k < 4; ++k) c.i[k] = min(max(i0 - 1 + k, 0), in - 1);
    return c;
}
// process-wide: align_corners and mode of PlanesSR's residual up-sampling (sr.hip: nvsr_set_sr_align_corners, nvsr_set_sr_plane_interp)
int sr_align_corners();
int sr_bicubic();


// f16-limb data / weight gradients of the SR network: the power of two that puts the largest |dy| of a gradient tensor (its bits in
// `absmax_bits`, absmax_kernel) into [2^12, 2^13).  Exponent field clamped to 254 (a tensor whose largest magnitude is below 2^-115 would
// otherwise ask for a scale beyond the largest finite power of two); an all-zero or non-finite tensor is not scaled.
__host__ __device__ inline float f16_gradient_scale(unsigned absmax_bits) {
    const int e = (int)((absmax_bits >> 23) & 0xffu);
    if (e == 0 || e == 255) return 1.0f;
    const int s = 254 + 12 - e;
    const unsigned bits = (unsigned)(s > 254 ? 254 : s) << 23;
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(bits);
#else
    float f; __builtin_memcpy(&f, &bits, 4); return f;
#endif
}


constexpr int K_PER_CHUNK = 18;                 // MFMA k-steps per 4-channel chunk (36 k / 2)
constexpr int FRAG_FLOATS = K_PER_CHUNK * 64;   // 1152 floats per (chunk, co-block)

// EPI_MASK_SCALE / EPI_ADD_CENTER are the two fused epilogues of a residual block's backward (sr_bwd.hip)
enum ConvEpilogue { EPI_NONE = 0, EPI_RELU = 1, EPI_RESIDUAL = 2, EPI_PIXEL_SHUFFLE = 3, EPI_MASK_SCALE = 4, EPI_ADD_CENTER = 5 };

struct ConvLayer { int Cin, Cout; };

inline int conv_ncb(int Cout) { return (Cout + 63) / 64 * 2; }
inline int conv_nchunks(int Cin) { return (Cin + 3) / 4; }
inline int64_t conv_packed_f32_floats(int Cin, int Cout) { return (int64_t)conv_nchunks(Cin) * conv_ncb(Cout) * FRAG_FLOATS; }
// bf16-limb fragments (sr.hip, conv3x3_limb_kernel): layers with Cin % 16 == 0 and a multiple of 256 (or at most 64) output channels carry, behind the f32
// fragments, [chunk of 16 input channels][co-block][tap 0..8][limb 0..2][lane][4 words] = 6912 words per (chunk, co-block)
constexpr int CL_FRAG_WORDS = 9 * 3 * 256;
inline bool conv_limb_eligible(int Cin, int Cout) { return Cin % 16 == 0 && (conv_ncb(Cout) % 8 == 0 || conv_ncb(Cout) == 2); }
inline int64_t conv_packed_limb_words(int Cin, int Cout) {
    return conv_limb_eligible(Cin, Cout) ? (int64_t)(Cin / 16) * conv_ncb(Cout) * CL_FRAG_WORDS : 0;
}
// bf16-limb fragments for v_mfma_f32_16x16x32_bf16 (sr.hip, conv3x3_limb16_kernel; round 3): layers with Cin % 32 == 0 and Cout % 128 == 0 carry a
// third region, [chunk of 32 input channels][16-channel co-block][tap 0..8][limb 0..2][lane][4 words]; lane (co = 16 cb + (l & 15),
// g = l >> 4) holds input channels 32 chunk + 8 g + 0..7 as bf16 pairs
inline bool conv_limb16_eligible(int Cin, int Cout) { return Cin % 32 == 0 && Cout % 128 == 0; }
inline int64_t conv_packed_limb16_words(int Cin, int Cout) {
    return conv_limb16_eligible(Cin, Cout) ? (int64_t)(Cin / 32) * (Cout / 16) * CL_FRAG_WORDS : 0;
}
// the same fragments in the 2-f16-limb arithmetic (limb_core.h: W 2^8 as hi + lo, round to nearest), a fourth region: [..][limb 0..1][lane][4 words]
inline int64_t conv_packed_f16_words(int Cin, int Cout) {
    return conv_limb16_eligible(Cin, Cout) ? (int64_t)(Cin / 32) * (Cout / 16) * (9 * 2 * 256) : 0;
}
inline int64_t conv_packed_floats(int Cin, int Cout) {
    return conv_packed_f32_floats(Cin, Cout) + conv_packed_limb_words(Cin, Cout) + conv_packed_limb16_words(Cin, Cout) + conv_packed_f16_words(Cin, Cout);
}
// Fragment kinds of a packed layer (bit k = region k of the blob: 0 f32, 1 bf16 limbs 32x32x16, 2 bf16 limbs 16x16x32, 3 f16 limbs 16x16x32) that
// launch_conv (sr.hip) READS for a launch in `arith` with the library's own choice of rows per tile (rows_per_tile 0: what every EDSR / PlanesSR entry
// point passes) -- the same predicates in the same order as the launcher's branches.  A training iteration re-packs both blobs of the network (the
// weights change every iteration): packing only these regions writes a fifth of the bytes (nvsr_pack_edsr_arith).  arith == NVSR_PACK_ALL_ARITHMETICS: every region.
constexpr unsigned CONV_KINDS_ALL = 15u;
inline unsigned conv_kinds_for(int Cin, int Cout, int arith, bool use_16x16x32, bool wide_rows8) {
    if (arith == NVSR_PACK_ALL_ARITHMETICS) return CONV_KINDS_ALL;
    if (arith == NVSR_ARITH_F32) return 1u;
    const bool wl = conv_limb_eligible(Cin, Cout), w16 = conv_limb16_eligible(Cin, Cout);
    const int ncb = conv_ncb(Cout);
    if (wl && ncb == 2) return 2u;                                              // narrow layer: conv3x3_limb_kernel<2, 1>
    if (w16 && use_16x16x32) return arith == NVSR_ARITH_F16X2 ? 8u : 4u;        // conv3x3_limb16_kernel
    (void)wide_rows8;                                                           // (the 8-row wide kernel reads region 1 like the 2..4-row one)
    if (wl) return 2u;                                                          // conv3x3_limb_kernel
    return 1u;                                                                  // no limb kernel eligible: exact-f32 kernels
}

// conv_input, (conv1, conv2) x nblocks, conv_mid, n_up x up-conv, conv_output  (state-dict order, models.py:802-816)
inline void edsr_layers(int Cin, int Cout, int hid, int nblocks, int n_up, ConvLayer* L, int* n) {
    int k = 0;
    L[k++] = {Cin, hid};
    for (int b = 0; b < 2 * nblocks; ++b) L[k++] = {hid, hid};
    L[k++] = {hid, hid};
    for (int u = 0; u < n_up; ++u) L[k++] = {hid, 4 * hid};
    L[k++] = {hid, Cout};
    *n = k;
}
constexpr int EDSR_MAX_LAYERS = 600;
inline bool edsr_geometry_ok(int nblocks, int n_up) { return nblocks >= 0 && nblocks <= 290 && n_up >= 0 && n_up <= 8; }

// Geometry of one EDSR application to an [Cin][H][W] input: per layer the input size, the epilogue, and where the layer's input lives
// in the activation record kept for the backward pass (layer 0 reads the caller's x).
struct EdsrPlan {
    int n;
    ConvLayer L[EDSR_MAX_LAYERS];
    int ih[EDSR_MAX_LAYERS], iw[EDSR_MAX_LAYERS];
    int epi[EDSR_MAX_LAYERS];          // forward epilogue of layer l
    int64_t act_off[EDSR_MAX_LAYERS];  // floats; input of layer l >= 1 inside the record
    int64_t acts_floats;               // size of the record
    int64_t max_tensor;                // largest activation (= largest gradient tensor) of the net, floats
    int Ho, Wo;                        // output size
};

inline int edsr_plan(int Cin, int Cout, int hid, int nblocks, int n_up, int H, int W, EdsrPlan* P) {
    if (!edsr_geometry_ok(nblocks, n_up) || Cin < 1 || Cout < 1 || hid < 1) return NVSR_ERR_SHAPE;
    edsr_layers(Cin, Cout, hid, nblocks, n_up, P->L, &P->n);
    int64_t h = H, w = W, off = 0, mx = 0;
    for (int l = 0; l < P->n; ++l) {
        if (h < 3 || w < 3 || h > 1 << 20 || w > 1 << 20) return NVSR_ERR_SHAPE;
        P->ih[l] = (int)h; P->iw[l] = (int)w;
        const bool in_blocks = l >= 1 && l <= 2 * nblocks;
        const bool up = l > 2 * nblocks + 1 && l < P->n - 1;
        P->epi[l] = in_blocks ? ((l & 1) ? EPI_RELU : EPI_RESIDUAL) : (up ? EPI_PIXEL_SHUFFLE : EPI_NONE);
        int64_t oh = h - 2, ow = w - 2, oc = P->L[l].Cout;
        if (up) { oh *= 2; ow *= 2; oc /= 4; }
        const int64_t out_floats = oc * oh * ow;
        if (out_floats > mx) mx = out_floats;
        if (l + 1 < P->n) { P->act_off[l + 1] = off; off += (out_floats + 3) / 4 * 4; }
        h = oh; w = ow;
    }
    P->act_off[0] = -1;
    P->acts_floats = off;
    P->max_tensor = mx;
    P->Ho = (int)h; P->Wo = (int)w;
    return NVSR_OK;
}

// PlanesSR's ROI arithmetic (models.py:902-905): roi = NULL (full plane) or [[ymin,xmin],[ymax,xmax]] in [-1,1] -> LR pixel bounds
inline void sr_roi(int R0, int R1, const float* roi, int* lo, int* hi) {
    lo[0] = lo[1] = 0; hi[0] = R0; hi[1] = R1;
    if (!roi) return;
    const int shape[2] = {R0, R1};
    for (int a = 0; a < 2; ++a) {
        const float mn = (float)shape[a] * (1.0f + roi[a]) / 2.0f, mx = (float)shape[a] * (1.0f + roi[2 + a]) / 2.0f;
        int l = (int)floorf(mn), hh = (int)ceilf(mx);
        l = l - 1 > 0 ? l - 1 : 0;
        hh = hh + 1 < shape[a] ? hh + 1 : shape[a];
        lo[a] = l; hi[a] = hh;
    }
}

// H, W: size of the tensor in memory; pad: virtual zero border (the kernel sees (H+2pad) x (W+2pad)).  Defined in sr.hip.
// batch: consecutive [C][H][W] planes in `in` / `skip` / `out`, all convolved with the same weights in one launch.
// ConvExec: per-call execution options of the convolutions (include/nvsr.h, the *_arith entry points).  arith = NVSR_ARITH_* or
// NVSR_ARITH_INHERIT (the process default of nvsr_set_conv_arithmetic); rows = 0 (cost model) or 2 / 3 / 4 rows per workgroup tile of the wide
// kernels (the parity tests force every instantiation).
struct ConvExec { int arith = -1; int rows = 0; const unsigned* in_absmax = nullptr; unsigned* out_absmax = nullptr; };      // out_absmax: ZEROED word that receives the bits of max |out| (the epilogues' atomicMax)
//      // in_absmax: launch_absmax of the input, if the caller has it already (f16 data gradients)
int conv_resolve_arith(int arith);      // INHERIT -> process default; sr.hip
// bits of max |x| over a tensor, in a device word that stays valid for the launches queued behind it (sr.hip): the power-of-two scale of an
// f16-limb gradient operand; NULL on a launch error
const unsigned* launch_absmax(const float* x, long n, hipStream_t stream, unsigned* owned = nullptr);
// A RAGGED batch: up to CONV_RAGGED_MAX planes of DIFFERENT sizes through one launch with the same weights -- the regions of interest of a
// scene's position planes in an SR training iteration (models.py:270-284: every plane has its own crop).  One 256-channel layer of one crop is
// 2-3 workgroup rounds with a last round a fifth full; the three crops together are 6-7 rounds.  The launch is a linear list of the planes'
// tiles, plane after plane (ConvRagged::tile0).  H, W: logical input size per plane INCLUDING the virtual border.
constexpr int CONV_RAGGED_MAX = 4;
struct ConvRagged {
    int n = 0;                                   // 0: not ragged (ConvParams' own tensors and batch strides apply)
    int H[CONV_RAGGED_MAX] = {0, 0, 0, 0}, W[CONV_RAGGED_MAX] = {0, 0, 0, 0};
    const float* in[CONV_RAGGED_MAX] = {nullptr, nullptr, nullptr, nullptr};
    float* out[CONV_RAGGED_MAX] = {nullptr, nullptr, nullptr, nullptr};
    const float* skip[CONV_RAGGED_MAX] = {nullptr, nullptr, nullptr, nullptr};
    // (filled by launch_conv) first workgroup of every plane in the launch's linear tile list -- the list has no empty tiles: a grid sized for the
    // largest plane would leave the smaller planes' unused tiles in a few XCDs' contiguous shares of the list (measured: two XCDs 28 % idle)
    unsigned tile0[CONV_RAGGED_MAX + 1] = {0, 0, 0, 0, 0};
};
// rag != NULL: in / skip / out / H / W / batch are ignored (rag->H, rag->W = sizes of the tensors in memory, the virtual border is added here);
// limb arithmetics only, and an f16 data gradient needs cx.in_absmax (one word for all planes: launch_absmax_ragged)
int launch_conv(const float* in, int Cin, int H, int W, const float* wpk, int Cout, int epilogue, const float* skip, float* out,
                hipStream_t stream, int pad = 0, int batch = 1, ConvExec cx = ConvExec{}, const ConvRagged* rag = nullptr);
// fragment blobs of nl consecutive layers (natural: their [Cout][Cin][3][3] weights one after the other) in 4 launches per 36 layers; transposed: the
// data gradients' fragments (nvsr_pack_conv3x3_dgrad per layer).  sr.hip
int pack_layers(const float* natural, const ConvLayer* layers, int nl, float* packed, int transposed, hipStream_t stream, int arith = NVSR_PACK_ALL_ARITHMETICS);
// max |x| over up to CONV_RAGGED_MAX tensors in one launch, into the caller's word
const unsigned* launch_absmax_ragged(int n, const float* const* x, const long* count, hipStream_t stream, unsigned* owned);

}  // namespace nvsr
