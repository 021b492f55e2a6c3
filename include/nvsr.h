/*
 * nvsr.h -- C ABI of the MI355X (gfx950) implementation of the Neural-Volume-Super-Resolution rendering hot path.
 *
 * The reference (pure PyTorch) has no FFI; its boundary for this path is the Python call surface used by
 * train_nerf.py.  Every entry point below replaces the ATen dispatches of one reference
 * function; the citation gives the reference file:line.  The host-side mirror of the reference interface lives in
 * neural-volume-super-resolution_amd/ and binds these symbols with ctypes.
 *
 * Conventions
 *   - all pointers are DEVICE pointers (fp32 unless stated) except `const nvsr_scene*`, a small host struct passed by value
 *     to the kernels; the library never allocates, never synchronises, and launches on `stream` (a hipStream_t);
 *   - tensors are dense row-major with the shapes given; "ray-major" [N,S] means S contiguous per ray;
 *   - return value: 0 = NVSR_OK, otherwise an nvsr_status; nothing is written on a shape error;
 *   - random inputs of the reference (stratified jitter t_rand, importance u, density noise) are explicit tensors so that
 *     train-mode results are reproducible (reference draws them on the CPU generator: train_utils.py:108,
 *     nerf_helpers.py:683, volume_rendering_utils.py:32).
 */
#ifndef NVSR_H
#define NVSR_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* nvsr_stream_t; /* hipStream_t */

typedef enum {
    NVSR_OK = 0,
    NVSR_ERR_SHAPE = 1,   /* argument out of the supported range */
    NVSR_ERR_LAUNCH = 2,  /* hipGetLastError() != hipSuccess after a launch */
    NVSR_ERR_NULL = 3,    /* required pointer is NULL */
    NVSR_ERR_ALIGN = 4    /* pointer not 16-byte aligned */
} nvsr_status;

/* Decoder geometry compiled into the MFMA kernels: TwoDimPlanesModel(dec_density_layers=4, dec_rgb_layers=4,
 * dec_channels=128, num_plane_channels=48, proj_combination='avg', viewdir_proj_combination='concat_pos'), the only
 * configuration the shipped YAMLs instantiate (config/TrainModels.yml, Feature_Planes_Only.yml; models.py:118-197). */
#define NVSR_PLANE_CHANNELS 48
#define NVSR_DEC_CHANNELS 128
#define NVSR_DEC_LAYERS 4
#define NVSR_DECODER_NATURAL_FLOATS 130564 /* state-dict order, see nvsr_pack_decoder */
#define NVSR_DECODER_PACKED_F32_FLOATS 130576 /* f32 MFMA-fragment order + biases/heads */
/* packed blob = [f32 fragments + biases/heads][bf16 fragments, 3 limbs per weight][bf16 fragments, 2 limbs per weight] (32-bit words) */
#define NVSR_DECODER_PACKED_FLOATS 453136
#define NVSR_DECODER_PACKED_BWD_FLOATS 487424 /* transposed layers for the backward pass (f32 fragments + 3-bf16-limb + 2-f16-limb fragments), see nvsr_pack_decoder_bwd */

/* One scene = 3 position planes + 1 view-direction plane, CHANNEL-LAST [H][W][48] (192 B per texel), the per-scene
 * normalisation box (models.py:261-268) and the plane projections rot_mats[d][:,1:] (models.py:471-497). */
typedef struct nvsr_scene {
    const float* planes[4]; /* device, channel-last; [0..2] position planes D0..D2, [3] view-direction plane */
    int32_t ph[4], pw[4];   /* rows (grid y) and columns (grid x) of each plane */
    float lo[5];            /* (float)box[0][i] */
    float range[5];         /* (float)(box[1][i] - box[0][i]), subtraction in double like the reference */
    float proj[3][6];       /* row-major 3x2: grid = n_xyz @ proj[d] */
} nvsr_scene;

int nvsr_version(void);

/* Arithmetic of the decoder GEMMs: the fused render pass (nvsr_render_pass*, N >= 16384 rays) and the training kernels
 * (nvsr_decode_rays*, nvsr_render_pass_backward_gates, nvsr_decoder_weight_grad -- what computes a gradient uses 3 limbs whenever a limb
 * mode is selected; a forward and the backward that consumes its gates / record must both run limb modes or both run f32).
 * Inputs, outputs, accumulation and everything outside the GEMMs are f32 in every mode.
 *   NVSR_ARITH_F32    v_mfma_f32_32x32x2_f32: exact f32 products, f32 accumulation -- the reference's arithmetic
 *   NVSR_ARITH_BF16X3 every f32 operand split exactly into 3 bf16 limbs (8 + 8 + 8 significant bits, by truncation), the products
 *                     Wh(xh + xm + xl) + Wm(xh + xm) + Wl xh on v_mfma_f32_32x32x16_bf16, 2.7x the f32 matrix rate.  Error of a dot
 *                     product of K terms, relative to sum |W_k||x_k|:
 *                       - the three dropped products: |Wm xl| + |Wl xm| + |Wl xl| < (2^-22 + 2^-22 + 2^-30) |W||x| = 2^-21 + 2^-30
 *                         (truncation limbs: |m| < 2^-7 |v|, |l| < 2^-15 |v|) -- about FOUR f32 roundings per product at worst, and
 *                         one-sided: truncated limbs carry the operand's sign, so the dropped part has the sign of W x and adds up
 *                         over K instead of averaging out;
 *                       - plus one f32 rounding of the running sum per MFMA that adds into it: 6 K / 16 of them (K / 2 for the f32
 *                         MFMA), <= 2^-24 each -- the products are exact, the sums are ordinary f32 sums.
 *                     Measured on adversarial operands (every mantissa 0x7FFFFF or 0x00FFFF, equal signs, K = 192; tests/
 *                     test_hip_round2.py::test_limb_error_bound): max error 9.8e-7 (2^-20), mean -1.8e-7, against 5.6e-7 / +1.8e-8 for
 *                     the exact-f32 kernel.  NOT bit-grade f32; every parity tolerance of the path holds in it (2e-5 on decoder outputs;
 *                     PSNR of the 800^2 frame vs the double-precision oracle 89.2 dB against 90.8 dB).  Default of rounds 1-3 for the
 *                     render pass; still what every training kernel runs (no exponent-range limits: gradients span many decades).
 *   NVSR_ARITH_F16X2  every f32 operand split into 2 f16 limbs by ROUNDING TO NEAREST: hi = RN(x) (11 significant bits), lo = RN(x - hi)
 *                     (x - hi is exact in f32); |x - hi - lo| <= 2^-23 |x| (one f32 ulp, and zero for three operands in four).  Products
 *                     Wh xh + Wh xl + Wl xh on v_mfma_f32_32x32x16_f16 (f16 x f16 products are exact in f32, f32 accumulation): 3 MFMAs
 *                     per f32 product block, 5.3x the f32 matrix rate and twice the rate of NVSR_ARITH_BF16X3.  Error per product
 *                     <= (2^-23 + 2^-23 + 2^-22) |W||x| = 2^-21 |W||x| at worst (the same bound as BF16X3), but with limbs of either sign:
 *                     the dropped Wl xl and the representation errors average out over K instead of adding up, and the sum takes half as
 *                     many f32 roundings (3 K / 16).  Measured against float64 it is as close as BF16X3 or closer (tests/
 *                     test_hip_round3.py::test_limb_gemm_error_bounds; frame PSNR and decoder-output errors in bench.py's line).
 *                     RANGE: f16 has 5 exponent bits, so the kernels carry static power-of-two scales (exact): weights are packed as
 *                     W 2^8 and activations / features are held as x 2^4.  Both limbs are normal numbers for 2^-10 <= |W| < 255 and
 *                     2^-6 <= |x| < 4094; below that the low limb is subnormal (honoured by the matrix pipe) and the error is absolute,
 *                     <= 2^-33 per weight and <= 2^-29 per activation; a weight >= 255 or an activation >= 4094 overflows to inf and
 *                     the pixel comes out NaN -- loud, never a wrong number (use BF16X3 or F32 for such a network).
 *                     The fused render pass (inference), the training forward (nvsr_decode_rays*) and the gate-driven backward
 *                     (nvsr_render_pass_backward_gates*), each with or without the weight-gradient record.  Gradients span many decades --
 *                     from ray to ray and, behind a surface, from sample to sample of one ray and between dL/dsigma and dL/drgb of one
 *                     sample -- but the 128 features of one chain of one POINT do not: the backward scales every point's density chain
 *                     and rgb chain by their own powers of two (exact: a point is a column of every product; the largest |dL/draw| of
 *                     the chain goes to [2^3, 2^4), undone when the chains meet and before the feature gradients are scattered) and
 *                     multiplies by UNSCALED transposed weights (low limb subnormal below |w| = 0.125: absolute error <= 2^-25, ~1e-6
 *                     of a typical weight).  Against the exact-f32 backward on the same gates its plane gradients are as close as the
 *                     3-limb backward's -- relative L2 ~1e-6, worst texel 0.1 % on the gradients of an opaque scene (round 3 scaled
 *                     whole wave tiles: 7e-6 / 1.3 % there; tests/test_hip_round4.py::test_f16_backward_with_the_dynamic_range_of_an_
 *                     opaque_ray, tests/test_hip_round3.py::test_f16_backward_matches_the_f32_backward); the
 *                     gradient half of the record is written UNSCALED, and so is the layer-input half by the recording
 *                     forward (round 4: a pass runs ONE arithmetic, forward and backward, with or without a record).  The weight-gradient
 *                     contraction (nvsr_decoder_weight_grad*) reads that f32 record and runs BF16X3 when F16X2 is selected.
 *                     The gates a forward publishes are signs of pre-activations: any limb backward consumes any limb forward's.
 * The mode is a per-call argument of the *_arith entry points below (NVSR_ARITH_INHERIT = the process default); every other entry point
 * uses the process default, whose initial value comes from the environment variable NVSR_DECODER_ARITHMETIC = f32 | bf16x3 | f16x2
 * and which nvsr_set_decoder_arithmetic changes.  Nothing but that default is process-global: calls with explicit modes are re-entrant
 * across threads and streams. */
#define NVSR_ARITH_INHERIT (-1)
#define NVSR_ARITH_INVALID (-2)     /* what nvsr_get_*_arithmetic returns when the environment variable holds an unknown string: every call
                                       that inherits the process default then fails with NVSR_ERR_SHAPE until nvsr_set_*_arithmetic is called */
#define NVSR_ARITH_F32 0
#define NVSR_ARITH_F16X2 2
#define NVSR_ARITH_BF16X3 3
#define NVSR_ARITH_DEFAULT NVSR_ARITH_F16X2
int nvsr_get_decoder_arithmetic(void);
int nvsr_set_decoder_arithmetic(int mode);
/* Range flag of NVSR_ARITH_F16X2 (round 4).  A launch in that mode whose operands leave the ranges above produces NaN -- which the caller
 * would otherwise have to find by reading its outputs.  With a device word registered here, every F16X2 launch that WRITES A NON-FINITE
 * RESULT also ORs a bit into that word (1 -- the fused render passes: a non-finite rgb / opacity of a ray; nvsr_decode_rays*: a non-finite raw
 * row; 2 -- the SR network: a non-finite value of an output plane inside its region of interest): the host mirror zeroes the word before a frame / an iteration, reads it back
 * asynchronously and re-renders in NVSR_ARITH_BF16X3 or raises (train_utils.py, training.py).  The reference renders any f32 model
 * (models.py:395-421); this is what keeps the drop-in's default arithmetic from ever returning NaN where the reference returns a number.
 * NULL (the initial state) disables the reporting.  The pointer is process-global like the default arithmetic and is read when a launch is
 * enqueued; the word must stay allocated until every launch enqueued while it was registered has finished.  Modes other than F16X2 never
 * touch it (they have no range limit; a NaN they return was a NaN in their inputs). */
int nvsr_set_range_flag(uint32_t* device_word);
uint32_t* nvsr_get_range_flag(void);
/* The arithmetic primitive alone (test hook, one wavefront): Y[32][32] = W[32][K] X[K][32] (row-major f32, K a multiple of 16) with the
 * operands split and multiplied exactly as the kernels of `arithmetic` do it (NVSR_ARITH_F32 | _BF16X3 | _F16X2, incl. the static scales of
 * F16X2) -- lets a test put chosen mantissas / magnitudes through the products that replace models.py:381-421's nn.Linear GEMMs. */
int nvsr_limb_gemm_probe(int arithmetic, int K, const float* W, const float* X, float* Y, nvsr_stream_t stream);
/* Same for the 3x3 convolutions of the SR network (process default + *_arith twins): forward and data gradient of the layers with
 * Cin % 16 == 0 and a multiple of 256, or at most 64, output channels (every layer of EDSR(256); other shapes always use the f32 kernel),
 * and every weight gradient: NVSR_ARITH_F32, NVSR_ARITH_BF16X3 (same error bound as above) or NVSR_ARITH_F16X2 -- the FORWARD convolutions
 * of the layers with Cin % 32 == 0 and Cout % 128 == 0 (EDSR's 65 trunk and 2 up-sampling convolutions, 97 % of the FLOPs) on 2 f16 limbs
 * (3 MFMAs per product block instead of 6; weights packed as W 2^8, the input patch held as x 2^4, same ranges and the same NaN-on-overflow
 * rule as above) -- and their data gradients too, with the power of two that puts the largest |dy| of the layer's whole gradient tensor into
 * [2^12, 2^13) in place of the static activation scale (one reduction per layer: gradients span many decades from layer to layer and step to
 * step, a tensor's values a few) and their weight gradients (X with the static scale, dy with its tensor's); the narrow input / output layers run
 * 3 bf16 limbs in that mode.
 * Environment: NVSR_CONV_ARITHMETIC = f32 | bf16x3 | f16x2. */
#define NVSR_CONV_ARITH_DEFAULT NVSR_ARITH_F16X2
int nvsr_get_conv_arithmetic(void);
int nvsr_set_conv_arithmetic(int mode);

/* ---- data layout ------------------------------------------------------------------------------------------------ */
/* [C,H,W] (reference plane layout, models.py:436-439) -> channel-last [H,W,C]; and back. */
int nvsr_plane_to_channel_last(const float* nchw, float* nhwc, int C, int H, int W, nvsr_stream_t stream);
int nvsr_plane_from_channel_last(const float* nhwc, float* nchw, int C, int H, int W, nvsr_stream_t stream);

/* natural blob = state-dict order (models.py:169-195): density_dec.0.{0..3}.{weight[out,in],bias}, fc_alpha.0.{weight,bias},
 * rgb_dec.0.{0..3}.{weight,bias}, fc_rgb.0.{weight,bias}  ->  packed blob consumed by the decode/render kernels. */
int nvsr_pack_decoder(const float* natural, float* packed, nvsr_stream_t stream);

/* ---- rays (nerf_helpers.py) -------------------------------------------------------------------------------------- */
/* get_ray_bundle (nerf_helpers.py:507-549).  c2w: 16 floats row-major (device).  ro, rd: [(H+2p)*(W+2p), 3].
 * focal_x = get_focal(f,'H'), focal_y = get_focal(f,'W') as the reference names them (:539-540). */
int nvsr_get_ray_bundle(int H, int W, double focal_x, double focal_y, const float* c2w, int padding, double offset,
                        float* ro, float* rd, nvsr_stream_t stream);
/* The rays of N selected pixels only: get_ray_bundle(...)[row, col] as train() uses it (train_nerf.py:814,842-844: the reference
 * generates all H*W rays every iteration, then gathers num_random_rays of them).  row_col [N,2] int32 (device); (row, col) may
 * lie outside the image (padding).  Bit-identical to the corresponding rows of nvsr_get_ray_bundle. */
int nvsr_get_ray_bundle_at(int H, int W, double focal_x, double focal_y, const float* c2w, double offset, int64_t N, const int32_t* row_col,
                           float* ro, float* rd, nvsr_stream_t stream);
/* Training pixels drawn on the device: entries [first, first + n) of a keyed pseudo-random PERMUTATION of range(total), total = H*W --
 * n distinct pixels, the first n draws of `np.random.choice(H*W, n, replace=False)` (train_nerf.py:836-838) with the device as the
 * generator (the reference permutes all H*W indices on the host every iteration: 640 000 for an 800x800 view, more than a whole step
 * of this library).  Ranks that pass the same key and disjoint [first, first + n) draw disjoint shares of one global batch.
 *   permutation: x -> F(x) repeated until F(x) < total ("cycle walking"), F = 8-round balanced Feistel network on 2*hb bits,
 *   hb = the smallest integer >= 1 with 2^(2 hb) >= total; halves L = x >> hb, R = x & (2^hb - 1); round r = 0..7:
 *     h = (uint32) R * 0x9E3779B1 + k_r;  h ^= h >> 15;  h *= 0x85EBCA77;  h ^= h >> 13;  h *= 0xC2B2AE3D;  h ^= h >> 16;
 *     (L, R) <- (R, L ^ (h & (2^hb - 1)));      F(x) = L << hb | R after round 7;
 *   k_r = high 32 bits of splitmix64(key + r)  (splitmix64(x): x += 0x9E3779B97F4A7C15; x = (x ^ x >> 30) * 0xBF58476D1CE4E5B9;
 *   x = (x ^ x >> 27) * 0x94D049BB133111EB; x ^ x >> 31).  Integer arithmetic only: bit-exact against the CPU restatement.
 * Index k of the permutation is pixel (row k % H, col k / H) -- the reference's column-by-column `coords` (train_nerf.py:818-828).
 * row_col [n,2] int32 (what nvsr_get_ray_bundle_at takes).  target (or NULL) [n,channels] = image[row, col, :] of image [H,W,channels]
 * (train_nerf.py:845 `target_s = img_target[select_inds...]`). */
int nvsr_sample_pixels(int64_t total, int H, int W, uint64_t key, int64_t first, int64_t n, const float* image, int channels,
                       int32_t* row_col, float* target, nvsr_stream_t stream);
/* The key of draw number `calls` of a sampler seeded with `seed`: splitmix64(splitmix64(seed) ^ calls) (host function, integer only).
 * DevicePixelSampler passes it to nvsr_sample_pixels; nvsr_sample_pixels_seq derives the same key on the device. */
uint64_t nvsr_sample_key(uint64_t seed, uint64_t calls);
/* nvsr_sample_pixels for launches replayed from a HIP graph (training.GraphedTrainStep): the key is nvsr_sample_key(state[0], state[1]) read
 * from DEVICE memory, and the launch advances state[1] by one when all its workgroups have read it -- replay k of one captured launch
 * draws what the k-th eager nvsr_sample_pixels call of a sampler with the same seed draws.  state: 4 x uint64 on the device
 * {seed, calls, 0 (arrival counter, must be 0 between launches), unused}; launches sharing a state must be ordered on one stream.  n >= 1. */
int nvsr_sample_pixels_seq(int64_t total, int H, int W, uint64_t* state, int64_t first, int64_t n, const float* image, int channels,
                           int32_t* row_col, float* target, nvsr_stream_t stream);
/* img2mse of two images against one target in one launch (train_nerf.py:893-905 computes F.mse_loss(rgb_coarse, target) and
 * F.mse_loss(rgb_fine, target) separately): losses[0] = mean((a - t)^2), losses[1] = mean((b - t)^2) (b may be NULL: losses[1] untouched);
 * g_a / g_b (or NULL) [n] = 2 (x - t) / n, the gradient of the loss with respect to its image.  n <= NVSR_MSE_PAIR_MAX_ELEMS elements
 * (one workgroup sums them; a training batch is 3 x 4096). */
#define NVSR_MSE_PAIR_MAX_ELEMS (1 << 22)
int nvsr_mse_pair(int64_t n, const float* a, const float* b, const float* target, float* losses, float* g_a, float* g_b, nvsr_stream_t stream);
/* the same two losses and their sum, losses3 = {mean((a - t)^2), mean((b - t)^2), their f32 sum} (train_nerf.py:905: loss = coarse_loss +
 * fine_loss) -- the three scalars an iteration reports, contiguous for one copy to the host; no gradients (nvsr_mse_pair_backward) */
int nvsr_mse_pair_sum(int64_t n, const float* a, const float* b, const float* target, float* losses3, nvsr_stream_t stream);
/* gradients of the two losses with the incoming gradients folded in: g_a = (2 (a - t) / n) * *scale_a, g_b = (2 (b - t) / n) * *scale_b
 * (device scalars; NULL = 1; g_a / g_b may be NULL) -- one launch where autograd would run two multiplies behind nvsr_mse_pair's gradients */
int nvsr_mse_pair_backward(int64_t n, const float* a, const float* b, const float* target, const float* scale_a, const float* scale_b, float* g_a,
                           float* g_b, nvsr_stream_t stream);
/* ndc_rays (nerf_helpers.py:578-605) */
int nvsr_ndc_rays(int H, int W, double focal, double near_, int64_t N, const float* ro, const float* rd, float* ro_out,
                  float* rd_out, nvsr_stream_t stream);
/* rays [N,11] = [ro, rd, near, far, viewdir = view_src/|view_src|]  (run_one_iter_of_nerf, train_utils.py:213-226) */
int nvsr_pack_rays(int64_t N, const float* ro, const float* rd, const float* view_src, double near_, double far_,
                   float* rays, nvsr_stream_t stream);
/* coarse depths + stratified jitter (predict_and_render_radiance, train_utils.py:95-109); t_rand [N,Nc] or NULL */
int nvsr_coarse_z(int64_t N, int Nc, const float* rays, int lindisp, const float* t_rand, float* z, nvsr_stream_t stream);

/* ---- importance sampling ------------------------------------------------------------------------------------------ */
/* sample_pdf_2 (nerf_helpers.py:668-702): bins [N,nb], weights [N,nb-1], u [N,ns] or NULL (det=True: linspace(0,1,ns))
 * -> samples [N,ns].  nb <= 256. */
int nvsr_sample_pdf(int64_t N, int nb, int ns, const float* bins, const float* weights, const float* u, float* samples,
                    nvsr_stream_t stream);
/* cumprod_exclusive (nerf_helpers.py:409-430): [N,n] rows -> out[:,0] = 1, out[:,i] = in[:,0] * ... * in[:,i-1]; in and out must not alias */
int nvsr_cumprod_exclusive(int64_t N, int n, const float* in, float* out, nvsr_stream_t stream);
/* autograd of cumprod_exclusive (the reference's helper is differentiable torch code, nerf_helpers.py:409-430): in, out = the forward's input
 * and output [N,n], g_out = dL/dout -> g_in = dL/din; exact for zeros in `in` (no division) */
int nvsr_cumprod_exclusive_backward(int64_t N, int n, const float* in, const float* out, const float* g_out, float* g_in,
                                    nvsr_stream_t stream);
/* torch.sort(x, dim=-1) values of [N,n] rows, n <= 512 (train_utils.py:155); in and out may alias */
int nvsr_sort_rows(int64_t N, int n, const float* in, float* out, nvsr_stream_t stream);
/* fused train_utils.py:144-155: z_mid, sample_pdf(z_mid, w[1:-1], Nf, det = (u == NULL)), sort(cat(z, samples)) -> [N,Nc+Nf] */
int nvsr_importance_resample(int64_t N, int Nc, int Nf, const float* z_coarse, const float* weights, const float* u,
                             float* z_fine, nvsr_stream_t stream);

/* the same with the un-jittered coarse depths of train_utils.py:95-100 recomputed from the rays' near / far (packed rays columns 6, 7)
 * instead of read: z_coarse of nvsr_coarse_z(..., t_rand = NULL) never has to exist (nvsr_render_rays does this for inference frames) */
int nvsr_importance_resample_rays(int64_t N, int Nc, int Nf, const float* rays, int lindisp, const float* weights, const float* u,
                                  float* z_fine, nvsr_stream_t stream);

/* ---- tri-plane decoder -------------------------------------------------------------------------------------------- */
/* TwoDimPlanesModel.forward (models.py:381-421): x [P,6] = [xyz, viewdir] -> out [P,4] = [rgb_raw, sigma_raw].  In the process's default
 * arithmetic (round 4: the limb modes run the training forward's kernel on tiles of 32 consecutive points; NVSR_ARITH_F32 the exact-f32
 * MFMA kernel); nvsr_triplane_decode_arith takes the mode per call. */
int nvsr_triplane_decode(const nvsr_scene* scene, const float* packed_decoder, int64_t P, const float* x, float* out,
                         nvsr_stream_t stream);
int nvsr_triplane_decode_arith(const nvsr_scene* scene, const float* packed_decoder, int64_t P, const float* x, float* out, int arithmetic,
                               nvsr_stream_t stream);

/* ---- compositing -------------------------------------------------------------------------------------------------- */
/* volume_render_radiance_field (volume_rendering_utils.py:6-51), mip_nerf=False.  raw [N,S,4], z [N,S], rd [N,3],
 * noise [N,S] or NULL (already scaled by radiance_field_noise_std).  weights/depth may be NULL. */
int nvsr_composite(int64_t N, int S, const float* raw, const float* z, const float* rd, const float* noise, int white_bkgd,
                   float* rgb, float* disp, float* acc, float* weights, float* depth, nvsr_stream_t stream);

/* same with the ray directions taken from packed rays [N,11] (columns 3..5) */
/* volume_render_radiance_field(..., mip_nerf=True) (volume_rendering_utils.py:19-26,41-42): raw [N,S,4] over the S intervals of the
 * edges z [N,S+1] -- every interval finite (no 1e10 tail), depth_map integrates the interval mid-points */
int nvsr_composite_mip(int64_t N, int S, const float* raw, const float* z, const float* rd, const float* noise, int white_bkgd,
                       float* rgb, float* disp, float* acc, float* weights /* [N,S] or NULL */, float* depth /* or NULL */,
                       nvsr_stream_t stream);
int nvsr_composite_rays(int64_t N, int S, const float* raw, const float* z, const float* rays, const float* noise, int white_bkgd,
                        float* rgb, float* disp, float* acc, float* weights, float* depth, nvsr_stream_t stream);

/* ---- fused per-ray render pass --------------------------------------------------------------------------------------
 * run_network + TwoDimPlanesModel.forward + volume_render_radiance_field for one pass (train_utils.py:111-139 or :156-180):
 * points ro + rd*z are generated, decoded and composited inside one kernel; no [N,S,*] intermediate reaches HBM.
 * rays [N,11], z [N,S] -> rgb [N,3], disp [N], acc [N]; weights [N,S] / depth [N] optional (NULL). */
int nvsr_render_pass(const nvsr_scene* scene, const float* packed_decoder, int64_t N, int S, const float* rays, const float* z,
                     const float* noise, int white_bkgd, float* rgb, float* disp, float* acc, float* weights, float* depth,
                     nvsr_stream_t stream);

/* Un-fused variant for small ray counts (training batches): decoder outputs raw [N,S,4] for all samples of all rays, tiled over
 * (ray block, sample) so that N*S/128 workgroup-steps fill the chip; feed raw to nvsr_composite. */
int nvsr_decode_rays(const nvsr_scene* scene, const float* packed_decoder, int64_t N, int S, const float* rays, const float* z,
                     float* raw, nvsr_stream_t stream);

/* predict_and_render_radiance (train_utils.py:71-182) for a ray chunk: coarse z -> coarse pass -> importance resample ->
 * fine pass.  workspace: nvsr_render_workspace_floats(N, Nc, Nf) floats.  Nf == 0: coarse only (fine outputs untouched).
 * Up to NVSR_FUSED_MIN_RAYS rays the passes run un-fused (nvsr_decode_rays + nvsr_composite), above it fused (nvsr_render_pass). */
#define NVSR_FUSED_MIN_RAYS 65536
/* the same threshold as the library was BUILT with (host mirrors that pick kernels per launch must use this, not a copy of the constant) */
int64_t nvsr_fused_min_rays(void);
int64_t nvsr_render_workspace_floats(int64_t N, int Nc, int Nf);
int nvsr_render_rays(const nvsr_scene* scene, const float* packed_coarse, const float* packed_fine, int64_t N, int Nc, int Nf,
                     const float* rays, int lindisp, int white_bkgd, const float* t_rand, const float* u,
                     const float* noise_coarse, const float* noise_fine, float* rgb_c, float* disp_c, float* acc_c,
                     float* rgb_f, float* disp_f, float* acc_f, float* workspace, nvsr_stream_t stream);

/* ---- feature-plane super-resolution (EDSR wrapped by PlanesSR) ------------------------------------------------------
 * All activations are [C][H][W] fp32 (the reference's layout); convolutions are 3x3, stride 1, no padding, no bias. */
/* one nn.Conv2d weight [Cout][Cin][3][3] -> MFMA fragment order; size = nvsr_conv3x3_packed_floats(Cin, Cout) */
int64_t nvsr_conv3x3_packed_floats(int Cin, int Cout);
int nvsr_pack_conv3x3(const float* w, int Cin, int Cout, float* packed, nvsr_stream_t stream);
/* out = epilogue(conv3x3(in)); epilogue 0 none | 1 ReLU | 2 residual: conv*0.1 + skip[..., 2:-2, 2:-2] with skip [Cout][H+2][W+2]
 * (_Residual_Block.forward, models.py:777-786) | 3 nn.PixelShuffle(2) (out [Cout/4][2(H-2)][2(W-2)]) */
int nvsr_conv3x3(const float* in, int Cin, int H, int W, const float* packed, int Cout, int epilogue, const float* skip, float* out,
                 nvsr_stream_t stream);
/* EDSR(in_channels, out_channels, hidden_size, n_blocks, scale_factor = 2^n_up, padding = 0)  (models.py:789-822).
 * natural = state-dict order: conv_input, residual.{b}.conv1, residual.{b}.conv2, conv_mid, upscale.{0,2,..}, conv_output. */
int64_t nvsr_edsr_natural_floats(int Cin, int Cout, int hid, int nblocks, int n_up);
int64_t nvsr_edsr_packed_floats(int Cin, int Cout, int hid, int nblocks, int n_up);
int nvsr_pack_edsr(const float* natural, int Cin, int Cout, int hid, int nblocks, int n_up, float* packed, nvsr_stream_t stream);
/* The same blob with only the fragment regions that a launch in `arithmetic` reads (NVSR_ARITH_INHERIT = the process default) -- for callers whose
 * weights change between launches: a training iteration (train_nerf.py:903-914: optimizer steps between forwards) re-packs both blobs of the network
 * every iteration, and of the four regions of a layer (f32 fragments, bf16 limbs for two MFMA shapes, f16 limbs) an arithmetic reads one.  The other
 * regions keep whatever `packed` held: such a blob serves launches in THAT arithmetic only (the *_arith entry points with the same value).
 * NVSR_PACK_ALL_ARITHMETICS = every region (what nvsr_pack_edsr does). */
#define NVSR_PACK_ALL_ARITHMETICS (-3)
int nvsr_pack_edsr_arith(const float* natural, int Cin, int Cout, int hid, int nblocks, int n_up, float* packed, int arithmetic, nvsr_stream_t stream);
int nvsr_edsr_out_size(int H, int W, int nblocks, int n_up, int* Ho, int* Wo);
int64_t nvsr_edsr_workspace_floats(int hid, int nblocks, int n_up, int H, int W);
int nvsr_edsr_forward(const float* x, int Cin, int H, int W, const float* packed, int Cout, int hid, int nblocks, int n_up, float* out,
                      float* workspace, nvsr_stream_t stream);
/* PlanesSR.forward (models.py:884-926): lr [C][R0][R1] -> out [C][sf*R0][sf*R1] = EDSR(replicate-padded crop)[over:-over] +
 * bilinear_x{sf}(lr) inside the ROI, NaN outside.  roi: NULL = full plane, else 4 HOST floats [[ymin,xmin],[ymax,xmax]] in [-1,1]
 * (models.py:278-279).  pad = EDSR.required_padding, over = HR_overpadding (models.py:836-842).  mean/std: optional [C] device
 * vectors (planes_{mean,std}_NON_LEARNED). */
/* plane_interp of the reference (config/TrainModels.yml:72,172): how a plane is sampled (grid_sample) and up-sampled (F.interpolate) */
#define NVSR_PLANE_INTERP_BILINEAR 0
#define NVSR_PLANE_INTERP_BICUBIC 1
/* align_corners of the bilinear residual F.interpolate(LR, scale_factor, 'bilinear', align_corners) (models.py:858-859; PlanesSR.align_corners
 * is the planes model's, :222): library state like the conv arithmetic, 1 (every shipped config) unless set; read by every nvsr_planes_sr*
 * call (forward and backward) when it is launched.  A binding with models of both kinds sets it before each call (ops.py does). */
int nvsr_set_sr_align_corners(int align_corners);
int nvsr_get_sr_align_corners(void);
/* ... and its mode (PlanesSR.plane_interp, models.py:858-859): NVSR_PLANE_INTERP_BILINEAR (every shipped config) | NVSR_PLANE_INTERP_BICUBIC */
int nvsr_set_sr_plane_interp(int plane_interp);
int nvsr_get_sr_plane_interp(void);
int64_t nvsr_planes_sr_workspace_floats(int C, int R0, int R1, int hid, int nblocks, int n_up, int pad, const float* roi);
int nvsr_planes_sr(const float* lr, int C, int R0, int R1, const float* packed, int hid, int nblocks, int n_up, int pad, int over,
                   const float* roi, const float* mean, const float* std_, float* out, float* workspace, nvsr_stream_t stream);

/* Batched variants: B planes of equal size share every launch (a single 200^2 plane fills the 256 CUs for only 1.2-1.8 workgroup
 * rounds per layer; the 3 position planes of a scene together for 3.6-5.4).  x [B][Cin][H][W] -> out [B][Cout][Ho][Wo];
 * workspace = B * nvsr_edsr_workspace_floats(...).  nvsr_planes_sr_batch: lr / out are HOST arrays of B device pointers,
 * workspace = B * nvsr_planes_sr_workspace_floats(...). */
int nvsr_edsr_forward_batch(const float* x, int B, int Cin, int H, int W, const float* packed, int Cout, int hid, int nblocks, int n_up,
                            float* out, float* workspace, nvsr_stream_t stream);
int nvsr_planes_sr_batch(const float* const* lr, int B, int C, int R0, int R1, const float* packed, int hid, int nblocks, int n_up, int pad,
                         int over, const float* roi, const float* mean, const float* stdv, float* const* out, float* workspace,
                         nvsr_stream_t stream);

/* ---- positional-encoding baseline (MipNeRF_baseline.yml; not on the tri-plane path) ------------------------------------ */
/* positional_encoding (nerf_helpers.py:552-575): x [P,D] -> [P, (include_input ? D : 0) + 2*D*L] = [x, sin(2^0 x), cos(2^0 x), ...] */
int nvsr_positional_encoding(int64_t P, int D, const float* x, int L, int include_input, float* out, nvsr_stream_t stream);
/* FlexibleNeRFModel.forward (models.py:83-108), use_viewdirs=True, num_layers_dir=1: x [P, dim_xyz+dim_dir] -> [P,4].
 * blob = state-dict order: layer1, layers_xyz.{i}, layers_dir.0, fc_alpha, fc_rgb, fc_feat, each {weight[out,in], bias}. */
int nvsr_flexible_nerf_forward(int64_t P, const float* x, int dim_xyz, int dim_dir, int hidden, int num_layers, int skip_every,
                               const float* blob, float* out, nvsr_stream_t stream);

/* ---- training: gradient with respect to the feature planes -------------------------------------------------------------
 * The reference differentiates run_one_iter_of_nerf with torch.autograd (train_nerf.py:860-903); with the decoder frozen
 * (Feature_Planes_Only.yml) the leaves are the planes.  z_samples carry no gradient (`.detach()`, train_utils.py:153). */
/* nvsr_render_pass that also returns the decoder outputs raw [N,S,4] (needed by nvsr_composite_backward); raw_out may be NULL */
int nvsr_render_pass_ex(const nvsr_scene* scene, const float* packed_decoder, int64_t N, int S, const float* rays, const float* z,
                        const float* noise, int white_bkgd, float* rgb, float* disp, float* acc, float* weights, float* depth,
                        float* raw_out, nvsr_stream_t stream);
/* transposed decoder layers in MFMA-fragment order (NVSR_DECODER_PACKED_BWD_FLOATS floats) from the natural blob */
int nvsr_pack_decoder_bwd(const float* natural, float* packed_bwd, nvsr_stream_t stream);
/* backward of volume_render_radiance_field (volume_rendering_utils.py:18-49): g_rgb [N,3], g_acc [N] or NULL -> g_raw [N,S,4]; S <= 512 */
int nvsr_composite_backward(int64_t N, int S, const float* raw, const float* z, const float* rd, const float* noise, int white_bkgd,
                            const float* g_rgb, const float* g_acc, float* g_raw, nvsr_stream_t stream);
/* the same with the gradient of depth_map (volume_rendering_utils.py:42-43: depth_map = sum_s w_s z_s; mip_nerf: z [N,S+1], z_s = interval
 * midpoints) as a third incoming gradient, g_depth [N] or NULL.  disp_map = 1 / max(1e-10, depth_map / acc_map) (:46) is a function of
 * depth_map and acc_map: its gradient is folded into g_depth / g_acc by the caller (ops.py `composite` does). */
int nvsr_composite_backward_depth(int64_t N, int S, const float* raw, const float* z, const float* rd, const float* noise, int white_bkgd,
                                  const float* g_rgb, const float* g_acc, const float* g_depth, int mip_nerf, float* g_raw, nvsr_stream_t stream);
/* nvsr_composite_backward_depth with the ray directions read out of packed rays [N,11] (columns 3..5 of nvsr_pack_rays' rows, what
 * nvsr_composite_rays reads in the forward): a training step's backward needs no [N,3] copy of them */
int nvsr_composite_backward_rays(int64_t N, int S, const float* raw, const float* z, const float* rays, const float* noise, int white_bkgd,
                                 const float* g_rgb, const float* g_acc, const float* g_depth, int mip_nerf, float* g_raw, nvsr_stream_t stream);
/* the same for nvsr_composite_mip (z [N,S+1]) */
int nvsr_composite_backward_mip(int64_t N, int S, const float* raw, const float* z, const float* rd, const float* noise, int white_bkgd,
                                const float* g_rgb, const float* g_acc, float* g_raw, nvsr_stream_t stream);
/* backward of one decode pass: g_raw [N,S,4] -> grad_planes[4] (host array of 4 device pointers, CHANNEL-LAST like the scene's
 * planes, accumulated with float atomics: zero them first) */
int nvsr_render_pass_backward(const nvsr_scene* scene, const float* packed_decoder, const float* packed_bwd, int64_t N, int S,
                              const float* rays, const float* z, const float* g_raw, float* const* grad_planes, nvsr_stream_t stream);

/* Faster path when the decoder is frozen: the training forward publishes every layer's ReLU gate (gates: N*S*32 uint32, 128 B
 * per point) and the backward runs the transposed layers only, without recomputing the forward. */
int nvsr_decode_rays_ex(const nvsr_scene* scene, const float* packed_decoder, int64_t N, int S, const float* rays, const float* z,
                        float* raw, uint32_t* gates /* or NULL */, float* record /* or NULL; needs gates */, nvsr_stream_t stream);
int nvsr_render_pass_backward_gates(const nvsr_scene* scene, const float* packed_decoder, const float* packed_bwd, int64_t N, int S,
                                    const float* rays, const float* z, const float* g_raw, const uint32_t* gates, float* const* grad_planes,
                                    float* view_ws /* or NULL, see below */, float* record /* or NULL */, nvsr_stream_t stream);

/* ---- training: gradient with respect to the decoder parameters ('decoder' in nerf.train.what, train_nerf.py:75-77) ------
 * torch.autograd's addmm backward (dW = delta^T @ input, db = sum delta) through models.py:169-195,395-421 becomes two calls:
 * the backward pass additionally RECORDS every layer's input and pre-activation gradient, then one contraction over all points
 * adds the weight / bias gradients into a blob in the natural (state-dict) order of nvsr_pack_decoder. */
/* floats of the record workspace of one pass (9.2 KB per point, one row per point: row = ray * S + s; every array of the record is allocated
 * for N * S rows rounded up to 8 + 32 DUMP rows -- round 6: the limb kernels write a tile's 32 rows as whole cache lines through an LDS stage and
 * send the padding points of a partial tile there; rows >= N * S are never read).  The half that the forward
 * layers produce (layer inputs) is written either by nvsr_decode_rays_ex(record) at forward time or by the recomputing
 * nvsr_render_pass_backward_ex(record); the gradient half by whichever backward entry receives the record. */
int64_t nvsr_decoder_record_floats(int64_t N, int S);
/* nvsr_render_pass_backward with optional outputs: grad_planes may be NULL (planes frozen) or hold NULL entries (that plane
 * frozen); record may be NULL (decoder frozen) or a workspace of nvsr_decoder_record_floats(N, S) floats, fully overwritten */
int nvsr_render_pass_backward_ex(const nvsr_scene* scene, const float* packed_decoder, const float* packed_bwd, int64_t N, int S,
                                 const float* rays, const float* z, const float* g_raw, float* const* grad_planes, float* record,
                                 float* view_ws, nvsr_stream_t stream);
/* view_ws (optional, nvsr_view_grad_workspace_floats(N, S) = N*S*48 floats): all samples of a ray hit the same 4 texels of the small
 * view-direction plane; with the workspace their gradients are first written as plain rows and summed per ray, and only N*4*48
 * atomics reach the plane (without it: N*S*4*48 atomics on 1 024 texels -- correct but 2x slower overall). */
int64_t nvsr_view_grad_workspace_floats(int64_t N, int S);
/* grad_natural [NVSR_DECODER_NATURAL_FLOATS] += gradient of this pass (float atomics: zero it before the first pass) */
int nvsr_decoder_weight_grad(int64_t N, int S, const float* record, float* grad_natural, nvsr_stream_t stream);


/* ---- training: gradients of the super-resolution CNN ('SR' in nerf.train.what) -------------------------------------------
 * torch.autograd through EDSR.forward (models.py:818-822), _Residual_Block.forward (:777-786), PlanesSR.forward (:884-926). */
/* weights [Cout][Cin][3][3] -> fragments of the conv's data gradient (nvsr_conv3x3_packed_floats(Cout, Cin) floats) */
int nvsr_pack_conv3x3_dgrad(const float* w, int Cin, int Cout, float* packed_dgrad, nvsr_stream_t stream);
/* data gradient of nvsr_conv3x3 (epilogue 0): dy [Cout][H-2][W-2] -> dx [Cin][H][W] */
int nvsr_conv3x3_dgrad(const float* dy, int Cin, int H, int W, const float* packed_dgrad, int Cout, float* dx, nvsr_stream_t stream);
/* weight gradient: dw [Cout][Cin][3][3] += scale * sum_{y,x} dy[co][y][x] * x[ci][y+ky][x+kx];  x [Cin][H][W], dy [Cout][H-2][W-2];
 * deterministic (fixed-order reduction of per-row-slab partial sums held in the workspace) */
int64_t nvsr_conv3x3_wgrad_workspace_floats(int Cin, int H, int W, int Cout);
int nvsr_conv3x3_wgrad(const float* dy, const float* x, int Cin, int H, int W, int Cout, float scale, float* dw, float* workspace,
                       nvsr_stream_t stream);
/* EDSR forward that keeps every layer's input in `acts` (nvsr_edsr_acts_floats floats) for nvsr_edsr_backward */
int64_t nvsr_edsr_acts_floats(int Cin, int Cout, int hid, int nblocks, int n_up, int H, int W);
int nvsr_edsr_forward_train(const float* x, int Cin, int H, int W, const float* packed, int Cout, int hid, int nblocks, int n_up, float* out,
                            float* acts, nvsr_stream_t stream);
int64_t nvsr_edsr_packed_dgrad_floats(int Cin, int Cout, int hid, int nblocks, int n_up);
int nvsr_pack_edsr_dgrad(const float* natural, int Cin, int Cout, int hid, int nblocks, int n_up, float* packed_dgrad, nvsr_stream_t stream);
int nvsr_pack_edsr_dgrad_arith(const float* natural, int Cin, int Cout, int hid, int nblocks, int n_up, float* packed_dgrad, int arithmetic,
                               nvsr_stream_t stream);      /* (see nvsr_pack_edsr_arith) */
int64_t nvsr_edsr_backward_workspace_floats(int Cin, int Cout, int hid, int nblocks, int n_up, int H, int W);
/* d_out [Cout][Ho][Wo] -> grad_natural (state-dict order) += weight gradients, dx [Cin][H][W] (or NULL) = input gradient.
 * Streams: everything the call enqueues is ordered on `stream`, and the library creates no stream of its own.  (A build with -DWG_REDUCE_LANE=1
 * runs the per-layer reductions of the weight gradients' partial sums on a library-owned side stream that `stream` waits for before the call
 * returns; measured and switched off: one more hardware queue in a process that already uses four costs more than the overlap gains, sr_bwd.hip.)
 * The same holds for nvsr_planes_sr_backward and nvsr_planes_sr_backward_batch_arith. */
int nvsr_edsr_backward(const float* x, int Cin, int H, int W, const float* acts, const float* packed_dgrad, int Cout, int hid, int nblocks,
                       int n_up, const float* d_out, float* grad_natural, float* dx, float* workspace, nvsr_stream_t stream);
/* PlanesSR: forward that keeps the prepared input + activation record, and its backward.  d_lr (or NULL: LR plane detached,
 * models.py:272) += gradient of the LR plane through the network input and the bilinear residual. */
int64_t nvsr_planes_sr_keep_floats(int C, int R0, int R1, int hid, int nblocks, int n_up, int pad, const float* roi);
int nvsr_planes_sr_train(const float* lr, int C, int R0, int R1, const float* packed, int hid, int nblocks, int n_up, int pad, int over,
                         const float* roi, const float* mean, const float* stdv, float* out, float* workspace, float* keep,
                         nvsr_stream_t stream);
int64_t nvsr_planes_sr_backward_workspace_floats(int C, int R0, int R1, int hid, int nblocks, int n_up, int pad, const float* roi);
int nvsr_planes_sr_backward(int C, int R0, int R1, const float* keep, const float* packed_dgrad, int hid, int nblocks, int n_up, int pad,
                            int over, const float* roi, const float* stdv, const float* d_out, float* grad_natural, float* d_lr,
                            float* workspace, nvsr_stream_t stream);

/* ---- decoder geometries other than the shipped one (inference) --------------------------------------------------------------------
 * TwoDimPlanesModel (models.py:118-434) is configurable: dec_channels, layer counts, skip_connect_every, num_plane_channels,
 * proj_combination, viewdir_proj_combination (config/TrainModels.yml:78,82,92 list alternatives).  The MFMA kernels above are compiled for
 * the shipped configuration; every other one the reference's own layer sizes admit runs through this generic path (plain kernels,
 * activations through HBM, exact-f32 MFMA layers): forward and backward. */
typedef struct nvsr_decoder_geometry {
    int32_t plane_channels;        /* num_plane_channels */
    int32_t viewdir_channels;      /* num_viewdir_plane_channels */
    int32_t hidden;                /* dec_channels */
    int32_t density_layers;        /* dec_density_layers */
    int32_t rgb_layers;            /* dec_rgb_layers */
    int32_t skip_connect_every;    /* 0 = None (models.py:203-207) */
    int32_t proj_combination;      /* 0 'sum' | 1 'avg' | 2 'concat'                              (models.py:355-361) */
    int32_t viewdir_combination;   /* 0 'sum' | 1 'avg' | 2 'mult' | 3 'concat' | 4 'concat_pos'  (models.py:363-379) */
} nvsr_decoder_geometry;
/* parameters in state-dict order: density_dec.0.{l}.{weight[out,in],bias}, fc_alpha.0.{weight,bias}, rgb_dec.0.{l}.{weight,bias},
 * fc_rgb.0.{weight,bias}; -1 for a geometry whose layer sizes are inconsistent in the reference itself (e.g. 'concat' view features on
 * summed position features) */
int64_t nvsr_generic_decoder_natural_floats(const nvsr_decoder_geometry* geometry);
int64_t nvsr_generic_decode_workspace_floats(const nvsr_decoder_geometry* geometry, int64_t P);
/* TwoDimPlanesModel.forward: x [P,6] = [xyz, viewdir] -> out [P,4].  scene: planes channel-last [H][W][plane_channels] (the view plane:
 * [H][W][viewdir_channels]); natural: the blob above (device). */
int nvsr_generic_decode(const nvsr_scene* scene, const nvsr_decoder_geometry* geometry, const float* natural, int64_t P, const float* x,
                        float* out, float* workspace, nvsr_stream_t stream);
/* Backward of nvsr_generic_decode (the reference: torch.autograd through models.py:381-421): d_out [P,4] -> the gradients of the
 * parameters (d_natural, the blob's layout; NULL = not wanted) and of the four planes (d_plane0..3, channel-last like the planes; NULL =
 * not wanted), all ACCUMULATED into what is there (float atomics: last-bit run-to-run differences, like grid_sampler_2d_backward).
 * The forward is recomputed chunk by chunk with every layer's output kept in the workspace. */
int64_t nvsr_generic_decode_backward_workspace_floats(const nvsr_decoder_geometry* geometry, int64_t P);
int nvsr_generic_decode_backward(const nvsr_scene* scene, const nvsr_decoder_geometry* geometry, const float* natural, int64_t P, const float* x,
                                 const float* d_out, float* d_natural, float* d_plane0, float* d_plane1, float* d_plane2, float* d_plane3,
                                 float* workspace, nvsr_stream_t stream);
/* ---- generic decoder, extended scene (round 4) ------------------------------------------------------------------------------------
 * The options of TwoDimPlanesModel that nvsr_scene cannot express, for the generic kernels only (the MFMA kernels of the shipped
 * geometry keep nvsr_scene):
 *   - any number of position planes with their projections (num_planes_or_rot_mats > 3: models.py:140,471-490 -- CoordProjector draws
 *     random orthonormal frames; grid_d = n_xyz @ rot_mats[d][:, 1:]); combine_pos_planes sums / averages / concatenates all of them;
 *   - grid_sample(align_corners=False) and grid_sample(mode='bicubic') (plane_interp, models.py:303-309,320-326; config/TrainModels.yml:72);
 *   - point_coords_noise (models.py:291-293): `coord_noise` [P,3] (device; NULL = none) is ADDED TO THE NORMALISED sample position
 *     before the projections -- the caller draws it (the reference: torch.normal(0, point_coords_noise * 2 / (1 + plane resolution)) on
 *     the CPU generator, once per model call, training only).
 * planes[0 .. num_position_planes-1] are the position planes, planes[num_position_planes] is the view-direction plane. */
#define NVSR_MAX_POSITION_PLANES 15
typedef struct nvsr_scene_ext {
    int32_t num_position_planes;                          /* 1 .. NVSR_MAX_POSITION_PLANES */
    int32_t align_corners;                                /* grid_sample's align_corners: 1 = True (every shipped config), 0 = False */
    int32_t plane_interp;                                 /* grid_sample's mode: NVSR_PLANE_INTERP_BILINEAR (every shipped config) | _BICUBIC */
    const float* planes[NVSR_MAX_POSITION_PLANES + 1];    /* device, channel-last */
    int32_t ph[NVSR_MAX_POSITION_PLANES + 1], pw[NVSR_MAX_POSITION_PLANES + 1];
    float lo[5], range[5];                                /* as in nvsr_scene */
    float proj[NVSR_MAX_POSITION_PLANES][6];              /* row-major 3x2 per position plane */
} nvsr_scene_ext;
int64_t nvsr_generic_decoder_natural_floats_ext(const nvsr_decoder_geometry* geometry, int num_position_planes);
int64_t nvsr_generic_decode_workspace_floats_ext(const nvsr_decoder_geometry* geometry, int num_position_planes, int64_t P);
int64_t nvsr_generic_decode_backward_workspace_floats_ext(const nvsr_decoder_geometry* geometry, int num_position_planes, int64_t P);
/* nvsr_generic_decode / nvsr_generic_decode_backward on an extended scene; d_planes: num_position_planes + 1 device pointers (host
 * array; NULL array or NULL entries = not wanted), accumulated into like d_plane0..3 */
int nvsr_generic_decode_ext(const nvsr_scene_ext* scene, const nvsr_decoder_geometry* geometry, const float* natural, int64_t P, const float* x,
                            const float* coord_noise, float* out, float* workspace, nvsr_stream_t stream);
int nvsr_generic_decode_backward_ext(const nvsr_scene_ext* scene, const nvsr_decoder_geometry* geometry, const float* natural, int64_t P,
                                     const float* x, const float* coord_noise, const float* d_out, float* d_natural, float* const* d_planes,
                                     float* workspace, nvsr_stream_t stream);

/* run_network's model input for a pass (train_utils.py:15-64,111): x [N*S,6] = [ro + rd * z, viewdir] from packed rays [N,11], z [N,S] */
int nvsr_ray_points(int64_t N, int S, const float* rays, const float* z, float* x, nvsr_stream_t stream);

/* ---- per-call arithmetic -----------------------------------------------------------------------------------------------------
 * Twins of the entry points above that run decoder GEMMs or SR convolutions, with the arithmetic as an explicit argument
 * (NVSR_ARITH_F32 | NVSR_ARITH_BF16X3 | NVSR_ARITH_F16X2 | NVSR_ARITH_INHERIT).  Same arguments, same
 * semantics; the un-suffixed entry points are these called with NVSR_ARITH_INHERIT.  A backward call must be given the mode of the
 * forward whose gates / record / activations it consumes (the host mirror stores it with the autograd context).
 * rows_per_tile (convolutions): 0 = the launcher's choice, 2 | 3 | 4 = force that row-tile instantiation of the wide kernels (two output
 * blocks per wave), 8 = the bf16-limb kernel's one-output-block x 8-row wave tile (round 3; equally fast; 2 | 3 | 4 | 8 give the same
 * bits), 16 = the bf16-limb kernel on v_mfma_f32_16x16x32_bf16 (round 3; layers with Cin % 32 == 0 and Cout % 128 == 0 only, their default:
 * the K dimension of an instruction spans 32 input channels, so the f32 accumulation order -- not the limb products -- differs from the
 * other instantiations; same tolerance against the oracle), 18 | 19 | 20 = that kernel with 2 | 3 | 4 rows forced, 22 = its 6-row tile
 * (NVSR_ARITH_F16X2 only)
 * (the results are identical; the parity tests force every instantiation). */
int nvsr_render_pass_arith(const nvsr_scene* scene, const float* packed_decoder, int64_t N, int S, const float* rays, const float* z,
                           const float* noise, int white_bkgd, float* rgb, float* disp, float* acc, float* weights, float* depth,
                           float* raw_out /* or NULL */, int arithmetic, nvsr_stream_t stream);
int nvsr_decode_rays_arith(const nvsr_scene* scene, const float* packed_decoder, int64_t N, int S, const float* rays, const float* z,
                           float* raw, uint32_t* gates /* or NULL */, float* record /* or NULL */, int arithmetic, nvsr_stream_t stream);
int nvsr_render_rays_arith(const nvsr_scene* scene, const float* packed_coarse, const float* packed_fine, int64_t N, int Nc, int Nf,
                           const float* rays, int lindisp, int white_bkgd, const float* t_rand, const float* u,
                           const float* noise_coarse, const float* noise_fine, float* rgb_c, float* disp_c, float* acc_c,
                           float* rgb_f, float* disp_f, float* acc_f, float* workspace, int arithmetic, nvsr_stream_t stream);
/* One decoder for both passes (models.fine.type == 'use_same': train_nerf.py:353-355 makes model_fine the coarse model itself).  Same
 * arguments and results as nvsr_render_rays_arith with packed_fine == packed_coarse; the fine pass evaluates the decoder on the Nf importance
 * samples only and takes the Nc coarse samples' outputs from the coarse pass (same decoder, same planes, same points: the reference
 * recomputes them, train_utils.py:155-170), then composites the merged list.  Taken for frames without stratified jitter on the fused limb
 * passes (N >= nvsr_fused_min_rays(), t_rand == NULL, a limb arithmetic); otherwise -- and with NVSR_NO_SHARED_DECODER in the environment --
 * it forwards to nvsr_render_rays_arith.  workspace: nvsr_render_shared_workspace_floats floats. */
int64_t nvsr_render_shared_workspace_floats(int64_t N, int Nc, int Nf);
int nvsr_render_rays_shared_arith(const nvsr_scene* scene, const float* packed_decoder, int64_t N, int Nc, int Nf, const float* rays, int lindisp,
                                  int white_bkgd, const float* t_rand, const float* u, const float* noise_coarse, const float* noise_fine,
                                  float* rgb_c, float* disp_c, float* acc_c, float* rgb_f, float* disp_f, float* acc_f, float* workspace,
                                  int arithmetic, nvsr_stream_t stream);
int nvsr_render_pass_backward_gates_arith(const nvsr_scene* scene, const float* packed_decoder, const float* packed_bwd, int64_t N, int S,
                                          const float* rays, const float* z, const float* g_raw, const uint32_t* gates,
                                          float* const* grad_planes, float* view_ws, float* record, int arithmetic, nvsr_stream_t stream);
int nvsr_decoder_weight_grad_arith(int64_t N, int S, const float* record, float* grad_natural, int arithmetic, nvsr_stream_t stream);
int nvsr_conv3x3_arith(const float* in, int Cin, int H, int W, const float* packed, int Cout, int epilogue, const float* skip, float* out,
                       int arithmetic, int rows_per_tile, nvsr_stream_t stream);
int nvsr_conv3x3_dgrad_arith(const float* dy, int Cin, int H, int W, const float* packed_dgrad, int Cout, float* dx, int arithmetic,
                             int rows_per_tile, nvsr_stream_t stream);
int nvsr_conv3x3_wgrad_arith(const float* dy, const float* x, int Cin, int H, int W, int Cout, float scale, float* dw, float* workspace,
                             int arithmetic, nvsr_stream_t stream);
int nvsr_edsr_forward_batch_arith(const float* x, int B, int Cin, int H, int W, const float* packed, int Cout, int hid, int nblocks, int n_up,
                                  float* out, float* workspace, int arithmetic, nvsr_stream_t stream);
int nvsr_edsr_forward_train_arith(const float* x, int Cin, int H, int W, const float* packed, int Cout, int hid, int nblocks, int n_up,
                                  float* out, float* acts, int arithmetic, nvsr_stream_t stream);
int nvsr_edsr_backward_arith(const float* x, int Cin, int H, int W, const float* acts, const float* packed_dgrad, int Cout, int hid,
                             int nblocks, int n_up, const float* d_out, float* grad_natural, float* dx, float* workspace, int arithmetic,
                             nvsr_stream_t stream);
int nvsr_planes_sr_arith(const float* lr, int C, int R0, int R1, const float* packed, int hid, int nblocks, int n_up, int pad, int over,
                         const float* roi, const float* mean, const float* std_, float* out, float* workspace, int arithmetic,
                         nvsr_stream_t stream);
int nvsr_planes_sr_batch_arith(const float* const* lr, int B, int C, int R0, int R1, const float* packed, int hid, int nblocks, int n_up,
                               int pad, int over, const float* roi, const float* mean, const float* stdv, float* const* out,
                               float* workspace, int arithmetic, nvsr_stream_t stream);
int nvsr_planes_sr_train_arith(const float* lr, int C, int R0, int R1, const float* packed, int hid, int nblocks, int n_up, int pad, int over,
                               const float* roi, const float* mean, const float* stdv, float* out, float* workspace, float* keep,
                               int arithmetic, nvsr_stream_t stream);
int nvsr_planes_sr_backward_arith(int C, int R0, int R1, const float* keep, const float* packed_dgrad, int hid, int nblocks, int n_up,
                                  int pad, int over, const float* roi, const float* stdv, const float* d_out, float* grad_natural,
                                  float* d_lr, float* workspace, int arithmetic, nvsr_stream_t stream);

/* ---- SR training on the regions of interest of B planes at once (round 5) -------------------------------------------------------------------
 * Replaces: the B calls `self.SR_model((plane_name, roi))` a training iteration makes from `TwoDimPlanesModel.planes()` (models.py:270-284 ->
 * PlanesSR.forward :884-926), one per position plane of the scene, and their autograd backward (train_nerf.py:903 `loss.backward()`).  Every
 * plane has its own region of interest, i.e. its own crop size; the network is the same.  One launch per layer convolves all B crops (a
 * ragged batch: one crop of a 200^2 plane leaves the last of its 2-3 workgroup rounds a fifth full), one weight-gradient pass and one reduction
 * per layer accumulates all planes' contributions, one magnitude reduction per gradient tensor serves all planes (f16 limbs).
 *   lr / out / d_out / d_lr   HOST arrays of B device pointers (B <= 4); planes [C][R0][R1], outputs / output gradients [C][sf R0][sf R1]
 *   rois                      B x 4 HOST floats, [[ymin,xmin],[ymax,xmax]] in [-1,1] per plane, or NULL (full planes)
 *   keep                      one device buffer of nvsr_planes_sr_batch_keep_floats floats: forward -> backward
 *   align_corners, plane_interp   of the residual up-sampling (PlanesSR.align_corners / .plane_interp), ARGUMENTS of the call (the
 *                             one-plane entry points read the process-wide nvsr_set_sr_* setting)
 *   d_lr                      NULL, or per plane NULL (LR plane detached: models.py:272) or [C][R0][R1], += the plane's gradient
 * Values: those of B one-plane calls up to the order of the weight-gradient sums and (f16 limbs) the shared power-of-two gradient scale. */
/* nvsr_planes_sr_batch_arith (evaluation: B equally sized planes, ONE region of interest or none) with the residual's settings as arguments */
int nvsr_planes_sr_batch_ex(const float* const* lr, int B, int C, int R0, int R1, const float* packed, int hid, int nblocks, int n_up, int pad,
                            int over, const float* roi, const float* mean, const float* stdv, float* const* out, float* workspace,
                            int arithmetic, int align_corners, int plane_interp, nvsr_stream_t stream);
int64_t nvsr_planes_sr_batch_keep_floats(int B, int C, int R0, int R1, int hid, int nblocks, int n_up, int pad, const float* rois);
int64_t nvsr_planes_sr_batch_workspace_floats(int B, int C, int R0, int R1, int hid, int nblocks, int n_up, int pad, const float* rois);
int64_t nvsr_planes_sr_batch_backward_workspace_floats(int B, int C, int R0, int R1, int hid, int nblocks, int n_up, int pad, const float* rois);
int nvsr_planes_sr_train_batch_arith(const float* const* lr, int B, int C, int R0, int R1, const float* packed, int hid, int nblocks, int n_up,
                                     int pad, int over, const float* rois, const float* mean, const float* stdv, float* const* out,
                                     float* workspace, float* keep, int arithmetic, int align_corners, int plane_interp, nvsr_stream_t stream);
int nvsr_planes_sr_backward_batch_arith(int B, int C, int R0, int R1, const float* keep, const float* packed_dgrad, int hid, int nblocks,
                                        int n_up, int pad, int over, const float* rois, const float* stdv, const float* const* d_out,
                                        float* grad_natural, float* const* d_lr, float* workspace, int arithmetic, int align_corners,
                                        int plane_interp, nvsr_stream_t stream);
/* The same with PROGRESS MARKS for a data-parallel caller (the 173 MB EDSR gradient is all-reduced every refinement iteration;
 * the reference has no distributed code -- models.py:789-822 under torch.autograd is what the marks follow).  The backward walks the layers from
 * conv_output to conv_input; grad_natural is in state-dict order, so the gradients become final from the END of the blob.  mark_events[i] (a
 * hipEvent_t of the caller) is recorded on `stream` as soon as the weight gradients of layer mark_layers[i] (0 = conv_input ... 2 nblocks + 4 =
 * conv_output with two upscale stages) and of every layer behind it are final: the caller's collective stream waits for it and reduces that part
 * of the blob while the remaining layers are still being computed.  Exact f32 / one plane: every mark is recorded at the end of the pass. */
int nvsr_planes_sr_backward_batch_marks(int B, int C, int R0, int R1, const float* keep, const float* packed_dgrad, int hid, int nblocks,
                                        int n_up, int pad, int over, const float* rois, const float* stdv, const float* const* d_out,
                                        float* grad_natural, float* const* d_lr, float* workspace, int arithmetic, int align_corners,
                                        int plane_interp, int n_marks, const int32_t* mark_layers, void* const* mark_events, nvsr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NVSR_H */
