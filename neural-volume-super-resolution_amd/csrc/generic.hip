// TwoDimPlanesModel.forward for ANY decoder geometry (inference): the fallback behind the MFMA kernels that are compiled for the shipped
// one (3 + 1 planes x 48 channels, 'avg' / 'concat_pos', 4 + 4 layers x 128).
//
// Reference: models.py:381-421 with the layer lists of the constructor :169-195, is_skip_layer :203-207, combine_pos_planes :355-361,
// combine_all_planes :363-379; the shipped YAMLs list dec_channels 256, proj_combination sum / concat, 24-channel planes and skip layers as
// alternatives (config/TrainModels.yml:78,82,92).
//
// Forward: three plain kernels, activations through HBM (this is the compatibility path, not the hot one):
//   generic_inputs_kernel   one thread per (point, input column): normalise, project, 4 bilinear taps per plane, combine -> the density
//                           decoder's input Xd [P][Kd] and the colour decoder's input Xr [P][Kr]
//   generic_linear_kernel   Y[P][M] = act(b + W[M][K1 + K2] . [X1[P][K1] | X2[P][K2]])  (X2 = the skip connection's second operand) on
//                           v_mfma_f32_32x32x2_f32 (exact f32 products, k in order), one 32 x 32 tile per wave, operands staged through LDS
//   ray_points_kernel       x[N*S][6] = [ro + rd * z, viewdir]  (run_network's input, train_utils.py:15-64,111)
// Backward (the reference differentiates the same forward with torch.autograd): the forward of a chunk is recomputed with every layer's
// output kept, then per layer, top down,
//   generic_wgrad_kernel    dW[M][K] += dZ^T . [X1 | X2],  db += column sums of dZ   (dZ = dY where the layer's output is > 0; the points
//                           are the contraction index: MFMA operands straight from memory, partial tiles added with float atomics)
//   generic_dgrad_kernel    dX[P][columns of W] (+)= dZ . W   (the forward's tiling with W read transposed)
// and generic_inputs_backward_kernel scatters the gradients of the two decoder inputs through the combination rule and the four
// bilinear taps into the channel-last gradient planes (float atomics, like grid_sampler_2d_backward).
#include "decode_core.h"

namespace nvsr {

// the lambdas of the input kernels read the scene out of the kernel arguments: kept as calls (the inliner's choice once the bicubic taps made them
// large) they receive the 700-byte scene through scratch memory and lose the planes' address space
#define GEN_INL __attribute__((always_inline))

struct GenGeom {
    int C, Cv, hidden, nd, nr, skip, proj, view;      // proj: 0 sum 1 avg 2 concat; view: 0 sum 1 avg 2 mult 3 concat 4 concat_pos
    int Kd, Kr;
    int np;                                           // position planes (num_density_planes, models.py:140): 3 unless a scene_ext says otherwise
};

// the scene of the generic kernels: up to NVSR_MAX_POSITION_PLANES position planes with their projections, then the view-direction plane;
// grid_sample's align_corners; the optional jitter of the normalised sample positions (models.py:291-293)
struct SceneDevN {
    const float* plane[NVSR_MAX_POSITION_PLANES + 1];
    int ph[NVSR_MAX_POSITION_PLANES + 1], pw[NVSR_MAX_POSITION_PLANES + 1];
    float lo[5], range[5];
    float proj[NVSR_MAX_POSITION_PLANES * 6];
    int np, align, bicubic;
};
struct GradPlanesN { float* g[NVSR_MAX_POSITION_PLANES + 1]; };

static SceneDevN gen_scene(const nvsr_scene* s) {
    SceneDevN d = {};
    for (int i = 0; i < 4; ++i) { d.plane[i] = s->planes[i]; d.ph[i] = s->ph[i]; d.pw[i] = s->pw[i]; }
    for (int i = 0; i < 5; ++i) { d.lo[i] = s->lo[i]; d.range[i] = s->range[i]; }
    for (int i = 0; i < 18; ++i) d.proj[i] = (&s->proj[0][0])[i];
    d.np = 3; d.align = 1; d.bicubic = 0;
    return d;
}
static SceneDevN gen_scene(const nvsr_scene_ext* s) {
    SceneDevN d = {};
    for (int i = 0; i <= s->num_position_planes; ++i) { d.plane[i] = s->planes[i]; d.ph[i] = s->ph[i]; d.pw[i] = s->pw[i]; }
    for (int i = 0; i < 5; ++i) { d.lo[i] = s->lo[i]; d.range[i] = s->range[i]; }
    for (int i = 0; i < 6 * s->num_position_planes; ++i) d.proj[i] = (&s->proj[0][0])[i];
    d.np = s->num_position_planes; d.align = s->align_corners ? 1 : 0; d.bicubic = s->plane_interp == NVSR_PLANE_INTERP_BICUBIC ? 1 : 0;
    return d;
}

__host__ __device__ inline bool gen_skip_layer(int layer_num, int skip) { return skip > 0 && layer_num % skip == 0 && layer_num > 0; }

static int gen_resolve(const nvsr_decoder_geometry* g, GenGeom* o, int np = 3) {
    if (!g) return NVSR_ERR_NULL;
    if (np < 1 || np > NVSR_MAX_POSITION_PLANES) return NVSR_ERR_SHAPE;
    o->np = np;
    o->C = g->plane_channels; o->Cv = g->viewdir_channels; o->hidden = g->hidden; o->nd = g->density_layers; o->nr = g->rgb_layers;
    o->skip = g->skip_connect_every; o->proj = g->proj_combination; o->view = g->viewdir_combination;
    if (o->C < 1 || o->C > 1024 || o->Cv < 1 || o->Cv > 1024 || o->hidden < 1 || o->hidden > 4096 || o->nd < 1 || o->nd > 64 || o->nr < 1 || o->nr > 64)
        return NVSR_ERR_SHAPE;
    if (o->skip < 0 || o->proj < 0 || o->proj > 2 || o->view < 0 || o->view > 4) return NVSR_ERR_SHAPE;
    // the combinations the reference's own layer sizes admit (models.py:186-190 vs :363-379)
    const bool concat_like = o->proj == 2 || o->view == 4;
    if (o->view == 3 && o->proj != 2) return NVSR_ERR_SHAPE;              // 'concat' view needs concatenated position features
    if (o->view <= 2 && (o->proj == 2 || o->Cv != o->C)) return NVSR_ERR_SHAPE;   // sum / avg / mult combine equal-sized vectors
    o->Kd = o->C * (o->proj == 2 ? np : 1);
    o->Kr = o->Cv + (concat_like ? np * o->C : 0);
    if (o->view <= 2) o->Kr = o->C;
    return NVSR_OK;
}

// in_features of decoder layer l (models.py:173-195)
static int gen_layer_in(const GenGeom& g, bool rgb, int l) {
    const int k0 = rgb ? g.Kr : g.Kd;
    if (l == 0) return k0;
    return g.hidden + (gen_skip_layer(l - 1, g.skip) ? k0 : 0);
}

struct GenTaps { int o[4]; float w[4]; };
__device__ __forceinline__ GenTaps gen_taps(int H, int W, int Cc, float gx, float gy, int align) {
    // grid_sample(padding_mode='border'): unnormalise, clip, floor; a clamped neighbour carries weight 0.  align_corners=True maps -1 / +1 to
    // the centres of the corner texels, (g + 1) (size - 1) / 2; False to their outer edges, (g + 1) size / 2 - 0.5 (the CPU kernel's
    // ComputeLocation: one product with the pre-divided scale, then the shift)
    const float mx = (float)(W - 1), my = (float)(H - 1);
    float x, y;
    if (align) { x = (gx + 1.0f) * (mx / 2.0f); y = (gy + 1.0f) * (my / 2.0f); }
    else { x = __fsub_rn(__fmul_rn(gx + 1.0f, (float)W / 2.0f), 0.5f); y = __fsub_rn(__fmul_rn(gy + 1.0f, (float)H / 2.0f), 0.5f); }
    x = fminf(mx, fmaxf(x, 0.0f));
    y = fminf(my, fmaxf(y, 0.0f));
    const float xw = floorf(x), yn = floorf(y);
    const float w = x - xw, e = 1.0f - w, n = y - yn, s = 1.0f - n;
    const int ix = (int)xw, iy = (int)yn, ix1 = min(ix + 1, W - 1), iy1 = min(iy + 1, H - 1);
    GenTaps t;
    t.w[0] = s * e; t.w[1] = s * w; t.w[2] = n * e; t.w[3] = n * w;
    t.o[0] = (iy * W + ix) * Cc; t.o[1] = (iy * W + ix1) * Cc; t.o[2] = (iy1 * W + ix) * Cc; t.o[3] = (iy1 * W + ix1) * Cc;
    return t;
}
__device__ __forceinline__ float gen_blend(const float* __restrict__ plane, const GenTaps& t, int c) {
    return fmaf(plane[t.o[3] + c], t.w[3], fmaf(plane[t.o[2] + c], t.w[2], fmaf(plane[t.o[1] + c], t.w[1], plane[t.o[0] + c] * t.w[0])));
}

// grid_sample(mode='bicubic', padding_mode='border') (ATen GridSampler.h / GridSamplerKernel.cpp): the coordinate is unnormalised but NOT
// clipped; the 4 x 4 neighbourhood starts at floor - 1 and every TAP's index is clipped to the plane (border padding); the weights are the cubic
// convolution kernel with A = -0.75 (UpSample.h: get_cubic_upsample_coefficients), rows first: out = sum_i cy[i] (sum_j cx[j] v[i][j]).
struct CubicTaps { int ix[4], iy[4]; float cx[4], cy[4]; };
__device__ __forceinline__ void cubic_coefficients(float t, float (&c)[4]) {
    const float A = -0.75f;
    const float x0 = t + 1.0f, x1 = t, x2 = 1.0f - t, x3 = x2 + 1.0f;
    c[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
    c[1] = ((A + 2.0f) * x1 - (A + 3.0f)) * x1 * x1 + 1.0f;
    c[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
    c[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
}
__device__ __forceinline__ CubicTaps gen_cubic_taps(int H, int W, float gx, float gy, int align) {
    float x, y;
    if (align) { x = (gx + 1.0f) * ((float)(W - 1) / 2.0f); y = (gy + 1.0f) * ((float)(H - 1) / 2.0f); }
    else { x = __fsub_rn(__fmul_rn(gx + 1.0f, (float)W / 2.0f), 0.5f); y = __fsub_rn(__fmul_rn(gy + 1.0f, (float)H / 2.0f), 0.5f); }
    const float fx = floorf(x), fy = floorf(y);
    CubicTaps t;
    cubic_coefficients(x - fx, t.cx);
    cubic_coefficients(y - fy, t.cy);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // (clip_coordinates on the float index, then the integer cast: the same texel as clamping the integer; NaN coordinates clamp to 0)
        const float xi = fminf((float)(W - 1), fmaxf(fx - 1.0f + (float)k, 0.0f)), yi = fminf((float)(H - 1), fmaxf(fy - 1.0f + (float)k, 0.0f));
        t.ix[k] = (int)xi; t.iy[k] = (int)yi;
    }
    return t;
}
__device__ __forceinline__ float gen_cubic_blend(const float* __restrict__ plane, int W, int Cc, const CubicTaps& t, int c) {
    float out = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float* row = plane + (long)t.iy[i] * W * Cc + c;
        const float r = row[t.ix[0] * Cc] * t.cx[0] + row[t.ix[1] * Cc] * t.cx[1] + row[t.ix[2] * Cc] * t.cx[2] + row[t.ix[3] * Cc] * t.cx[3];
        out += r * t.cy[i];
    }
    return out;
}
// one feature of one plane in either interpolation
__device__ __forceinline__ float gen_sample(const SceneDevN& sc, int d, int Cc, float gx, float gy, int c) {
    if (sc.bicubic) return gen_cubic_blend(sc.plane[d], sc.pw[d], Cc, gen_cubic_taps(sc.ph[d], sc.pw[d], gx, gy, sc.align), c);
    return gen_blend(sc.plane[d], gen_taps(sc.ph[d], sc.pw[d], Cc, gx, gy, sc.align), c);
}
// ... and its transpose: v times the tap weights, added into the gradient plane
__device__ __forceinline__ void gen_scatter(const SceneDevN& sc, int d, int Cc, float gx, float gy, int c, float v, float* __restrict__ g) {
    if (!g || v == 0.0f) return;
    if (sc.bicubic) {
        const CubicTaps t = gen_cubic_taps(sc.ph[d], sc.pw[d], gx, gy, sc.align);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float w = t.cx[j] * t.cy[i];
                if (w != 0.0f) unsafeAtomicAdd(g + ((long)t.iy[i] * sc.pw[d] + t.ix[j]) * Cc + c, v * w);
            }
        return;
    }
    const GenTaps t = gen_taps(sc.ph[d], sc.pw[d], Cc, gx, gy, sc.align);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (t.w[i] != 0.0f) unsafeAtomicAdd(g + t.o[i] + c, v * t.w[i]);
}

// normalised sample position (models.py:261-268) + the optional jitter the reference adds to it in training (:291-293)
__device__ __forceinline__ void gen_position(const SceneDevN& sc, const float* q, const float* noise, long p, float& n0, float& n1, float& n2) {
    n0 = norm_coord(q[0], sc.lo[0], sc.range[0]); n1 = norm_coord(q[1], sc.lo[1], sc.range[1]); n2 = norm_coord(q[2], sc.lo[2], sc.range[2]);
    if (noise) { n0 = __fadd_rn(n0, noise[3 * p]); n1 = __fadd_rn(n1, noise[3 * p + 1]); n2 = __fadd_rn(n2, noise[3 * p + 2]); }
}

__global__ void generic_inputs_kernel(SceneDevN sc, GenGeom g, long P, const float* __restrict__ x, const float* __restrict__ noise,
                                      float* __restrict__ Xd, float* __restrict__ Xr) {
    const int K = g.Kd + g.Kr;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P * K) return;
    const long p = idx / K;
    const int j = (int)(idx - p * K);
    const float* q = x + p * 6;
    // (cart2az_el nerf_helpers.py:492-496)
    float n0, n1, n2;
    gen_position(sc, q, noise, p, n0, n1, n2);
    auto pos_feat = [&](int d, int c) GEN_INL {
        const float* M = sc.proj + 6 * d;
        return gen_sample(sc, d, g.C, n0 * M[0] + n1 * M[2] + n2 * M[4], n0 * M[1] + n1 * M[3] + n2 * M[5], c);
    };
    auto view_feat = [&](int c) GEN_INL {
        const float az = atan2f(q[4], q[3]);
        const float el = atan2f(q[5], sqrtf(__fadd_rn(__fmul_rn(q[3], q[3]), __fmul_rn(q[4], q[4]))));
        return gen_sample(sc, g.np, g.Cv, norm_coord(az, sc.lo[3], sc.range[3]), norm_coord(el, sc.lo[4], sc.range[4]), c);
    };
    auto combined_pos = [&](int c) GEN_INL {                       // combine_pos_planes, column c of its result (stack(...).sum(0) / .mean(0): plane by plane)
        if (g.proj == 2) return pos_feat(c / g.C, c % g.C);
        float s = pos_feat(0, c);
        for (int d = 1; d < g.np; ++d) s = __fadd_rn(s, pos_feat(d, c));
        return g.proj == 1 ? __fdiv_rn(s, (float)g.np) : s;
    };
    if (j < g.Kd) { Xd[p * g.Kd + j] = combined_pos(j); return; }
    const int c = j - g.Kd;
    float v;
    if (g.view == 4) v = c < g.np * g.C ? pos_feat(c / g.C, c % g.C) : view_feat(c - g.np * g.C);       // cat(pos_planes + [viewdir])
    else if (g.view == 3) v = c < g.Kd ? combined_pos(c) : view_feat(c - g.Kd);                        // cat([combined, viewdir])
    else {
        const float pp = combined_pos(c), vv = view_feat(c);
        v = g.view == 0 ? __fadd_rn(pp, vv) : g.view == 1 ? __fdiv_rn(__fadd_rn(pp, vv), 2.0f) : __fmul_rn(pp, __fadd_rn(1.0f, vv));
    }
    Xr[p * g.Kr + c] = v;
}

// Y[p][yoff + m] (row stride ldy) = act(b[m] + sum_k W[m][k] X[p][k]),  X = [X1 | X2] with K1 + K2 columns, W row-major [M][K1 + K2]
constexpr int GL_PTS = 128, GL_OUT = 32, GL_K = 32, GL_LD = GL_K + 1;
template <bool RELU>
__global__ __launch_bounds__(256) void generic_linear_kernel(long P, int M, int K1, int K2, const float* __restrict__ X1, const float* __restrict__ X2,
                                                            const float* __restrict__ Wt, const float* __restrict__ b, float* __restrict__ Y, int ldy,
                                                            int yoff) {
    __shared__ float xs[GL_PTS * GL_LD], ws[GL_OUT * GL_LD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 31, h = lane >> 5;
    const long p0 = (long)blockIdx.x * GL_PTS;
    const int o0 = blockIdx.y * GL_OUT, K = K1 + K2;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    for (int k0 = 0; k0 < K; k0 += GL_K) {
        __syncthreads();
        for (int e = tid; e < GL_PTS * GL_K; e += 256) {
            const int pt = e / GL_K, k = e % GL_K;
            const long p = p0 + pt;
            const int kk = k0 + k;
            float v = 0.0f;
            if (p < P && kk < K) v = kk < K1 ? X1[p * K1 + kk] : X2[p * K2 + (kk - K1)];
            xs[pt * GL_LD + k] = v;
        }
        for (int e = tid; e < GL_OUT * GL_K; e += 256) {
            const int o = e / GL_K, k = e % GL_K;
            ws[o * GL_LD + k] = (o0 + o < M && k0 + k < K) ? Wt[(long)(o0 + o) * K + k0 + k] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < GL_K / 2; ++s)      // A: lane (m, h) = W[o0 + m][2s + h];  B: lane (n, h) = X[point n of this wave][2s + h]
            acc = mfma32(ws[n * GL_LD + 2 * s + h], xs[(wave * 32 + n) * GL_LD + 2 * s + h], acc);
    }
    const long p = p0 + wave * 32 + n;
    if (p >= P) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = o0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < M) {
            float v = __fadd_rn(acc[r], b[m]);
            if (RELU) v = fmaxf(v, 0.0f);
            Y[p * ldy + yoff + m] = v;
        }
    }
}

__global__ void ray_points_kernel(long N, int S, const float* __restrict__ rays, const float* __restrict__ z, float* __restrict__ x) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * S) return;
    const float* r = rays + (i / S) * 11;
    const float zc = z[i];
    float* o = x + i * 6;
    o[0] = __fadd_rn(r[0], __fmul_rn(r[3], zc));
    o[1] = __fadd_rn(r[1], __fmul_rn(r[4], zc));
    o[2] = __fadd_rn(r[2], __fmul_rn(r[5], zc));
    o[3] = r[8]; o[4] = r[9]; o[5] = r[10];
}

// dX[p][xoff + k] (row stride ldx; ACCUM: +=) = sum_m dZ[p][m] W[m][koff + k], k < Kn;  dZ[p][m] = dY[p * ldd + doff + m], zeroed where
// Hmask[p][m] <= 0 (Hmask = the layer's post-ReLU output, NULL for the linear heads);  W row-major [M][K]
template <bool ACCUM>
__global__ __launch_bounds__(256) void generic_dgrad_kernel(long P, int M, int K, int koff, int Kn, const float* __restrict__ dY, int ldd, int doff,
                                                           const float* __restrict__ Hmask, const float* __restrict__ Wt, float* __restrict__ dX,
                                                           int ldx, int xoff) {
    __shared__ float xs[GL_PTS * GL_LD], ws[GL_OUT * GL_LD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 31, h = lane >> 5;
    const long p0 = (long)blockIdx.x * GL_PTS;
    const int o0 = blockIdx.y * GL_OUT;                    // first output column (k) of this tile
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    for (int m0 = 0; m0 < M; m0 += GL_K) {
        __syncthreads();
        for (int e = tid; e < GL_PTS * GL_K; e += 256) {
            const int pt = e / GL_K, mm = m0 + e % GL_K;
            const long p = p0 + pt;
            float v = 0.0f;
            if (p < P && mm < M) {
                v = dY[p * ldd + doff + mm];
                if (Hmask && !(Hmask[p * M + mm] > 0.0f)) v = 0.0f;
            }
            xs[pt * GL_LD + e % GL_K] = v;
        }
        for (int e = tid; e < GL_OUT * GL_K; e += 256) {
            const int o = e % GL_OUT, mm = m0 + e / GL_OUT;      // consecutive threads -> consecutive k of one row of W
            ws[o * GL_LD + e / GL_OUT] = (o0 + o < Kn && mm < M) ? Wt[(long)mm * K + koff + o0 + o] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < GL_K / 2; ++s)      // A: lane (k, h) = W[m0 + 2s + h][k];  B: lane (n, h) = dZ[point n of this wave][m0 + 2s + h]
            acc = mfma32(ws[n * GL_LD + 2 * s + h], xs[(wave * 32 + n) * GL_LD + 2 * s + h], acc);
    }
    const long p = p0 + wave * 32 + n;
    if (p >= P) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k = o0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (k < Kn) {
            float* q = dX + p * ldx + xoff + k;
            *q = ACCUM ? __fadd_rn(*q, acc[r]) : acc[r];
        }
    }
}

// dW[m][k] += sum_p dZ[p][m] Xc[p][k] for k < K = K1 + K2 (Xc = [X1 | X2]),  db[m] += sum_p dZ[p][m] (the column k == K of the tile grid,
// whose operand is 1).  Workgroup = one 32 x 32 tile of (m, k) x a slab of GW_PTS points, a quarter of the slab per wave.
constexpr int GW_PTS = 2048;
__global__ __launch_bounds__(256) void generic_wgrad_kernel(long P, int M, int K1, int K2, const float* __restrict__ dY, int ldd, int doff,
                                                           const float* __restrict__ Hmask, const float* __restrict__ X1, const float* __restrict__ X2,
                                                           float* __restrict__ dW, float* __restrict__ db) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * 32, k0 = blockIdx.y * 32, K = K1 + K2;
    const long pa = (long)blockIdx.z * GW_PTS + wave * (GW_PTS / 4), pb = min(pa + GW_PTS / 4, P);
    const int m = m0 + n, k = k0 + n;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    for (long ps = pa; ps < pb; ps += 2) {                   // (uniform trip count: every lane takes part in every MFMA)
        const long p = ps + h;
        float a = 0.0f, b = 0.0f;
        if (p < pb) {
            if (m < M) {
                a = dY[p * ldd + doff + m];
                if (Hmask && !(Hmask[p * M + m] > 0.0f)) a = 0.0f;
            }
            if (k < K1) b = X1[p * K1 + k];
            else if (k < K) b = X2[p * K2 + (k - K1)];
            else if (k == K) b = 1.0f;
        }
        acc = mfma32(a, b, acc);                              // A: lane (m, h) = dZ[p][m];  B: lane (k, h) = Xc[p][k]
    }
    if (k > K) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int mr = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (mr < M) unsafeAtomicAdd(k < K ? dW + (long)mr * K + k : db + mr, acc[r]);
    }
}

// gradients of the decoder inputs -> gradient planes: the transpose of generic_inputs_kernel, one thread per (point, input column)
__global__ void generic_inputs_backward_kernel(SceneDevN sc, GenGeom g, long P, const float* __restrict__ x, const float* __restrict__ noise,
                                               const float* __restrict__ dXd, const float* __restrict__ dXr, GradPlanesN gp) {
    const int K = g.Kd + g.Kr;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P * K) return;
    const long p = idx / K;
    const int j = (int)(idx - p * K);
    const float* q = x + p * 6;
    float n0, n1, n2;
    gen_position(sc, q, noise, p, n0, n1, n2);
    auto pos_grid = [&](int d, float& gx, float& gy) GEN_INL {
        const float* M = sc.proj + 6 * d;
        gx = n0 * M[0] + n1 * M[2] + n2 * M[4]; gy = n0 * M[1] + n1 * M[3] + n2 * M[5];
    };
    auto view_grid = [&](float& gx, float& gy) GEN_INL {
        const float az = atan2f(q[4], q[3]);
        const float el = atan2f(q[5], sqrtf(__fadd_rn(__fmul_rn(q[3], q[3]), __fmul_rn(q[4], q[4]))));
        gx = norm_coord(az, sc.lo[3], sc.range[3]); gy = norm_coord(el, sc.lo[4], sc.range[4]);
    };
    auto scatter_plane = [&](int d, int c, float v) GEN_INL {         // position plane d, channel c
        float gx, gy;
        pos_grid(d, gx, gy);
        gen_scatter(sc, d, g.C, gx, gy, c, v, gp.g[d]);
    };
    auto scatter_view = [&](int c, float v) GEN_INL {
        float gx, gy;
        view_grid(gx, gy);
        gen_scatter(sc, g.np, g.Cv, gx, gy, c, v, gp.g[g.np]);
    };
    auto pos_value = [&](int d, int c) GEN_INL { float gx, gy; pos_grid(d, gx, gy); return gen_sample(sc, d, g.C, gx, gy, c); };
    auto scatter_pos = [&](int c, float v) GEN_INL {                 // combine_pos_planes transposed, column c of its result
        if (g.proj == 2) { scatter_plane(c / g.C, c % g.C, v); return; }
        const float f = g.proj == 1 ? v / (float)g.np : v;
        for (int d = 0; d < g.np; ++d) scatter_plane(d, c, f);
    };
    auto combined_pos = [&](int c) GEN_INL {
        if (g.proj == 2) return pos_value(c / g.C, c % g.C);
        float s = pos_value(0, c);
        for (int d = 1; d < g.np; ++d) s = __fadd_rn(s, pos_value(d, c));
        return g.proj == 1 ? __fdiv_rn(s, (float)g.np) : s;
    };
    if (j < g.Kd) { scatter_pos(j, dXd[p * g.Kd + j]); return; }
    const int c = j - g.Kd;
    const float v = dXr[p * g.Kr + c];
    if (g.view == 4) {
        if (c < g.np * g.C) scatter_plane(c / g.C, c % g.C, v);
        else scatter_view(c - g.np * g.C, v);
    } else if (g.view == 3) {
        if (c < g.Kd) scatter_pos(c, v);
        else scatter_view(c - g.Kd, v);
    } else if (g.view == 0) {
        scatter_pos(c, v);
        scatter_view(c, v);
    } else if (g.view == 1) {
        scatter_pos(c, v * 0.5f);
        scatter_view(c, v * 0.5f);
    } else {                                                 // mult: pp * (1 + vv)
        float gx, gy;
        view_grid(gx, gy);
        const float pp = combined_pos(c), vv = gen_sample(sc, g.np, g.Cv, gx, gy, c);
        scatter_pos(c, v * __fadd_rn(1.0f, vv));
        scatter_view(c, v * pp);
    }
}

constexpr long GEN_CHUNK = 1L << 20;       // points per pass of the layer stack (bounds the activation workspace)
constexpr long GEN_BWD_CHUNK = 1L << 17;   // the backward keeps every layer's output of a chunk

}  // namespace nvsr

using namespace nvsr;

extern "C" {

static int64_t gen_natural_floats(const nvsr_decoder_geometry* geom, int np) {
    GenGeom g;
    if (gen_resolve(geom, &g, np)) return -1;
    int64_t n = 0;
    for (int l = 0; l < g.nd; ++l) n += (int64_t)g.hidden * gen_layer_in(g, false, l) + g.hidden;
    n += g.hidden + 1;
    for (int l = 0; l < g.nr; ++l) n += (int64_t)g.hidden * gen_layer_in(g, true, l) + g.hidden;
    n += 3 * g.hidden + 3;
    return n;
}
int64_t nvsr_generic_decoder_natural_floats(const nvsr_decoder_geometry* geom) { return gen_natural_floats(geom, 3); }
int64_t nvsr_generic_decoder_natural_floats_ext(const nvsr_decoder_geometry* geom, int num_position_planes) { return gen_natural_floats(geom, num_position_planes); }

static int64_t gen_workspace_floats(const nvsr_decoder_geometry* geom, int np, int64_t P) {
    GenGeom g;
    if (gen_resolve(geom, &g, np) || P < 0) return -1;
    const int64_t c = P < GEN_CHUNK ? P : GEN_CHUNK;
    return c * (int64_t)(g.Kd + g.Kr + 2 * g.hidden);
}
int64_t nvsr_generic_decode_workspace_floats(const nvsr_decoder_geometry* geom, int64_t P) { return gen_workspace_floats(geom, 3, P); }
int64_t nvsr_generic_decode_workspace_floats_ext(const nvsr_decoder_geometry* geom, int num_position_planes, int64_t P) {
    return gen_workspace_floats(geom, num_position_planes, P);
}

static int gen_check_scene(const SceneDevN& sc, const GenGeom& g) {
    for (int d = 0; d <= g.np; ++d) {
        if (!sc.plane[d]) return NVSR_ERR_NULL;
        const int64_t cc = d < g.np ? g.C : g.Cv;
        if (sc.ph[d] < 1 || sc.pw[d] < 1 || (int64_t)sc.ph[d] * sc.pw[d] * cc >= (int64_t)1 << 31) return NVSR_ERR_SHAPE;
    }
    return NVSR_OK;
}
static int gen_check_ext(const nvsr_scene_ext* scene) {
    if (!scene) return NVSR_ERR_NULL;
    if (scene->num_position_planes < 1 || scene->num_position_planes > NVSR_MAX_POSITION_PLANES) return NVSR_ERR_SHAPE;
    return NVSR_OK;
}

static int gen_decode(const SceneDevN& sc, const GenGeom& g, const float* natural, int64_t P, const float* x, const float* coord_noise, float* out,
                      float* workspace, hipStream_t stream) {
    if (!natural || !x || !out || !workspace) return NVSR_ERR_NULL;
    if (int e = gen_check_scene(sc, g)) return e;
    if (P < 0) return NVSR_ERR_SHAPE;
    for (int64_t a = 0; a < P; a += GEN_CHUNK) {
        const long n = (long)((P - a) < GEN_CHUNK ? (P - a) : GEN_CHUNK);
        float* Xd = workspace;
        float* Xr = Xd + n * g.Kd;
        float* H[2] = {Xr + n * g.Kr, Xr + n * g.Kr + n * g.hidden};
        const long threads = n * (g.Kd + g.Kr);
        hipLaunchKernelGGL(generic_inputs_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, sc, g, n, x + a * 6,
                           coord_noise ? coord_noise + a * 3 : nullptr, Xd, Xr);
        if (int e = NVSR_CHECK_LAUNCH()) return e;
        const float* w = natural;
        const dim3 pts((unsigned)((n + GL_PTS - 1) / GL_PTS));
        for (int dec = 0; dec < 2; ++dec) {
            const bool rgb = dec == 1;
            const float* in0 = rgb ? Xr : Xd;
            const int k0 = rgb ? g.Kr : g.Kd, nl = rgb ? g.nr : g.nd;
            const float* cur = in0;
            int cur_k = k0;
            for (int l = 0; l < nl; ++l) {
                const bool skip = l > 0 && gen_skip_layer(l - 1, g.skip);       // x = cat(x, input) in front of this layer (models.py:397-399)
                const int K1 = cur_k, K2 = skip ? k0 : 0;
                float* y = H[l & 1];
                hipLaunchKernelGGL(generic_linear_kernel<true>, dim3(pts.x, (g.hidden + GL_OUT - 1) / GL_OUT), dim3(256), 0, stream, n, g.hidden, K1, K2,
                                   cur, in0, w, w + (long)g.hidden * (K1 + K2), y, g.hidden, 0);
                if (int e = NVSR_CHECK_LAUNCH()) return e;
                w += (long)g.hidden * (K1 + K2) + g.hidden;
                cur = y; cur_k = g.hidden;
            }
            const int M = rgb ? 3 : 1;                                          // fc_rgb -> out[:, 0:3], fc_alpha -> out[:, 3]
            hipLaunchKernelGGL(generic_linear_kernel<false>, dim3(pts.x, 1), dim3(256), 0, stream, n, M, g.hidden, 0, cur, cur, w, w + (long)M * g.hidden,
                               out + a * 4, 4, rgb ? 0 : 3);
            if (int e = NVSR_CHECK_LAUNCH()) return e;
            w += (long)M * g.hidden + M;
        }
    }
    return NVSR_OK;
}
int nvsr_generic_decode(const nvsr_scene* scene, const nvsr_decoder_geometry* geom, const float* natural, int64_t P, const float* x, float* out,
                        float* workspace, nvsr_stream_t stream) {
    GenGeom g;
    if (int e = gen_resolve(geom, &g)) return e;
    if (!scene) return NVSR_ERR_NULL;
    return gen_decode(gen_scene(scene), g, natural, P, x, nullptr, out, workspace, (hipStream_t)stream);
}
int nvsr_generic_decode_ext(const nvsr_scene_ext* scene, const nvsr_decoder_geometry* geom, const float* natural, int64_t P, const float* x,
                            const float* coord_noise, float* out, float* workspace, nvsr_stream_t stream) {
    if (int e = gen_check_ext(scene)) return e;
    GenGeom g;
    if (int e = gen_resolve(geom, &g, scene->num_position_planes)) return e;
    return gen_decode(gen_scene(scene), g, natural, P, x, coord_noise, out, workspace, (hipStream_t)stream);
}

static int64_t gen_bwd_floats_per_point(const GenGeom& g) { return 2 * (int64_t)(g.Kd + g.Kr) + (int64_t)(g.nd + g.nr + 2) * g.hidden; }

static int64_t gen_bwd_workspace_floats(const nvsr_decoder_geometry* geom, int np, int64_t P) {
    GenGeom g;
    if (gen_resolve(geom, &g, np) || P < 0) return -1;
    return (P < GEN_BWD_CHUNK ? P : GEN_BWD_CHUNK) * gen_bwd_floats_per_point(g);
}
int64_t nvsr_generic_decode_backward_workspace_floats(const nvsr_decoder_geometry* geom, int64_t P) { return gen_bwd_workspace_floats(geom, 3, P); }
int64_t nvsr_generic_decode_backward_workspace_floats_ext(const nvsr_decoder_geometry* geom, int num_position_planes, int64_t P) {
    return gen_bwd_workspace_floats(geom, num_position_planes, P);
}

static int gen_decode_backward(const SceneDevN& sc, const GenGeom& g, const float* natural, int64_t P, const float* x, const float* coord_noise,
                               const float* d_out, float* d_natural, const GradPlanesN& gp, float* workspace, hipStream_t stream) {
    if (!natural || !x || !d_out || !workspace) return NVSR_ERR_NULL;
    if (int e = gen_check_scene(sc, g)) return e;
    if (P < 0 || g.nd + g.nr > 128) return NVSR_ERR_SHAPE;
    bool want_planes = false;
    for (int d = 0; d <= g.np; ++d) want_planes = want_planes || gp.g[d] != nullptr;
    // offsets of the layers in the natural blob (state-dict order: density layers, fc_alpha, rgb layers, fc_rgb; weight then bias)
    long woff[2][65];
    {
        long o = 0;
        for (int dec = 0; dec < 2; ++dec) {
            const int nl = dec ? g.nr : g.nd;
            for (int l = 0; l < nl; ++l) { woff[dec][l] = o; o += (long)g.hidden * gen_layer_in(g, dec, l) + g.hidden; }
            woff[dec][nl] = o;
            o += (long)(dec ? 3 : 1) * g.hidden + (dec ? 3 : 1);
        }
    }
    for (int64_t a = 0; a < P; a += GEN_BWD_CHUNK) {
        const long n = (long)((P - a) < GEN_BWD_CHUNK ? (P - a) : GEN_BWD_CHUNK);
        float* Xin[2] = {workspace, workspace + n * g.Kd};
        float* Hs[2] = {Xin[1] + n * g.Kr, Xin[1] + n * g.Kr + (long)g.nd * n * g.hidden};          // [layer][n][hidden] per decoder
        float* dXin[2] = {Hs[1] + (long)g.nr * n * g.hidden, Hs[1] + (long)g.nr * n * g.hidden + n * g.Kd};
        float* dH[2] = {dXin[1] + n * g.Kr, dXin[1] + n * g.Kr + n * g.hidden};
        const float* xa = x + a * 6;
        const float* da = d_out + a * 4;
        const long threads = n * (g.Kd + g.Kr);
        const float* na = coord_noise ? coord_noise + a * 3 : nullptr;
        hipLaunchKernelGGL(generic_inputs_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, sc, g, n, xa, na, Xin[0], Xin[1]);
        if (int e = NVSR_CHECK_LAUNCH()) return e;
        const dim3 pts((unsigned)((n + GL_PTS - 1) / GL_PTS));
        const unsigned slabs = (unsigned)((n + GW_PTS - 1) / GW_PTS);
        if (hipMemsetAsync(dXin[0], 0, sizeof(float) * n * (g.Kd + g.Kr), stream) != hipSuccess) return NVSR_ERR_LAUNCH;
        for (int dec = 0; dec < 2; ++dec) {
            const float* in0 = Xin[dec];
            const int k0 = dec ? g.Kr : g.Kd, nl = dec ? g.nr : g.nd, Mh = dec ? 3 : 1, hoff = dec ? 0 : 3;
            // forward with every layer's output kept
            const float* cur = in0;
            int cur_k = k0;
            for (int l = 0; l < nl; ++l) {
                const bool skip = l > 0 && gen_skip_layer(l - 1, g.skip);
                const int K1 = cur_k, K2 = skip ? k0 : 0;
                const float* w = natural + woff[dec][l];
                float* y = Hs[dec] + (long)l * n * g.hidden;
                hipLaunchKernelGGL(generic_linear_kernel<true>, dim3(pts.x, (g.hidden + GL_OUT - 1) / GL_OUT), dim3(256), 0, stream, n, g.hidden, K1, K2,
                                   cur, in0, w, w + (long)g.hidden * (K1 + K2), y, g.hidden, 0);
                if (int e = NVSR_CHECK_LAUNCH()) return e;
                cur = y; cur_k = g.hidden;
            }
            // head: d_out[:, hoff : hoff + Mh] -> dH[0]; its weight gradient from the last layer's output
            const float* wh = natural + woff[dec][nl];
            if (d_natural) {
                float* gw = d_natural + woff[dec][nl];
                hipLaunchKernelGGL(generic_wgrad_kernel, dim3(1, (g.hidden + 1 + 31) / 32, slabs), dim3(256), 0, stream, n, Mh, g.hidden, 0, da, 4, hoff,
                                   (const float*)nullptr, cur, cur, gw, gw + (long)Mh * g.hidden);
                if (int e = NVSR_CHECK_LAUNCH()) return e;
            }
            hipLaunchKernelGGL(generic_dgrad_kernel<false>, dim3(pts.x, (g.hidden + GL_OUT - 1) / GL_OUT), dim3(256), 0, stream, n, Mh, g.hidden, 0, g.hidden,
                               da, 4, hoff, (const float*)nullptr, wh, dH[0], g.hidden, 0);
            if (int e = NVSR_CHECK_LAUNCH()) return e;
            int cb = 0;
            for (int l = nl - 1; l >= 0; --l) {
                const bool skip = l > 0 && gen_skip_layer(l - 1, g.skip);
                const int K1 = l == 0 ? k0 : g.hidden, K2 = skip ? k0 : 0, K = K1 + K2;
                const float* X1 = l == 0 ? in0 : Hs[dec] + (long)(l - 1) * n * g.hidden;
                const float* Hl = Hs[dec] + (long)l * n * g.hidden;
                const float* w = natural + woff[dec][l];
                if (d_natural) {
                    float* gw = d_natural + woff[dec][l];
                    hipLaunchKernelGGL(generic_wgrad_kernel, dim3((g.hidden + 31) / 32, (K + 1 + 31) / 32, slabs), dim3(256), 0, stream, n, g.hidden, K1, K2,
                                       dH[cb], g.hidden, 0, Hl, X1, in0, gw, gw + (long)g.hidden * K);
                    if (int e = NVSR_CHECK_LAUNCH()) return e;
                }
                if (l > 0) {
                    hipLaunchKernelGGL(generic_dgrad_kernel<false>, dim3(pts.x, (K1 + GL_OUT - 1) / GL_OUT), dim3(256), 0, stream, n, g.hidden, K, 0, K1,
                                       dH[cb], g.hidden, 0, Hl, w, dH[cb ^ 1], g.hidden, 0);
                    if (int e = NVSR_CHECK_LAUNCH()) return e;
                    if (skip && want_planes) {
                        hipLaunchKernelGGL(generic_dgrad_kernel<true>, dim3(pts.x, (K2 + GL_OUT - 1) / GL_OUT), dim3(256), 0, stream, n, g.hidden, K, K1, K2,
                                           dH[cb], g.hidden, 0, Hl, w, dXin[dec], k0, 0);
                        if (int e = NVSR_CHECK_LAUNCH()) return e;
                    }
                } else if (want_planes) {
                    hipLaunchKernelGGL(generic_dgrad_kernel<true>, dim3(pts.x, (K1 + GL_OUT - 1) / GL_OUT), dim3(256), 0, stream, n, g.hidden, K, 0, K1,
                                       dH[cb], g.hidden, 0, Hl, w, dXin[dec], k0, 0);
                    if (int e = NVSR_CHECK_LAUNCH()) return e;
                }
                cb ^= 1;
            }
        }
        if (want_planes) {
            hipLaunchKernelGGL(generic_inputs_backward_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, sc, g, n, xa, na, dXin[0],
                               dXin[1], gp);
            if (int e = NVSR_CHECK_LAUNCH()) return e;
        }
    }
    return NVSR_OK;
}
int nvsr_generic_decode_backward(const nvsr_scene* scene, const nvsr_decoder_geometry* geom, const float* natural, int64_t P, const float* x,
                                 const float* d_out, float* d_natural, float* d_plane0, float* d_plane1, float* d_plane2, float* d_plane3,
                                 float* workspace, nvsr_stream_t stream) {
    GenGeom g;
    if (int e = gen_resolve(geom, &g)) return e;
    if (!scene) return NVSR_ERR_NULL;
    GradPlanesN gp = {};
    gp.g[0] = d_plane0; gp.g[1] = d_plane1; gp.g[2] = d_plane2; gp.g[3] = d_plane3;
    return gen_decode_backward(gen_scene(scene), g, natural, P, x, nullptr, d_out, d_natural, gp, workspace, (hipStream_t)stream);
}
int nvsr_generic_decode_backward_ext(const nvsr_scene_ext* scene, const nvsr_decoder_geometry* geom, const float* natural, int64_t P, const float* x,
                                     const float* coord_noise, const float* d_out, float* d_natural, float* const* d_planes, float* workspace,
                                     nvsr_stream_t stream) {
    if (int e = gen_check_ext(scene)) return e;
    GenGeom g;
    if (int e = gen_resolve(geom, &g, scene->num_position_planes)) return e;
    GradPlanesN gp = {};
    if (d_planes)
        for (int d = 0; d <= scene->num_position_planes; ++d) gp.g[d] = d_planes[d];
    return gen_decode_backward(gen_scene(scene), g, natural, P, x, coord_noise, d_out, d_natural, gp, workspace, (hipStream_t)stream);
}

int nvsr_ray_points(int64_t N, int S, const float* rays, const float* z, float* x, nvsr_stream_t stream) {
    if (!rays || !z || !x) return NVSR_ERR_NULL;
    if (N < 0 || S < 1) return NVSR_ERR_SHAPE;
    if (N == 0) return NVSR_OK;
    const long n = (long)N * S;
    hipLaunchKernelGGL(ray_points_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long)N, S, rays, z, x);
    return NVSR_CHECK_LAUNCH();
}

}  // extern "C"
