// TwoDimPlanesModel.forward for ANY decoder geometry (inference): the fallback behind the MFMA kernels that are compiled for the shipped
// one (3 + 1 planes x 48 channels, 'avg' / 'concat_pos', 4 + 4 layers x 128).
//
// Reference: models.py:381-421 with the layer lists of the constructor :169-195, is_skip_layer :203-207, combine_pos_planes :355-361,
// combine_all_planes :363-379; the shipped YAMLs list dec_channels 256, proj_combination sum / concat, 24-channel planes and skip layers as
// alternatives (config/TrainModels.yml:78,82,92).
//
// Three plain kernels, activations through HBM (this is the compatibility path, not the hot one):
//   generic_inputs_kernel   one thread per (point, input column): normalise, project, 4 bilinear taps per plane, combine -> the density
//                           decoder's input Xd [P][Kd] and the colour decoder's input Xr [P][Kr]
//   generic_linear_kernel   Y[P][M] = act(b + W[M][K1 + K2] . [X1[P][K1] | X2[P][K2]])  (X2 = the skip connection's second operand) on
//                           v_mfma_f32_32x32x2_f32 (exact f32 products, k in order), one 32 x 32 tile per wave, operands staged through LDS
//   ray_points_kernel       x[N*S][6] = [ro + rd * z, viewdir]  (run_network's input, train_utils.py:15-64,111)
#include "decode_core.h"

namespace nvsr {

struct GenGeom {
    int C, Cv, hidden, nd, nr, skip, proj, view;      // proj: 0 sum 1 avg 2 concat; view: 0 sum 1 avg 2 mult 3 concat 4 concat_pos
    int Kd, Kr;
};

__host__ __device__ inline bool gen_skip_layer(int layer_num, int skip) { return skip > 0 && layer_num % skip == 0 && layer_num > 0; }

static int gen_resolve(const nvsr_decoder_geometry* g, GenGeom* o) {
    if (!g) return NVSR_ERR_NULL;
    o->C = g->plane_channels; o->Cv = g->viewdir_channels; o->hidden = g->hidden; o->nd = g->density_layers; o->nr = g->rgb_layers;
    o->skip = g->skip_connect_every; o->proj = g->proj_combination; o->view = g->viewdir_combination;
    if (o->C < 1 || o->C > 1024 || o->Cv < 1 || o->Cv > 1024 || o->hidden < 1 || o->hidden > 4096 || o->nd < 1 || o->nd > 64 || o->nr < 1 || o->nr > 64)
        return NVSR_ERR_SHAPE;
    if (o->skip < 0 || o->proj < 0 || o->proj > 2 || o->view < 0 || o->view > 4) return NVSR_ERR_SHAPE;
    // the combinations the reference's own layer sizes admit (models.py:186-190 vs :363-379)
    const bool concat_like = o->proj == 2 || o->view == 4;
    if (o->view == 3 && o->proj != 2) return NVSR_ERR_SHAPE;              // 'concat' view needs concatenated position features
    if (o->view <= 2 && (o->proj == 2 || o->Cv != o->C)) return NVSR_ERR_SHAPE;   // sum / avg / mult combine equal-sized vectors
    o->Kd = o->C * (o->proj == 2 ? 3 : 1);
    o->Kr = o->Cv + (concat_like ? 3 * o->C : 0);
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
__device__ __forceinline__ GenTaps gen_taps(int H, int W, int Cc, float gx, float gy) {
    // grid_sample(align_corners=True, padding_mode='border'): unnormalise, clip, floor; a clamped neighbour carries weight 0
    const float mx = (float)(W - 1), my = (float)(H - 1);
    float x = (gx + 1.0f) * (mx / 2.0f), y = (gy + 1.0f) * (my / 2.0f);
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

__global__ void generic_inputs_kernel(SceneDev sc, GenGeom g, long P, const float* __restrict__ x, float* __restrict__ Xd, float* __restrict__ Xr) {
    const int K = g.Kd + g.Kr;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P * K) return;
    const long p = idx / K;
    const int j = (int)(idx - p * K);
    const float* q = x + p * 6;
    // normalised coordinates (models.py:261-268; cart2az_el nerf_helpers.py:492-496)
    const float n0 = norm_coord(q[0], sc.lo[0], sc.range[0]), n1 = norm_coord(q[1], sc.lo[1], sc.range[1]), n2 = norm_coord(q[2], sc.lo[2], sc.range[2]);
    auto pos_feat = [&](int d, int c) {
        const float* M = sc.proj + 6 * d;
        const GenTaps t = gen_taps(sc.ph[d], sc.pw[d], g.C, n0 * M[0] + n1 * M[2] + n2 * M[4], n0 * M[1] + n1 * M[3] + n2 * M[5]);
        return gen_blend(sc.plane[d], t, c);
    };
    auto view_feat = [&](int c) {
        const float az = atan2f(q[4], q[3]);
        const float el = atan2f(q[5], sqrtf(__fadd_rn(__fmul_rn(q[3], q[3]), __fmul_rn(q[4], q[4]))));
        const GenTaps t = gen_taps(sc.ph[3], sc.pw[3], g.Cv, norm_coord(az, sc.lo[3], sc.range[3]), norm_coord(el, sc.lo[4], sc.range[4]));
        return gen_blend(sc.plane[3], t, c);
    };
    auto combined_pos = [&](int c) {                       // combine_pos_planes, column c of its result
        if (g.proj == 2) return pos_feat(c / g.C, c % g.C);
        const float s = __fadd_rn(__fadd_rn(pos_feat(0, c), pos_feat(1, c)), pos_feat(2, c));
        return g.proj == 1 ? __fdiv_rn(s, 3.0f) : s;
    };
    if (j < g.Kd) { Xd[p * g.Kd + j] = combined_pos(j); return; }
    const int c = j - g.Kd;
    float v;
    if (g.view == 4) v = c < 3 * g.C ? pos_feat(c / g.C, c % g.C) : view_feat(c - 3 * g.C);             // cat(pos_planes + [viewdir])
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

constexpr long GEN_CHUNK = 1L << 20;       // points per pass of the layer stack (bounds the activation workspace)

}  // namespace nvsr

using namespace nvsr;

extern "C" {

int64_t nvsr_generic_decoder_natural_floats(const nvsr_decoder_geometry* geom) {
    GenGeom g;
    if (gen_resolve(geom, &g)) return -1;
    int64_t n = 0;
    for (int l = 0; l < g.nd; ++l) n += (int64_t)g.hidden * gen_layer_in(g, false, l) + g.hidden;
    n += g.hidden + 1;
    for (int l = 0; l < g.nr; ++l) n += (int64_t)g.hidden * gen_layer_in(g, true, l) + g.hidden;
    n += 3 * g.hidden + 3;
    return n;
}

int64_t nvsr_generic_decode_workspace_floats(const nvsr_decoder_geometry* geom, int64_t P) {
    GenGeom g;
    if (gen_resolve(geom, &g) || P < 0) return -1;
    const int64_t c = P < GEN_CHUNK ? P : GEN_CHUNK;
    return c * (int64_t)(g.Kd + g.Kr + 2 * g.hidden);
}

int nvsr_generic_decode(const nvsr_scene* scene, const nvsr_decoder_geometry* geom, const float* natural, int64_t P, const float* x, float* out,
                        float* workspace, nvsr_stream_t stream_) {
    GenGeom g;
    if (int e = gen_resolve(geom, &g)) return e;
    if (!scene || !natural || !x || !out || !workspace) return NVSR_ERR_NULL;
    for (int d = 0; d < 4; ++d) {
        if (!scene->planes[d]) return NVSR_ERR_NULL;
        const int64_t cc = d < 3 ? g.C : g.Cv;
        if (scene->ph[d] < 1 || scene->pw[d] < 1 || (int64_t)scene->ph[d] * scene->pw[d] * cc >= (int64_t)1 << 31) return NVSR_ERR_SHAPE;
    }
    if (P < 0) return NVSR_ERR_SHAPE;
    hipStream_t stream = (hipStream_t)stream_;
    const SceneDev sc = to_dev(scene);
    for (int64_t a = 0; a < P; a += GEN_CHUNK) {
        const long n = (long)((P - a) < GEN_CHUNK ? (P - a) : GEN_CHUNK);
        float* Xd = workspace;
        float* Xr = Xd + n * g.Kd;
        float* H[2] = {Xr + n * g.Kr, Xr + n * g.Kr + n * g.hidden};
        const long threads = n * (g.Kd + g.Kr);
        hipLaunchKernelGGL(generic_inputs_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, sc, g, n, x + a * 6, Xd, Xr);
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

int nvsr_ray_points(int64_t N, int S, const float* rays, const float* z, float* x, nvsr_stream_t stream) {
    if (!rays || !z || !x) return NVSR_ERR_NULL;
    if (N < 0 || S < 1) return NVSR_ERR_SHAPE;
    if (N == 0) return NVSR_OK;
    const long n = (long)N * S;
    hipLaunchKernelGGL(ray_points_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long)N, S, rays, z, x);
    return NVSR_CHECK_LAUNCH();
}

}  // extern "C"
