/*
 * nvsr_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, CPU restatement of the volumetric-rendering hot path of
 * princeton-computational-imaging/Neural-Volume-Super-Resolution (pure PyTorch, fp32).
 * It is the checker for the HIP kernels: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The product path never calls into it.
 *
 * Parity status: PINNED.  The reference has no tests of its own, so every
 * function below is checked against outputs of the reference itself, generated in the build
 * container by tests/golden/gen_golden.py and committed under tests/golden/ (tests/test_oracle.py).
 *
 * Each function cites the reference file:line it follows (paths relative to the upstream repo).
 * All arithmetic is fp32 like the reference; dot products (Linear / conv) accumulate in ORC_ACC
 * (double by default: the reference's sgemm/conv summation order is unspecified, so the checker
 * sits at the centre of all fp32 orderings; build with -DORC_ACC=float for the timed CPU baseline).
 *
 * Build: see oracle/Makefile (gcc -O3 -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef ORC_ACC
#define ORC_ACC double
#endif
typedef ORC_ACC acc_t;

#define ORC_EXPORT __attribute__((visibility("default")))

ORC_EXPORT int orc_acc_is_double(void) { return sizeof(acc_t) == 8; }

/* ------------------------------------------------------------------------------------------
 * get_ray_bundle  (nerf_helpers.py:507-549, meshgrid_xy :396-406, get_focal :432-437)
 *   ii = col + off (- pad), jj = row + off (- pad); dir = [(ii - W/2)/f_x, -(jj - H/2)/f_y, -1]
 *   rd[r] = sum_k dir[k] * c2w[r][k];  ro = c2w[:3,3].  Directions are NOT normalised.
 * f_x = get_focal(focal,'H'), f_y = get_focal(focal,'W') (the reference's naming, :539-540).
 * Output rows: (H+2*pad) x (W+2*pad) x 3.
 */
ORC_EXPORT void orc_get_ray_bundle(int H, int W, double focal_x, double focal_y, const float* c2w /*4x4*/,
                                   int pad, double off, float* ro, float* rd) {
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    const float fx = (float)focal_x, fy = (float)focal_y;
    const float hw = (float)(W * 0.5), hh = (float)(H * 0.5);
    for (int r = 0; r < Hp; ++r)
        for (int c = 0; c < Wp; ++c) {
            float ii = (float)c + (float)off;
            float jj = (float)r + (float)off;
            if (pad > 0) { ii = ii - (float)pad; jj = jj - (float)pad; }
            const float d0 = (ii - hw) / fx;
            const float d1 = -(jj - hh) / fy;
            const float d2 = -1.0f;
            float* o = ro + ((size_t)r * Wp + c) * 3;
            float* d = rd + ((size_t)r * Wp + c) * 3;
            for (int k = 0; k < 3; ++k) {
                volatile float p0 = d0 * c2w[k * 4 + 0];
                volatile float p1 = d1 * c2w[k * 4 + 1];
                volatile float p2 = d2 * c2w[k * 4 + 2];
                volatile float s0 = 0.0f + p0;   /* torch.sum starts from +0.0: (-0.0) + (-0.0) + (-0.0) must give +0.0 */
                volatile float s = s0 + p1;
                d[k] = s + p2;
                o[k] = c2w[k * 4 + 3];
            }
        }
}

/* ndc_rays (nerf_helpers.py:578-605) */
ORC_EXPORT void orc_ndc_rays(int H, int W, double focal, double near_, int N, const float* ro, const float* rd,
                             float* ro_out, float* rd_out) {
    const float nr = (float)near_;
    const float sx = (float)(-1.0 / (W / (2.0 * focal)));
    const float sy = (float)(-1.0 / (H / (2.0 * focal)));
    const float two_near = (float)(2.0 * near_);
    const float m_two_near = (float)(-2.0 * near_);
    for (int i = 0; i < N; ++i) {
        const float* o = ro + 3 * i;
        const float* d = rd + 3 * i;
        const float t = -(nr + o[2]) / d[2];
        const float ox = o[0] + t * d[0], oy = o[1] + t * d[1], oz = o[2] + t * d[2];
        ro_out[3 * i + 0] = sx * ox / oz;
        ro_out[3 * i + 1] = sy * oy / oz;
        ro_out[3 * i + 2] = 1.0f + two_near / oz;
        rd_out[3 * i + 0] = sx * (d[0] / d[2] - ox / oz);
        rd_out[3 * i + 1] = sy * (d[1] / d[2] - oy / oz);
        rd_out[3 * i + 2] = m_two_near / oz;
    }
}

/* linspace as torch computes it for float (steps n): step=(end-start)/(n-1); i<n/2 ? start+i*step : end-(n-1-i)*step */
static inline float orc_linspace01(int i, int n) {
    if (n == 1) return 0.0f;
    const float step = 1.0f / (float)(n - 1);
    return (i < n / 2) ? (0.0f + step * (float)i) : (1.0f - step * (float)(n - 1 - i));
}

/* coarse depths, stratified jitter (train_utils.py:95-109) */
ORC_EXPORT void orc_coarse_z(int N, int Nc, const float* near_, const float* far_, int lindisp, int perturb,
                             const float* t_rand /*[N,Nc] or NULL*/, float* z /*[N,Nc]*/) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        float* zi = z + (size_t)i * Nc;
        const float nr = near_[i], fr = far_[i];
        for (int s = 0; s < Nc; ++s) {
            const float t = orc_linspace01(s, Nc);
            if (!lindisp)
                zi[s] = nr * (1.0f - t) + fr * t;
            else
                zi[s] = 1.0f / (1.0f / nr * (1.0f - t) + 1.0f / fr * t);
        }
        if (perturb) {
            float* tmp = (float*)malloc(sizeof(float) * (size_t)Nc);
            for (int s = 0; s < Nc; ++s) {
                const float lower = (s == 0) ? zi[0] : 0.5f * (zi[s] + zi[s - 1]);
                const float upper = (s == Nc - 1) ? zi[Nc - 1] : 0.5f * (zi[s + 1] + zi[s]);
                tmp[s] = lower + (upper - lower) * t_rand[(size_t)i * Nc + s];
            }
            memcpy(zi, tmp, sizeof(float) * (size_t)Nc);
            free(tmp);
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Tri-plane decoder: TwoDimPlanesModel.forward (models.py:381-421)
 *   cart2az_el (nerf_helpers.py:492-496), normalize_coords (models.py:261-268),
 *   CoordProjector.forward (models.py:495-497), project_xyz / project_viewdir (models.py:289-326):
 *   grid_sample(bilinear, align_corners=True, padding_mode='border'),
 *   combine_pos_planes 'avg' (:358-359), combine_all_planes 'concat_pos' (:379), decoders (:395-421).
 */
typedef struct {
    int C;                 /* plane channels (48) */
    int hidden;            /* dec_channels (128) */
    int n_density_layers;  /* 4 */
    int n_rgb_layers;      /* 4 */
    const float* blob;     /* state-dict order: density_dec.{i}.{weight[out,in],bias}, fc_alpha.{weight[1,h],bias},
                              rgb_dec.{i}.{weight,bias}, fc_rgb.{weight[3,h],bias} */
} orc_decoder;

static size_t orc_decoder_floats(const orc_decoder* d) {
    const size_t h = d->hidden, C = d->C;
    size_t n = C * h + h + (size_t)(d->n_density_layers - 1) * (h * h + h) + h + 1;
    n += 4 * C * h + h + (size_t)(d->n_rgb_layers - 1) * (h * h + h) + 3 * h + 3;
    return n;
}
ORC_EXPORT long orc_decoder_blob_floats(int C, int hidden, int nd, int nr) {
    orc_decoder d = {C, hidden, nd, nr, NULL};
    return (long)orc_decoder_floats(&d);
}

/* unnormalise + clip for align_corners=True, border padding (ATen GridSamplerKernel.cpp ComputeLocation) */
static inline float orc_grid_loc(float g, int size) {
    const float scaling = (float)(size - 1) / 2.0f;
    float x = (g + 1.0f) * scaling;
    const float mx = (float)(size - 1);
    x = fminf(mx, fmaxf(x, 0.0f));
    return x;
}

/* bilinear sample of one NCHW plane [C,Hp,Wp] at normalised (gx -> width, gy -> height) */
static void orc_sample_plane(const float* plane, int C, int Hp, int Wp, float gx, float gy, float* out) {
    const float x = orc_grid_loc(gx, Wp), y = orc_grid_loc(gy, Hp);
    const float xw = floorf(x), yn = floorf(y);
    const float w = x - xw, e = 1.0f - w, n = y - yn, s = 1.0f - n;
    const float nw = s * e, ne = s * w, sw = n * e, se = n * w;
    const int ix = (int)xw, iy = (int)yn;
    const int x1ok = (ix + 1 <= Wp - 1), y1ok = (iy + 1 <= Hp - 1);
    const size_t HW = (size_t)Hp * Wp;
    for (int c = 0; c < C; ++c) {
        const float* p = plane + (size_t)c * HW;
        float v = p[(size_t)iy * Wp + ix] * nw;
        if (x1ok) v += p[(size_t)iy * Wp + ix + 1] * ne;
        if (y1ok) v += p[(size_t)(iy + 1) * Wp + ix] * sw;
        if (x1ok && y1ok) v += p[(size_t)(iy + 1) * Wp + ix + 1] * se;
        out[c] = v;
    }
}

/* y[o] = relu?(b[o] + sum_k x[k] W[o][k]) */
static void orc_linear(const float* W, const float* b, const float* x, int in, int out, int relu, float* y) {
    for (int o = 0; o < out; ++o) {
        acc_t a = 0;
        const float* w = W + (size_t)o * in;
        for (int k = 0; k < in; ++k) a += (acc_t)x[k] * (acc_t)w[k];
        float v = (float)a + b[o];
        y[o] = (relu && v < 0.0f) ? 0.0f : v;
    }
}

typedef struct {
    const float* planes[4]; /* NCHW [C,H,W] each */
    int ph[4], pw[4];
    float lo[5], range[5];  /* box[0] and (box[1]-box[0]) cast to f32 */
    float proj[3][6];       /* rot_mats[d][:,1:] row-major 3x2 */
} orc_scene;

ORC_EXPORT void orc_scene_init(orc_scene* sc, const float* p0, const float* p1, const float* p2, const float* pv,
                               const int* hw /*8 ints*/, const double* box /*[2,5]*/, const float* rot /*[3,3,3] or NULL*/) {
    sc->planes[0] = p0; sc->planes[1] = p1; sc->planes[2] = p2; sc->planes[3] = pv;
    for (int d = 0; d < 4; ++d) { sc->ph[d] = hw[2 * d]; sc->pw[d] = hw[2 * d + 1]; }
    for (int i = 0; i < 5; ++i) { sc->lo[i] = (float)box[i]; sc->range[i] = (float)(box[5 + i] - box[i]); }
    static const float def_rot[3][9] = {{1,0,0, 0,1,0, 0,0,1}, {0,1,0, 1,0,0, 0,0,1}, {0,1,0, 0,0,1, 1,0,0}};
    for (int d = 0; d < 3; ++d)
        for (int k = 0; k < 3; ++k)
            for (int c = 0; c < 2; ++c) sc->proj[d][k * 2 + c] = rot ? rot[d * 9 + k * 3 + 1 + c] : def_rot[d][k * 3 + 1 + c];
}
ORC_EXPORT int orc_scene_sizeof(void) { return (int)sizeof(orc_scene); }

/* One point.  feats (optional) receives [f0|f1|f2|fview|density_in] = 5*C floats, n5 (optional) 5 floats. */
static void orc_decode_point(const orc_scene* sc, const orc_decoder* dec, const float* x6, float* out4,
                             float* feats, float* n5_out) {
    const int C = dec->C, h = dec->hidden;
    float x5[5], n5[5];
    x5[0] = x6[0]; x5[1] = x6[1]; x5[2] = x6[2];
    x5[3] = atan2f(x6[4], x6[3]);
    x5[4] = atan2f(x6[5], sqrtf(x6[3] * x6[3] + x6[4] * x6[4]));
    for (int i = 0; i < 5; ++i) n5[i] = 2.0f * (x5[i] - sc->lo[i]) / sc->range[i] - 1.0f;
    if (n5_out) memcpy(n5_out, n5, sizeof(n5));
    float f[4 * 64 + 64]; /* C <= 64 */
    float* rgb_in = f;
    for (int d = 0; d < 3; ++d) {
        const float* M = sc->proj[d];
        const float gx = n5[0] * M[0] + n5[1] * M[2] + n5[2] * M[4];
        const float gy = n5[0] * M[1] + n5[1] * M[3] + n5[2] * M[5];
        orc_sample_plane(sc->planes[d], C, sc->ph[d], sc->pw[d], gx, gy, rgb_in + d * C);
    }
    orc_sample_plane(sc->planes[3], C, sc->ph[3], sc->pw[3], n5[3], n5[4], rgb_in + 3 * C);
    float* dens_in = f + 4 * C;
    for (int c = 0; c < C; ++c) dens_in[c] = ((rgb_in[c] + rgb_in[C + c]) + rgb_in[2 * C + c]) / 3.0f;
    if (feats) memcpy(feats, f, sizeof(float) * 5 * (size_t)C);

    float a[512], b[512];
    const float* p = dec->blob;
    const float* cur = dens_in;
    int in = C;
    float* dst = a;
    for (int l = 0; l < dec->n_density_layers; ++l) {
        orc_linear(p, p + (size_t)h * in, cur, in, h, 1, dst);
        p += (size_t)h * in + h;
        cur = dst; dst = (dst == a) ? b : a; in = h;
    }
    float alpha;
    orc_linear(p, p + h, cur, h, 1, 0, &alpha);
    p += h + 1;
    cur = rgb_in; in = 4 * C; dst = a;
    for (int l = 0; l < dec->n_rgb_layers; ++l) {
        orc_linear(p, p + (size_t)h * in, cur, in, h, 1, dst);
        p += (size_t)h * in + h;
        cur = dst; dst = (dst == a) ? b : a; in = h;
    }
    orc_linear(p, p + 3 * h, cur, h, 3, 0, out4);
    out4[3] = alpha;
}

ORC_EXPORT void orc_triplane_decode(const orc_scene* sc, const orc_decoder* dec, long P, const float* x /*[P,6]*/,
                                    float* out /*[P,4]*/, float* feats /*[P,5C] or NULL*/, float* n5 /*[P,5] or NULL*/) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < P; ++i)
        orc_decode_point(sc, dec, x + 6 * i, out + 4 * i, feats ? feats + (size_t)i * 5 * dec->C : NULL, n5 ? n5 + 5 * i : NULL);
}

/* ------------------------------------------------------------------------------------------
 * volume_render_radiance_field (volume_rendering_utils.py:6-51) + cumprod_exclusive (nerf_helpers.py:409-430)
 */
static void orc_composite_ray(int S, const float* raw /*[S,4]*/, const float* z, const float* rd3, const float* noise,
                              int white, float* rgb3, float* disp, float* acc, float* weights, float* depth) {
    const float nrm = sqrtf(rd3[0] * rd3[0] + rd3[1] * rd3[1] + rd3[2] * rd3[2]);
    float T = 1.0f, r = 0, g = 0, b = 0, dep = 0, ac = 0;
    for (int s = 0; s < S; ++s) {
        const float dist = ((s == S - 1) ? 1e10f : (z[s + 1] - z[s])) * nrm;
        float sig = raw[4 * s + 3] + (noise ? noise[s] : 0.0f);
        sig = sig > 0.0f ? sig : 0.0f;
        const float alpha = 1.0f - expf(-sig * dist);
        const float w = alpha * T;
        T = T * (1.0f - alpha + 1e-10f);
        if (weights) weights[s] = w;
        r += w * (1.0f / (1.0f + expf(-raw[4 * s + 0])));
        g += w * (1.0f / (1.0f + expf(-raw[4 * s + 1])));
        b += w * (1.0f / (1.0f + expf(-raw[4 * s + 2])));
        dep += w * z[s];
        ac += w;
    }
    { const float q = dep / ac; *disp = 1.0f / ((q != q) ? q : fmaxf(1e-10f, q)); } /* torch.max propagates NaN (acc == 0) */
    if (white) { r += 1.0f - ac; g += 1.0f - ac; b += 1.0f - ac; }
    rgb3[0] = r; rgb3[1] = g; rgb3[2] = b;
    *acc = ac; *depth = dep;
}

ORC_EXPORT void orc_composite(long N, int S, const float* raw, const float* z, const float* rd, const float* noise,
                              int white, float* rgb, float* disp, float* acc, float* weights, float* depth) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < N; ++i)
        orc_composite_ray(S, raw + (size_t)i * S * 4, z + (size_t)i * S, rd + 3 * i, noise ? noise + (size_t)i * S : NULL,
                          white, rgb + 3 * i, disp + i, acc + i, weights ? weights + (size_t)i * S : NULL, depth + i);
}

/* mip_nerf=True branch (volume_rendering_utils.py:19-26,41-42): z [S+1] interval edges, no 1e10 tail, depth over the interval mid-points */
ORC_EXPORT void orc_composite_mip(long N, int S, const float* raw, const float* z, const float* rd, const float* noise,
                                  int white, float* rgb, float* disp, float* acc, float* weights, float* depth) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < N; ++i) {
        const float* rw = raw + (size_t)i * S * 4;
        const float* zz = z + (size_t)i * (S + 1);
        const float* d3 = rd + 3 * i;
        const float nrm = sqrtf(d3[0] * d3[0] + d3[1] * d3[1] + d3[2] * d3[2]);
        float T = 1.0f, r = 0, g = 0, b = 0, dep = 0, ac = 0;
        for (int s = 0; s < S; ++s) {
            const float dist = (zz[s + 1] - zz[s]) * nrm;
            float sig = rw[4 * s + 3] + (noise ? noise[(size_t)i * S + s] : 0.0f);
            sig = sig > 0.0f ? sig : 0.0f;
            const float alpha = 1.0f - expf(-sig * dist);
            const float w = alpha * T;
            T = T * (1.0f - alpha + 1e-10f);
            if (weights) weights[(size_t)i * S + s] = w;
            r += w * (1.0f / (1.0f + expf(-rw[4 * s + 0])));
            g += w * (1.0f / (1.0f + expf(-rw[4 * s + 1])));
            b += w * (1.0f / (1.0f + expf(-rw[4 * s + 2])));
            dep += w * (0.5f * (zz[s] + zz[s + 1]));
            ac += w;
        }
        { const float q = dep / ac; disp[i] = 1.0f / ((q != q) ? q : fmaxf(1e-10f, q)); }
        if (white) { r += 1.0f - ac; g += 1.0f - ac; b += 1.0f - ac; }
        rgb[3 * i] = r; rgb[3 * i + 1] = g; rgb[3 * i + 2] = b;
        acc[i] = ac; depth[i] = dep;
    }
}

ORC_EXPORT void orc_cumprod_exclusive(long N, int S, const float* in, float* out) {
    for (long i = 0; i < N; ++i) {
        float T = 1.0f;
        for (int s = 0; s < S; ++s) { out[i * S + s] = T; T *= in[i * S + s]; }
    }
}

/* ------------------------------------------------------------------------------------------
 * sample_pdf_2 (nerf_helpers.py:668-702).  bins [N,nb], weights [N,nb-1], u [N,ns] (the caller supplies u:
 * linspace(0,1,ns) for det=True, the CPU generator's rand otherwise) -> samples [N,ns]
 */
static void orc_sample_pdf_ray(int nb, int ns, const float* bins, const float* w, const float* u, float* out) {
    float cdf[1024];
    float sum = 0.0f;
    for (int i = 0; i < nb - 1; ++i) sum += (w[i] + 1e-5f);
    cdf[0] = 0.0f;
    float run = 0.0f;
    for (int i = 0; i < nb - 1; ++i) { run += (w[i] + 1e-5f) / sum; cdf[i + 1] = run; }
    for (int j = 0; j < ns; ++j) {
        /* searchsorted(cdf, u, right=True): first index with cdf[idx] > u */
        int lo = 0, hi = nb;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] > u[j]) hi = mid; else lo = mid + 1; }
        const int below = lo - 1 > 0 ? lo - 1 : 0;
        const int above = lo < nb - 1 ? lo : nb - 1;
        float denom = cdf[above] - cdf[below];
        if (denom < 1e-5f) denom = 1.0f;
        const float t = (u[j] - cdf[below]) / denom;
        out[j] = bins[below] + t * (bins[above] - bins[below]);
    }
}

ORC_EXPORT void orc_sample_pdf(long N, int nb, int ns, const float* bins, const float* weights, const float* u, float* samples) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < N; ++i)
        orc_sample_pdf_ray(nb, ns, bins + (size_t)i * nb, weights + (size_t)i * (nb - 1), u + (size_t)i * ns, samples + (size_t)i * ns);
}

static int orc_cmp_float(const void* a, const void* b) {
    const float x = *(const float*)a, y = *(const float*)b;
    return (x > y) - (x < y);
}
/* sort(cat(z_vals, z_samples)) along the last dim (train_utils.py:155) */
ORC_EXPORT void orc_sort_rows(long N, int n, float* data) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < N; ++i) qsort(data + (size_t)i * n, (size_t)n, sizeof(float), orc_cmp_float);
}

/* ------------------------------------------------------------------------------------------
 * predict_and_render_radiance + run_network (train_utils.py:15-182) on packed rays [N,11] =
 * [ro, rd, near, far, viewdir] (run_one_iter_of_nerf, train_utils.py:213-226).
 * t_rand/u/noise_* are the explicit random inputs (NULL: perturb off / det=True / no noise).
 */
typedef struct {
    int num_coarse, num_fine, lindisp, perturb, white_background;
} orc_render_cfg;

ORC_EXPORT void orc_pack_rays(long N, const float* ro, const float* rd, const float* dirs_for_view, double near_, double far_, float* rays) {
    for (long i = 0; i < N; ++i) {
        float* r = rays + 11 * i;
        const float* v = dirs_for_view + 3 * i;
        memcpy(r, ro + 3 * i, 12); memcpy(r + 3, rd + 3 * i, 12);
        r[6] = (float)near_; r[7] = (float)far_;
        const float nrm = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        r[8] = v[0] / nrm; r[9] = v[1] / nrm; r[10] = v[2] / nrm;
    }
}

ORC_EXPORT void orc_render_rays(const orc_scene* sc, const orc_decoder* coarse, const orc_decoder* fine, const orc_render_cfg* cfg,
                                long N, const float* rays, const float* t_rand, const float* u, const float* noise_c,
                                const float* noise_f, float* rgb_c, float* disp_c, float* acc_c, float* rgb_f, float* disp_f,
                                float* acc_f, float* z_fine_out /*[N,Nc+Nf] or NULL*/, float* weights_c_out /*[N,Nc] or NULL*/) {
    const int Nc = cfg->num_coarse, Nf = cfg->num_fine, St = Nc + Nf;
#pragma omp parallel
    {
        float* z = (float*)malloc(sizeof(float) * (size_t)(St + 4));
        float* zs = (float*)malloc(sizeof(float) * (size_t)(Nf + 4));
        float* zm = (float*)malloc(sizeof(float) * (size_t)(Nc + 4));
        float* raw = (float*)malloc(sizeof(float) * 4 * (size_t)(St + 4));
        float* w = (float*)malloc(sizeof(float) * (size_t)(St + 4));
        float* ud = (float*)malloc(sizeof(float) * (size_t)(Nf + 4));
#pragma omp for schedule(dynamic, 16)
        for (long i = 0; i < N; ++i) {
            const float* r = rays + 11 * i;
            orc_coarse_z(1, Nc, r + 6, r + 7, cfg->lindisp, cfg->perturb, t_rand ? t_rand + (size_t)i * Nc : NULL, z);
            float x6[6], depth;
            x6[3] = r[8]; x6[4] = r[9]; x6[5] = r[10];
            for (int s = 0; s < Nc; ++s) {
                for (int k = 0; k < 3; ++k) x6[k] = r[k] + r[3 + k] * z[s];
                orc_decode_point(sc, coarse, x6, raw + 4 * s, NULL, NULL);
            }
            orc_composite_ray(Nc, raw, z, r + 3, noise_c ? noise_c + (size_t)i * Nc : NULL, cfg->white_background,
                              rgb_c + 3 * i, disp_c + i, acc_c + i, w, &depth);
            if (weights_c_out) memcpy(weights_c_out + (size_t)i * Nc, w, sizeof(float) * (size_t)Nc);
            if (Nf <= 0) continue;
            for (int s = 0; s < Nc - 1; ++s) zm[s] = 0.5f * (z[s + 1] + z[s]);
            const float* ui;
            if (u) ui = u + (size_t)i * Nf;
            else { for (int j = 0; j < Nf; ++j) ud[j] = orc_linspace01(j, Nf); ui = ud; }
            orc_sample_pdf_ray(Nc - 1, Nf, zm, w + 1, ui, zs);
            memcpy(z + Nc, zs, sizeof(float) * (size_t)Nf);
            qsort(z, (size_t)St, sizeof(float), orc_cmp_float);
            if (z_fine_out) memcpy(z_fine_out + (size_t)i * St, z, sizeof(float) * (size_t)St);
            for (int s = 0; s < St; ++s) {
                for (int k = 0; k < 3; ++k) x6[k] = r[k] + r[3 + k] * z[s];
                orc_decode_point(sc, fine, x6, raw + 4 * s, NULL, NULL);
            }
            orc_composite_ray(St, raw, z, r + 3, noise_f ? noise_f + (size_t)i * St : NULL, cfg->white_background,
                              rgb_f + 3 * i, disp_f + i, acc_f + i, NULL, &depth);
        }
        free(z); free(zs); free(zm); free(raw); free(w); free(ud);
    }
}

/* One pass of run_network + volume_render_radiance_field on given depths z [N,S] (train_utils.py:111-139 / :156-180) */
ORC_EXPORT void orc_render_given_z(const orc_scene* sc, const orc_decoder* dec, long N, int S, const float* rays, const float* z,
                                   const float* noise, int white, float* rgb, float* disp, float* acc, float* weights /*or NULL*/,
                                   float* depth, float* raw_out /*[N,S,4] or NULL*/) {
#pragma omp parallel
    {
        float* raw = (float*)malloc(sizeof(float) * 4 * (size_t)S);
        float* w = (float*)malloc(sizeof(float) * (size_t)S);
#pragma omp for schedule(dynamic, 16)
        for (long i = 0; i < N; ++i) {
            const float* r = rays + 11 * i;
            const float* zi = z + (size_t)i * S;
            float x6[6];
            x6[3] = r[8]; x6[4] = r[9]; x6[5] = r[10];
            for (int s = 0; s < S; ++s) {
                for (int k = 0; k < 3; ++k) x6[k] = r[k] + r[3 + k] * zi[s];
                orc_decode_point(sc, dec, x6, raw + 4 * s, NULL, NULL);
            }
            if (raw_out) memcpy(raw_out + (size_t)i * S * 4, raw, sizeof(float) * 4 * (size_t)S);
            orc_composite_ray(S, raw, zi, r + 3, noise ? noise + (size_t)i * S : NULL, white, rgb + 3 * i, disp + i, acc + i,
                              w, depth + i);
            if (weights) memcpy(weights + (size_t)i * S, w, sizeof(float) * (size_t)S);
        }
        free(raw); free(w);
    }
}

/* ------------------------------------------------------------------------------------------
 * EDSR (models.py:769-822): 3x3 valid convolutions without bias, residual blocks with a centre-cropped
 * identity and x0.1 scaling, PixelShuffle(2) upscaling.
 */
/* out[co][y][x] = sum_{ci,ky,kx} w[co][ci][ky][kx] in[ci][y+ky][x+kx];   in [Ci,H,W] -> out [Co,H-2,W-2] */
ORC_EXPORT void orc_conv3x3_valid(const float* in, int Ci, int H, int W, const float* wgt, int Co, int relu, float* out) {
    const int Ho = H - 2, Wo = W - 2;
#pragma omp parallel
    {
        acc_t* row = (acc_t*)malloc(sizeof(acc_t) * (size_t)Wo);
#pragma omp for collapse(2) schedule(static)
        for (int co = 0; co < Co; ++co)
            for (int y = 0; y < Ho; ++y) {
                for (int x = 0; x < Wo; ++x) row[x] = 0;
                for (int ci = 0; ci < Ci; ++ci)
                    for (int ky = 0; ky < 3; ++ky) {
                        const float* ip = in + ((size_t)ci * H + (y + ky)) * W;
                        const float* wp = wgt + (((size_t)co * Ci + ci) * 3 + ky) * 3;
                        const acc_t w0 = wp[0], w1 = wp[1], w2 = wp[2];
                        for (int x = 0; x < Wo; ++x) row[x] += w0 * (acc_t)ip[x] + w1 * (acc_t)ip[x + 1] + w2 * (acc_t)ip[x + 2];
                    }
                float* op = out + ((size_t)co * Ho + y) * Wo;
                for (int x = 0; x < Wo; ++x) { float v = (float)row[x]; op[x] = (relu && v < 0) ? 0.0f : v; }
            }
        free(row);
    }
}

/* nn.PixelShuffle(2): in [4C,H,W] -> out [C,2H,2W] */
ORC_EXPORT void orc_pixel_shuffle2(const float* in, int C, int H, int W, float* out) {
    for (int c = 0; c < C; ++c)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x)
                for (int i = 0; i < 2; ++i)
                    for (int j = 0; j < 2; ++j)
                        out[((size_t)c * 2 * H + 2 * y + i) * 2 * W + 2 * x + j] = in[((size_t)(c * 4 + i * 2 + j) * H + y) * W + x];
}

/* weights blob order = state-dict order: conv_input [hid,Cin,3,3], residual.{b}.conv1, .conv2 [hid,hid,3,3],
 * conv_mid, upscale.{0,2,..} [4hid,hid,3,3], conv_output [Cout,hid,3,3].  scale_factor = 2^n_up. */
ORC_EXPORT long orc_edsr_blob_floats(int Cin, int Cout, int hid, int nblocks, int n_up) {
    return 9L * ((long)hid * Cin + 2L * nblocks * hid * hid + (long)hid * hid + (long)n_up * 4 * hid * hid + (long)Cout * hid);
}
ORC_EXPORT void orc_edsr_out_size(int H, int W, int nblocks, int n_up, int* Ho, int* Wo) {
    int h = H - 2 - 4 * nblocks - 2, w = W - 2 - 4 * nblocks - 2;
    for (int u = 0; u < n_up; ++u) { h = (h - 2) * 2; w = (w - 2) * 2; }
    *Ho = h - 2; *Wo = w - 2;
}

ORC_EXPORT void orc_edsr_forward(const float* x, int Cin, int H, int W, const float* blob, int Cout, int hid, int nblocks,
                                 int n_up, float* out) {
    const float* p = blob;
    int h = H - 2, w = W - 2;
    float* cur = (float*)malloc(sizeof(float) * (size_t)hid * h * w);
    orc_conv3x3_valid(x, Cin, H, W, p, hid, 0, cur);
    p += 9 * (size_t)hid * Cin;
    for (int b = 0; b < nblocks; ++b) {          /* _Residual_Block.forward (models.py:777-786) */
        float* t1 = (float*)malloc(sizeof(float) * (size_t)hid * (h - 2) * (w - 2));
        orc_conv3x3_valid(cur, hid, h, w, p, hid, 1, t1);
        p += 9 * (size_t)hid * hid;
        float* t2 = (float*)malloc(sizeof(float) * (size_t)hid * (h - 4) * (w - 4));
        orc_conv3x3_valid(t1, hid, h - 2, w - 2, p, hid, 0, t2);
        p += 9 * (size_t)hid * hid;
        for (int c = 0; c < hid; ++c)
            for (int y = 0; y < h - 4; ++y)
                for (int xx = 0; xx < w - 4; ++xx) {
                    float* o = t2 + ((size_t)c * (h - 4) + y) * (w - 4) + xx;
                    *o = *o * 0.1f + cur[((size_t)c * h + y + 2) * w + xx + 2];
                }
        free(t1); free(cur);
        cur = t2; h -= 4; w -= 4;
    }
    {
        float* t = (float*)malloc(sizeof(float) * (size_t)hid * (h - 2) * (w - 2));
        orc_conv3x3_valid(cur, hid, h, w, p, hid, 0, t);
        p += 9 * (size_t)hid * hid;
        free(cur); cur = t; h -= 2; w -= 2;
    }
    for (int uidx = 0; uidx < n_up; ++uidx) {
        float* t = (float*)malloc(sizeof(float) * 4 * (size_t)hid * (h - 2) * (w - 2));
        orc_conv3x3_valid(cur, hid, h, w, p, 4 * hid, 0, t);
        p += 9 * 4 * (size_t)hid * hid;
        free(cur);
        h -= 2; w -= 2;
        cur = (float*)malloc(sizeof(float) * (size_t)hid * 4 * h * w);
        orc_pixel_shuffle2(t, hid, h, w, cur);
        free(t);
        h *= 2; w *= 2;
    }
    orc_conv3x3_valid(cur, hid, h, w, p, Cout, 0, out);
    free(cur);
}

/* F.interpolate(mode='bilinear', align_corners=True, scale_factor=sf) (models.py:858-859) on [C,H,W] */
ORC_EXPORT void orc_upsample_bilinear_ac(const float* in, int C, int H, int W, int sf, float* out) {
    const int Ho = H * sf, Wo = W * sf;
    const float sh = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.0f;
    const float sw = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.0f;
#pragma omp parallel for collapse(2) schedule(static)
    for (int c = 0; c < C; ++c)
        for (int oy = 0; oy < Ho; ++oy) {
            const float fy = sh * (float)oy;
            const int y0 = (int)fy;
            const int yp = (y0 < H - 1) ? 1 : 0;
            const float ly1 = fy - (float)y0, ly0 = 1.0f - ly1;
            for (int ox = 0; ox < Wo; ++ox) {
                const float fx = sw * (float)ox;
                const int x0 = (int)fx;
                const int xp = (x0 < W - 1) ? 1 : 0;
                const float lx1 = fx - (float)x0, lx0 = 1.0f - lx1;
                const float* p = in + ((size_t)c * H + y0) * W + x0;
                out[((size_t)c * Ho + oy) * Wo + ox] =
                    ly0 * (lx0 * p[0] + lx1 * p[xp]) + ly1 * (lx0 * p[(size_t)yp * W] + lx1 * p[(size_t)yp * W + xp]);
            }
        }
}

/* PlanesSR.forward (models.py:884-926).  roi = NULL: full plane (eval).  roi = [[ymin,xmin],[ymax,xmax]] in [-1,1]
 * (training ROI path): the area outside the ROI is NaN.  pad = inner_model.required_padding, over = HR_overpadding
 * (models.py:836-842).  mean/std (optional) = planes_{mean,std}_NON_LEARNED.  out [C, sf*R0, sf*R1]. */
ORC_EXPORT void orc_planes_sr(const float* lr, int C, int R0, int R1, const float* blob, int hid, int nblocks, int n_up,
                              int pad, int over, const float* roi, const float* mean, const float* std_, float* out) {
    const int sf = 1 << n_up;
    int lo[2] = {0, 0}, hi[2] = {R0, R1};
    const int shape[2] = {R0, R1};
    if (roi)
        for (int a = 0; a < 2; ++a) {
            const float mn = (float)shape[a] * (1.0f + roi[a]) / 2.0f, mx = (float)shape[a] * (1.0f + roi[2 + a]) / 2.0f;
            int l = (int)floorf(mn), h = (int)ceilf(mx);
            l = l - 1 > 0 ? l - 1 : 0;
            h = h + 1 < shape[a] ? h + 1 : shape[a];
            lo[a] = l; hi[a] = h;
        }
    const int ch = hi[0] - lo[0], cw = hi[1] - lo[1];
    const int Hp = ch + 2 * pad, Wp = cw + 2 * pad;
    float* x = (float*)malloc(sizeof(float) * (size_t)C * Hp * Wp);
    /* crop (with as much real context as available, models.py:906-911) + replicate pad (:912-914) == clamped gather */
    for (int c = 0; c < C; ++c)
        for (int y = 0; y < Hp; ++y) {
            int sy = lo[0] - pad + y; sy = sy < 0 ? 0 : (sy > R0 - 1 ? R0 - 1 : sy);
            for (int xx = 0; xx < Wp; ++xx) {
                int sx = lo[1] - pad + xx; sx = sx < 0 ? 0 : (sx > R1 - 1 ? R1 - 1 : sx);
                float v = lr[((size_t)c * R0 + sy) * R1 + sx];
                if (mean) v = (v - mean[c]) / std_[c];
                x[((size_t)c * Hp + y) * Wp + xx] = v;
            }
        }
    int Ho, Wo;
    orc_edsr_out_size(Hp, Wp, nblocks, n_up, &Ho, &Wo);
    float* diff = (float*)malloc(sizeof(float) * (size_t)C * Ho * Wo);
    orc_edsr_forward(x, C, Hp, Wp, blob, C, hid, nblocks, n_up, diff);
    free(x);
    float* res = (float*)malloc(sizeof(float) * (size_t)C * R0 * sf * R1 * sf);
    orc_upsample_bilinear_ac(lr, C, R0, R1, sf, res);
    const size_t HRh = (size_t)R0 * sf, HRw = (size_t)R1 * sf;
    for (size_t i = 0; i < (size_t)C * HRh * HRw; ++i) out[i] = NAN;
    for (int c = 0; c < C; ++c)
        for (int y = 0; y < ch * sf; ++y)
            for (int xx = 0; xx < cw * sf; ++xx) {
                const size_t o = ((size_t)c * HRh + (size_t)lo[0] * sf + y) * HRw + (size_t)lo[1] * sf + xx;
                out[o] = diff[((size_t)c * Ho + y + over) * Wo + xx + over] + res[o];
            }
    free(diff); free(res);
}

/* ------------------------------------------------------------------------------------------
 * Backward of the SR network (the reference: torch.autograd through EDSR.forward models.py:818-822, _Residual_Block.forward
 * :777-786, PlanesSR.forward :884-926 when 'SR' is in nerf.train.what).  Pinned against tests/golden/g14_sr_grads.npz.
 */
/* backward of orc_conv3x3_valid: dy [Co,H-2,W-2] -> dx [Ci,H,W] (overwritten; may be NULL), dw [Co,Ci,3,3] += (double; may be NULL) */
ORC_EXPORT void orc_conv3x3_valid_backward(const float* in, int Ci, int H, int W, const float* wgt, int Co, const float* dy, float* dx,
                                           double* dw) {
    const int Ho = H - 2, Wo = W - 2;
    if (dx) {
#pragma omp parallel
        {
            double* row = (double*)malloc(sizeof(double) * (size_t)W);
#pragma omp for collapse(2) schedule(static)
            for (int ci = 0; ci < Ci; ++ci)
                for (int y = 0; y < H; ++y) {
                    for (int x = 0; x < W; ++x) row[x] = 0;
                    for (int co = 0; co < Co; ++co)
                        for (int ky = 0; ky < 3; ++ky) {
                            const int yo = y - ky;
                            if (yo < 0 || yo >= Ho) continue;
                            const float* gp = dy + ((size_t)co * Ho + yo) * Wo;
                            const float* wp = wgt + (((size_t)co * Ci + ci) * 3 + ky) * 3;
                            for (int kx = 0; kx < 3; ++kx) {
                                const double wv = wp[kx];
                                for (int xo = 0; xo < Wo; ++xo) row[xo + kx] += wv * (double)gp[xo];
                            }
                        }
                    float* op = dx + ((size_t)ci * H + y) * W;
                    for (int x = 0; x < W; ++x) op[x] = (float)row[x];
                }
            free(row);
        }
    }
    if (dw) {
#pragma omp parallel for collapse(2) schedule(static)
        for (int co = 0; co < Co; ++co)
            for (int ci = 0; ci < Ci; ++ci) {
                double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                for (int y = 0; y < Ho; ++y) {
                    const float* gp = dy + ((size_t)co * Ho + y) * Wo;
                    for (int ky = 0; ky < 3; ++ky) {
                        const float* ip = in + ((size_t)ci * H + (y + ky)) * W;
                        for (int x = 0; x < Wo; ++x) {
                            const double g = gp[x];
                            a[ky * 3 + 0] += g * (double)ip[x];
                            a[ky * 3 + 1] += g * (double)ip[x + 1];
                            a[ky * 3 + 2] += g * (double)ip[x + 2];
                        }
                    }
                }
                double* q = dw + ((size_t)co * Ci + ci) * 9;
                for (int t = 0; t < 9; ++t) q[t] += a[t];
            }
    }
}

/* d_out [Cout,Ho,Wo] -> d_blob (same order as the weights blob, overwritten) and dx [Cin,H,W] (may be NULL) */
ORC_EXPORT void orc_edsr_backward(const float* x, int Cin, int H, int W, const float* blob, int Cout, int hid, int nblocks, int n_up,
                                  const float* d_out, float* d_blob, float* dx) {
    const int nl = 1 + 2 * nblocks + 1 + n_up + 1;
    /* forward, keeping every conv input */
    const float** in = (const float**)calloc((size_t)nl, sizeof(float*));
    int* ih = (int*)calloc((size_t)nl, sizeof(int)); int* iw = (int*)calloc((size_t)nl, sizeof(int));
    int* ci_ = (int*)calloc((size_t)nl, sizeof(int)); int* co_ = (int*)calloc((size_t)nl, sizeof(int));
    const float** wl = (const float**)calloc((size_t)nl, sizeof(float*));
    const float* p = blob;
    int k = 0, h = H, w = W;
    const float* cur = x;
    #define ORC_LAYER(CI, CO) do { in[k] = cur; ih[k] = h; iw[k] = w; ci_[k] = (CI); co_[k] = (CO); wl[k] = p; p += 9 * (size_t)(CI) * (CO); } while (0)
    {
        ORC_LAYER(Cin, hid);
        float* t = (float*)malloc(sizeof(float) * (size_t)hid * (h - 2) * (w - 2));
        orc_conv3x3_valid(cur, Cin, h, w, wl[k], hid, 0, t);
        cur = t; h -= 2; w -= 2; ++k;
    }
    for (int b = 0; b < nblocks; ++b) {
        const float* blk_in = cur;
        ORC_LAYER(hid, hid);
        float* t1 = (float*)malloc(sizeof(float) * (size_t)hid * (h - 2) * (w - 2));
        orc_conv3x3_valid(cur, hid, h, w, wl[k], hid, 1, t1);
        cur = t1; ++k; h -= 2; w -= 2;
        ORC_LAYER(hid, hid);
        float* t2 = (float*)malloc(sizeof(float) * (size_t)hid * (h - 2) * (w - 2));
        orc_conv3x3_valid(cur, hid, h, w, wl[k], hid, 0, t2);
        ++k; h -= 2; w -= 2;
        for (int c = 0; c < hid; ++c)
            for (int y = 0; y < h; ++y)
                for (int xx = 0; xx < w; ++xx) {
                    float* o = t2 + ((size_t)c * h + y) * w + xx;
                    *o = *o * 0.1f + blk_in[((size_t)c * (h + 4) + y + 2) * (w + 4) + xx + 2];
                }
        cur = t2;
    }
    {
        ORC_LAYER(hid, hid);
        float* t = (float*)malloc(sizeof(float) * (size_t)hid * (h - 2) * (w - 2));
        orc_conv3x3_valid(cur, hid, h, w, wl[k], hid, 0, t);
        cur = t; h -= 2; w -= 2; ++k;
    }
    for (int u = 0; u < n_up; ++u) {
        ORC_LAYER(hid, 4 * hid);
        float* t = (float*)malloc(sizeof(float) * 4 * (size_t)hid * (h - 2) * (w - 2));
        orc_conv3x3_valid(cur, hid, h, w, wl[k], 4 * hid, 0, t);
        h -= 2; w -= 2; ++k;
        float* sh = (float*)malloc(sizeof(float) * 4 * (size_t)hid * h * w);
        orc_pixel_shuffle2(t, hid, h, w, sh);
        free(t);
        cur = sh; h *= 2; w *= 2;
    }
    ORC_LAYER(hid, Cout);
    ++k;
    #undef ORC_LAYER
    /* backward */
    const size_t nblob = (size_t)(p - blob);
    double* dwd = (double*)calloc(nblob, sizeof(double));
    float* g = (float*)malloc(sizeof(float) * (size_t)Cout * (h - 2) * (w - 2));
    memcpy(g, d_out, sizeof(float) * (size_t)Cout * (h - 2) * (w - 2));
    int l = nl - 1;
    #define ORC_BACK(NEED_DX) do { \
        float* gx = (NEED_DX) ? (float*)malloc(sizeof(float) * (size_t)ci_[l] * ih[l] * iw[l]) : NULL; \
        orc_conv3x3_valid_backward(in[l], ci_[l], ih[l], iw[l], wl[l], co_[l], g, gx, dwd + (wl[l] - blob)); \
        free(g); g = gx; } while (0)
    ORC_BACK(1); --l;                                          /* conv_output */
    for (int u = n_up - 1; u >= 0; --u) {                      /* PixelShuffle^T, then the up-conv */
        const int hs = ih[l] - 2, ws = iw[l] - 2;              /* conv output size before the shuffle */
        float* gu = (float*)malloc(sizeof(float) * 4 * (size_t)hid * hs * ws);
        for (int c = 0; c < hid; ++c)
            for (int y = 0; y < hs; ++y)
                for (int xx = 0; xx < ws; ++xx)
                    for (int i = 0; i < 2; ++i)
                        for (int j = 0; j < 2; ++j)
                            gu[((size_t)(c * 4 + i * 2 + j) * hs + y) * ws + xx] = g[((size_t)c * 2 * hs + 2 * y + i) * 2 * ws + 2 * xx + j];
        free(g); g = gu;
        ORC_BACK(1); --l;
    }
    ORC_BACK(1); --l;                                          /* conv_mid */
    for (int b = nblocks - 1; b >= 0; --b) {
        /* y = 0.1 * conv2(relu(conv1(x))) + crop(x):  g = dL/dy [hid, hh, ww] */
        const int hh = ih[l] - 2, ww = iw[l] - 2;
        float* gy = (float*)malloc(sizeof(float) * (size_t)hid * hh * ww);
        memcpy(gy, g, sizeof(float) * (size_t)hid * hh * ww);
        for (size_t i = 0; i < (size_t)hid * hh * ww; ++i) g[i] *= 0.1f;
        ORC_BACK(1);                                           /* conv2: g -> d t1 */
        const float* t1 = in[l];
        for (size_t i = 0; i < (size_t)hid * ih[l] * iw[l]; ++i) if (!(t1[i] > 0.0f)) g[i] = 0.0f;   /* ReLU */
        --l;
        ORC_BACK(1);                                           /* conv1: -> d x (block input) */
        for (int c = 0; c < hid; ++c)
            for (int y = 0; y < hh; ++y)
                for (int xx = 0; xx < ww; ++xx) g[((size_t)c * ih[l] + y + 2) * iw[l] + xx + 2] += gy[((size_t)c * hh + y) * ww + xx];
        free(gy);
        --l;
    }
    ORC_BACK(dx != NULL);                                      /* conv_input */
    #undef ORC_BACK
    if (dx) { memcpy(dx, g, sizeof(float) * (size_t)Cin * H * W); free(g); }
    for (size_t i = 0; i < nblob; ++i) d_blob[i] = (float)dwd[i];
    free(dwd);
    for (int i = 1; i < nl; ++i) free((void*)in[i]);
    free(in); free(ih); free(iw); free(ci_); free(co_); free(wl);
}

/* backward of orc_planes_sr: d_out [C, sf*R0, sf*R1] (entries outside the ROI are ignored) -> d_blob (overwritten), d_lr [C,R0,R1]
 * (overwritten; NULL = LR plane detached): through the network input (clamped gather, optional normalisation) and through the
 * bilinear residual (models.py:858-859,917) */
ORC_EXPORT void orc_planes_sr_backward(const float* lr, int C, int R0, int R1, const float* blob, int hid, int nblocks, int n_up,
                                       int pad, int over, const float* roi, const float* mean, const float* std_, const float* d_out,
                                       float* d_blob, float* d_lr) {
    const int sf = 1 << n_up;
    int lo[2] = {0, 0}, hi[2] = {R0, R1};
    const int shape[2] = {R0, R1};
    if (roi)
        for (int a = 0; a < 2; ++a) {
            const float mn = (float)shape[a] * (1.0f + roi[a]) / 2.0f, mx = (float)shape[a] * (1.0f + roi[2 + a]) / 2.0f;
            int l = (int)floorf(mn), h = (int)ceilf(mx);
            l = l - 1 > 0 ? l - 1 : 0;
            h = h + 1 < shape[a] ? h + 1 : shape[a];
            lo[a] = l; hi[a] = h;
        }
    const int ch = hi[0] - lo[0], cw = hi[1] - lo[1];
    const int Hp = ch + 2 * pad, Wp = cw + 2 * pad;
    float* x = (float*)malloc(sizeof(float) * (size_t)C * Hp * Wp);
    for (int c = 0; c < C; ++c)
        for (int y = 0; y < Hp; ++y) {
            int sy = lo[0] - pad + y; sy = sy < 0 ? 0 : (sy > R0 - 1 ? R0 - 1 : sy);
            for (int xx = 0; xx < Wp; ++xx) {
                int sx = lo[1] - pad + xx; sx = sx < 0 ? 0 : (sx > R1 - 1 ? R1 - 1 : sx);
                float v = lr[((size_t)c * R0 + sy) * R1 + sx];
                if (mean) v = (v - mean[c]) / std_[c];
                x[((size_t)c * Hp + y) * Wp + xx] = v;
            }
        }
    int Ho, Wo;
    orc_edsr_out_size(Hp, Wp, nblocks, n_up, &Ho, &Wo);
    const size_t HRh = (size_t)R0 * sf, HRw = (size_t)R1 * sf;
    float* d_diff = (float*)calloc((size_t)C * Ho * Wo, sizeof(float));
    for (int c = 0; c < C; ++c)
        for (int y = 0; y < ch * sf; ++y)
            for (int xx = 0; xx < cw * sf; ++xx)
                d_diff[((size_t)c * Ho + y + over) * Wo + xx + over] = d_out[((size_t)c * HRh + (size_t)lo[0] * sf + y) * HRw + (size_t)lo[1] * sf + xx];
    float* dxp = d_lr ? (float*)malloc(sizeof(float) * (size_t)C * Hp * Wp) : NULL;
    orc_edsr_backward(x, C, Hp, Wp, blob, C, hid, nblocks, n_up, d_diff, d_blob, dxp);
    free(x); free(d_diff);
    if (!d_lr) return;
    double* acc = (double*)calloc((size_t)C * R0 * R1, sizeof(double));
    for (int c = 0; c < C; ++c)
        for (int y = 0; y < Hp; ++y) {
            int sy = lo[0] - pad + y; sy = sy < 0 ? 0 : (sy > R0 - 1 ? R0 - 1 : sy);
            for (int xx = 0; xx < Wp; ++xx) {
                int sx = lo[1] - pad + xx; sx = sx < 0 ? 0 : (sx > R1 - 1 ? R1 - 1 : sx);
                double v = dxp[((size_t)c * Hp + y) * Wp + xx];
                if (mean) v /= (double)std_[c];
                acc[((size_t)c * R0 + sy) * R1 + sx] += v;
            }
        }
    free(dxp);
    /* residual path: bilinear x sf, align_corners=True, over the ROI */
    const float shh = HRh > 1 ? (float)(R0 - 1) / (float)(HRh - 1) : 0.0f;
    const float sww = HRw > 1 ? (float)(R1 - 1) / (float)(HRw - 1) : 0.0f;
    for (int c = 0; c < C; ++c)
        for (size_t oy = (size_t)lo[0] * sf; oy < (size_t)hi[0] * sf; ++oy) {
            const float fy = shh * (float)oy;
            const int y0 = (int)fy, yp = (y0 < R0 - 1) ? 1 : 0;
            const float ly1 = fy - (float)y0, ly0 = 1.0f - ly1;
            for (size_t ox = (size_t)lo[1] * sf; ox < (size_t)hi[1] * sf; ++ox) {
                const float fx = sww * (float)ox;
                const int x0 = (int)fx, xp = (x0 < R1 - 1) ? 1 : 0;
                const float lx1 = fx - (float)x0, lx0 = 1.0f - lx1;
                const double g = d_out[((size_t)c * HRh + oy) * HRw + ox];
                double* q = acc + ((size_t)c * R0 + y0) * R1 + x0;
                q[0] += g * ly0 * lx0; q[xp] += g * ly0 * lx1;
                q[(size_t)yp * R1] += g * ly1 * lx0; q[(size_t)yp * R1 + xp] += g * ly1 * lx1;
            }
        }
    for (size_t i = 0; i < (size_t)C * R0 * R1; ++i) d_lr[i] = (float)acc[i];
    free(acc);
}

/* ------------------------------------------------------------------------------------------
 * positional_encoding (nerf_helpers.py:552-575): [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), ...]
 */
ORC_EXPORT void orc_positional_encoding(long P, int D, const float* x, int L, int include_input, float* out) {
    const int stride = (include_input ? D : 0) + 2 * D * L;
    for (long i = 0; i < P; ++i) {
        float* o = out + (size_t)i * stride;
        if (include_input) { memcpy(o, x + (size_t)i * D, sizeof(float) * (size_t)D); o += D; }
        for (int l = 0; l < L; ++l) {
            const float f = (float)ldexp(1.0, l);
            for (int k = 0; k < D; ++k) o[k] = sinf(f * x[(size_t)i * D + k]);
            for (int k = 0; k < D; ++k) o[D + k] = cosf(f * x[(size_t)i * D + k]);
            o += 2 * D;
        }
    }
}

/* FlexibleNeRFModel.forward (models.py:83-108), use_viewdirs=True, num_layers_dir=1, xyz_input_2_dir=False.
 * blob = state-dict order: layer1.{w,b}, layers_xyz.{i}.{w,b}, layers_dir.0.{w,b}, fc_alpha.{w,b}, fc_rgb.{w,b}, fc_feat.{w,b} */
ORC_EXPORT void orc_flexible_nerf(long P, const float* x /*[P,dim_xyz+dim_dir]*/, int dim_xyz, int dim_dir, int hidden,
                                  int num_layers, int skip_every, const float* blob, float* out /*[P,4]*/) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < P; ++i) {
        const float* xi = x + (size_t)i * (dim_xyz + dim_dir);
        float a[1024], b[1024];
        const float* p = blob;
        orc_linear(p, p + (size_t)hidden * dim_xyz, xi, dim_xyz, hidden, 0, a);
        p += (size_t)hidden * dim_xyz + hidden;
        float* cur = a; float* nxt = b;
        int width = hidden;
        for (int l = 0; l < num_layers - 1; ++l) {
            int in = hidden;
            if (l % skip_every == 0 && l > 0) { memcpy(cur + width, xi, sizeof(float) * (size_t)dim_xyz); in = hidden + dim_xyz; }
            orc_linear(p, p + (size_t)hidden * in, cur, in, hidden, 1, nxt);
            p += (size_t)hidden * in + hidden;
            float* t = cur; cur = nxt; nxt = t;
        }
        const int hd = hidden / 2;
        const float* dir_w = p; p += (size_t)hd * (dim_dir + hidden) + hd;
        const float* alpha_w = p; p += hidden + 1;
        const float* rgb_w = p; p += 3 * (size_t)hd + 3;
        const float* feat_w = p;
        float feat[1024 + 64];
        orc_linear(feat_w, feat_w + (size_t)hidden * hidden, cur, hidden, hidden, 1, feat);
        float alpha;
        orc_linear(alpha_w, alpha_w + hidden, cur, hidden, 1, 0, &alpha);
        memcpy(feat + hidden, xi + dim_xyz, sizeof(float) * (size_t)dim_dir);
        float hdir[512];
        orc_linear(dir_w, dir_w + (size_t)hd * (dim_dir + hidden), feat, dim_dir + hidden, hd, 1, hdir);
        orc_linear(rgb_w, rgb_w + 3 * (size_t)hd, hdir, hd, 3, 0, out + 4 * i);
        out[4 * i + 3] = alpha;
    }
}

/* ------------------------------------------------------------------------------------------
 * Backward of one training step with respect to the feature planes (the reference gets it from torch.autograd through
 * train_utils.py:185-282; decoder frozen as in Feature_Planes_Only.yml).  Analytic chain rule in double:
 *   composite (volume_rendering_utils.py:18-49) -> sigma/rgb heads -> ReLU MLPs (models.py:395-421) -> 'avg' / 'concat_pos'
 *   combination (models.py:355-379) -> transposed bilinear taps of grid_sample (models.py:303-326).
 * Pinned against tests/golden/g11_grads.npz (autograd of the reference itself).
 */
typedef struct {
    int ix, iy, x1ok, y1ok;
    double w[4]; /* nw, ne, sw, se */
} orc_taps;

static void orc_plane_taps(int Hp, int Wp, float gx, float gy, orc_taps* t) {
    const float x = orc_grid_loc(gx, Wp), y = orc_grid_loc(gy, Hp);
    const float xw = floorf(x), yn = floorf(y);
    const float w = x - xw, e = 1.0f - w, n = y - yn, s = 1.0f - n;
    t->w[0] = s * e; t->w[1] = s * w; t->w[2] = n * e; t->w[3] = n * w;
    t->ix = (int)xw; t->iy = (int)yn;
    t->x1ok = (t->ix + 1 <= Wp - 1); t->y1ok = (t->iy + 1 <= Hp - 1);
}

#define ORC_MAXH 256
typedef struct {
    orc_taps taps[4];
    double in_rgb[4 * 64], in_den[64];
    double h_den[8][ORC_MAXH], h_rgb[8][ORC_MAXH]; /* post-ReLU activations per layer */
    double raw[4];
} orc_act;

static void orc_lin_d(const float* W, const float* b, const double* x, int in, int out, int relu, double* y) {
    for (int o = 0; o < out; ++o) {
        double a = b[o];
        const float* w = W + (size_t)o * in;
        for (int k = 0; k < in; ++k) a += x[k] * (double)w[k];
        y[o] = (relu && a < 0) ? 0.0 : a;
    }
}

static void orc_forward_store(const orc_scene* sc, const orc_decoder* dec, const float* x6, orc_act* A) {
    const int C = dec->C, h = dec->hidden;
    float x5[5], n5[5];
    x5[0] = x6[0]; x5[1] = x6[1]; x5[2] = x6[2];
    x5[3] = atan2f(x6[4], x6[3]);
    x5[4] = atan2f(x6[5], sqrtf(x6[3] * x6[3] + x6[4] * x6[4]));
    for (int i = 0; i < 5; ++i) n5[i] = 2.0f * (x5[i] - sc->lo[i]) / sc->range[i] - 1.0f;
    for (int d = 0; d < 4; ++d) {
        float gx, gy;
        if (d < 3) {
            const float* M = sc->proj[d];
            gx = n5[0] * M[0] + n5[1] * M[2] + n5[2] * M[4];
            gy = n5[0] * M[1] + n5[1] * M[3] + n5[2] * M[5];
        } else { gx = n5[3]; gy = n5[4]; }
        orc_taps* t = &A->taps[d];
        orc_plane_taps(sc->ph[d], sc->pw[d], gx, gy, t);
        const int Hp = sc->ph[d], Wp = sc->pw[d];
        const size_t HW = (size_t)Hp * Wp;
        for (int c = 0; c < C; ++c) {
            const float* p = sc->planes[d] + (size_t)c * HW;
            double v = p[(size_t)t->iy * Wp + t->ix] * t->w[0];
            if (t->x1ok) v += p[(size_t)t->iy * Wp + t->ix + 1] * t->w[1];
            if (t->y1ok) v += p[(size_t)(t->iy + 1) * Wp + t->ix] * t->w[2];
            if (t->x1ok && t->y1ok) v += p[(size_t)(t->iy + 1) * Wp + t->ix + 1] * t->w[3];
            A->in_rgb[d * C + c] = v;
        }
    }
    for (int c = 0; c < C; ++c) A->in_den[c] = (A->in_rgb[c] + A->in_rgb[C + c] + A->in_rgb[2 * C + c]) / 3.0;
    const float* p = dec->blob;
    const double* cur = A->in_den;
    int in = C;
    for (int l = 0; l < dec->n_density_layers; ++l) {
        orc_lin_d(p, p + (size_t)h * in, cur, in, h, 1, A->h_den[l]);
        p += (size_t)h * in + h; cur = A->h_den[l]; in = h;
    }
    orc_lin_d(p, p + h, cur, h, 1, 0, &A->raw[3]);
    p += h + 1;
    cur = A->in_rgb; in = 4 * C;
    for (int l = 0; l < dec->n_rgb_layers; ++l) {
        orc_lin_d(p, p + (size_t)h * in, cur, in, h, 1, A->h_rgb[l]);
        p += (size_t)h * in + h; cur = A->h_rgb[l]; in = h;
    }
    orc_lin_d(p, p + 3 * h, cur, h, 3, 0, A->raw);
}

/* g_raw[4] -> scatter-add into gp[4] (double, NCHW like the planes) */
/* gw (optional): d/d(decoder parameters), accumulated in the blob's own layout (weight[out,in], bias per layer) */
static void orc_backward_point(const orc_scene* sc, const orc_decoder* dec, const orc_act* A, const double* g_raw, double* const* gp,
                               double* gw) {
    const int C = dec->C, h = dec->hidden, nd = dec->n_density_layers, nr = dec->n_rgb_layers;
    const float* Wd[8]; const float* Wr[8];
    const float* p = dec->blob;
    int in = C;
    for (int l = 0; l < nd; ++l) { Wd[l] = p; p += (size_t)h * in + h; in = h; }
    const float* Wa = p; p += h + 1;
    in = 4 * C;
    for (int l = 0; l < nr; ++l) { Wr[l] = p; p += (size_t)h * in + h; in = h; }
    const float* Wc = p;
    double g[ORC_MAXH], gn[ORC_MAXH], g_in_rgb[4 * 64], g_in_den[64];
    /* rgb branch */
    if (gw) {
        double* q = gw + (Wc - dec->blob);
        for (int c = 0; c < 3; ++c) {
            for (int f = 0; f < h; ++f) q[(size_t)c * h + f] += g_raw[c] * A->h_rgb[nr - 1][f];
            q[(size_t)3 * h + c] += g_raw[c];
        }
    }
    for (int f = 0; f < h; ++f) {
        double a = 0;
        for (int c = 0; c < 3; ++c) a += (double)Wc[(size_t)c * h + f] * g_raw[c];
        g[f] = A->h_rgb[nr - 1][f] > 0 ? a : 0.0;
    }
    for (int l = nr - 1; l >= 1; --l) {
        if (gw) {   /* g = dL/d(pre-activation of layer l); its input is the post-ReLU output of layer l-1 */
            double* q = gw + (Wr[l] - dec->blob);
            for (int o = 0; o < h; ++o) {
                if (g[o] == 0.0) continue;
                for (int k = 0; k < h; ++k) q[(size_t)o * h + k] += g[o] * A->h_rgb[l - 1][k];
                q[(size_t)h * h + o] += g[o];
            }
        }
        for (int k = 0; k < h; ++k) {
            double a = 0;
            for (int o = 0; o < h; ++o) a += (double)Wr[l][(size_t)o * h + k] * g[o];
            gn[k] = A->h_rgb[l - 1][k] > 0 ? a : 0.0;
        }
        memcpy(g, gn, sizeof(double) * (size_t)h);
    }
    if (gw) {
        double* q = gw + (Wr[0] - dec->blob);
        for (int o = 0; o < h; ++o) {
            if (g[o] == 0.0) continue;
            for (int k = 0; k < 4 * C; ++k) q[(size_t)o * 4 * C + k] += g[o] * A->in_rgb[k];
            q[(size_t)h * 4 * C + o] += g[o];
        }
    }
    for (int k = 0; k < 4 * C; ++k) {
        double a = 0;
        for (int o = 0; o < h; ++o) a += (double)Wr[0][(size_t)o * 4 * C + k] * g[o];
        g_in_rgb[k] = a;
    }
    /* density branch */
    if (gw) {
        double* q = gw + (Wa - dec->blob);
        for (int f = 0; f < h; ++f) q[f] += g_raw[3] * A->h_den[nd - 1][f];
        q[h] += g_raw[3];
    }
    for (int f = 0; f < h; ++f) g[f] = A->h_den[nd - 1][f] > 0 ? (double)Wa[f] * g_raw[3] : 0.0;
    for (int l = nd - 1; l >= 1; --l) {
        if (gw) {
            double* q = gw + (Wd[l] - dec->blob);
            for (int o = 0; o < h; ++o) {
                if (g[o] == 0.0) continue;
                for (int k = 0; k < h; ++k) q[(size_t)o * h + k] += g[o] * A->h_den[l - 1][k];
                q[(size_t)h * h + o] += g[o];
            }
        }
        for (int k = 0; k < h; ++k) {
            double a = 0;
            for (int o = 0; o < h; ++o) a += (double)Wd[l][(size_t)o * h + k] * g[o];
            gn[k] = A->h_den[l - 1][k] > 0 ? a : 0.0;
        }
        memcpy(g, gn, sizeof(double) * (size_t)h);
    }
    if (gw) {
        double* q = gw + (Wd[0] - dec->blob);
        for (int o = 0; o < h; ++o) {
            if (g[o] == 0.0) continue;
            for (int k = 0; k < C; ++k) q[(size_t)o * C + k] += g[o] * A->in_den[k];
            q[(size_t)h * C + o] += g[o];
        }
    }
    for (int k = 0; k < C; ++k) {
        double a = 0;
        for (int o = 0; o < h; ++o) a += (double)Wd[0][(size_t)o * C + k] * g[o];
        g_in_den[k] = a;
    }
    for (int d = 0; d < 4; ++d) {
        const orc_taps* t = &A->taps[d];
        const int Hp = sc->ph[d], Wp = sc->pw[d];
        const size_t HW = (size_t)Hp * Wp;
        for (int c = 0; c < C; ++c) {
            const double gf = g_in_rgb[d * C + c] + (d < 3 ? g_in_den[c] / 3.0 : 0.0);
            double* q = gp[d] + (size_t)c * HW;
            q[(size_t)t->iy * Wp + t->ix] += t->w[0] * gf;
            if (t->x1ok) q[(size_t)t->iy * Wp + t->ix + 1] += t->w[1] * gf;
            if (t->y1ok) q[(size_t)(t->iy + 1) * Wp + t->ix] += t->w[2] * gf;
            if (t->x1ok && t->y1ok) q[(size_t)(t->iy + 1) * Wp + t->ix + 1] += t->w[3] * gf;
        }
    }
}

/* d(rgb, acc)/d raw of one ray; g_rgb[3], g_acc upstream; g_raw [S,4] out */
static void orc_composite_backward_ray(int S, const double* raw /*[S,4]*/, const float* z, const float* rd3, const float* noise, int white,
                                       const double* g_rgb, double g_acc, double* g_raw) {
    const double nrm = sqrt((double)rd3[0] * rd3[0] + (double)rd3[1] * rd3[1] + (double)rd3[2] * rd3[2]);
    double* alpha = (double*)malloc(sizeof(double) * 6 * (size_t)S);
    double *T = alpha + S, *w = alpha + 2 * S, *gw = alpha + 3 * S, *dist = alpha + 4 * S, *sig = alpha + 5 * S;
    double Tr = 1.0;
    for (int s = 0; s < S; ++s) {
        dist[s] = ((s == S - 1) ? 1e10 : ((double)z[s + 1] - (double)z[s])) * nrm;
        double sg = raw[4 * s + 3] + (noise ? (double)noise[s] : 0.0);
        sig[s] = sg;
        sg = sg > 0 ? sg : 0;
        alpha[s] = 1.0 - exp(-sg * dist[s]);
        T[s] = Tr;
        w[s] = alpha[s] * Tr;
        Tr *= (1.0 - alpha[s] + 1e-10);
    }
    const double ga = g_acc - (white ? (g_rgb[0] + g_rgb[1] + g_rgb[2]) : 0.0);
    double suffix = 0.0;   /* sum_{k>s} w_k * dL/dw_k */
    for (int s = S - 1; s >= 0; --s) {
        double c[3];
        for (int k = 0; k < 3; ++k) c[k] = 1.0 / (1.0 + exp(-raw[4 * s + k]));
        gw[s] = g_rgb[0] * c[0] + g_rgb[1] * c[1] + g_rgb[2] * c[2] + ga;
        for (int k = 0; k < 3; ++k) g_raw[4 * s + k] = w[s] * g_rgb[k] * c[k] * (1.0 - c[k]);
        const double g_alpha = T[s] * gw[s] - suffix / (1.0 - alpha[s] + 1e-10);
        g_raw[4 * s + 3] = (sig[s] > 0) ? g_alpha * dist[s] * (1.0 - alpha[s]) : 0.0;
        suffix += w[s] * gw[s];
    }
    free(alpha);
}

/* One train step: forward recomputed in double, gradient of (sum g_rgb_c . rgb_c + sum g_rgb_f . rgb_f) wrt the 4 planes.
 * grad planes: NCHW float [C,H,W] each (overwritten). */
static void orc_render_backward_impl(const orc_scene* sc, const orc_decoder* coarse, const orc_decoder* fine, const orc_render_cfg* cfg,
                                     long N, const float* rays, const float* t_rand, const float* u, const float* noise_c,
                                     const float* noise_f, const float* g_rgb_c, const float* g_rgb_f, const float* z_fine_in,
                                     float* gp0, float* gp1, float* gp2, float* gpv, float* gdec_c, float* gdec_f) {
    const int Nc = cfg->num_coarse, Nf = cfg->num_fine, St = Nc + Nf, C = coarse->C;
    float* gout[4] = {gp0, gp1, gp2, gpv};
    double* gwc = gdec_c ? (double*)calloc(orc_decoder_floats(coarse), sizeof(double)) : NULL;
    double* gwf = (gdec_f && fine) ? (double*)calloc(orc_decoder_floats(fine), sizeof(double)) : NULL;
    double* gp[4];
    for (int d = 0; d < 4; ++d) gp[d] = (double*)calloc((size_t)C * sc->ph[d] * sc->pw[d], sizeof(double));
    orc_act* acts = (orc_act*)malloc(sizeof(orc_act) * (size_t)(St > Nc ? St : Nc));
    float* z = (float*)malloc(sizeof(float) * (size_t)(St + 4));
    float* zs = (float*)malloc(sizeof(float) * (size_t)(Nf + 4));
    float* zm = (float*)malloc(sizeof(float) * (size_t)(Nc + 4));
    float* rawf = (float*)malloc(sizeof(float) * 4 * (size_t)(St + 4));
    double* rawd = (double*)malloc(sizeof(double) * 4 * (size_t)(St + 4));
    double* graw = (double*)malloc(sizeof(double) * 4 * (size_t)(St + 4));
    float* w = (float*)malloc(sizeof(float) * (size_t)(St + 4));
    float* ud = (float*)malloc(sizeof(float) * (size_t)(Nf + 4));
    for (long i = 0; i < N; ++i) {
        const float* r = rays + 11 * i;
        float x6[6], rgb3[3], dsp, ac, dep;
        x6[3] = r[8]; x6[4] = r[9]; x6[5] = r[10];
        orc_coarse_z(1, Nc, r + 6, r + 7, cfg->lindisp, cfg->perturb, t_rand ? t_rand + (size_t)i * Nc : NULL, z);
        for (int s = 0; s < Nc; ++s) {
            for (int k = 0; k < 3; ++k) x6[k] = r[k] + r[3 + k] * z[s];
            orc_forward_store(sc, coarse, x6, &acts[s]);
            for (int k = 0; k < 4; ++k) { rawd[4 * s + k] = acts[s].raw[k]; rawf[4 * s + k] = (float)acts[s].raw[k]; }
        }
        if (g_rgb_c) {
            const double g3[3] = {g_rgb_c[3 * i], g_rgb_c[3 * i + 1], g_rgb_c[3 * i + 2]};
            orc_composite_backward_ray(Nc, rawd, z, r + 3, noise_c ? noise_c + (size_t)i * Nc : NULL, cfg->white_background, g3, 0.0, graw);
            for (int s = 0; s < Nc; ++s) orc_backward_point(sc, coarse, &acts[s], graw + 4 * s, gp, gwc);
        }
        if (Nf <= 0 || !g_rgb_f) continue;
        if (z_fine_in) {
            memcpy(z, z_fine_in + (size_t)i * St, sizeof(float) * (size_t)St);
        } else {
            orc_composite_ray(Nc, rawf, z, r + 3, noise_c ? noise_c + (size_t)i * Nc : NULL, cfg->white_background, rgb3, &dsp, &ac, w, &dep);
            for (int s = 0; s < Nc - 1; ++s) zm[s] = 0.5f * (z[s + 1] + z[s]);
            const float* ui;
            if (u) ui = u + (size_t)i * Nf;
            else { for (int j = 0; j < Nf; ++j) ud[j] = orc_linspace01(j, Nf); ui = ud; }
            orc_sample_pdf_ray(Nc - 1, Nf, zm, w + 1, ui, zs);      /* detached: no gradient through the depths (train_utils.py:153) */
            memcpy(z + Nc, zs, sizeof(float) * (size_t)Nf);
            qsort(z, (size_t)St, sizeof(float), orc_cmp_float);
        }
        for (int s = 0; s < St; ++s) {
            for (int k = 0; k < 3; ++k) x6[k] = r[k] + r[3 + k] * z[s];
            orc_forward_store(sc, fine, x6, &acts[s]);
            for (int k = 0; k < 4; ++k) rawd[4 * s + k] = acts[s].raw[k];
        }
        const double g3[3] = {g_rgb_f[3 * i], g_rgb_f[3 * i + 1], g_rgb_f[3 * i + 2]};
        orc_composite_backward_ray(St, rawd, z, r + 3, noise_f ? noise_f + (size_t)i * St : NULL, cfg->white_background, g3, 0.0, graw);
        for (int s = 0; s < St; ++s) orc_backward_point(sc, fine, &acts[s], graw + 4 * s, gp, gwf);
    }
    for (int d = 0; d < 4; ++d) {
        const size_t n = (size_t)C * sc->ph[d] * sc->pw[d];
        if (gout[d]) for (size_t k = 0; k < n; ++k) gout[d][k] = (float)gp[d][k];
        free(gp[d]);
    }
    if (gwc) { for (size_t k = 0; k < orc_decoder_floats(coarse); ++k) gdec_c[k] = (float)gwc[k]; free(gwc); }
    if (gwf) { for (size_t k = 0; k < orc_decoder_floats(fine); ++k) gdec_f[k] = (float)gwf[k]; free(gwf); }
    free(acts); free(z); free(zs); free(zm); free(rawf); free(rawd); free(graw); free(w); free(ud);
}

ORC_EXPORT void orc_render_backward(const orc_scene* sc, const orc_decoder* coarse, const orc_decoder* fine, const orc_render_cfg* cfg,
                                    long N, const float* rays, const float* t_rand, const float* u, const float* noise_c,
                                    const float* noise_f, const float* g_rgb_c /*[N,3] or NULL*/, const float* g_rgb_f /*[N,3] or NULL*/,
                                    const float* z_fine_in /*[N,Nc+Nf] or NULL: use these fine depths instead of resampling*/,
                                    float* gp0, float* gp1, float* gp2, float* gpv) {
    orc_render_backward_impl(sc, coarse, fine, cfg, N, rays, t_rand, u, noise_c, noise_f, g_rgb_c, g_rgb_f, z_fine_in, gp0, gp1, gp2, gpv,
                             NULL, NULL);
}

/* Same step, additionally d/d(decoder parameters) of both models (what: ['decoder'], train_nerf.py:75-77), each in the blob's
 * state-dict order.  Pinned against tests/golden/g13_decoder_grads.npz.  Plane outputs may be NULL. */
ORC_EXPORT void orc_render_backward_dec(const orc_scene* sc, const orc_decoder* coarse, const orc_decoder* fine, const orc_render_cfg* cfg,
                                        long N, const float* rays, const float* t_rand, const float* u, const float* noise_c,
                                        const float* noise_f, const float* g_rgb_c, const float* g_rgb_f, const float* z_fine_in,
                                        float* gp0, float* gp1, float* gp2, float* gpv, float* gdec_c, float* gdec_f) {
    orc_render_backward_impl(sc, coarse, fine, cfg, N, rays, t_rand, u, noise_c, noise_f, g_rgb_c, g_rgb_f, z_fine_in, gp0, gp1, gp2, gpv,
                             gdec_c, gdec_f);
}
