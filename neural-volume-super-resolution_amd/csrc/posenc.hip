// positional_encoding (nerf_helpers.py:552-575) and the positional-encoding NeRF MLP FlexibleNeRFModel.forward
// (models.py:83-108) -- the baseline model of MipNeRF_baseline.yml, kept for interface completeness (a12, a13).
// Not on the tri-plane hot path: one thread per point, activations in LDS rows, weights through the scalar cache.
#include "nvsr_common.h"

namespace nvsr {

__global__ void posenc_kernel(long P, int D, const float* __restrict__ x, int L, int include_input, float* __restrict__ out) {
    const int stride = (include_input ? D : 0) + 2 * D * L;
    const long n = P * stride;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long p = i / stride;
    int c = (int)(i - p * stride);
    if (include_input) {
        if (c < D) { out[i] = x[p * D + c]; return; }
        c -= D;
    }
    const int l = c / (2 * D), r = c - l * 2 * D;
    const float v = __fmul_rn(ldexpf(1.0f, l), x[p * D + (r < D ? r : r - D)]);   // func(2.0 ** i * tensor)
    out[i] = (r < D) ? sinf(v) : cosf(v);
}

constexpr int NERF_TPB = 64;
constexpr int NERF_MAXW = 256;   // widest activation row (hidden + dim_xyz skip, or hidden + dim_dir)

// y[o] = act(b[o] + sum_k x[k] W[o][k]); x, y are this thread's LDS rows
__device__ __forceinline__ void dense(const float* __restrict__ W, const float* __restrict__ b, const float* x, int in, int out, bool relu,
                                      float* y) {
    for (int o = 0; o < out; ++o) {
        float a = 0.0f;
        const float* w = W + (long)o * in;
        for (int k = 0; k < in; ++k) a = fmaf(x[k], w[k], a);
        a += b[o];
        y[o] = (relu && a < 0.0f) ? 0.0f : a;
    }
}

__global__ __launch_bounds__(NERF_TPB) void flexible_nerf_kernel(long P, const float* __restrict__ x, int dim_xyz, int dim_dir, int hidden,
                                                                 int num_layers, int skip_every, const float* __restrict__ blob,
                                                                 float* __restrict__ out) {
    __shared__ float bufA[NERF_TPB][NERF_MAXW + 1], bufB[NERF_TPB][NERF_MAXW + 1];
    const long p = (long)blockIdx.x * NERF_TPB + threadIdx.x;
    if (p >= P) return;
    const float* xi = x + p * (dim_xyz + dim_dir);
    float* cur = bufA[threadIdx.x];
    float* nxt = bufB[threadIdx.x];
    const float* w = blob;
    for (int k = 0; k < dim_xyz; ++k) nxt[k] = xi[k];
    dense(w, w + (long)hidden * dim_xyz, nxt, dim_xyz, hidden, false, cur);           // layer1: no activation (models.py:88)
    w += (long)hidden * dim_xyz + hidden;
    for (int l = 0; l < num_layers - 1; ++l) {
        int in = hidden;
        if (l % skip_every == 0 && l > 0) {                                             // models.py:90-95
            for (int k = 0; k < dim_xyz; ++k) cur[hidden + k] = xi[k];
            in = hidden + dim_xyz;
        }
        dense(w, w + (long)hidden * in, cur, in, hidden, true, nxt);
        w += (long)hidden * in + hidden;
        float* t = cur; cur = nxt; nxt = t;
    }
    const int hd = hidden / 2;
    const float* dir_w = w;   w += (long)hd * (dim_dir + hidden) + hd;
    const float* alpha_w = w; w += hidden + 1;
    const float* rgb_w = w;   w += 3 * hd + 3;
    const float* feat_w = w;
    float alpha;
    dense(alpha_w, alpha_w + hidden, cur, hidden, 1, false, &alpha);
    dense(feat_w, feat_w + (long)hidden * hidden, cur, hidden, hidden, true, nxt);     // feat = relu(fc_feat(x))
    for (int k = 0; k < dim_dir; ++k) nxt[hidden + k] = xi[dim_xyz + k];                // cat(feat, view)
    dense(dir_w, dir_w + (long)hd * (dim_dir + hidden), nxt, dim_dir + hidden, hd, true, cur);
    float rgb[3];
    dense(rgb_w, rgb_w + 3 * hd, cur, hd, 3, false, rgb);
    out[p * 4 + 0] = rgb[0]; out[p * 4 + 1] = rgb[1]; out[p * 4 + 2] = rgb[2]; out[p * 4 + 3] = alpha;
}

}  // namespace nvsr

using namespace nvsr;

extern "C" {

int nvsr_positional_encoding(int64_t P, int D, const float* x, int L, int include_input, float* out, nvsr_stream_t stream) {
    if (!x || !out) return NVSR_ERR_NULL;
    if (P < 0 || D < 1 || L < 0 || L > 30) return NVSR_ERR_SHAPE;
    const int64_t n = P * ((include_input ? D : 0) + 2 * D * L);
    if (n == 0) return NVSR_OK;
    hipLaunchKernelGGL(posenc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long)P, D, x, L, include_input, out);
    return NVSR_CHECK_LAUNCH();
}

int nvsr_flexible_nerf_forward(int64_t P, const float* x, int dim_xyz, int dim_dir, int hidden, int num_layers, int skip_every,
                               const float* blob, float* out, nvsr_stream_t stream) {
    if (!x || !blob || !out) return NVSR_ERR_NULL;
    if (P < 0 || dim_xyz < 1 || dim_dir < 1 || hidden < 2 || hidden % 2 || num_layers < 1 || skip_every < 1) return NVSR_ERR_SHAPE;
    if (hidden + dim_xyz > NERF_MAXW || hidden + dim_dir > NERF_MAXW) return NVSR_ERR_SHAPE;
    if (P == 0) return NVSR_OK;
    hipLaunchKernelGGL(flexible_nerf_kernel, dim3((unsigned)((P + NERF_TPB - 1) / NERF_TPB)), dim3(NERF_TPB), 0, (hipStream_t)stream, (long)P, x,
                       dim_xyz, dim_dir, hidden, num_layers, skip_every, blob, out);
    return NVSR_CHECK_LAUNCH();
}

}  // extern "C"
