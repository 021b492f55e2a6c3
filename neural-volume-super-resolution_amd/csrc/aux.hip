// Ray generation, depth sampling, inverse-CDF importance sampling, row sort, compositing and plane re-layout kernels (gfx950).
// These are the HBM-bound stages of the path: one wavefront (64 lanes) per ray, coalesced row reads, wave scans in registers.
//
// Reference functions replaced (upstream paths): get_ray_bundle nerf_helpers.py:507-549, ndc_rays :578-605,
// sample_pdf_2 :668-702, cumprod_exclusive :409-430, predict_and_render_radiance train_utils.py:95-109,144-155,
// run_one_iter_of_nerf train_utils.py:213-226, volume_render_radiance_field volume_rendering_utils.py:6-51.
#include <cstdlib>
#include "nvsr_common.h"
#include "wave_scan.h"

namespace nvsr {

constexpr int WPB = 4;  // waves per block for the wave-per-ray kernels

// ---- plane layout ------------------------------------------------------------------------------------------------------
// [C, HW] -> [HW, C]: a block moves 64 pixels x all channels through LDS so that both sides are coalesced.
__global__ void to_channel_last_kernel(const float* __restrict__ in, float* __restrict__ out, int Cc, long HW) {
    extern __shared__ float tile[];  // [Cc][65]
    const long p0 = (long)blockIdx.x * 64;
    const int npx = (int)min((long)64, HW - p0);
    for (int i = threadIdx.x; i < Cc * 64; i += blockDim.x) {
        const int c = i >> 6, px = i & 63;
        if (px < npx) tile[c * 65 + px] = in[(long)c * HW + p0 + px];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < npx * Cc; i += blockDim.x) {
        const int px = i / Cc, c = i - px * Cc;
        out[p0 * Cc + i] = tile[c * 65 + px];
    }
}
__global__ void from_channel_last_kernel(const float* __restrict__ in, float* __restrict__ out, int Cc, long HW) {
    extern __shared__ float tile[];
    const long p0 = (long)blockIdx.x * 64;
    const int npx = (int)min((long)64, HW - p0);
    for (int i = threadIdx.x; i < npx * Cc; i += blockDim.x) {
        const int px = i / Cc, c = i - px * Cc;
        tile[c * 65 + px] = in[p0 * Cc + i];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < Cc * 64; i += blockDim.x) {
        const int c = i >> 6, px = i & 63;
        if (px < npx) out[(long)c * HW + p0 + px] = tile[c * 65 + px];
    }
}

// ---- rays --------------------------------------------------------------------------------------------------------------
__global__ void ray_bundle_kernel(int H, int W, float fx, float fy, const float* __restrict__ c2w, int pad, float off,
                                  float* __restrict__ ro, float* __restrict__ rd) {
    const int Wp = W + 2 * pad;
    const long n = (long)(H + 2 * pad) * Wp;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r = (int)(i / Wp), c = (int)(i - (long)r * Wp);
    float ii = __fadd_rn((float)c, off), jj = __fadd_rn((float)r, off);
    if (pad > 0) { ii = __fsub_rn(ii, (float)pad); jj = __fsub_rn(jj, (float)pad); }
    const float d0 = __fdiv_rn(__fsub_rn(ii, (float)(W * 0.5)), fx);
    const float d1 = -__fdiv_rn(__fsub_rn(jj, (float)(H * 0.5)), fy);
    const float d2 = -1.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        // torch.sum over the 3 products: accumulator starts at +0.0 (so an all-(-0.0) row sums to +0.0, which matters to atan2 in
        // cart2az_el), left to right, no FMA contraction -- bit-exact with the reference on CPU including the sign of zero
        rd[i * 3 + k] = __fadd_rn(__fadd_rn(__fadd_rn(0.0f, __fmul_rn(d0, c2w[k * 4 + 0])), __fmul_rn(d1, c2w[k * 4 + 1])), __fmul_rn(d2, c2w[k * 4 + 2]));
        ro[i * 3 + k] = c2w[k * 4 + 3];
    }
}

// get_ray_bundle(...)[rows, cols] of train_nerf.py:814,842-844 without generating the other H*W - N rays: same arithmetic per
// ray as ray_bundle_kernel (bit-identical), driven by the selected pixel coordinates
__global__ void ray_bundle_at_kernel(int H, int W, float fx, float fy, const float* __restrict__ c2w, float off, long N,
                                     const int* __restrict__ rc, float* __restrict__ ro, float* __restrict__ rd) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int r = rc[2 * i], c = rc[2 * i + 1];
    const float ii = __fadd_rn((float)c, off), jj = __fadd_rn((float)r, off);
    const float d0 = __fdiv_rn(__fsub_rn(ii, (float)(W * 0.5)), fx);
    const float d1 = -__fdiv_rn(__fsub_rn(jj, (float)(H * 0.5)), fy);
    const float d2 = -1.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        rd[i * 3 + k] = __fadd_rn(__fadd_rn(__fadd_rn(0.0f, __fmul_rn(d0, c2w[k * 4 + 0])), __fmul_rn(d1, c2w[k * 4 + 1])), __fmul_rn(d2, c2w[k * 4 + 2]));
        ro[i * 3 + k] = c2w[k * 4 + 3];
    }
}

// ---- training inputs ---------------------------------------------------------------------------------------------------
// Keyed bijection of [0, 2^(2*hb)): an 8-round balanced Feistel network over two hb-bit halves (nvsr.h: nvsr_sample_pixels states the
// round function bit by bit; the CPU checker restates it).  Walking the cycle until the value is below `total` restricts it to a
// bijection of [0, total): entries [first, first + n) of that permutation are n DISTINCT pixels, uniform over the keys.
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ __forceinline__ unsigned long long feistel8(unsigned long long x, int hb, const unsigned* __restrict__ rk) {
    const unsigned mask = (hb >= 32) ? 0xFFFFFFFFu : ((1u << hb) - 1u);
    unsigned L = (unsigned)(x >> hb) & mask, R = (unsigned)x & mask;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        unsigned h = R * 0x9E3779B1u + rk[r];
        h ^= h >> 15; h *= 0x85EBCA77u;
        h ^= h >> 13; h *= 0xC2B2AE3Du;
        h ^= h >> 16;
        const unsigned t = L ^ (h & mask);
        L = R; R = t;
    }
    return ((unsigned long long)L << hb) | R;
}

__global__ void sample_pixels_kernel(long total, int hb, int H, int Wd, unsigned long long key, long first, long n,
                                     const float* __restrict__ image, int C, int* __restrict__ rc, float* __restrict__ target) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned rk[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) rk[r] = (unsigned)(splitmix64(key + (unsigned long long)r) >> 32);
    unsigned long long x = (unsigned long long)(first + i);
    do { x = feistel8(x, hb, rk); } while (x >= (unsigned long long)total);      // (x started below total: its cycle comes back below it)
    // coords = stack(meshgrid_xy(arange(H), arange(W)), -1).reshape(-1, 2) enumerates pixels column by column (train_nerf.py:818-828)
    const int row = (int)(x % (unsigned long long)H), col = (int)(x / (unsigned long long)H);
    rc[2 * i] = row; rc[2 * i + 1] = col;
    if (target) {
        const float* px = image + ((long)row * Wd + col) * C;
        for (int c = 0; c < C; ++c) target[i * C + c] = px[c];
    }
}

// key of draw number `calls` of a sampler seeded with `seed` (nvsr.h: nvsr_sample_key)
__host__ __device__ inline unsigned long long sample_key(unsigned long long seed, unsigned long long calls) {
    auto mix = [](unsigned long long x) {
        x += 0x9E3779B97F4A7C15ull;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        return x ^ (x >> 31);
    };
    return mix(mix(seed) ^ calls);
}

// nvsr_sample_pixels with the key derived on the device from state = {seed, calls, arrivals, -}: a launch that a HIP graph replays draws
// a new batch every replay.  Every thread reads `calls` before its workgroup arrives at the counter; the workgroup that arrives last
// (all others have read) advances `calls` and resets the counter for the next launch on the stream.
__global__ void sample_pixels_seq_kernel(long total, int hb, int H, int Wd, unsigned long long* __restrict__ state, long first, long n,
                                         const float* __restrict__ image, int C, int* __restrict__ rc, float* __restrict__ target) {
    const unsigned long long seed = __hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long calls = __hip_atomic_load(state + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long key = sample_key(seed, calls);
    __syncthreads();                       // every thread of the workgroup holds `calls` in a register
    if (threadIdx.x == 0) {
        const unsigned long long prev = __hip_atomic_fetch_add(state + 2, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1 == (unsigned long long)gridDim.x) {
            __hip_atomic_store(state + 2, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(state + 1, calls + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned rk[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) rk[r] = (unsigned)(splitmix64(key + (unsigned long long)r) >> 32);
    unsigned long long x = (unsigned long long)(first + i);
    do { x = feistel8(x, hb, rk); } while (x >= (unsigned long long)total);
    const int row = (int)(x % (unsigned long long)H), col = (int)(x / (unsigned long long)H);
    rc[2 * i] = row; rc[2 * i + 1] = col;
    if (target) {
        const float* px = image + ((long)row * Wd + col) * C;
        for (int c = 0; c < C; ++c) target[i * C + c] = px[c];
    }
}

// img2mse of the coarse and the fine image against one target in ONE launch, with the gradients the backward will want
// (train_nerf.py:893-905: two F.mse_loss calls; here 2 (x - t) / n is written beside the forward sums): one 1024-thread workgroup.
// with_sum: losses[2] = losses[0] + losses[1] (the iteration's loss, train_nerf.py:905: coarse_loss + fine_loss -- one f32 addition, like torch's)
__global__ void __launch_bounds__(1024) mse_pair_kernel(long n, const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ t,
                                                        float* __restrict__ losses, float* __restrict__ ga, float* __restrict__ gb, int with_sum) {
    __shared__ float red[2][16];
    const float inv = 1.0f / (float)n, two_inv = 2.0f / (float)n;
    float sa = 0.0f, sb = 0.0f;
    for (long i = threadIdx.x; i < n; i += 1024) {
        const float tv = t[i];
        const float da = a[i] - tv;
        sa += da * da;
        if (ga) ga[i] = two_inv * da;
        if (b) {
            const float db = b[i] - tv;
            sb += db * db;
            if (gb) gb[i] = two_inv * db;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sa += __shfl_xor(sa, o); sb += __shfl_xor(sb, o); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = sa; red[1][wave] = sb; }
    __syncthreads();
    if (threadIdx.x < 2) {
        float s = 0.0f;
        for (int w = 0; w < 16; ++w) s += red[threadIdx.x][w];
        if (threadIdx.x == 0 || b) losses[threadIdx.x] = s * inv;
        if (with_sum && b) {          // (threads 0 and 1 are lanes of one wave)
            const float other = __shfl_xor(s * inv, 1);
            if (threadIdx.x == 0) losses[2] = __fadd_rn(s * inv, other);
        }
    }
}

// gradients of the two losses of mse_pair_kernel with the incoming gradients folded in: ga = (2 (a - t) / n) * *scale_a, gb likewise (a NULL scale
// is 1; the product is rounded after the gradient, like the two torch multiplies this launch replaces)
__global__ void __launch_bounds__(256) mse_pair_backward_kernel(long n, const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ t,
                                                                const float* __restrict__ scale_a, const float* __restrict__ scale_b,
                                                                float* __restrict__ ga, float* __restrict__ gb) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float two_inv = 2.0f / (float)n;
    const float tv = t[i];
    if (ga) { const float g = __fmul_rn(two_inv, a[i] - tv); ga[i] = scale_a ? __fmul_rn(g, *scale_a) : g; }
    if (gb) { const float g = __fmul_rn(two_inv, b[i] - tv); gb[i] = scale_b ? __fmul_rn(g, *scale_b) : g; }
}

__global__ void ndc_rays_kernel(float sx, float sy, float nr, float two_near, float m_two_near, long N, const float* __restrict__ ro,
                                const float* __restrict__ rd, float* __restrict__ ro_out, float* __restrict__ rd_out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float o0 = ro[3 * i], o1 = ro[3 * i + 1], o2 = ro[3 * i + 2];
    const float d0 = rd[3 * i], d1 = rd[3 * i + 1], d2 = rd[3 * i + 2];
    const float t = __fdiv_rn(-__fadd_rn(nr, o2), d2);
    const float ox = __fadd_rn(o0, __fmul_rn(t, d0)), oy = __fadd_rn(o1, __fmul_rn(t, d1)), oz = __fadd_rn(o2, __fmul_rn(t, d2));
    ro_out[3 * i + 0] = __fdiv_rn(__fmul_rn(sx, ox), oz);
    ro_out[3 * i + 1] = __fdiv_rn(__fmul_rn(sy, oy), oz);
    ro_out[3 * i + 2] = __fadd_rn(1.0f, __fdiv_rn(two_near, oz));
    rd_out[3 * i + 0] = __fmul_rn(sx, __fsub_rn(__fdiv_rn(d0, d2), __fdiv_rn(ox, oz)));
    rd_out[3 * i + 1] = __fmul_rn(sy, __fsub_rn(__fdiv_rn(d1, d2), __fdiv_rn(oy, oz)));
    rd_out[3 * i + 2] = __fdiv_rn(m_two_near, oz);
}

__global__ void pack_rays_kernel(long N, const float* __restrict__ ro, const float* __restrict__ rd, const float* __restrict__ vs,
                                 float near_, float far_, float* __restrict__ rays) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float* r = rays + 11 * i;
    const float v0 = vs[3 * i], v1 = vs[3 * i + 1], v2 = vs[3 * i + 2];
    const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(v0, v0), __fmul_rn(v1, v1)), __fmul_rn(v2, v2)));
    r[0] = ro[3 * i]; r[1] = ro[3 * i + 1]; r[2] = ro[3 * i + 2];
    r[3] = rd[3 * i]; r[4] = rd[3 * i + 1]; r[5] = rd[3 * i + 2];
    r[6] = near_; r[7] = far_;
    r[8] = __fdiv_rn(v0, nrm); r[9] = __fdiv_rn(v1, nrm); r[10] = __fdiv_rn(v2, nrm);
}


__global__ void coarse_z_kernel(long N, int Nc, const float* __restrict__ rays, int lindisp, const float* __restrict__ t_rand,
                                float* __restrict__ z) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * Nc) return;
    const long ray = i / Nc;
    const int s = (int)(i - ray * Nc);
    const float nr = rays[ray * 11 + 6], fr = rays[ray * 11 + 7];
    const float zc = coarse_depth(nr, fr, s, Nc, lindisp);
    if (!t_rand) { z[i] = zc; return; }
    const float lower = (s == 0) ? zc : __fmul_rn(0.5f, __fadd_rn(zc, coarse_depth(nr, fr, s - 1, Nc, lindisp)));
    const float upper = (s == Nc - 1) ? zc : __fmul_rn(0.5f, __fadd_rn(coarse_depth(nr, fr, s + 1, Nc, lindisp), zc));
    z[i] = __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), t_rand[i]));
}

// ---- inverse-CDF importance sampling: one wave per ray -------------------------------------------------------------------
// cdf (nb entries) is built in LDS by a wave scan; every lane then inverts it for samples lane, lane+64, ...
// `bins`/`w` may come from global memory or LDS.  Returns through out_fn(j, sample).
template <class LoadBin, class LoadW, class Store>
__device__ __forceinline__ void sample_pdf_wave(int nb, int ns, LoadBin bin, LoadW wgt, const float* __restrict__ u, float* cdf /*LDS[nb]*/,
                                                int lane, Store store) {
    const int nw = nb - 1;
    // pdf = (w + 1e-5) / sum(w + 1e-5)
    float part = 0.0f;
    for (int i = lane; i < nw; i += 64) part += __fadd_rn(wgt(i), 1e-5f);
    const float total = wave_sum(part);
    float carry = 0.0f;
    if (lane == 0) cdf[0] = 0.0f;
    for (int base = 0; base < nw; base += 64) {
        const int i = base + lane;
        const float p = (i < nw) ? __fdiv_rn(__fadd_rn(wgt(i), 1e-5f), total) : 0.0f;
        const float sc = wave_scan_add(p, lane) + carry;
        if (i < nw) cdf[i + 1] = sc;
        carry = __shfl(sc, 63);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): cdf visible to the whole wave
    for (int j = lane; j < ns; j += 64) {
        const float uj = u ? u[j] : linspace01(j, ns);
        int lo = 0, hi = nb;                 // searchsorted(cdf, u, right=True): first index with cdf[idx] > u
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] > uj) hi = mid; else lo = mid + 1; }
        const int below = max(lo - 1, 0), above = min(lo, nb - 1);
        const float c0 = cdf[below], c1 = cdf[above];
        float denom = __fsub_rn(c1, c0);
        if (denom < 1e-5f) denom = 1.0f;
        const float t = __fdiv_rn(__fsub_rn(uj, c0), denom);
        const float b0 = bin(below), b1 = bin(above);
        store(j, __fadd_rn(b0, __fmul_rn(t, __fsub_rn(b1, b0))));
    }
}

__global__ __launch_bounds__(WPB * 64) void sample_pdf_kernel(long N, int nb, int ns, const float* __restrict__ bins,
                                                             const float* __restrict__ weights, const float* __restrict__ u,
                                                             float* __restrict__ samples) {
    __shared__ float cdf_s[WPB][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long ray = (long)blockIdx.x * WPB + wave;
    if (ray >= N) return;
    const float* b = bins + ray * nb;
    const float* w = weights + ray * (nb - 1);
    float* out = samples + ray * ns;
    sample_pdf_wave(nb, ns, [&](int i) { return b[i]; }, [&](int i) { return w[i]; }, u ? u + ray * ns : nullptr, cdf_s[wave], lane,
                    [&](int j, float v) { out[j] = v; });
}

// rank sort of n <= 512 values held in LDS: rank = #smaller + #equal-with-lower-index (values only matter, like torch.sort).  NaNs order
// like torch.sort's: after every number, among themselves by index (round 3: with plain < / == every NaN got rank 0 and the row lost
// elements).
__device__ __forceinline__ void rank_sort_wave(const float* vals /*LDS[n]*/, int n, float* __restrict__ out, int lane) {
    for (int e = lane; e < n; e += 64) {
        const float v = vals[e];
        const bool vnan = v != v;
        int rank = 0;
        for (int k = 0; k < n; ++k) {
            const float o = vals[k];
            const bool onan = o != o;
            rank += (o < v) || (!onan && vnan) || ((o == v || (onan && vnan)) && k < e);
        }
        out[rank] = v;
    }
}

// sort(cat(a, b)) of two runs held in LDS (all = [a[0..na) | b[0..nb)]), same ranks as rank_sort_wave.  The coarse depths are always
// non-decreasing (linspace, stratified jitter) and so are the inverse-CDF samples of a sorted u (perturb off): an element of a sorted run
// has rank = its own index + a count in the other run, a binary search instead of n comparisons.  Run a sorted: a[i] -> i + #(b < a[i])
// and b[j] -> #(a <= b[j]) + (j when b is sorted as well, else its rank among b).  Anything else (a NaN, a caller's unsorted depths)
// takes the general rank sort.
__device__ __forceinline__ void merge_sort_wave(const float* all /*LDS[na+nb]*/, int na, int nb, float* __restrict__ out, int lane) {
    const float* a = all;
    const float* b = all + na;
    bool oka = true, okb = true, finite = true;
    for (int i = lane; i + 1 < na; i += 64) oka = oka && (a[i] <= a[i + 1]);
    for (int j = lane; j + 1 < nb; j += 64) okb = okb && (b[j] <= b[j + 1]);
    for (int j = lane; j < na + nb; j += 64) finite = finite && (all[j] == all[j]);
    const bool sa = __builtin_amdgcn_ballot_w64(!oka) == 0, sb = __builtin_amdgcn_ballot_w64(!okb) == 0;
    // (a NaN anywhere -- the searches below count with < / <= -- takes the NaN-aware rank sort)
    if (!sa || __builtin_amdgcn_ballot_w64(!finite) != 0) { rank_sort_wave(all, na + nb, out, lane); return; }
    for (int i = lane; i < na; i += 64) {
        const float v = a[i];
        int cnt = 0;
        if (sb) {
            int lo = 0, hi = nb;                                   // first index with b[idx] >= v
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (b[mid] < v) lo = mid + 1; else hi = mid; }
            cnt = lo;
        } else {
            for (int k = 0; k < nb; ++k) cnt += b[k] < v;
        }
        out[i + cnt] = v;
    }
    for (int j = lane; j < nb; j += 64) {
        const float v = b[j];
        int lo = 0, hi = na;                                       // first index with a[idx] > v
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] <= v) lo = mid + 1; else hi = mid; }
        int rank = j;
        if (!sb) {
            rank = 0;
            for (int k = 0; k < nb; ++k) { const float o = b[k]; rank += (o < v) || (o == v && k < j); }
        }
        out[lo + rank] = v;
    }
}

// cumprod_exclusive (nerf_helpers.py:409-430): out[.., 0] = 1, out[.., i] = prod_{k < i} in[.., k], multiplied left to right like torch.cumprod's
// sequential CPU scan (the fused passes carry this product in a register; this is the stand-alone helper of the reference's surface)
__global__ void cumprod_exclusive_kernel(long N, int n, const float* __restrict__ in, float* __restrict__ out) {
    const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= N) return;
    const float* a = in + row * n;
    float* o = out + row * n;
    float p = 1.0f, run = 1.0f;
    for (int i = 0; i < n; ++i) {
        const float v = a[i];
        o[i] = (i == 0) ? 1.0f : p;
        run = (i == 0) ? v : __fmul_rn(run, v);     // inclusive cumprod up to i
        p = run;
    }
}

// backward of cumprod_exclusive: out_i = prod_{k<i} x_k  =>  dL/dx_j = out_j * R_j,  R_j = sum_{i>j} g_i prod_{j<k<i} x_k, i.e.
// R_{n-1} = 0, R_j = g_{j+1} + x_{j+1} R_{j+1} -- no division, so zeros in x are exact (torch.cumprod's backward special-cases them)
__global__ void cumprod_exclusive_backward_kernel(long N, int n, const float* __restrict__ in, const float* __restrict__ out,
                                                  const float* __restrict__ g_out, float* __restrict__ g_in) {
    const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= N) return;
    const float* x = in + row * n;
    const float* o = out + row * n;
    const float* g = g_out + row * n;
    float* gi = g_in + row * n;
    float R = 0.0f;
    for (int j = n - 1; j >= 0; --j) {
        gi[j] = __fmul_rn(o[j], R);
        R = fmaf(x[j], R, g[j]);
    }
}

__global__ __launch_bounds__(WPB * 64) void sort_rows_kernel(long N, int n, const float* __restrict__ in, float* __restrict__ out) {
    __shared__ float vals[WPB][512];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * WPB + wave;
    if (row >= N) return;
    for (int i = lane; i < n; i += 64) vals[wave][i] = in[row * n + i];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    rank_sort_wave(vals[wave], n, out + row * n, lane);
}

// The inference shape of the resampler (round 3): deterministic u (linspace), Nc <= 64 coarse depths in all[0..Nc), Nf <= 128 samples.
// The general path below spends ~700 VALU instructions per ray, most of them in data-dependent binary-search loops (it is VALU-issue-bound:
// one ray per ~2 000 cycles and SIMD with 10 waves resident).  Here every search is either a fixed 6-step branch-free one or replaced:
//   * searchsorted(cdf, u, right): lo = #(cdf <= u), 6 steps over the cdf padded to 64 entries with +inf (valid because the pdf is checked
//     to be non-negative and NaN-free, i.e. the cdf is non-decreasing);
//   * rank of sample j in the merged row = j + #(a <= b_j): b_j lies in bin `below` whose edges are mid-points of a, so the count is
//     below + 1 plus at most two compares -- then VERIFIED against the definition (a[k-1] <= b_j < a[k]);
//   * rank of coarse depth i = i + #(b < a_i) = i + #{j : #(a <= b_j) <= i} (a is sorted): a 65-bin histogram of the counts above (LDS
//     atomics) and one wave scan, no search at all.
// Same samples (same operations in the same order as sample_pdf_wave) and the same ranks as merge_sort_wave, i.e. the same bits as
// torch.sort(cat(z, samples)).  Returns false -- nothing written -- when a precondition fails (a NaN, a negative weight, unsorted depths or
// samples, a failed verification): the caller then runs the general path.
__device__ __forceinline__ bool resample_fast_wave(int Nc, int Nf, float* all, float* zm, float* cdf, const float* __restrict__ w, int lane,
                                                   float* __restrict__ out) {
    const int nb = Nc - 1, nw = Nc - 2;
    const float a_i = lane < Nc ? all[lane] : 0.0f;
    const float a_n = lane + 1 < Nc ? all[lane + 1] : 0.0f;
    bool ok = !(lane + 1 < Nc) || (a_i <= a_n);                          // coarse depths sorted (and not NaN)
    if (lane < nb) zm[lane] = __fmul_rn(0.5f, __fadd_rn(a_n, a_i));
    // pdf / cdf exactly as sample_pdf_wave (one 64-lane pass: nw <= 62)
    const float wi = lane < nw ? __fadd_rn(w[lane + 1], 1e-5f) : 0.0f;
    const float total = wave_sum(wi);
    const float p = lane < nw ? __fdiv_rn(wi, total) : 0.0f;
    ok = ok && (p >= 0.0f);                                              // (false for NaN)
    const float sc = wave_scan_add(p, lane) + 0.0f;
    if (lane == 0) cdf[0] = 0.0f;
    if (lane < nw) cdf[lane + 1] = sc;
    if (lane >= nb) cdf[lane] = __builtin_inff();                        // pad to 64 entries
    int* hist = reinterpret_cast<int*>(cdf) + 64;                        // 65 bins behind the cdf
    hist[lane] = 0;
    if (lane == 0) hist[64] = 0;
    if (__builtin_amdgcn_ballot_w64(!ok) != 0) return false;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    float smp[2];
    int cntb[2];
    bool good = true;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int j = lane + 64 * q;
        const bool live = j < Nf;
        const float uj = linspace01(live ? j : 0, Nf);
        int lo = 0;
#pragma unroll
        for (int step = 32; step > 0; step >>= 1) lo += (cdf[lo + step - 1] <= uj) ? step : 0;
        const int below = max(lo - 1, 0), above = min(lo, nb - 1);
        const float c0 = cdf[below], c1 = cdf[above];
        float denom = __fsub_rn(c1, c0);
        if (denom < 1e-5f) denom = 1.0f;
        const float t = __fdiv_rn(__fsub_rn(uj, c0), denom);
        const float b0 = zm[below], b1 = zm[above];
        const float v = __fadd_rn(b0, __fmul_rn(t, __fsub_rn(b1, b0)));
        smp[q] = v;
        if (live) all[Nc + j] = v;
        // #(a <= v): a[0..below] <= zm[below] <= v; then at most a[below + 1], a[below + 2]
        int k = min(below + 1, Nc);
        k += (k < Nc && all[k] <= v) ? 1 : 0;
        k += (k < Nc && all[k] <= v) ? 1 : 0;
        const bool lo_ok = (k == 0) || (all[k - 1] <= v), hi_ok = (k == Nc) || !(all[k] <= v);
        good = good && (!live || (lo_ok && hi_ok && v == v));
        cntb[q] = k;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    // samples sorted?  (always, for a non-decreasing cdf and increasing u -- checked, not assumed)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int j = lane + 64 * q;
        if (j + 1 < Nf) good = good && (smp[q] <= all[Nc + j + 1]);
    }
    if (__builtin_amdgcn_ballot_w64(!good) != 0) return false;
#pragma unroll
    for (int q = 0; q < 2; ++q)
        if (lane + 64 * q < Nf) atomicAdd(&hist[cntb[q]], 1);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    const int below_or_at = (int)wave_scan_add((float)hist[lane], lane);   // #{j : cnt_b[j] <= lane} (<= 128: exact in f32)
    if (lane < Nc) out[lane + below_or_at] = a_i;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int j = lane + 64 * q;
        if (j < Nf) out[j + cntb[q]] = smp[q];
    }
    return true;
}

// train_utils.py:144-155 fused: z_mid -> sample_pdf(z_mid, w[1:-1]) -> sort(cat(z, samples))
// zc == NULL: the coarse depths are the un-jittered ones of train_utils.py:95-100 and are recomputed from the ray's near / far (packed rays
// columns 6, 7) -- bit for bit what nvsr_coarse_z writes -- so that an inference frame never stores them
#ifndef NVSR_RESAMPLE_GENERAL_ONLY
#define NVSR_RESAMPLE_GENERAL_ONLY 0
#endif
__global__ __launch_bounds__(WPB * 64) void importance_resample_kernel(long N, int Nc, int Nf, const float* __restrict__ zc,
                                                                      const float* __restrict__ rays, int lindisp,
                                                                      const float* __restrict__ weights,
                                                                      const float* __restrict__ u, float* __restrict__ zf, int force_general,
                                                                      float* __restrict__ z_new) {
    __shared__ float cdf_s[WPB][256];
    __shared__ float zmid_s[WPB][256];
    __shared__ float all_s[WPB][512];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long ray = (long)blockIdx.x * WPB + wave;
    if (ray >= N) return;
    const float* w = weights + ray * Nc;
    float* all = all_s[wave];
    float* zm = zmid_s[wave];
    if (zc) {
        const float* z = zc + ray * Nc;
        for (int i = lane; i < Nc; i += 64) all[i] = z[i];
    } else {
        const float nr = rays[ray * 11 + 6], fr = rays[ray * 11 + 7];
        for (int i = lane; i < Nc; i += 64) all[i] = coarse_depth(nr, fr, i, Nc, lindisp);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
#if !NVSR_RESAMPLE_GENERAL_ONLY
    if (!u && Nc >= 3 && Nc <= 64 && Nf >= 1 && Nf <= 128 && !force_general)
        if (resample_fast_wave(Nc, Nf, all, zm, cdf_s[wave], w, lane, zf + ray * (Nc + Nf))) {
            if (z_new)                                                   // (the fast path leaves the samples behind the coarse depths)
                for (int j = lane; j < Nf; j += 64) z_new[ray * Nf + j] = all[Nc + j];
            return;
        }
    __builtin_amdgcn_wave_barrier();
#endif
    for (int i = lane; i < Nc - 1; i += 64) zm[i] = __fmul_rn(0.5f, __fadd_rn(all[i + 1], all[i]));
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    sample_pdf_wave(Nc - 1, Nf, [&](int i) { return zm[i]; }, [&](int i) { return w[i + 1]; }, u ? u + ray * Nf : nullptr,
                    cdf_s[wave], lane, [&](int j, float v) { all[Nc + j] = v; });
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    if (z_new)
        for (int j = lane; j < Nf; j += 64) z_new[ray * Nf + j] = all[Nc + j];
    merge_sort_wave(all, Nc, Nf, zf + ray * (Nc + Nf), lane);
}

// Shared-decoder fine pass (models.fine.type == 'use_same', train_nerf.py:353-355: model_fine IS model_coarse): the fine pass evaluates the
// decoder at sort(cat(z_coarse, z_samples)) (train_utils.py:155-170), a third of which it has already evaluated -- same decoder, same planes,
// same points -- in the coarse pass.  The decoder runs on the new samples only; this kernel merges the two sorted depth lists of a ray and
// gathers the two lists of decoder outputs into the merged order, which is then composited.  Positions: a coarse depth goes behind the new
// samples that are smaller, a new sample behind the coarse depths that are smaller or equal (at equal depths the two points are the same
// point and their outputs the same numbers).  Unsorted input (NaN weights) falls back to counting.  One wave per ray.
__global__ __launch_bounds__(WPB * 64) void shared_merge_kernel(long N, int Nc, int Nf, const float* __restrict__ rays, int lindisp,
                                                               const float* __restrict__ z_new, const float* __restrict__ raw_c,
                                                               const float* __restrict__ raw_new, float* __restrict__ z_m, float* __restrict__ raw_m) {
    __shared__ float a_s[WPB][256], b_s[WPB][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long ray = (long)blockIdx.x * WPB + wave;
    if (ray >= N) return;
    float* a = a_s[wave];
    float* b = b_s[wave];
    const float nr = rays[ray * 11 + 6], fr = rays[ray * 11 + 7];
    for (int i = lane; i < Nc; i += 64) a[i] = coarse_depth(nr, fr, i, Nc, lindisp);
    for (int j = lane; j < Nf; j += 64) b[j] = z_new[ray * Nf + j];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    bool sorted = true;
    for (int i = lane; i + 1 < Nc; i += 64) sorted = sorted && (a[i] <= a[i + 1]);
    for (int j = lane; j + 1 < Nf; j += 64) sorted = sorted && (b[j] <= b[j + 1]);
    const bool fast = __builtin_amdgcn_ballot_w64(!sorted) == 0;
    float* zo = z_m + ray * (Nc + Nf);
    f32x4* ro = reinterpret_cast<f32x4*>(raw_m) + ray * (Nc + Nf);
    const f32x4* rc = reinterpret_cast<const f32x4*>(raw_c) + ray * Nc;
    const f32x4* rn = reinterpret_cast<const f32x4*>(raw_new) + ray * Nf;
    for (int i = lane; i < Nc; i += 64) {
        const float v = a[i];
        int cnt = 0, rank = i;
        if (fast) {                                                      // #{j : b[j] < v}
            int lo = 0, hi = Nf;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (b[mid] < v) lo = mid + 1; else hi = mid; }
            cnt = lo;
        } else {
            for (int j = 0; j < Nf; ++j) cnt += b[j] < v ? 1 : 0;
            rank = 0;
            for (int k = 0; k < Nc; ++k) rank += (a[k] < v || (a[k] == v && k < i)) ? 1 : 0;
        }
        const int pos = min(rank + cnt, Nc + Nf - 1);
        zo[pos] = v;
        ro[pos] = rc[i];
    }
    for (int j = lane; j < Nf; j += 64) {
        const float v = b[j];
        int cnt = 0, rank = j;
        if (fast) {                                                      // #{i : a[i] <= v}
            int lo = 0, hi = Nc;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] <= v) lo = mid + 1; else hi = mid; }
            cnt = lo;
        } else {
            for (int i = 0; i < Nc; ++i) cnt += a[i] <= v ? 1 : 0;
            rank = 0;
            for (int k = 0; k < Nf; ++k) rank += (b[k] < v || (b[k] == v && k < j)) ? 1 : 0;
        }
        const int pos = min(rank + cnt, Nc + Nf - 1);
        zo[pos] = v;
        ro[pos] = rn[j];
    }
}

// ---- compositing: one wave per ray, samples across lanes, transmittance by a wave scan ------------------------------------
__global__ __launch_bounds__(WPB * 64) void composite_kernel(long N, int S, const float* __restrict__ raw, const float* __restrict__ z,
                                                            const float* __restrict__ rd, int rd_stride, const float* __restrict__ noise, int white,
                                                            float* __restrict__ rgb, float* __restrict__ disp, float* __restrict__ acc,
                                                            float* __restrict__ weights, float* __restrict__ depth, int mip) {
    // mip (volume_rendering_utils.py:19-26,41-42, the Mip-NeRF baseline's intervals): z holds S + 1 interval edges per ray, every sample has
    // its own finite interval (no 1e10 tail) and the depth map integrates the interval mid-points
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long ray = (long)blockIdx.x * WPB + wave;
    if (ray >= N) return;
    const float d0 = rd[ray * rd_stride], d1 = rd[ray * rd_stride + 1], d2 = rd[ray * rd_stride + 2];
    const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d0, d0), __fmul_rn(d1, d1)), __fmul_rn(d2, d2)));
    const float* zr = z + ray * (S + (mip ? 1 : 0));
    const f32x4* rr = reinterpret_cast<const f32x4*>(raw) + ray * S;
    float Tcarry = 1.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f, dep = 0.0f, ac = 0.0f;
    for (int base = 0; base < S; base += 64) {
        const int s = base + lane;
        const bool in = s < S;
        float w = 0.0f, zs = 0.0f, fac = 1.0f;
        f32x4 rv = {0.0f, 0.0f, 0.0f, 0.0f};
        if (in) {
            zs = zr[s];
            rv = rr[s];
            const float dist = __fmul_rn((mip || s + 1 < S) ? __fsub_rn(zr[s + 1], zs) : 1e10f, nrm);
            if (mip) zs = __fmul_rn(0.5f, __fadd_rn(zs, zr[s + 1]));
            float sig = rv[3];
            if (noise) sig = __fadd_rn(sig, noise[ray * S + s]);
            sig = fmaxf(sig, 0.0f);
            const float alpha = __fsub_rn(1.0f, expf(-__fmul_rn(sig, dist)));
            fac = __fadd_rn(__fsub_rn(1.0f, alpha), 1e-10f);
            w = alpha;
        }
        const float incl = wave_scan_mul(fac, lane);          // prod_{k<=lane} fac_k
        float excl = __shfl_up(incl, 1);
        if (lane == 0) excl = 1.0f;
        const float T = __fmul_rn(excl, Tcarry);
        Tcarry = __fmul_rn(__shfl(incl, 63), Tcarry);
        w = __fmul_rn(w, T);
        if (in) {
            if (weights) weights[ray * S + s] = w;
            cr += w * (1.0f / (1.0f + expf(-rv[0])));
            cg += w * (1.0f / (1.0f + expf(-rv[1])));
            cb += w * (1.0f / (1.0f + expf(-rv[2])));
            dep += w * zs;
            ac += w;
        }
    }
    cr = wave_sum(cr); cg = wave_sum(cg); cb = wave_sum(cb); dep = wave_sum(dep); ac = wave_sum(ac);
    if (lane == 0) {
        const float q = dep / ac;
        disp[ray] = 1.0f / ((q != q) ? q : fmaxf(1e-10f, q));
        if (white) { const float bg = 1.0f - ac; cr += bg; cg += bg; cb += bg; }
        rgb[ray * 3] = cr; rgb[ray * 3 + 1] = cg; rgb[ray * 3 + 2] = cb;
        acc[ray] = ac;
        if (depth) depth[ray] = dep;
    }
}

}  // namespace nvsr

using namespace nvsr;

static inline unsigned blocks_for(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

extern "C" {

int nvsr_version(void) { return 500; }   // 5xx: round 5 (nvsr_planes_sr_*_batch_arith / nvsr_planes_sr_batch_ex: SR training on the regions of interest of B planes at once);
                                         // 4xx: round 4 (range flag, device-keyed pixel sampler, tile-pair training forward, gate bit layout, per-point backward scales;
                                         // 410: nvsr_scene_ext / *_ext generic entry points, nvsr_set_sr_align_corners, nvsr_render_rays_shared_arith)
int64_t nvsr_fused_min_rays(void) { return NVSR_FUSED_MIN_RAYS; }

int nvsr_plane_to_channel_last(const float* nchw, float* nhwc, int Cc, int H, int W, nvsr_stream_t stream) {
    if (!nchw || !nhwc) return NVSR_ERR_NULL;
    if (Cc < 1 || Cc > 256 || H < 1 || W < 1) return NVSR_ERR_SHAPE;
    const long HW = (long)H * W;
    hipLaunchKernelGGL(to_channel_last_kernel, dim3(blocks_for(HW, 64)), dim3(256), Cc * 65 * sizeof(float), (hipStream_t)stream,
                       nchw, nhwc, Cc, HW);
    return NVSR_CHECK_LAUNCH();
}

int nvsr_plane_from_channel_last(const float* nhwc, float* nchw, int Cc, int H, int W, nvsr_stream_t stream) {
    if (!nchw || !nhwc) return NVSR_ERR_NULL;
    if (Cc < 1 || Cc > 256 || H < 1 || W < 1) return NVSR_ERR_SHAPE;
    const long HW = (long)H * W;
    hipLaunchKernelGGL(from_channel_last_kernel, dim3(blocks_for(HW, 64)), dim3(256), Cc * 65 * sizeof(float), (hipStream_t)stream,
                       nhwc, nchw, Cc, HW);
    return NVSR_CHECK_LAUNCH();
}

int nvsr_get_ray_bundle(int H, int W, double focal_x, double focal_y, const float* c2w, int padding, double offset, float* ro,
                        float* rd, nvsr_stream_t stream) {
    if (!c2w || !ro || !rd) return NVSR_ERR_NULL;
    if (H < 1 || W < 1 || padding < 0) return NVSR_ERR_SHAPE;
    const int64_t n = (int64_t)(H + 2 * padding) * (W + 2 * padding);
    hipLaunchKernelGGL(ray_bundle_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, H, W, (float)focal_x,
                       (float)focal_y, c2w, padding, (float)offset, ro, rd);
    return NVSR_CHECK_LAUNCH();
}

int nvsr_get_ray_bundle_at(int H, int W, double focal_x, double focal_y, const float* c2w, double offset, int64_t N, const int32_t* row_col,
                           float* ro, float* rd, nvsr_stream_t stream) {
    if (!c2w || !row_col || !ro || !rd) return NVSR_ERR_NULL;
    if (H < 1 || W < 1 || N < 0) return NVSR_ERR_SHAPE;
    if (N == 0) return NVSR_OK;
    hipLaunchKernelGGL(ray_bundle_at_kernel, dim3(blocks_for(N, 256)), dim3(256), 0, (hipStream_t)stream, H, W, (float)focal_x, (float)focal_y,
                       c2w, (float)offset, (long)N, row_col, ro, rd);
    return NVSR_CHECK_LAUNCH();
}

int nvsr_sample_pixels(int64_t total, int H, int W, uint64_t key, int64_t first, int64_t n, const float* image, int channels,
                       int32_t* row_col, float* target, nvsr_stream_t stream) {
    if (!row_col || (target && !image)) return NVSR_ERR_NULL;
    if (H < 1 || W < 1 || total != (int64_t)H * W || first < 0 || n < 0 || first + n > total || (target && channels < 1)) return NVSR_ERR_SHAPE;
    if (n == 0) return NVSR_OK;
    int hb = 1;
    while (hb < 31 && (1ll << (2 * hb)) < total) ++hb;          // 2^(2 hb) >= total, at most 4 total: < 4 walks expected
    hipLaunchKernelGGL(sample_pixels_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (long)total, hb, H, W,
                       (unsigned long long)key, (long)first, (long)n, image, channels, row_col, target);
    return NVSR_CHECK_LAUNCH();
}

uint64_t nvsr_sample_key(uint64_t seed, uint64_t calls) { return sample_key(seed, calls); }

int nvsr_sample_pixels_seq(int64_t total, int H, int W, uint64_t* state, int64_t first, int64_t n, const float* image, int channels,
                           int32_t* row_col, float* target, nvsr_stream_t stream) {
    if (!state || !row_col || (target && !image)) return NVSR_ERR_NULL;
    if (H < 1 || W < 1 || total != (int64_t)H * W || first < 0 || n < 1 || first + n > total || (target && channels < 1)) return NVSR_ERR_SHAPE;
    int hb = 1;
    while (hb < 31 && (1ll << (2 * hb)) < total) ++hb;
    hipLaunchKernelGGL(sample_pixels_seq_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (long)total, hb, H, W,
                       reinterpret_cast<unsigned long long*>(state), (long)first, (long)n, image, channels, row_col, target);
    return NVSR_CHECK_LAUNCH();
}

int nvsr_mse_pair(int64_t n, const float* a, const float* b, const float* target, float* losses, float* g_a, float* g_b, nvsr_stream_t stream) {
    if (!a || !target || !losses || (g_b && !b)) return NVSR_ERR_NULL;
    if (n < 1 || n > NVSR_MSE_PAIR_MAX_ELEMS) return NVSR_ERR_SHAPE;
    hipLaunchKernelGGL(mse_pair_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (long)n, a, b, target, losses, g_a, g_b, 0);
    return NVSR_CHECK_LAUNCH();
}

int nvsr_mse_pair_sum(int64_t n, const float* a, const float* b, const float* target, float* losses3, nvsr_stream_t stream) {
    if (!a || !b || !target || !losses3) return NVSR_ERR_NULL;
    if (n < 1 || n > NVSR_MSE_PAIR_MAX_ELEMS) return NVSR_ERR_SHAPE;
    hipLaunchKernelGGL(mse_pair_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (long)n, a, b, target, losses3, (float*)nullptr, (float*)nullptr, 1);
    return NVSR_CHECK_LAUNCH();
}

int nvsr_mse_pair_backward(int64_t n, const float* a, const float* b, const float* target, const float* scale_a, const float* scale_b, float* g_a,
                           float* g_b, nvsr_stream_t stream) {
    if (!target || (g_a && !a) || (g_b && !b)) return NVSR_ERR_NULL;
    if (n < 1 || n > NVSR_MSE_PAIR_MAX_ELEMS) return NVSR_ERR_SHAPE;
    if (!g_a && !g_b) return NVSR_OK;
    hipLaunchKernelGGL(mse_pair_backward_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (long)n, a, b, target, scale_a, scale_b,
                       g_a, g_b);
    return NVSR_CHECK_LAUNCH();
}

int nvsr_ndc_rays(int H, int W, double focal, double near_, int64_t N, const float* ro, const float* rd, float* ro_out, float* rd_out,
                  nvsr_stream_t stream) {
    if (!ro || !rd || !ro_out || !rd_out) return NVSR_ERR_NULL;
    if (N < 0 || H < 1 || W < 1) return NVSR_ERR_SHAPE;
    if (N == 0) return NVSR_OK;
    const float sx = (float)(-1.0 / (W / (2.0 * focal))), sy = (float)(-1.0 / (H / (2.0 * focal)));
    hipLaunchKernelGGL(ndc_rays_kernel, dim3(blocks_for(N, 256)), dim3(256), 0, (hipStream_t)stream, sx, sy, (float)near_,
                       (float)(2.0 * near_), (float)(-2.0 * near_), (long)N, ro, rd, ro_out, rd_out);
    return NVSR_CHECK_LAUNCH();
}

int nvsr_pack_rays(int64_t N, const float* ro, const float* rd, const float* view_src, double near_, double far_, float* rays,
                   nvsr_stream_t stream) {
    if (!ro || !rd || !view_src || !rays) return NVSR_ERR_NULL;
    if (N < 0) return NVSR_ERR_SHAPE;
    if (N == 0) return NVSR_OK;
    hipLaunchKernelGGL(pack_rays_kernel, dim3(blocks_for(N, 256)), dim3(256), 0, (hipStream_t)stream, (long)N, ro, rd, view_src,
                       (float)near_, (float)far_, rays);
    return NVSR_CHECK_LAUNCH();
}

int nvsr_coarse_z(int64_t N, int Nc, const float* rays, int lindisp, const float* t_rand, float* z, nvsr_stream_t stream) {
    if (!rays || !z) return NVSR_ERR_NULL;
    if (N < 0 || Nc < 1) return NVSR_ERR_SHAPE;
    if (N == 0) return NVSR_OK;
    hipLaunchKernelGGL(coarse_z_kernel, dim3(blocks_for(N * Nc, 256)), dim3(256), 0, (hipStream_t)stream, (long)N, Nc, rays, lindisp,
                       t_rand, z);
    return NVSR_CHECK_LAUNCH();
}

int nvsr_sample_pdf(int64_t N, int nb, int ns, const float* bins, const float* weights, const float* u, float* samples,
                    nvsr_stream_t stream) {
    if (!bins || !weights || !samples) return NVSR_ERR_NULL;
    if (N < 0 || nb < 2 || nb > 256 || ns < 1) return NVSR_ERR_SHAPE;
    if (N == 0) return NVSR_OK;
    hipLaunchKernelGGL(sample_pdf_kernel, dim3(blocks_for(N, WPB)), dim3(WPB * 64), 0, (hipStream_t)stream, (long)N, nb, ns, bins,
                       weights, u, samples);
    return NVSR_CHECK_LAUNCH();
}

int nvsr_cumprod_exclusive(int64_t N, int n, const float* in, float* out, nvsr_stream_t stream) {
    if (!in || !out) return NVSR_ERR_NULL;
    if (N < 0 || n < 1) return NVSR_ERR_SHAPE;
    if (N == 0) return NVSR_OK;
    hipLaunchKernelGGL(cumprod_exclusive_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long)N, n, in, out);
    return NVSR_CHECK_LAUNCH();
}

int nvsr_cumprod_exclusive_backward(int64_t N, int n, const float* in, const float* out, const float* g_out, float* g_in, nvsr_stream_t stream) {
    if (!in || !out || !g_out || !g_in) return NVSR_ERR_NULL;
    if (N < 0 || n < 1) return NVSR_ERR_SHAPE;
    if (N == 0) return NVSR_OK;
    hipLaunchKernelGGL(cumprod_exclusive_backward_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long)N, n, in, out,
                       g_out, g_in);
    return NVSR_CHECK_LAUNCH();
}

int nvsr_sort_rows(int64_t N, int n, const float* in, float* out, nvsr_stream_t stream) {
    if (!in || !out) return NVSR_ERR_NULL;
    if (N < 0 || n < 1 || n > 512) return NVSR_ERR_SHAPE;
    if (N == 0) return NVSR_OK;
    hipLaunchKernelGGL(sort_rows_kernel, dim3(blocks_for(N, WPB)), dim3(WPB * 64), 0, (hipStream_t)stream, (long)N, n, in, out);
    return NVSR_CHECK_LAUNCH();
}

// NVSR_RESAMPLE_GENERAL=1 (environment, read once): every ray takes the general path of importance_resample_kernel (A/B timing)
static int resample_general_only() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("NVSR_RESAMPLE_GENERAL"); v = (e && e[0] == '1') ? 1 : 0; }
    return v;
}

int nvsr_importance_resample(int64_t N, int Nc, int Nf, const float* z_coarse, const float* weights, const float* u, float* z_fine,
                             nvsr_stream_t stream) {
    if (!z_coarse || !weights || !z_fine) return NVSR_ERR_NULL;
    if (N < 0 || Nc < 3 || Nc > 256 || Nf < 1 || Nf > 256) return NVSR_ERR_SHAPE;
    if (N == 0) return NVSR_OK;
    hipLaunchKernelGGL(importance_resample_kernel, dim3(blocks_for(N, WPB)), dim3(WPB * 64), 0, (hipStream_t)stream, (long)N, Nc, Nf,
                       z_coarse, (const float*)nullptr, 0, weights, u, z_fine, resample_general_only(), (float*)nullptr);
    return NVSR_CHECK_LAUNCH();
}

static int importance_resample_rays_impl(int64_t N, int Nc, int Nf, const float* rays, int lindisp, const float* weights, const float* u,
                                         float* z_fine, float* z_new, nvsr_stream_t stream) {
    if (!rays || !weights || !z_fine) return NVSR_ERR_NULL;
    if (N < 0 || Nc < 3 || Nc > 256 || Nf < 1 || Nf > 256) return NVSR_ERR_SHAPE;
    if (N == 0) return NVSR_OK;
    hipLaunchKernelGGL(importance_resample_kernel, dim3(blocks_for(N, WPB)), dim3(WPB * 64), 0, (hipStream_t)stream, (long)N, Nc, Nf,
                       (const float*)nullptr, rays, lindisp, weights, u, z_fine, resample_general_only(), z_new);
    return NVSR_CHECK_LAUNCH();
}
int nvsr_importance_resample_rays(int64_t N, int Nc, int Nf, const float* rays, int lindisp, const float* weights, const float* u,
                                  float* z_fine, nvsr_stream_t stream) {
    return importance_resample_rays_impl(N, Nc, Nf, rays, lindisp, weights, u, z_fine, nullptr, stream);
}

int nvsr_composite(int64_t N, int S, const float* raw, const float* z, const float* rd, const float* noise, int white_bkgd, float* rgb,
                   float* disp, float* acc, float* weights, float* depth, nvsr_stream_t stream) {
    if (!raw || !z || !rd || !rgb || !disp || !acc) return NVSR_ERR_NULL;
    if (!aligned16(raw)) return NVSR_ERR_ALIGN;
    if (N < 0 || S < 1) return NVSR_ERR_SHAPE;
    if (N == 0) return NVSR_OK;
    hipLaunchKernelGGL(composite_kernel, dim3(blocks_for(N, WPB)), dim3(WPB * 64), 0, (hipStream_t)stream, (long)N, S, raw, z, rd, 3, noise,
                       white_bkgd, rgb, disp, acc, weights, depth, 0);
    return NVSR_CHECK_LAUNCH();
}

/* mip_nerf=True: raw [N,S,4] over the S intervals of z [N,S+1] */
int nvsr_composite_mip(int64_t N, int S, const float* raw, const float* z, const float* rd, const float* noise, int white_bkgd, float* rgb,
                       float* disp, float* acc, float* weights, float* depth, nvsr_stream_t stream) {
    if (!raw || !z || !rd || !rgb || !disp || !acc) return NVSR_ERR_NULL;
    if (!aligned16(raw)) return NVSR_ERR_ALIGN;
    if (N < 0 || S < 1) return NVSR_ERR_SHAPE;
    if (N == 0) return NVSR_OK;
    hipLaunchKernelGGL(composite_kernel, dim3(blocks_for(N, WPB)), dim3(WPB * 64), 0, (hipStream_t)stream, (long)N, S, raw, z, rd, 3, noise,
                       white_bkgd, rgb, disp, acc, weights, depth, 1);
    return NVSR_CHECK_LAUNCH();
}

/* same, reading the directions out of packed rays [N,11] (columns 3..5) */
int nvsr_composite_rays(int64_t N, int S, const float* raw, const float* z, const float* rays, const float* noise, int white_bkgd,
                        float* rgb, float* disp, float* acc, float* weights, float* depth, nvsr_stream_t stream) {
    if (!raw || !z || !rays || !rgb || !disp || !acc) return NVSR_ERR_NULL;
    if (!aligned16(raw)) return NVSR_ERR_ALIGN;
    if (N < 0 || S < 1) return NVSR_ERR_SHAPE;
    if (N == 0) return NVSR_OK;
    hipLaunchKernelGGL(composite_kernel, dim3(blocks_for(N, WPB)), dim3(WPB * 64), 0, (hipStream_t)stream, (long)N, S, raw, z, rays + 3, 11,
                       noise, white_bkgd, rgb, disp, acc, weights, depth, 0);
    return NVSR_CHECK_LAUNCH();
}

int64_t nvsr_render_workspace_floats(int64_t N, int Nc, int Nf) {
    // z_c [N,Nc], w_c [N,Nc], z_f [N,Nc+Nf], each rounded up to 4 floats so that every sub-buffer (raw_ws included) stays 16-byte aligned
    const int64_t base = 2 * round4(N * (int64_t)Nc) + (Nf > 0 ? round4(N * (int64_t)(Nc + Nf)) : 0);
    return base + (N < NVSR_FUSED_MIN_RAYS ? 4 * N * (int64_t)(Nc + (Nf > 0 ? Nf : 0)) : 0);   // + raw [N,S,4] for the un-fused path
}

extern "C" int nvsr_internal_resolve_decoder_arith(int arithmetic);      // render.hip
extern "C" int nvsr_render_pass3_coarse_z_launch(int limbs, const nvsr_scene* scene, const float* packed_decoder, int64_t N, int S, const float* rays,
                                                 int lindisp, const float* noise, int white_bkgd, float* rgb, float* disp, float* acc,
                                                 float* weights, float* depth, float* raw_out, nvsr_stream_t stream);

static int render_one_pass(const nvsr_scene* scene, const float* packed, int64_t N, int S, const float* rays, const float* z,
                           const float* noise, int white, float* rgb, float* disp, float* acc, float* weights, float* raw_ws,
                           int arithmetic, nvsr_stream_t stream) {
    if (N >= NVSR_FUSED_MIN_RAYS)
        return nvsr_render_pass_arith(scene, packed, N, S, rays, z, noise, white, rgb, disp, acc, weights, nullptr, nullptr, arithmetic, stream);
    // few rays: one workgroup per 128 rays would leave most CUs idle -> tile over samples, composite separately
    if (int e = nvsr_decode_rays_arith(scene, packed, N, S, rays, z, raw_ws, nullptr, nullptr, arithmetic, stream)) return e;
    // rd = rays[:, 3:6]: the composite kernel reads rd with stride 3, so pass a strided view through a tiny repack
    return nvsr_composite_rays(N, S, raw_ws, z, rays, noise, white, rgb, disp, acc, weights, nullptr, stream);
}

int nvsr_render_rays(const nvsr_scene* scene, const float* packed_coarse, const float* packed_fine, int64_t N, int Nc, int Nf,
                     const float* rays, int lindisp, int white_bkgd, const float* t_rand, const float* u, const float* noise_coarse,
                     const float* noise_fine, float* rgb_c, float* disp_c, float* acc_c, float* rgb_f, float* disp_f, float* acc_f,
                     float* workspace, nvsr_stream_t stream) {
    return nvsr_render_rays_arith(scene, packed_coarse, packed_fine, N, Nc, Nf, rays, lindisp, white_bkgd, t_rand, u, noise_coarse, noise_fine,
                                  rgb_c, disp_c, acc_c, rgb_f, disp_f, acc_f, workspace, NVSR_ARITH_INHERIT, stream);
}

int nvsr_render_rays_arith(const nvsr_scene* scene, const float* packed_coarse, const float* packed_fine, int64_t N, int Nc, int Nf,
                           const float* rays, int lindisp, int white_bkgd, const float* t_rand, const float* u, const float* noise_coarse,
                           const float* noise_fine, float* rgb_c, float* disp_c, float* acc_c, float* rgb_f, float* disp_f, float* acc_f,
                           float* workspace, int arithmetic, nvsr_stream_t stream) {
    if (!workspace) return NVSR_ERR_NULL;
    if (!aligned16(workspace)) return NVSR_ERR_ALIGN;
    if (N < 0 || Nc < 1 || Nf < 0) return NVSR_ERR_SHAPE;
    if (Nf > 0 && (Nc < 3 || Nc > 256 || Nf > 256)) return NVSR_ERR_SHAPE;
    if (Nf > 0 && (!packed_fine || !rgb_f || !disp_f || !acc_f)) return NVSR_ERR_NULL;
    if (N == 0) return NVSR_OK;
    float* z_c = workspace;
    float* w_c = z_c + round4(N * (int64_t)Nc);
    float* z_f = w_c + round4(N * (int64_t)Nc);
    float* raw_ws = z_f + (Nf > 0 ? round4(N * (int64_t)(Nc + Nf)) : 0);
    int e;
    // Inference frames (no stratified jitter) on the fused limb passes: the coarse depths are a function of (near, far, s) -- the coarse
    // pass and the resampler compute them in registers, the [N,Nc] depth tensor (164 MB at 800 x 800 x 64) is never written or read
    const int arith = nvsr_internal_resolve_decoder_arith(arithmetic);
    const bool in_kernel_z = !t_rand && Nf > 0 && N >= NVSR_FUSED_MIN_RAYS && arith > 0 && !getenv("NVSR_RENDER_V1") && !getenv("NVSR_STORE_COARSE_Z");
    if (in_kernel_z) {
        if (!scene || !packed_coarse || !rays || !rgb_c || !disp_c || !acc_c) return NVSR_ERR_NULL;
        for (int d = 0; d < 4; ++d) {
            if (!scene->planes[d]) return NVSR_ERR_NULL;
            if (!aligned16(scene->planes[d])) return NVSR_ERR_ALIGN;
            if (scene->ph[d] < 1 || scene->pw[d] < 1 || (int64_t)scene->ph[d] * scene->pw[d] * NVSR_PLANE_CHANNELS >= (int64_t)1 << 31) return NVSR_ERR_SHAPE;
        }
        if (!aligned16(packed_coarse)) return NVSR_ERR_ALIGN;
        e = nvsr_render_pass3_coarse_z_launch(arith, scene, packed_coarse, N, Nc, rays, lindisp, noise_coarse, white_bkgd, rgb_c, disp_c, acc_c, w_c,
                                              nullptr, nullptr, stream);
        if (e) return e;
        e = nvsr_importance_resample_rays(N, Nc, Nf, rays, lindisp, w_c, u, z_f, stream);
        if (e) return e;
        return render_one_pass(scene, packed_fine, N, Nc + Nf, rays, z_f, noise_fine, white_bkgd, rgb_f, disp_f, acc_f, nullptr, raw_ws, arithmetic, stream);
    }
    e = nvsr_coarse_z(N, Nc, rays, lindisp, t_rand, z_c, stream);
    if (e) return e;
    e = render_one_pass(scene, packed_coarse, N, Nc, rays, z_c, noise_coarse, white_bkgd, rgb_c, disp_c, acc_c, Nf > 0 ? w_c : nullptr,
                        raw_ws, arithmetic, stream);
    if (e || Nf <= 0) return e;
    e = nvsr_importance_resample(N, Nc, Nf, z_c, w_c, u, z_f, stream);
    if (e) return e;
    return render_one_pass(scene, packed_fine, N, Nc + Nf, rays, z_f, noise_fine, white_bkgd, rgb_f, disp_f, acc_f, nullptr, raw_ws, arithmetic, stream);
}

// ---- one decoder for both passes (models.fine.type == 'use_same') ----------------------------------------------------------------------
// does this call take the shared path?  (a frame without stratified jitter on the fused limb passes -- the conditions of in_kernel_z above)
static bool shared_path(int64_t N, int Nf, const float* t_rand, int arithmetic) {
    return !t_rand && Nf > 0 && N >= NVSR_FUSED_MIN_RAYS && nvsr_internal_resolve_decoder_arith(arithmetic) > 0 && !getenv("NVSR_RENDER_V1") &&
           !getenv("NVSR_STORE_COARSE_Z") && !getenv("NVSR_NO_SHARED_DECODER");
}
int64_t nvsr_render_shared_workspace_floats(int64_t N, int Nc, int Nf) {
    // w_c [N,Nc], z_new [N,Nf], z_m [N,Nc+Nf], raw_c [N,Nc,4], raw_new [N,Nf,4], raw_m [N,Nc+Nf,4] -- and never less than the two-decoder path
    const int64_t S = (int64_t)Nc + (Nf > 0 ? Nf : 0);
    const int64_t shared = round4(N * (int64_t)Nc) + round4(N * (int64_t)Nf) + round4(N * S) + 4 * N * (int64_t)Nc + 4 * N * (int64_t)Nf + 4 * N * S;
    const int64_t plain = nvsr_render_workspace_floats(N, Nc, Nf);
    return shared > plain ? shared : plain;
}
int nvsr_render_rays_shared_arith(const nvsr_scene* scene, const float* packed, int64_t N, int Nc, int Nf, const float* rays, int lindisp,
                                  int white_bkgd, const float* t_rand, const float* u, const float* noise_coarse, const float* noise_fine,
                                  float* rgb_c, float* disp_c, float* acc_c, float* rgb_f, float* disp_f, float* acc_f, float* workspace,
                                  int arithmetic, nvsr_stream_t stream) {
    if (!shared_path(N, Nf, t_rand, arithmetic))
        return nvsr_render_rays_arith(scene, packed, packed, N, Nc, Nf, rays, lindisp, white_bkgd, t_rand, u, noise_coarse, noise_fine, rgb_c, disp_c,
                                      acc_c, rgb_f, disp_f, acc_f, workspace, arithmetic, stream);
    if (!workspace || !scene || !packed || !rays || !rgb_c || !disp_c || !acc_c || !rgb_f || !disp_f || !acc_f) return NVSR_ERR_NULL;
    if (!aligned16(workspace) || !aligned16(packed)) return NVSR_ERR_ALIGN;
    if (Nc < 3 || Nc > 256 || Nf > 256) return NVSR_ERR_SHAPE;
    for (int d = 0; d < 4; ++d) {
        if (!scene->planes[d]) return NVSR_ERR_NULL;
        if (!aligned16(scene->planes[d])) return NVSR_ERR_ALIGN;
        if (scene->ph[d] < 1 || scene->pw[d] < 1 || (int64_t)scene->ph[d] * scene->pw[d] * NVSR_PLANE_CHANNELS >= (int64_t)1 << 31) return NVSR_ERR_SHAPE;
    }
    const int64_t S = (int64_t)Nc + Nf;
    float* w_c = workspace;
    float* z_new = w_c + round4(N * (int64_t)Nc);
    float* z_m = z_new + round4(N * (int64_t)Nf);
    float* raw_c = z_m + round4(N * S);
    float* raw_new = raw_c + 4 * N * (int64_t)Nc;
    float* raw_m = raw_new + 4 * N * (int64_t)Nf;
    const int arith = nvsr_internal_resolve_decoder_arith(arithmetic);
    int e = nvsr_render_pass3_coarse_z_launch(arith, scene, packed, N, Nc, rays, lindisp, noise_coarse, white_bkgd, rgb_c, disp_c, acc_c, w_c, nullptr,
                                              raw_c, stream);
    if (e) return e;
    if ((e = importance_resample_rays_impl(N, Nc, Nf, rays, lindisp, w_c, u, z_m, z_new, stream))) return e;
    // the decoder on the new samples only (its own compositing lands in the fine outputs and is overwritten below)
    if ((e = nvsr_render_pass_arith(scene, packed, N, Nf, rays, z_new, nullptr, white_bkgd, rgb_f, disp_f, acc_f, nullptr, nullptr, raw_new, arithmetic, stream)))
        return e;
    hipLaunchKernelGGL(shared_merge_kernel, dim3(blocks_for(N, WPB)), dim3(WPB * 64), 0, (hipStream_t)stream, (long)N, Nc, Nf, rays, lindisp, z_new,
                       raw_c, raw_new, z_m, raw_m);
    if ((e = NVSR_CHECK_LAUNCH())) return e;
    return nvsr_composite_rays(N, (int)S, raw_m, z_m, rays, noise_fine, white_bkgd, rgb_f, disp_f, acc_f, nullptr, nullptr, stream);
}

}  // extern "C"
