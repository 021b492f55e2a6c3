// VERDICT r5 item 5, "tools/ form first": would the fused render pass gain from v_mfma_f32_16x16x32_f16 at two waves per SIMD?
// tools/mfma_power_roof.hip measured BARE streams: 16x16x32 at two waves per SIMD sustains +14 % over 32x32x16 (1 837 vs 1 609 TFLOP/s, f16, random
// operands, power-managed clock).  The render pass is not a bare stream: per MFMA it reads weight fragments from LDS, splits the activations into
// two f16 limbs, applies bias + ReLU, and its B operands are post-ReLU activations (half zeros).  This bench runs the DECODER CHAIN of the pass --
// hidden layers 128 -> 128 in the 2-f16-limb arithmetic (3 MFMA products per block), weights resident in LDS, limb splits and bias + ReLU on the
// vector unit, FILL more VALU instructions per 32 MFMA-cycles standing in for gathers / heads / compositing -- in the three structures:
//   A  32x32x16, one wave per SIMD, two 32-point tiles per wave (a weight fragment feeds 2 tiles)          = render_pass3_kernel<2>
//   B  32x32x16, two waves per SIMD, one 32-point tile per wave                                            = decode_rays_limb_kernel
//   C  16x16x32, two waves per SIMD, one 32-point tile per wave as two 16-point halves (a fragment feeds both)
// and prints point-layers per second and the executed MFMA rate.  Same work per point in all three; only instruction shape and occupancy differ.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_sidework_roof.hip -o scratch/mfma_sidework_roof && scratch/mfma_sidework_roof
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0)
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0)
#ifndef FILL
#define FILL 2          // extra VALU per 32 MFMA-cycles and tile (the pass has ~5.4 VALU per 32x32x16 MFMA in all; split + ReLU here are ~2.7)
#endif

__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ float urand(unsigned s) { return (float)(hash(s) >> 8) * (1.0f / 8388608.0f) - 1.0f; }       // [-1, 1)
__device__ __forceinline__ unsigned f16_pair(float e1, float e0) {
    const f16x2 v = __builtin_convertvector(f32x2{e0, e1}, f16x2);
    return __builtin_bit_cast(unsigned, v);
}
template <int HI>
__device__ __forceinline__ float f16_rest(float x, unsigned pair) {
    float r;
    if (HI) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pair), "v"(x));
    else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pair), "v"(x));
    return r;
}
// 8 f32 values -> hi / lo f16 limbs (4 words each): 4 cvt_pk + 8 fma_mix + 4 cvt_pk = 16 VALU
__device__ __forceinline__ void split8(const float* e, u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        hi[j] = f16_pair(e[2 * j + 1], e[2 * j]);
        const float r0 = f16_rest<0>(e[2 * j], hi[j]), r1 = f16_rest<1>(e[2 * j + 1], hi[j]);
        lo[j] = f16_pair(r1, r0);
    }
}

// weights of one layer as f16 limb fragments in LDS: [K-block][out block][limb hi / lo][lane][4 words] = 64 KB
constexpr int W_WORDS = 8 * 4 * 2 * 256;
__device__ __forceinline__ void fill_weights(unsigned* lds, int tpb) {
    for (int i = threadIdx.x; i < W_WORDS; i += tpb) {
        const float w0 = urand(2u * i) * 0.15f, w1 = urand(2u * i + 1u) * 0.15f;
        const bool lo = (i >> 8) & 1;
        lds[i] = lo ? f16_pair(w1 * 4.8e-4f, w0 * 4.8e-4f) : f16_pair(w1, w0);       // low limbs: 2^-11 of the value
    }
    __syncthreads();
}

// ---- A / B: 32x32x16, TILES tiles of 32 points per wave --------------------------------------------------------------------------------
template <int WPS, int TILES>
__global__ __launch_bounds__(256 * WPS, 1) void chain32(float* out, int layers) {
    __shared__ __attribute__((aligned(16))) unsigned lds[W_WORDS];
    fill_weights(lds, 256 * WPS);
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const u32x4* wv = reinterpret_cast<const u32x4*>(lds) + lane;
    f32x16 acc[TILES][4], act[TILES][4];
    float inj[TILES], fill[TILES][4];
    for (int k = 0; k < TILES; ++k) {
        inj[k] = urand(t * 7u + k);
        for (int b = 0; b < 4; ++b) {
            fill[k][b] = urand(t * 13u + 4 * k + b);
            for (int r = 0; r < 16; ++r) act[k][b][r] = fmaxf(urand(t * 64u + 16 * b + r + 1000 * k), 0.0f);
        }
    }
    for (int l = 0; l < layers; ++l) {
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            u32x4 xh[TILES], xl[TILES];
#pragma unroll
            for (int k = 0; k < TILES; ++k) {
                float e[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) e[i] = act[k][kb >> 1][8 * (kb & 1) + i];
                split8(e, xh[k], xl[k]);
            }
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {
                const u32x4 ah = wv[((kb * 4 + ob) * 2 + 0) * 64], al = wv[((kb * 4 + ob) * 2 + 1) * 64];
#pragma unroll
                for (int k = 0; k < TILES; ++k) {
                    if (kb == 0) {
                        const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                        acc[k][ob] = MFMA32(ah, xl[k], zero);
                    } else acc[k][ob] = MFMA32(ah, xl[k], acc[k][ob]);
                    acc[k][ob] = MFMA32(al, xh[k], acc[k][ob]);
                    acc[k][ob] = MFMA32(ah, xh[k], acc[k][ob]);
#pragma unroll
                    for (int f = 0; f < 3 * FILL; ++f) fill[k][f & 3] = fmaf(fill[k][f & 3], 0.999f, inj[k]);      // stand-in side work
                }
            }
        }
        // bias + ReLU: act = max(acc + inj, 0)     (2 VALU per element)
#pragma unroll
        for (int k = 0; k < TILES; ++k)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) act[k][b][r] = fmaxf(acc[k][b][r] + inj[k], 0.0f);
    }
    float s = 0.0f;
    for (int k = 0; k < TILES; ++k)
        for (int b = 0; b < 4; ++b) { s += fill[k][b]; for (int r = 0; r < 16; ++r) s += act[k][b][r]; }
    if (s == 12345.678f || out == nullptr) out[t] = s;
    if (blockIdx.x == 0 && threadIdx.x < 64) out[threadIdx.x] = act[0][0][threadIdx.x & 15];      // (a sample of the state: bounded? half zero?)
}

// ---- C: 16x16x32, two 16-point halves per wave ---------------------------------------------------------------------------------------------
// C/D of 16x16: lane (n = l & 15, g = l >> 4) holds rows 4 g + r of a 16-feature block; K-block q (32 features) takes, for lane group g, the
// 8 values {block 2q rows 4g..4g+3, block 2q+1 rows 4g..4g+3}: layers chain through registers once the weights are packed in that k-order.
template <int WPS>
__global__ __launch_bounds__(256 * WPS, 1) void chain16(float* out, int layers) {
    __shared__ __attribute__((aligned(16))) unsigned lds[W_WORDS];      // [K-block 4][out block 8][limb][lane][4 words]: the same 64 KB
    fill_weights(lds, 256 * WPS);
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const u32x4* wv = reinterpret_cast<const u32x4*>(lds) + lane;
    f32x4 acc[2][8], act[2][8];
    float inj[2], fill[2][4];
    for (int k = 0; k < 2; ++k) {
        inj[k] = urand(t * 7u + k);
        for (int b = 0; b < 4; ++b) fill[k][b] = urand(t * 13u + 4 * k + b);
        for (int b = 0; b < 8; ++b)
            for (int r = 0; r < 4; ++r) act[k][b][r] = fmaxf(urand(t * 64u + 4 * b + r + 1000 * k), 0.0f);
    }
    for (int l = 0; l < layers; ++l) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            u32x4 xh[2], xl[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float e[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) { e[i] = act[k][2 * q][i]; e[4 + i] = act[k][2 * q + 1][i]; }
                split8(e, xh[k], xl[k]);
            }
#pragma unroll
            for (int ob = 0; ob < 8; ++ob) {
                const u32x4 ah = wv[((q * 8 + ob) * 2 + 0) * 64], al = wv[((q * 8 + ob) * 2 + 1) * 64];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    if (q == 0) {
                        const f32x4 zero = {0, 0, 0, 0};
                        acc[k][ob] = MFMA16(ah, xl[k], zero);
                    } else acc[k][ob] = MFMA16(ah, xl[k], acc[k][ob]);
                    acc[k][ob] = MFMA16(al, xh[k], acc[k][ob]);
                    acc[k][ob] = MFMA16(ah, xh[k], acc[k][ob]);
                }
                // (3 x 2 MFMAs of 16 cycles = 96 cycles = 3 MFMA32: the same FILL per 32 MFMA-cycles and 32-point tile)
#pragma unroll
                for (int f = 0; f < 3 * FILL; ++f) fill[0][f & 3] = fmaf(fill[0][f & 3], 0.999f, inj[0]);
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int b = 0; b < 8; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) act[k][b][r] = fmaxf(acc[k][b][r] + inj[k], 0.0f);
    }
    float s = 0.0f;
    for (int k = 0; k < 2; ++k) {
        for (int b = 0; b < 4; ++b) s += fill[k][b];
        for (int b = 0; b < 8; ++b) for (int r = 0; r < 4; ++r) s += act[k][b][r];
    }
    if (s == 12345.678f || out == nullptr) out[t] = s;
    if (blockIdx.x == 0 && threadIdx.x < 64) out[threadIdx.x] = act[0][0][threadIdx.x & 3];
}

template <class K>
static void run(const char* name, K kernel, int wps, int points_per_wave) {
    float* out;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, layers = 6000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f, last = 0.0f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kernel, dim3(cus), dim3(256 * wps), 0, 0, out, layers);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        last = ms; if (rep >= 2 && ms < best) best = ms;
    }
    float h[64];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    int zeros = 0; float mx = 0.0f;
    for (int i = 0; i < 64; ++i) { zeros += h[i] == 0.0f; mx = h[i] > mx ? h[i] : mx; }
    const double point_layers = (double)cus * 4 * wps * points_per_wave * layers;
    const double flop = point_layers * 2.0 * 128 * 128 * 3;                      // executed MFMA FLOP (3 limb products)
    printf("%-72s %8.2f ms (last %8.2f)  %7.2f G point-layers/s  %7.1f TFLOP/s executed f16 MFMA = %5.1f %% of 2516.6   [state: max %.2f, %d/64 zero]\n", name, best, last,
           point_layers / best / 1e6, flop / best / 1e9, 100.0 * flop / best / 1e9 / 2516.6, mx, zeros);
    hipFree(out);
}

int main() {
    printf("decoder chain 128 -> 128, 2 f16 limbs (3 products), weights in LDS, splits + bias/ReLU + FILL=%d extra VALU per 32 MFMA-cycles and tile\n", FILL);
    run("A  32x32x16, 1 wave / SIMD, 2 tiles per wave (render_pass3)", chain32<1, 2>, 1, 64);
    run("B  32x32x16, 2 waves / SIMD, 1 tile per wave (decode_rays_limb)", chain32<2, 1>, 2, 32);
    run("C  16x16x32, 2 waves / SIMD, 1 tile per wave as 2 x 16 points", chain16<2>, 2, 32);
    run("C1 16x16x32, 1 wave / SIMD, 1 tile per wave as 2 x 16 points", chain16<1>, 1, 32);
    return 0;
}
