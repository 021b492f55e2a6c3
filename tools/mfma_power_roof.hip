// The SUSTAINED rate of v_mfma_f32_32x32x16_bf16 on the whole chip: bare MFMA streams from registers, nothing else in the loop, long enough
// for the clock to settle (the dense peak of the guide, 2 516.6 TFLOP/s, is 256 CUs x 4 SIMDs x 1 024 FLOP/clk x 2.4 GHz; under this load the
// chip does not hold 2.4 GHz).  Operands: pseudo-random bf16 values (a hash of lane / register / iteration, |x| in [0.5, 2)) or all zeros --
// the rate depends on the data (toggling = power).  What a kernel built from 3-limb products can reach is this number / 6.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_power_roof.hip -o scratch/mfma_power_roof && scratch/mfma_power_roof
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// -DROOF_F16=1: the same streams with v_mfma_f32_{32x32x16,16x16x32}_f16 (the 2-f16-limb arithmetic's instruction): an 11-bit significand
// multiplier array toggles more than bf16's 8-bit one
#ifndef ROOF_F16
#define ROOF_F16 0
#endif
#if ROOF_F16
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0)
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0)
#define ROOF_TYPE "f16"
#else
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0)
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0)
#define ROOF_TYPE "bf16"
#endif

__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// two bf16 values with random sign / mantissa and exponent 126..127 (|v| in [0.5, 2))
__device__ __forceinline__ unsigned rnd_pair(unsigned s) {
    const unsigned r = hash(s);
#if ROOF_F16
    // f16: random sign and 10-bit mantissa, exponent field 14..15 (|v| in [0.5, 2))
    const unsigned l16 = (r & 0x83ffu) | (0x3800u + ((r >> 2) & 0x400u)), h16 = ((r >> 16) & 0x83ffu) | (0x3800u + ((r >> 18) & 0x400u));
    return l16 | (h16 << 16);
#endif
    const unsigned lo = (r & 0x807fu) | (0x3f00u + ((r >> 8) & 0x80u)), hi = ((r >> 16) & 0x807fu) | (0x3f00u + ((r >> 24) & 0x80u));
    return lo | (hi << 16);
}

template <int WAVES_PER_SIMD, int NACC>
__global__ __launch_bounds__(256 * WAVES_PER_SIMD, 1) void roof(float* out, int iters, int zeros) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int w = 0; w < 4; ++w) {
            a[i][w] = zeros ? 0u : rnd_pair(t * 64u + i * 8u + w);
            b[i][w] = zeros ? 0u : rnd_pair(t * 64u + 32u + i * 8u + w);
        }
    f32x16 acc[NACC];
    for (int k = 0; k < NACC; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int k = 0; k < NACC; ++k)
                acc[k] = MFMA32(a[(u + k) & 3], b[(u * 3 + k) & 3], acc[k]);
        if (!zeros) {           // keep the accumulators bounded (values stay O(1): random-sign products) and the operands changing
            a[it & 3][it & 3] ^= 0x00010001u;
        }
    }
    float s = 0.0f;
    for (int k = 0; k < NACC; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
    if (s == 12345.678f) out[t] = s;       // (never true: keeps the loop alive)
}

// the same with v_mfma_f32_16x16x32_bf16 (half the FLOP per instruction, 4 accumulator registers instead of 16)
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int WAVES_PER_SIMD, int NACC>
__global__ __launch_bounds__(256 * WAVES_PER_SIMD, 1) void roof16(float* out, int iters, int zeros) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int w = 0; w < 4; ++w) {
            a[i][w] = zeros ? 0u : rnd_pair(t * 64u + i * 8u + w);
            b[i][w] = zeros ? 0u : rnd_pair(t * 64u + 32u + i * 8u + w);
        }
    f32x4 acc[NACC];
    for (int k = 0; k < NACC; ++k) for (int r = 0; r < 4; ++r) acc[k][r] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int k = 0; k < NACC; ++k)
                acc[k] = MFMA16(a[(u + k) & 3], b[(u * 3 + k) & 3], acc[k]);
        if (!zeros) a[it & 3][it & 3] ^= 0x00010001u;
    }
    float s = 0.0f;
    for (int k = 0; k < NACC; ++k) for (int r = 0; r < 4; ++r) s += acc[k][r];
    if (s == 12345.678f) out[t] = s;
}
template <int WPS, int NACC>
static void run16(const char* name, int zeros) {
    float* out;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int dev_cus = p.multiProcessorCount, iters = 40000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f, last = 0.0f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((roof16<WPS, NACC>), dim3(dev_cus), dim3(256 * WPS), 0, 0, out, iters, zeros);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        last = ms; if (rep >= 2 && ms < best) best = ms;
    }
    const double mfmas = (double)dev_cus * 4 * WPS * iters * 16 * NACC;
    const double flop = mfmas * 2.0 * 16 * 16 * 32;
    printf("16x16x32 %-35s %s: %8.2f ms (last %8.2f)  %7.1f TFLOP/s dense " ROOF_TYPE " = %5.1f %% of 2516.6 ; / 6 = %6.1f, / 3 = %6.1f TFLOP/s of limb f32 work ; %.2f ns per MFMA and SIMD -> %.2f GHz x (16 cycles)\n",
           name, zeros ? "zeros " : "random", best, last, flop / best / 1e9, 100.0 * flop / best / 1e9 / 2516.6, flop / best / 1e9 / 6.0, flop / best / 1e9 / 3.0,
           best * 1e6 / (iters * 16.0 * NACC * WPS), 16.0 / (best * 1e6 / (iters * 16.0 * NACC * WPS)));
    hipFree(out);
}

template <int WPS, int NACC>
static void run(const char* name, int zeros) {
    float* out;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    int dev_cus = 256;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    dev_cus = p.multiProcessorCount;
    const int iters = 40000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f, last = 0.0f;
    for (int rep = 0; rep < 6; ++rep) {        // the first launches run while the clock is still settling: report the last and the best
        hipEventRecord(e0);
        hipLaunchKernelGGL((roof<WPS, NACC>), dim3(dev_cus), dim3(256 * WPS), 0, 0, out, iters, zeros);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        last = ms; if (rep >= 2 && ms < best) best = ms;
    }
    const double mfmas = (double)dev_cus * 4 * WPS * iters * 16 * NACC;          // per launch
    const double flop = mfmas * 2.0 * 32 * 32 * 16;
    printf("%-44s %s: %8.2f ms (last %8.2f)  %7.1f TFLOP/s dense " ROOF_TYPE " = %5.1f %% of 2516.6 ; / 6 = %6.1f, / 3 = %6.1f TFLOP/s of limb f32 work ; %.2f ns per MFMA and SIMD -> %.2f GHz x (32 cycles)\n",
           name, zeros ? "zeros " : "random", best, last, flop / best / 1e9, 100.0 * flop / best / 1e9 / 2516.6, flop / best / 1e9 / 6.0, flop / best / 1e9 / 3.0,
           best * 1e6 / (iters * 16.0 * NACC * WPS), 32.0 / (best * 1e6 / (iters * 16.0 * NACC * WPS)));
    hipFree(out);
}

int main() {
    run<1, 4>("1 wave / SIMD, 4 accumulators", 0);
    run<1, 4>("1 wave / SIMD, 4 accumulators", 1);
    run<1, 1>("1 wave / SIMD, 1 accumulator (dependent chain)", 0);
    run<2, 4>("2 waves / SIMD, 4 accumulators", 0);
    run16<1, 8>("1 wave / SIMD, 8 accumulators", 0);
    run16<1, 8>("1 wave / SIMD, 8 accumulators", 1);
    run16<1, 1>("1 wave / SIMD, 1 accumulator", 0);
    run16<2, 8>("2 waves / SIMD, 8 accumulators", 0);
    return 0;
}
