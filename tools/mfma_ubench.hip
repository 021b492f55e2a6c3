#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// MODE 0: register-only MFMA chain (4 accumulators, dependent in groups of 4)
// MODE 1: + one ds_read_b128 per 4 MFMAs (A operand from LDS), waits as hipcc places them
// MODE 2: like 1 with accumulators interleaved
// MODE 3: like 1 plus NV independent VALU instructions in the shadow of every MFMA (the "side work" of render2.hip)
// MODE 4: like 3, the VALU instructions read a second accumulator set (v_accvgpr_read when that set lives in AGPRs)
template <int MODE, int NV = 2>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = 1e-3f * (i & 15);
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const int lane = threadIdx.x & 63;
    float b0 = 1.0f + lane * 1e-3f, b1 = 0.5f, b2 = 0.25f, b3 = 0.125f;
    const f32x4* wv = reinterpret_cast<const f32x4*>(lds) + lane;
    float side[16];
    for (int i = 0; i < 16; ++i) side[i] = 1.0f + i;
    f32x16 acc2[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc2[a][r] = 0.5f * r;
    if (MODE == 4) for (int a = 0; a < 4; ++a) acc2[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, b1, acc2[a], 0, 0, 0);
    f32x4 nexta = wv[0];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int g = 0; g < 64; ++g) {
                const int ib = g & 3;
                acc[ib] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, b1, acc[ib], 0, 0, 0);
                acc[ib] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1, b2, acc[ib], 0, 0, 0);
                acc[ib] = __builtin_amdgcn_mfma_f32_32x32x2f32(b2, b3, acc[ib], 0, 0, 0);
                acc[ib] = __builtin_amdgcn_mfma_f32_32x32x2f32(b3, b0, acc[ib], 0, 0, 0);
            }
        } else if (MODE == 1) {
            f32x4 a = wv[0];
#pragma unroll
            for (int g = 0; g < 64; ++g) {
                const int ib = g & 3;
                acc[ib] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b0, acc[ib], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                const f32x4 an = wv[((g + 1) & 63) * 64];
                __builtin_amdgcn_sched_barrier(0);
                acc[ib] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b1, acc[ib], 0, 0, 0);
                acc[ib] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b2, acc[ib], 0, 0, 0);
                acc[ib] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b3, acc[ib], 0, 0, 0);
                a = an;
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (MODE >= 3) {
            f32x4 a = wv[0];
#pragma unroll
            for (int g = 0; g < 64; ++g) {
                const int ib = g & 3;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[ib] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b0, acc[ib], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (j == 0) { const f32x4 an = wv[((g + 1) & 63) * 64]; side[15] += an[0]; a = (j == 3) ? an : a; nexta = an; }
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        const int e = (4 * g + j + 5 * v) & 15;          // independent fillers (a chain on one register prices latency)
                        if (MODE == 4) side[e] = fmaxf(acc2[(g >> 2) & 3][e], side[e]);
                        // volatile asm: a plain C++ expression here is moved OUT of the MFMA gaps by hipcc (it is independent of the MFMAs and
                        // sched_barrier only binds the machine scheduler), which then measures exposed VALU time, not VALU in a shadow
                        else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(side[e]) : "v"(side[(e + 1) & 15]));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                a = nexta;
            }
        } else {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                f32x4 a[4];
#pragma unroll
                for (int ib = 0; ib < 4; ++ib) a[ib] = wv[(g * 4 + ib) * 64];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int ib = 0; ib < 4; ++ib) acc[ib] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ib][j], b0, acc[ib], 0, 0, 0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r] + acc2[a][r];
    for (int i = 0; i < 16; ++i) s += side[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (float)(t1 - t0);
}
template <int MODE, int NV = 2>
void run(const char* name, int blocks) {
    float* out; hipMalloc(&out, ((1 << 20) + 16) * 4);
    const int iters = 200;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE, NV><<<blocks, 256>>>(out, 10);
    hipEventRecord(a);
    k<MODE, NV><<<blocks, 256>>>(out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    float cyc; hipMemcpy(&cyc, out + (1 << 20), 4, hipMemcpyDeviceToHost);
    const double mf = 256.0 * iters;   // MFMAs per wave
    printf("%-28s blocks %4d: %.3f ms  memtime ticks/MFMA(wave0) %.1f   TFLOP/s %.1f\n", name, blocks, ms, cyc / mf,
           blocks * 4 * mf * 4096 / ms / 1e9);
    hipFree(out);
}
int main() {
    for (int blocks : {256, 512}) {
        run<0>("regs only", blocks);
        run<1>("ds_read per 4 (prefetch)", blocks);
        run<2>("ds_read x4 per 16", blocks);
        run<3, 2>("+2 VALU per MFMA", blocks);
        run<3, 6>("+6 VALU per MFMA", blocks);
        run<3, 10>("+10 VALU per MFMA", blocks);
        run<3, 12>("+12 VALU per MFMA", blocks);
        run<3, 14>("+14 VALU per MFMA", blocks);
        run<4, 1>("+1 VALU reading 2nd acc set", blocks);
    }
    return 0;
}
