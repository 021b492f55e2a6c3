// Microbenchmark for the split-operand decoder layer: an exact f32 product as a sum of bf16 x bf16 products on the bf16 matrix pipe.
//   x = h + m + l   (three bf16 limbs by truncation: 8 + 8 + 8 significant bits, exact)
//   W x ~= Wh(xh + xm + xl) + Wm(xh + xm) + Wl xh      (6 of the 9 limb products; the dropped ones are <= 2^-24 |W||x|)
// One hidden layer [128 x 128] for a 32-point tile = 8 K-blocks x 4 output blocks x 6 v_mfma_f32_32x32x16_bf16, activations chained in
// registers (the 8 values of K-block (ib, half) in a lane are acc[ib][8*half .. 8*half+7] of the previous layer).
// hipcc --offload-arch=gfx950 -O3 tools/limb_ubench.hip -o /tmp/limb_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned hi_pair(float e1, float e0) {      // {bf16 trunc(e1), bf16 trunc(e0)}
    return __builtin_amdgcn_perm(__float_as_uint(e1), __float_as_uint(e0), 0x07060302u);
}
__device__ __forceinline__ float rem16(float x) { return x - __uint_as_float(__float_as_uint(x) & 0xffff0000u); }

template <int LIMBS>
struct Limbs { u32x4 v[LIMBS]; };

// slot-sliced split of the 8 values e[0..7]: pair j = slot / 6 handles e[2j], e[2j+1] in 6 slices
template <int LIMBS>
struct SplitPend { float r[2]; };
template <int LIMBS>
__device__ __forceinline__ void split_slot(int slot, const float (&e)[8], Limbs<LIMBS>& out, SplitPend<LIMBS>& p) {
    constexpr int SL = LIMBS == 3 ? 6 : 3;
    const int j = slot / SL, st = slot % SL;
    if (j >= 4) return;
    const float e0 = e[2 * j], e1 = e[2 * j + 1];
    if (st == 0) { out.v[0][j] = hi_pair(e1, e0); if (LIMBS > 1) p.r[0] = rem16(e0); }
    if (st == 1 && LIMBS > 1) { p.r[1] = rem16(e1); }
    if (st == 2 && LIMBS > 1) { out.v[1][j] = hi_pair(p.r[1], p.r[0]); }
    if (st == 3 && LIMBS > 2) { p.r[0] = rem16(p.r[0]); }
    if (st == 4 && LIMBS > 2) { p.r[1] = rem16(p.r[1]); }
    if (st == 5 && LIMBS > 2) { out.v[2][j] = hi_pair(p.r[1], p.r[0]); }
}

constexpr int nprod(int limbs) { return limbs == 3 ? 6 : limbs == 2 ? 3 : 1; }

// MODE 0: MFMAs + A-fragment reads only (B limbs constant)   MODE 1: + split of the next K-block in the gaps
// MODE 2: + NV extra VALU per gap (the other tile's side work)
template <int LIMBS, int MODE, int NV, int ORDER = 0>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned lds[24576];      // one layer of 3-limb fragments: 8 kb x 4 ob x 3 limbs x 1 KB
    for (int i = threadIdx.x; i < 24576; i += 256) lds[i] = 0x3c003c00u + (i & 7);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x16 acc[4], act[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) { acc[a][r] = 0.f; act[a][r] = 1.0f + 1e-3f * (lane + r + 16 * a); }
    float side[16];
    for (int i = 0; i < 16; ++i) side[i] = 1.0f + i;
    const u32x4* wv = reinterpret_cast<const u32x4*>(lds) + lane;
    Limbs<LIMBS> cur, nxt;
    for (int t = 0; t < LIMBS; ++t) { cur.v[t] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}; nxt.v[t] = cur.v[t]; }
    SplitPend<LIMBS> sp;
    sp.r[0] = sp.r[1] = 0.f;
    constexpr int NP = nprod(LIMBS);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (ORDER == 0) {
        u32x4 fa[LIMBS], fn[LIMBS];
#pragma unroll
        for (int t = 0; t < LIMBS; ++t) fa[t] = wv[t * 64];
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            float e[8];
            const int kn = (kb + 1) & 7;
#pragma unroll
            for (int i = 0; i < 8; ++i) e[i] = act[kn >> 1][8 * (kn & 1) + i];
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {
                const int q = kb * 4 + ob;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    // product order: (Wh,xl) (Wh,xm) (Wh,xh) (Wm,xm) (Wm,xh) (Wl,xh)   [2 limbs: (Wh,xl) (Wh,xh) (Wl,xh)]
                    int wa, xb;
                    if (LIMBS == 3) { const int WA[6] = {0, 0, 0, 1, 1, 2}, XB[6] = {2, 1, 0, 1, 0, 0}; wa = WA[p]; xb = XB[p]; }
                    else if (LIMBS == 2) { const int WA[3] = {0, 0, 1}, XB[3] = {1, 0, 0}; wa = WA[p]; xb = XB[p]; }
                    else { wa = 0; xb = 0; }
                    acc[ob] = mfma_bf16(fa[wa], cur.v[xb], acc[ob]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (p < LIMBS) fn[p] = wv[(((q + 1) & 31) * LIMBS + p) * 64];
                    if (MODE >= 1) split_slot<LIMBS>(ob * NP + p, e, nxt, sp);
                    if (MODE >= 2) {
#pragma unroll
                        for (int v = 0; v < NV; ++v) {
                            const int x = (q * NP + p + v) & 15;
                            asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(side[x]) : "v"(side[(x + 1) & 15]));
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int t = 0; t < LIMBS; ++t) fa[t] = fn[t];
            }
            if (MODE >= 1) cur = nxt;
        }
        } else {
            u32x4 fa[4][LIMBS], fn[4][LIMBS];
#pragma unroll
            for (int ob = 0; ob < 4; ++ob)
#pragma unroll
                for (int t = 0; t < LIMBS; ++t) fa[ob][t] = wv[(ob * LIMBS + t) * 64];
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                float e[8];
                const int kn = (kb + 1) & 7;
#pragma unroll
                for (int i = 0; i < 8; ++i) e[i] = act[kn >> 1][8 * (kn & 1) + i];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
#pragma unroll
                    for (int ob = 0; ob < 4; ++ob) {
                        int wa, xb;
                        if (LIMBS == 3) { const int WA[6] = {0, 0, 0, 1, 1, 2}, XB[6] = {2, 1, 0, 1, 0, 0}; wa = WA[p]; xb = XB[p]; }
                        else if (LIMBS == 2) { const int WA[3] = {0, 0, 1}, XB[3] = {1, 0, 0}; wa = WA[p]; xb = XB[p]; }
                        else { wa = 0; xb = 0; }
                        acc[ob] = mfma_bf16(fa[ob][wa], cur.v[xb], acc[ob]);
                        __builtin_amdgcn_sched_barrier(0);
                        const int slot = p * 4 + ob;
                        if (slot < 4 * LIMBS) fn[slot / LIMBS][slot % LIMBS] = wv[((((kb + 1) & 7) * 4 + slot / LIMBS) * LIMBS + slot % LIMBS) * 64];
                        if (MODE >= 1) split_slot<LIMBS>(slot, e, nxt, sp);
                        if (MODE >= 2) {
#pragma unroll
                            for (int v = 0; v < NV; ++v) {
                                const int x = (kb * 24 + slot + v) & 15;
                                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(side[x]) : "v"(side[(x + 1) & 15]));
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int ob = 0; ob < 4; ++ob)
#pragma unroll
                    for (int t = 0; t < LIMBS; ++t) fa[ob][t] = fn[ob][t];
                if (MODE >= 1) cur = nxt;
            }
        }
        // next layer: act = ReLU-ish of acc (kept trivial here), acc restarts
        if (MODE >= 1 && (it & 7) == 7) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) { act[a][r] = fmaxf(acc[a][r] * 1e-6f, 0.5f); acc[a][r] = 0.f; }
        }
    }
    float s = 0;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int i = 0; i < 16; ++i) s += side[i];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 19] = (float)(t1 - t0);
}

template <int LIMBS, int MODE, int NV, int ORDER = 0>
void run(const char* name, int blocks) {
    float* out; hipMalloc(&out, (1 << 20) * 4);
    const int iters = 400;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<LIMBS, MODE, NV, ORDER><<<blocks, 256>>>(out, 10);
    hipEventRecord(a);
    k<LIMBS, MODE, NV, ORDER><<<blocks, 256>>>(out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double mf = 32.0 * nprod(LIMBS) * iters;   // MFMAs per wave
    float cyc; hipMemcpy(&cyc, out + (1 << 19), 4, hipMemcpyDeviceToHost);
    // one tile-layer = 128 x 128 x 32 MACs
    printf("limbs %d %-34s blocks %4d: %.3f ms  ns/MFMA %.2f  memtime ticks/MFMA %.1f   us per tile-layer %.3f   f32-equivalent TFLOP/s %.1f\n", LIMBS, name, blocks,
           ms, ms * 1e6 / mf, cyc / mf, ms * 1e3 / iters, blocks * 4.0 * iters * 2.0 * 128 * 128 * 32 / ms / 1e9);
    hipFree(out);
}
int main() {
    const int blocks = 256;
    run<3, 0, 0>("mfma + frags", blocks);
    run<3, 1, 0>("+ split", blocks);
    run<3, 2, 1>("+ split + 1 VALU/gap", blocks);
    run<3, 2, 2>("+ split + 2 VALU/gap", blocks);
    run<3, 2, 3>("+ split + 3 VALU/gap", blocks);
    run<3, 2, 4>("+ split + 4 VALU/gap", blocks);
    run<3, 2, 5>("+ split + 5 VALU/gap", blocks);
    run<3, 2, 2, 1>("interleaved: + split + 2", blocks);
    run<3, 2, 4, 1>("interleaved: + split + 4", blocks);
    run<2, 0, 0>("mfma + frags", blocks);
    run<2, 1, 0>("+ split", blocks);
    run<2, 2, 1>("+ split + 1 VALU/gap", blocks);
    run<2, 2, 2>("+ split + 2 VALU/gap", blocks);
    run<2, 2, 3>("+ split + 3 VALU/gap", blocks);
    run<2, 2, 4>("+ split + 4 VALU/gap", blocks);
    return 0;
}
