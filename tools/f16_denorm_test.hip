// Does v_mfma_f32_32x32x16_f16 honour subnormal f16 inputs on gfx950, and does v_cvt_pk_f16_f32 produce them?  (limb_core.h, LIMBS = 2:
// the low limb of a small operand is a subnormal.)   hipcc --offload-arch=gfx950 -O2 tools/f16_denorm_test.hip -o /tmp/f16_denorm && /tmp/f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float a, float b, float* out) {
    f16x2_t pa = __builtin_convertvector(f32x2_t{a, a}, f16x2_t), pb = __builtin_convertvector(f32x2_t{b, b}, f16x2_t);
    f16x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = pa[i & 1]; B[i] = pb[i & 1]; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)pa[0]; out[2] = (float)pb[0]; }
}
int main() {
    float* d; hipMalloc(&d, 64);
    const float cases[][2] = {{1.0f, 1.0f}, {0x1p-20f, 1.0f}, {1.0f, 0x1p-20f}, {0x1p-24f, 1.0f}, {0x1.8p-16f, 0x1p4f}, {0x1p-20f, 0x1p-20f}};
    for (auto& c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, c[0], c[1], d);
        float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        printf("a = %a (f16 %a)  b = %a (f16 %a)  mfma sum of 16 = %a  expected %a  %s\n", c[0], h[1], c[1], h[2], h[0], 16.0 * (double)h[1] * (double)h[2],
               h[0] == (float)(16.0 * (double)h[1] * (double)h[2]) ? "ok" : "DIFFERS");
    }
    return 0;
}
