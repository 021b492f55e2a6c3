// Throughput of the bilinear-tap gather of a 32-point tile from a channel-last plane ([H][W][48] f32, 192 B per texel), by lane mapping.
//   MODE 0  lane = (point, half): every lane reads its point's 96-B half of a texel with 6 dwordx4 loads per tap (render2 / render3)
//   MODE 1  16 lanes per point (12 active), lane c reads bytes 16c..16c+15 of the texel: one load instruction covers 4 whole texels
//   MODE 2  like 0 with dwordx2 loads (12 per tap) -- to see whether the cost is per instruction, per lane or per byte
// Points of a tile are neighbours on the plane (0.36 texel apart, like adjacent pixels of an 800^2 view on an 800^2 plane); every
// (wave, step) starts at a pseudo-random texel.  One 256-thread workgroup per CU unless WG2.
// hipcc --offload-arch=gfx950 -O3 tools/gather_ubench.hip -o /tmp/gather_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int R = 800;

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const float* __restrict__ plane, float* out, int steps, float spacing) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned seed = (blockIdx.x * 4 + wave) * 2654435761u + 12345u;
    f32x4 acc = {0, 0, 0, 0};
    for (int s = 0; s < steps; ++s) {
        seed = seed * 1664525u + 1013904223u;
        const float bx = (float)((seed >> 8) % (R - 16)), by = (float)((seed >> 18) % (R - 2));
        if (MODE == 0 || MODE == 2) {
            const int pt = lane & 31, h = lane >> 5;
            const float x = bx + spacing * pt, y = by + 0.37f;
            const int ix = (int)x, iy = (int)y;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float* p = plane + ((long)(iy + (t >> 1)) * R + ix + (t & 1)) * 48 + 24 * h;
                if (MODE == 0) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) acc += reinterpret_cast<const f32x4*>(p)[i];
                } else {
#pragma unroll
                    for (int i = 0; i < 12; ++i) { const f32x2 v = reinterpret_cast<const f32x2*>(p)[i]; acc[0] += v[0]; acc[1] += v[1]; }
                }
            }
        } else if (MODE == 3 || MODE == 4) {
            // 4 lanes per point read 64 contiguous bytes per instruction: MODE 3 the 4 lanes are a quad (4q..4q+3), MODE 4 they are
            // 16 apart (n, n+16, n+32, n+48: the B-operand layout of the 16x16x32 MFMA); 16 points per instruction, 3 instructions per tap
            const int q = MODE == 3 ? (lane >> 2) : (lane & 15), r = MODE == 3 ? (lane & 3) : (lane >> 4);
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int pt = 16 * g + q;
                const float x = bx + spacing * pt, y = by + 0.37f;
                const int ix = (int)x, iy = (int)y;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float* p = plane + ((long)(iy + (t >> 1)) * R + ix + (t & 1)) * 48 + 4 * r;
#pragma unroll
                    for (int i = 0; i < 3; ++i) acc += *reinterpret_cast<const f32x4*>(p + 16 * i);
                }
            }
        } else {
            const int c = lane & 15, sub = lane >> 4;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int pt = 4 * g + sub;
                const float x = bx + spacing * pt, y = by + 0.37f;
                const int ix = (int)x, iy = (int)y;
                if (c < 12) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float* p = plane + ((long)(iy + (t >> 1)) * R + ix + (t & 1)) * 48 + 4 * c;
                        acc += *reinterpret_cast<const f32x4*>(p);
                    }
                }
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int MODE>
void run(const char* name, const float* plane, float* out, int blocks, float spacing) {
    const int steps = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 256>>>(plane, out, 50, spacing);
    hipEventRecord(a);
    k<MODE><<<blocks, 256>>>(plane, out, steps, spacing);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)blocks * 4 * steps * 32 * 4 * 192;     // algorithmic: 32 points x 4 taps x 192 B per wave-step
    printf("%-44s spacing %.2f blocks %4d: %.3f ms   %.1f GB/s per CU   %.2f TB/s chip   %.2f us per tile-gather per CU-wave\n", name, spacing, blocks, ms,
           bytes / blocks * (blocks > 256 ? blocks / 256.0 : 1.0) / ms / 1e6, bytes / ms / 1e9, ms * 1e3 / steps);
}
int main() {
    float *plane, *out;
    const size_t n = (size_t)R * R * 48;
    hipMalloc(&plane, n * 4 * 3);
    hipMemset(plane, 0, n * 4 * 3);
    hipMalloc(&out, 1 << 22);
    for (float sp : {0.36f, 2.0f}) {
        for (int blocks : {256, 512}) {
            run<0>("lane = (point, half), 6 x dwordx4 per tap", plane, out, blocks, sp);
            run<1>("16 lanes per texel, 1 x dwordx4 per tap", plane, out, blocks, sp);
            run<2>("lane = (point, half), 12 x dwordx2 per tap", plane, out, blocks, sp);
            run<3>("quad of lanes per point, 64 B per quad", plane, out, blocks, sp);
            run<4>("lanes n, n+16, n+32, n+48 per point", plane, out, blocks, sp);
        }
    }
    return 0;
}
