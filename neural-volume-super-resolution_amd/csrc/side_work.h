// Side-work slices shared by the two-tiles-per-wave render kernels (render2.hip: f32 matrix pipe, render3.hip: bf16 limbs): everything
// non-MFMA of one tile, cut into pieces that are issued in the MFMA shadows of the other tile.
#pragma once
#include "tile_pair.h"

namespace nvsr {

// two single-tap buffers (48 registers): the accumulators of both tiles fill the 256 AGPRs, everything the VALU touches
// (features, blends, tap data) has to fit the 256 architectural VGPRs
struct RawTaps2 { f32x4 r0[HALF_C / 4], r1[HALF_C / 4]; };

// ---- side-work slices ---------------------------------------------------------------------------------------------------
// gather of one plane for one tile inside a 24-group block:
//   g 0,1   : loads of taps nw, ne (6 dwordx4 each)            g 8..10 : F  = nw*T0 + ne*T1      (8 channels per group)
//   g 11,12 : loads of taps sw, se into the same buffers        g 20..22: F += sw*T2 + se*T3
// i.e. ~8 groups (2000 cycles) between a load and its use.
struct GatherJob {
    const float* plane;
    Taps t;
};

template <int G0 = 0>
__device__ __forceinline__ void gather_side(int g_, int j, const GatherJob& job, int h, RawTaps2& rt, float (&F)[HALF_C]) {
    const int g = g_ - G0;                                                // the 24-group schedule starts at group G0 of the block
    if ((g == 0 || g == 1 || g == 11 || g == 12) && j < 3) {            // 2 of the tap's 6 loads per slot
        const int off = (g == 0) ? job.t.o00 : (g == 1) ? job.t.o01 : (g == 11) ? job.t.o10 : job.t.o11;
        const f32x4* p = reinterpret_cast<const f32x4*>(job.plane + off + HALF_C * h);
#pragma unroll
        for (int i = 2 * j; i < 2 * j + 2; ++i) {
            if (g == 0 || g == 11) rt.r0[i] = p[i]; else rt.r1[i] = p[i];
        }
    }
    if (g >= 8 && g <= 10) {                                              // 2 channels per slot
#pragma unroll
        for (int k = 2 * j; k < 2 * j + 2; ++k) {
            const int c = 8 * (g - 8) + k, i = c >> 2, jj = c & 3;
            F[c] = fmaf(rt.r1[i][jj], job.t.ne, rt.r0[i][jj] * job.t.nw);
        }
    }
    if (g >= 20 && g <= 22) {
#pragma unroll
        for (int k = 2 * j; k < 2 * j + 2; ++k) {
            const int c = 8 * (g - 20) + k, i = c >> 2, jj = c & 3;
            F[c] = fmaf(rt.r1[i][jj], job.t.se, fmaf(rt.r0[i][jj], job.t.sw, F[c]));
        }
    }
}

// partial dot products of NH heads: group g handles the 4 features of (ib, q) = (g >> 2, g & 3), g < 16: weights read in slot 0, FMAs in slot 2
template <int NH>
struct HeadPend { f32x4 w[NH]; };
template <int NH>
__device__ __forceinline__ void heads_side(int g, int j, const float* w, int h, const f32x16 (&in)[4], float (&s)[NH], HeadPend<NH>& pend) {
    if (g >= 0 && g < 16) {
        if (j == 0) {
#pragma unroll
            for (int n = 0; n < NH; ++n) pend.w[n] = *reinterpret_cast<const f32x4*>(w + n * HID + g * 8 + h * 4);
        }
        if (j == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int n = 0; n < NH; ++n) s[n] = fmaf(in[g >> 2][4 * (g & 3) + e], pend.w[n][e], s[n]);
        }
    }
}
// one sample of volume_render_radiance_field for one tile (volume_rendering_utils.py:18-45)
template <class TileT>
__device__ __forceinline__ void composite_sample(TileT& t, float nrm, float noise, bool last) {
    const float dist = __fmul_rn(last ? 1e10f : __fsub_rn(t.zn, t.zc), nrm);
    // relu(sigma + noise) by compare + select: a NaN stays a NaN like torch.relu's (volume_rendering_utils.py:30) instead of becoming an
    // empty sample (v_max_f32 returns its non-NaN operand)
    const float sn = __fadd_rn(t.raw[3], noise);
    const float sig = sn < 0.0f ? 0.0f : sn;
    const float alpha = __fsub_rn(1.0f, expf(-__fmul_rn(sig, dist)));
    const float w = __fmul_rn(alpha, t.T);
    t.T = __fmul_rn(t.T, __fadd_rn(__fsub_rn(1.0f, alpha), 1e-10f));
    t.cr = __fadd_rn(t.cr, __fmul_rn(w, 1.0f / (1.0f + expf(-t.raw[0]))));
    t.cg = __fadd_rn(t.cg, __fmul_rn(w, 1.0f / (1.0f + expf(-t.raw[1]))));
    t.cb = __fadd_rn(t.cb, __fmul_rn(w, 1.0f / (1.0f + expf(-t.raw[2]))));
    t.dep = __fadd_rn(t.dep, __fmul_rn(w, t.zc));
    t.ac = __fadd_rn(t.ac, w);
    t.raw[3] = w;     // hand the weight back to the caller (stored when requested)
}

__device__ __forceinline__ Taps pos_taps2(const SceneDev& sc, int d, float n0, float n1, float n2) {
    const float* M = sc.proj + 6 * d;
    return make_taps(sc, d, n0 * M[0] + n1 * M[2] + n2 * M[4], n0 * M[1] + n1 * M[3] + n2 * M[5]);
}

}  // namespace nvsr
