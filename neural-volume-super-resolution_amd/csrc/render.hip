// Tri-plane decoder + fused per-ray render pass for gfx950 (MI355X).
//
// Reference functions replaced (upstream paths): TwoDimPlanesModel.forward models.py:381-421 (normalize_coords :261-268,
// CoordProjector :495-497, project_xyz/project_viewdir :289-326, combine_* :355-379), run_network train_utils.py:15-64,
// volume_render_radiance_field volume_rendering_utils.py:6-51.
//
// Design:
//   * one workgroup = 8 waves = 256 points per step (32 points per wave, one point per lane-pair l / l+32);
//   * the decoder runs on v_mfma_f32_32x32x2_f32 (exact fp32): D[feature][point] = W[feature][k] * H[k][point];
//     the accumulator (C/D) register layout of one layer IS the B-operand layout of the next, so activations never
//     leave registers; bias is the accumulator's initial value, ReLU is applied in place;
//   * weights (518 KB) stream L2 -> LDS as pre-permuted A fragments by LDS-DMA (global_load_lds, 16 B/lane) into a
//     2 x 64 KB ring, one chunk (<= 64 KB) ahead of the MFMAs that consume it; biases and the two heads stay in LDS;
//   * plane features: each lane gathers 24 of the 48 channels (96 contiguous bytes) of the 4 bilinear taps straight into the
//     B-fragment registers (channel-last planes), blends in registers;
//   * the render pass walks the samples of its 32 rays front to back, so transmittance is a running product in a register
//     (the reference's exclusive cumprod, in the same order) and only per-ray results are written.
#include <stdio.h>
#include <cstdlib>

#include <cstring>

#include "decode_core.h"

namespace nvsr {

// Workgroup = 4 waves (one per SIMD); two workgroups fit one CU (2 x 80 KB of LDS, 2 waves per SIMD at <= 256 VGPRs).
constexpr int TPB = 256;
constexpr int NWAVES = TPB / 64;
constexpr int PTS_PER_WG = NWAVES * 32;

// =====================================================================================================================
// TwoDimPlanesModel.forward on an explicit point list x[P,6] -> out[P,4]
// =====================================================================================================================
__global__ __launch_bounds__(TPB, 1) void triplane_decode_kernel(SceneDev sc, const float* __restrict__ packed, long P,
                                                                 const float* __restrict__ x, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    NVSR_RACE_PROBE_DELAY(lds);      // (probe builds only, nvsr_common.h)
    RingState rs{packed, lds, 0, (int)(threadIdx.x >> 6), (int)(threadIdx.x & 63), (threadIdx.x >> 6) * 1024u + (threadIdx.x & 63) * 16u};
    decode_prologue<NWAVES>(rs);
    const long ntiles = (P + PTS_PER_WG - 1) / PTS_PER_WG;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {   // uniform trip count per workgroup
        long p = tile * PTS_PER_WG + rs.wave * 32 + (rs.lane & 31);
        const bool valid = p < P;
        if (!valid) p = P - 1;
        const float* xp = x + p * 6;
        const Taps vt = view_taps(sc, xp[3], xp[4], xp[5]);
        float raw[4];
        decode_step<NWAVES>(sc, rs, xp[0], xp[1], xp[2], vt, raw);
        if (valid && rs.lane < 32) *reinterpret_cast<f32x4*>(out + p * 4) = f32x4{raw[0], raw[1], raw[2], raw[3]};
    }
}

// =====================================================================================================================
// Decoder over the samples of a ray block WITHOUT fusing the compositing: raw [N,S,4].  Used when there are too few rays to
// fill the chip with one workgroup per 128 rays (training batches: 4096 rays = 32 workgroups): tiles are (ray block, sample)
// pairs, so N*S/128 workgroup-steps are spread over the whole grid; nvsr_composite then consumes raw.
// =====================================================================================================================
template <bool MASKS, bool RECORD>
__global__ __launch_bounds__(TPB, 1) void decode_rays_kernel(SceneDev sc, const float* __restrict__ packed, long N, int S,
                                                             const float* __restrict__ rays, const float* __restrict__ z,
                                                             float* __restrict__ raw_out, unsigned* __restrict__ gates, DecRecord rec) {
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    NVSR_RACE_PROBE_DELAY(lds);      // (probe builds only, nvsr_common.h)
    RingState rs{packed, lds, 0, (int)(threadIdx.x >> 6), (int)(threadIdx.x & 63), (threadIdx.x >> 6) * 1024u + (threadIdx.x & 63) * 16u};
    decode_prologue<NWAVES>(rs);
    const long nrb = (N + PTS_PER_WG - 1) / PTS_PER_WG;
    const long ntiles = nrb * S;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {   // uniform trip count per workgroup
        const long rb = tile / S;
        const int s = (int)(tile - rb * S);
        long ray = rb * PTS_PER_WG + rs.wave * 32 + (rs.lane & 31);
        const bool valid = ray < N;
        if (!valid) ray = N - 1;
        const float* r = rays + ray * 11;
        const float zc = z[ray * S + s];
        const Taps vt = view_taps(sc, r[8], r[9], r[10]);
        float raw[4];
        // gate record of this lane: [point ray*S+s][lane half][16 words]; padding lanes of the last ray block rewrite ray N-1's
        // record with identical values
        unsigned* gl = MASKS ? gates + ((ray * S + s) * 2 + (rs.lane >> 5)) * 16 : nullptr;
        decode_step<NWAVES, MASKS, RECORD>(sc, rs, __fadd_rn(r[0], __fmul_rn(r[3], zc)), __fadd_rn(r[1], __fmul_rn(r[4], zc)),
                                           __fadd_rn(r[2], __fmul_rn(r[5], zc)), vt, raw, gl, &rec, record_row(ray, s, N, S), valid);
        if (valid && rs.lane < 32) *reinterpret_cast<f32x4*>(raw_out + (ray * S + s) * 4) = f32x4{raw[0], raw[1], raw[2], raw[3]};
    }
}

// =====================================================================================================================
// Fused render pass: rays [N,11], depths z [N,S] -> per-ray rgb / disp / acc (/ weights / depth)
// =====================================================================================================================
// Per-ray constants live in LDS (16 floats per ray, 8 KB per workgroup) and are re-read every step: kept in VGPRs they push
// the 2016-MFMA step body over 256 registers, and a spill reload is a vector-memory load whose wait also waits for the
// (older, in-order) LDS-DMA of the next weight chunk -- ~2 us per reload point.
constexpr int RAY_FLOATS = 16;
constexpr int RENDER_LDS_FLOATS = LDS_FLOATS + PTS_PER_WG * RAY_FLOATS;
static_assert(RENDER_LDS_FLOATS * 4 <= 80 * 1024, "two workgroups must fit one CU's 160 KB of LDS");

__global__ __launch_bounds__(TPB, 1) void render_pass_kernel(SceneDev sc, const float* __restrict__ packed, long N, int S,
                                                             const float* __restrict__ rays, const float* __restrict__ z,
                                                             const float* __restrict__ noise, int white,
                                                             float* __restrict__ rgb, float* __restrict__ disp,
                                                             float* __restrict__ acc, float* __restrict__ weights,
                                                             float* __restrict__ depth, float* __restrict__ raw_out) {
    __shared__ __attribute__((aligned(16))) float lds[RENDER_LDS_FLOATS];
    NVSR_RACE_PROBE_DELAY(lds);      // (probe builds only, nvsr_common.h)
    RingState rs{packed, lds, 0, (int)(threadIdx.x >> 6), (int)(threadIdx.x & 63), (threadIdx.x >> 6) * 1024u + (threadIdx.x & 63) * 16u};
    decode_prologue<NWAVES>(rs);

    const long ray0 = (long)blockIdx.x * PTS_PER_WG + rs.wave * 32 + (rs.lane & 31);
    const bool valid = ray0 < N;
    const long ray = valid ? ray0 : N - 1;
    float* rc = lds + LDS_FLOATS + (rs.wave * 32 + (rs.lane & 31)) * RAY_FLOATS;
    {
        const float* r = rays + ray * 11;
        const float dx = r[3], dy = r[4], dz = r[5];
        const Taps vt = view_taps(sc, r[8], r[9], r[10]);
        // dists * ||rd||  (volume_rendering_utils.py:27)
        const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
        if (rs.lane < 32) {
            reinterpret_cast<f32x4*>(rc)[0] = f32x4{r[0], r[1], r[2], dx};
            reinterpret_cast<f32x4*>(rc)[1] = f32x4{dy, dz, nrm, 0.0f};
            reinterpret_cast<f32x4*>(rc)[2] = f32x4{__int_as_float(vt.o00), __int_as_float(vt.o01), __int_as_float(vt.o10), __int_as_float(vt.o11)};
            reinterpret_cast<f32x4*>(rc)[3] = f32x4{vt.nw, vt.ne, vt.sw, vt.se};
        }
    }
    const float* zr = z + ray * S;
    float T = 1.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f, dep = 0.0f, ac = 0.0f;
    float zc = zr[0];
#if NVSR_ABLATE & 16
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
    for (int s = 0; s < S; ++s) {
        // next depth / this sample's noise: issued before this step's DMAs so that their waits do not include them
        const float zn = (s + 1 < S) ? zr[s + 1] : 0.0f;
        const float nzs = noise ? noise[ray * S + s] : 0.0f;
        const f32x4 c0 = reinterpret_cast<const f32x4*>(rc)[0], c1 = reinterpret_cast<const f32x4*>(rc)[1];
        const f32x4 c2 = reinterpret_cast<const f32x4*>(rc)[2], c3 = reinterpret_cast<const f32x4*>(rc)[3];
        Taps vt;
        vt.o00 = __float_as_int(c2[0]); vt.o01 = __float_as_int(c2[1]); vt.o10 = __float_as_int(c2[2]); vt.o11 = __float_as_int(c2[3]);
        vt.nw = c3[0]; vt.ne = c3[1]; vt.sw = c3[2]; vt.se = c3[3];
        float raw[4];
        decode_step<NWAVES>(sc, rs, __fadd_rn(c0[0], __fmul_rn(c0[3], zc)), __fadd_rn(c0[1], __fmul_rn(c1[0], zc)),
                    __fadd_rn(c0[2], __fmul_rn(c1[1], zc)), vt, raw);
        // volume_render_radiance_field, one sample (volume_rendering_utils.py:18-45)
        const float nrm = reinterpret_cast<const f32x4*>(rc)[1][2];
        const float dist = __fmul_rn((s + 1 < S) ? __fsub_rn(zn, zc) : 1e10f, nrm);
        const float sig = fmaxf(__fadd_rn(raw[3], nzs), 0.0f);
        const float alpha = __fsub_rn(1.0f, expf(-__fmul_rn(sig, dist)));
        const float w = __fmul_rn(alpha, T);
        T = __fmul_rn(T, __fadd_rn(__fsub_rn(1.0f, alpha), 1e-10f));
        cr = __fadd_rn(cr, __fmul_rn(w, 1.0f / (1.0f + expf(-raw[0]))));
#if NVSR_ABLATE & 64
        cg = raw[1];   // debug: cycles of one hidden half (128 MFMAs)
#else
        cg = __fadd_rn(cg, __fmul_rn(w, 1.0f / (1.0f + expf(-raw[1]))));
#endif
        cb = __fadd_rn(cb, __fmul_rn(w, 1.0f / (1.0f + expf(-raw[2]))));
        dep = __fadd_rn(dep, __fmul_rn(w, zc));
        ac = __fadd_rn(ac, w);
        if (weights && valid && rs.lane < 32) weights[ray * S + s] = w;
        if (raw_out && valid && rs.lane < 32) *reinterpret_cast<f32x4*>(raw_out + (ray * S + s) * 4) = f32x4{raw[0], raw[1], raw[2], raw[3]};
        zc = zn;
    }
    ring_sync();
#if NVSR_ABLATE & 16
    if (threadIdx.x == 0) acc[N + blockIdx.x] = (NVSR_ABLATE & 64) ? cg : (float)(__builtin_amdgcn_s_memtime() - t_begin) / (float)S;   // debug build
#endif
    if (valid && rs.lane < 32) {
        const float q = dep / ac;                       // NaN when acc == 0, like torch.max(1e-10, nan)
        disp[ray] = 1.0f / ((q != q) ? q : fmaxf(1e-10f, q));
        if (white) { const float bg = 1.0f - ac; cr += bg; cg += bg; cb += bg; }
        rgb[ray * 3 + 0] = cr; rgb[ray * 3 + 1] = cg; rgb[ray * 3 + 2] = cb;
        acc[ray] = ac;
        if (depth) depth[ray] = dep;
    }
}

// =====================================================================================================================
// natural (state-dict) blob -> packed blob
// =====================================================================================================================
__device__ __forceinline__ int bias_src(int layer, int f) {
    if (layer == 0) return N_DEN_B0 + f;
    if (layer < 4) return N_DEN_W1 + (layer - 1) * N_HID_STRIDE + HID * HID + f;
    if (layer == 4) return N_RGB_B0 + f;
    return N_RGB_W1 + (layer - 5) * N_HID_STRIDE + HID * HID + f;
}

__global__ void pack_decoder_kernel(const float* __restrict__ nat, float* __restrict__ packed) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= NVSR_DECODER_PACKED_F32_FLOATS) return;
    int src = -1;
    if (idx < P_SMALL) {
        const int j = idx & 3, lane = (idx >> 2) & 63, ib = (idx >> 8) & 3;
        const int i = 32 * ib + (lane & 31), h = lane >> 5;
        if (idx < P_DEN0) {                                     // rgb layer 0: [p][q][ib][lane][j]
            const int p = idx / P_PLANE_FLOATS, q = (idx % P_PLANE_FLOATS) >> 10;
            src = N_RGB_W0 + i * (4 * C) + C * p + HALF_C * h + 4 * q + j;
        } else if (idx < P_DEN1) {                              // density layer 0
            const int q = (idx - P_DEN0) >> 10;
            src = N_DEN_W0 + i * C + HALF_C * h + 4 * q + j;
        } else {                                                // hidden layers: [kb][q][ib][lane][j]
            const bool is_rgb = idx >= P_RGB1;
            const int rem0 = idx - (is_rgb ? P_RGB1 : P_DEN1);
            const int l = rem0 / P_HID_FLOATS, rem = rem0 % P_HID_FLOATS;
            const int kb = rem >> 12, q = (rem >> 10) & 3;
            const int k = 32 * kb + 8 * q + 4 * h + j;
            src = (is_rgb ? N_RGB_W1 : N_DEN_W1) + l * N_HID_STRIDE + i * HID + k;
        }
    } else {
        const int s = idx - P_SMALL;
        if (s < S_HEAD_B) {                                     // [vec][ib][q][h][j]
            const int vec = s >> 7, rem = s & 127;
            const int ib = rem >> 5, q = (rem >> 3) & 3, h = (rem >> 2) & 1, j = rem & 3;
            const int f = 32 * ib + 8 * q + 4 * h + j;
            if (vec < 8) src = bias_src(vec, f);
            else if (vec == 8) src = N_ALPHA_W + f;
            else src = N_FCRGB_W + (vec - 9) * HID + f;
        } else if (s == S_HEAD_B) src = N_ALPHA_B;
        else if (s < S_HEAD_B + 4) src = N_FCRGB_B + (s - S_HEAD_B - 1);
    }
    packed[idx] = src >= 0 ? nat[src] : 0.0f;
}

}  // namespace nvsr

using namespace nvsr;

extern "C" int nvsr_render_pass2_launch(const nvsr_scene* scene, const float* packed_decoder, int64_t N, int S, const float* rays,
                                        const float* z, const float* noise, int white_bkgd, float* rgb, float* disp, float* acc,
                                        float* weights, float* depth, float* raw_out, nvsr_stream_t stream);
extern "C" int nvsr_render_pass3_launch(int limbs, const nvsr_scene* scene, const float* packed_decoder, int64_t N, int S, const float* rays,
                                        const float* z, const float* noise, int white_bkgd, float* rgb, float* disp, float* acc,
                                        float* weights, float* depth, float* raw_out, nvsr_stream_t stream);
extern "C" int nvsr_pack_decoder_limbs_launch(const float* natural, float* packed, nvsr_stream_t stream);
extern "C" int nvsr_decode_rays_limb_launch(int limbs, const nvsr_scene* scene, const float* packed_decoder, int64_t N, int S, const float* rays,
                                            const float* z, float* raw, uint32_t* gates, float* record, nvsr_stream_t stream);

extern "C" int nvsr_decode_rays_pair_launch(const nvsr_scene* scene, const float* packed_decoder, int64_t N, int S, const float* rays, const float* z,
                                            float* raw, uint32_t* gates, nvsr_stream_t stream);

// arithmetic of the fused render pass (process-wide): -1 = not yet read from the environment
static int g_decoder_arithmetic = -1;

extern "C" {

// "f32" | "bf16x3" | "f16x2" -> the mode; unset or empty -> dflt; anything else -> NVSR_ARITH_INVALID (said once on stderr): every call that
// inherits the process default then fails with NVSR_ERR_SHAPE instead of silently running the default (round 3 renamed bf16x2 -> f16x2:
// a stale NVSR_DECODER_ARITHMETIC=bf16x2 must not quietly select another arithmetic)
extern "C" int nvsr_internal_parse_arith_env(const char* name, int dflt) {
    const char* e = getenv(name);
    if (!e || !*e) return dflt;
    if (!strcmp(e, "f32")) return NVSR_ARITH_F32;
    if (!strcmp(e, "bf16x3")) return NVSR_ARITH_BF16X3;
    if (!strcmp(e, "f16x2")) return NVSR_ARITH_F16X2;
    fprintf(stderr, "nvsr: %s=%s is not one of f32 | bf16x3 | f16x2\n", name, e);
    return NVSR_ARITH_INVALID;
}

int nvsr_get_decoder_arithmetic(void) {
    if (g_decoder_arithmetic == -1) g_decoder_arithmetic = nvsr_internal_parse_arith_env("NVSR_DECODER_ARITHMETIC", NVSR_ARITH_DEFAULT);
    return g_decoder_arithmetic;
}

/* NVSR_ARITH_INHERIT -> the process default; anything that is not a mode -> -1 */
int nvsr_internal_resolve_decoder_arith(int arithmetic) {
    if (arithmetic == NVSR_ARITH_INHERIT) arithmetic = nvsr_get_decoder_arithmetic();
    return (arithmetic == NVSR_ARITH_F32 || arithmetic == NVSR_ARITH_BF16X3 || arithmetic == NVSR_ARITH_F16X2) ? arithmetic : -1;
}

static uint32_t* g_range_flag = nullptr;
int nvsr_set_range_flag(uint32_t* device_word) { g_range_flag = device_word; return NVSR_OK; }
uint32_t* nvsr_get_range_flag(void) { return g_range_flag; }

int nvsr_set_decoder_arithmetic(int mode) {
    if (mode != NVSR_ARITH_F32 && mode != NVSR_ARITH_BF16X3 && mode != NVSR_ARITH_F16X2) return NVSR_ERR_SHAPE;
    g_decoder_arithmetic = mode;
    return NVSR_OK;
}

int nvsr_pack_decoder(const float* natural, float* packed, nvsr_stream_t stream) {
    if (!natural || !packed) return NVSR_ERR_NULL;
    if (!aligned16(packed)) return NVSR_ERR_ALIGN;
    const int n = NVSR_DECODER_PACKED_F32_FLOATS;
    hipLaunchKernelGGL(pack_decoder_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, natural, packed);
    if (hipGetLastError() != hipSuccess) return NVSR_ERR_LAUNCH;
    return nvsr_pack_decoder_limbs_launch(natural, packed, stream);
}

static int check_scene(const nvsr_scene* s) {
    if (!s) return NVSR_ERR_NULL;
    for (int d = 0; d < 4; ++d) {
        if (!s->planes[d]) return NVSR_ERR_NULL;
        if (!aligned16(s->planes[d])) return NVSR_ERR_ALIGN;
        if (s->ph[d] < 1 || s->pw[d] < 1 || (int64_t)s->ph[d] * s->pw[d] * NVSR_PLANE_CHANNELS >= (int64_t)1 << 31) return NVSR_ERR_SHAPE;
    }
    return NVSR_OK;
}

int nvsr_triplane_decode(const nvsr_scene* scene, const float* packed_decoder, int64_t P, const float* x, float* out,
                         nvsr_stream_t stream) {
    return nvsr_triplane_decode_arith(scene, packed_decoder, P, x, out, NVSR_ARITH_INHERIT, stream);
}

int nvsr_triplane_decode_arith(const nvsr_scene* scene, const float* packed_decoder, int64_t P, const float* x, float* out, int arithmetic,
                               nvsr_stream_t stream) {
    const int arith = nvsr_internal_resolve_decoder_arith(arithmetic);
    if (arith < 0) return NVSR_ERR_SHAPE;
    if (int e = check_scene(scene)) return e;
    if (!packed_decoder || !x || !out) return NVSR_ERR_NULL;
    if (!aligned16(packed_decoder) || !aligned16(out)) return NVSR_ERR_ALIGN;
    if (P < 0) return NVSR_ERR_SHAPE;
    if (P == 0) return NVSR_OK;
    // limb arithmetics: the training forward's kernel on the point list (z = NULL, one sample per "ray": tiles of 32 consecutive points)
    if (arith != NVSR_ARITH_F32)
        return nvsr_decode_rays_limb_launch(arith == NVSR_ARITH_F16X2 ? 2 : 3, scene, packed_decoder, P, 1, x, nullptr, out, nullptr, nullptr, stream);
    const int64_t ntiles = (P + PTS_PER_WG - 1) / PTS_PER_WG;
    const int grid = (int)(ntiles < 2048 ? ntiles : 2048);
    hipLaunchKernelGGL(triplane_decode_kernel, dim3(grid), dim3(TPB), 0, (hipStream_t)stream, to_dev(scene), packed_decoder,
                       (long)P, x, out);
    return NVSR_CHECK_LAUNCH();
}

int nvsr_decode_rays(const nvsr_scene* scene, const float* packed_decoder, int64_t N, int S, const float* rays, const float* z,
                     float* raw, nvsr_stream_t stream) {
    return nvsr_decode_rays_arith(scene, packed_decoder, N, S, rays, z, raw, nullptr, nullptr, NVSR_ARITH_INHERIT, stream);
}

int nvsr_decode_rays_ex(const nvsr_scene* scene, const float* packed_decoder, int64_t N, int S, const float* rays, const float* z,
                        float* raw, uint32_t* gates, float* record, nvsr_stream_t stream) {
    return nvsr_decode_rays_arith(scene, packed_decoder, N, S, rays, z, raw, gates, record, NVSR_ARITH_INHERIT, stream);
}

int nvsr_decode_rays_arith(const nvsr_scene* scene, const float* packed_decoder, int64_t N, int S, const float* rays, const float* z,
                           float* raw, uint32_t* gates, float* record, int arithmetic, nvsr_stream_t stream) {
    const int arith = nvsr_internal_resolve_decoder_arith(arithmetic);
    if (arith < 0) return NVSR_ERR_SHAPE;
    if (int e = check_scene(scene)) return e;
    if (!packed_decoder || !rays || !z || !raw) return NVSR_ERR_NULL;
    if (!aligned16(packed_decoder) || !aligned16(raw) || !aligned16(gates) || !aligned16(record)) return NVSR_ERR_ALIGN;
    if (N < 0 || S < 1 || S > 4096) return NVSR_ERR_SHAPE;
    if (N == 0) return NVSR_OK;
    if (record && !gates) return NVSR_ERR_NULL;      // the record is consumed together with the gates
    // limb matrix pipe (decode_limb.hip): 3 bf16 limbs, or -- NVSR_ARITH_F16X2 without a weight-gradient record -- 2 f16 limbs (the gates are
    // signs of pre-activations: the 3-limb backward consumes them whichever forward arithmetic found them)
    // NVSR_ARITH_F16X2 without a record: the tile-pair kernel (decode_pair.hip; same gates, same raw to the rounding of the heads' summation
    // order) where it is the faster one -- from four 32-sample chunks per ray on (same-box A/B, 4 096 rays: S = 128 0.395-0.408 against
    // 0.408-0.413 ms, S = 64 0.224-0.238 against 0.219-0.227; profiles/r04_pair_forward_ablation.txt).  NVSR_DECODE_PAIR=0 / =1 force one kernel.
    if (arith == NVSR_ARITH_F16X2 && !record && S > 32) {
        static const char* env = getenv("NVSR_DECODE_PAIR");
        const bool pair = env ? !strcmp(env, "1") : S >= 128;
        if (pair) return nvsr_decode_rays_pair_launch(scene, packed_decoder, N, S, rays, z, raw, gates, stream);
    }
    if (arith != NVSR_ARITH_F32)
        return nvsr_decode_rays_limb_launch(arith == NVSR_ARITH_F16X2 ? 2 : 3, scene, packed_decoder, N, S, rays, z, raw, gates, record, stream);
    const int64_t ntiles = ((N + PTS_PER_WG - 1) / PTS_PER_WG) * S;
    const int grid = (int)(ntiles < 2048 ? ntiles : 2048);
    if (record)
        hipLaunchKernelGGL((decode_rays_kernel<true, true>), dim3(grid), dim3(TPB), 0, (hipStream_t)stream, to_dev(scene), packed_decoder, (long)N,
                           S, rays, z, raw, gates, make_record(record, (long)N, S));
    else if (gates)
        hipLaunchKernelGGL((decode_rays_kernel<true, false>), dim3(grid), dim3(TPB), 0, (hipStream_t)stream, to_dev(scene), packed_decoder, (long)N,
                           S, rays, z, raw, gates, DecRecord{});
    else
        hipLaunchKernelGGL((decode_rays_kernel<false, false>), dim3(grid), dim3(TPB), 0, (hipStream_t)stream, to_dev(scene), packed_decoder,
                           (long)N, S, rays, z, raw, (unsigned*)nullptr, DecRecord{});
    return NVSR_CHECK_LAUNCH();
}

int nvsr_render_pass_ex(const nvsr_scene* scene, const float* packed_decoder, int64_t N, int S, const float* rays, const float* z,
                        const float* noise, int white_bkgd, float* rgb, float* disp, float* acc, float* weights, float* depth,
                        float* raw_out, nvsr_stream_t stream) {
    return nvsr_render_pass_arith(scene, packed_decoder, N, S, rays, z, noise, white_bkgd, rgb, disp, acc, weights, depth, raw_out,
                                  NVSR_ARITH_INHERIT, stream);
}

int nvsr_render_pass_arith(const nvsr_scene* scene, const float* packed_decoder, int64_t N, int S, const float* rays, const float* z,
                           const float* noise, int white_bkgd, float* rgb, float* disp, float* acc, float* weights, float* depth,
                           float* raw_out, int arithmetic, nvsr_stream_t stream) {
    const int arith = nvsr_internal_resolve_decoder_arith(arithmetic);
    if (arith < 0) return NVSR_ERR_SHAPE;
    if (int e = check_scene(scene)) return e;
    if (!packed_decoder || !rays || !z || !rgb || !disp || !acc) return NVSR_ERR_NULL;
    if (!aligned16(packed_decoder) || (raw_out && !aligned16(raw_out))) return NVSR_ERR_ALIGN;
    if (N < 0 || S < 1 || S > 4096) return NVSR_ERR_SHAPE;
    if (N == 0) return NVSR_OK;
    if (N >= 16384 && !getenv("NVSR_RENDER_V1")) {   // two-tiles-per-wave kernels; NVSR_RENDER_V1=1 selects the first-generation kernel
        if (arith != NVSR_ARITH_F32)                 // bf16-limb matrix pipe (render3.hip)
            return nvsr_render_pass3_launch(arith, scene, packed_decoder, N, S, rays, z, noise, white_bkgd, rgb, disp, acc, weights, depth, raw_out, stream);
        return nvsr_render_pass2_launch(scene, packed_decoder, N, S, rays, z, noise, white_bkgd, rgb, disp, acc, weights, depth, raw_out, stream);
    }
    const int64_t grid = (N + PTS_PER_WG - 1) / PTS_PER_WG;
    if (grid > 0x7fffffff) return NVSR_ERR_SHAPE;
    hipLaunchKernelGGL(render_pass_kernel, dim3((unsigned)grid), dim3(TPB), 0, (hipStream_t)stream, to_dev(scene), packed_decoder,
                       (long)N, S, rays, z, noise, white_bkgd, rgb, disp, acc, weights, depth, raw_out);
    return NVSR_CHECK_LAUNCH();
}

int nvsr_render_pass(const nvsr_scene* scene, const float* packed_decoder, int64_t N, int S, const float* rays, const float* z,
                     const float* noise, int white_bkgd, float* rgb, float* disp, float* acc, float* weights, float* depth,
                     nvsr_stream_t stream) {
    return nvsr_render_pass_arith(scene, packed_decoder, N, S, rays, z, noise, white_bkgd, rgb, disp, acc, weights, depth, nullptr, NVSR_ARITH_INHERIT,
                                  stream);
}

}  // extern "C"
