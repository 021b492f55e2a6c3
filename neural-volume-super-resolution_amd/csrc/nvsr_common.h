// Shared device-side definitions for the gfx950 kernels (not part of the public ABI).
#pragma once
// ---- race probe (tools/race_probe.sh; NEVER the product) ---------------------------------------------------------------------------------
// Round 5 found a data race that four rounds of parity tests had passed over: the backward kernels read head weights out of LDS in front of the
// first workgroup barrier; alone on a GPU the waves of a workgroup start together and the race never shows -- and even a late wave usually finds
// the RIGHT bytes, left behind by the previous workgroup of the same kernel.  -DNVSR_RACE_PROBE=1 builds a probe library in which, in EVERY kernel
// that fills LDS cooperatively, the first wave starts by filling the kernel's whole LDS array with NaN patterns (behind a start-of-kernel barrier of the probe's own) and all
// other waves then start ~50 us LATE (real-time counter): whatever the first wave reads from LDS before a barrier has published it is then NaN with certainty, and the parity
// tests fail instead of passing by luck.  NVSR_RACE_PROBE_DELAY(lds) stands right behind the kernel's __shared__ array.
#ifndef NVSR_RACE_PROBE
#define NVSR_RACE_PROBE 0
#endif
#ifndef NVSR_RP_LO            // (bisection of a finding: poison only the words [NVSR_RP_LO, NVSR_RP_HI) of the array)
#define NVSR_RP_LO 0u
#define NVSR_RP_HI 0xffffffffu
#endif
#ifndef NVSR_RP_PATTERN
#define NVSR_RP_PATTERN 0x7fc00000u
#endif
#if NVSR_RACE_PROBE
#define NVSR_RACE_PROBE_DELAY(ARR)                                                                                          \
    do {                                                                                                                    \
        if (NVSR_RACE_PROBE == 1) {                  /* (2: the late start alone, no poison) */                             \
            /* the first wave fills the array (stores through the array itself = ds_write), THEN a start-of-kernel barrier, THEN the    \
               others sleep: a sleeping wave wakes after a few microseconds at most, and a poison store that lands behind another    \
               wave's real store would be the probe's own race (seen: s_sleep is far shorter than its 64 x 127 clocks suggest) */    \
            if ((threadIdx.x >> 6) == 0)                                                                                    \
                for (unsigned nvsr_rp_ = NVSR_RP_LO + threadIdx.x; nvsr_rp_ < sizeof(ARR) / 4 && nvsr_rp_ < NVSR_RP_HI; nvsr_rp_ += 64) \
                    (ARR)[nvsr_rp_] = __builtin_bit_cast(__typeof__((ARR)[0] + 0), NVSR_RP_PATTERN);                        \
            __syncthreads();                                                                                                \
        }                                                                                                                   \
        if ((threadIdx.x >> 6) != 0) {               /* ~50 us by the 100 MHz real-time counter (s_sleep alone proved far shorter) */ \
            const unsigned long long nvsr_rp_t0_ = __builtin_amdgcn_s_memrealtime();                                        \
            while (__builtin_amdgcn_s_memrealtime() - nvsr_rp_t0_ < 5000ull) __builtin_amdgcn_s_sleep(127);                 \
        }                                                                                                                   \
    } while (0)
#else
#define NVSR_RACE_PROBE_DELAY(ARR) do { } while (0)
#endif

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nvsr.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define NVSR_CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? NVSR_OK : NVSR_ERR_LAUNCH)

namespace nvsr {

constexpr int C = NVSR_PLANE_CHANNELS;  // 48 channels per plane texel
constexpr int HID = NVSR_DEC_CHANNELS;  // 128 decoder width
constexpr int HALF_C = C / 2;           // channels per lane-half (lane>>5) in the MFMA B-fragment

// ---- natural (state-dict order) blob offsets, in floats ----------------------------------------------------------
constexpr int N_DEN_W0 = 0;
constexpr int N_DEN_B0 = HID * C;                               // 6144
constexpr int N_DEN_W1 = N_DEN_B0 + HID;                        // 6272
constexpr int N_HID_STRIDE = HID * HID + HID;                   // 16512
constexpr int N_ALPHA_W = N_DEN_W1 + 3 * N_HID_STRIDE;          // 55808
constexpr int N_ALPHA_B = N_ALPHA_W + HID;                      // 55936
constexpr int N_RGB_W0 = N_ALPHA_B + 1;                         // 55937
constexpr int N_RGB_B0 = N_RGB_W0 + HID * 4 * C;                // 80513
constexpr int N_RGB_W1 = N_RGB_B0 + HID;                        // 80641
constexpr int N_FCRGB_W = N_RGB_W1 + 3 * N_HID_STRIDE;          // 130177
constexpr int N_FCRGB_B = N_FCRGB_W + 3 * HID;                  // 130561
static_assert(N_FCRGB_B + 3 == NVSR_DECODER_NATURAL_FLOATS, "natural blob size");

// ---- packed blob: A-operand fragments of v_mfma_f32_32x32x2_f32 in consumption order ------------------------------
// A fragment unit = 256 floats = [lane 0..63][j 0..3]: lane l holds W[32*ib + (l&31)][k(l>>5, j)], read with one
// ds_read_b128 per lane (conflict-free, lane-linear) and consumed by 4 consecutive MFMAs.
//   feature layers  (K = 48 per plane):  [plane p][q 0..5][ib 0..3][lane][j]   k = 48p + 24h + 4q + j
//   hidden layers   (K = 128)         :  [kb 0..3][q 0..3][ib 0..3][lane][j]   k = 32kb + 8q + 4h + j
// (h = lane>>5; the k of a hidden layer is exactly the C/D register layout of the previous layer's accumulator, so
//  activations chain through registers without any lane movement.)
constexpr int P_PLANE_FLOATS = 6 * 4 * 256;                     // 6144  (24 KB) one plane's share of a feature layer
constexpr int P_HID_FLOATS = HID * HID;                         // 16384 (64 KB)
constexpr int P_RGB0 = 0;                                       // 4 planes
constexpr int P_DEN0 = P_RGB0 + 4 * P_PLANE_FLOATS;             // 24576
constexpr int P_DEN1 = P_DEN0 + P_PLANE_FLOATS;                 // 30720
constexpr int P_RGB1 = P_DEN1 + 3 * P_HID_FLOATS;               // 79872
constexpr int P_SMALL = P_RGB1 + 3 * P_HID_FLOATS;              // 129024
// small region: biases [layer: den0..3, rgb0..3][ib][q][h][j], fc_alpha weight, fc_rgb weights (same order), head biases
constexpr int S_BIAS = 0;                                       // 8 * 128
constexpr int S_ALPHA_W = 8 * HID;                              // 1024
constexpr int S_RGB_W = S_ALPHA_W + HID;                        // 1152
constexpr int S_HEAD_B = S_RGB_W + 3 * HID;                     // 1536: alpha_b, rgb_b[3]
constexpr int S_F16_POISON = S_HEAD_B + 4;                      // 0, or NaN when a weight does not fit the f16 limbs (render3.hip packer)
constexpr int SMALL_FLOATS = 1552;                              // padded to a multiple of 16 B
static_assert(P_SMALL + SMALL_FLOATS == NVSR_DECODER_PACKED_F32_FLOATS, "packed blob size");

struct SceneDev {
    const float* plane[4];
    int ph[4], pw[4];
    float lo[5], range[5];
    float proj[18];
    // (W-1), (H-1) and their halves as floats: kernel arguments live in SGPRs; computed in the kernel they become
    // loop-invariant VGPRs that hipcc hoists out of the sample loop and spills
    float mx[4], my[4], hx[4], hy[4];
};

inline SceneDev to_dev(const nvsr_scene* s) {
    SceneDev d;
    for (int i = 0; i < 4; ++i) { d.plane[i] = s->planes[i]; d.ph[i] = s->ph[i]; d.pw[i] = s->pw[i]; }
    for (int i = 0; i < 5; ++i) { d.lo[i] = s->lo[i]; d.range[i] = s->range[i]; }
    for (int i = 0; i < 18; ++i) d.proj[i] = (&s->proj[0][0])[i];
    for (int i = 0; i < 4; ++i) {
        d.mx[i] = (float)(s->pw[i] - 1); d.my[i] = (float)(s->ph[i] - 1);
        d.hx[i] = d.mx[i] / 2.0f; d.hy[i] = d.my[i] / 2.0f;
    }
    return d;
}

// ---- activation / delta record of one training pass (decoder-weight gradients) ---------------------------------------------
// One row per decoded point, row q = s * N + ray (sample-major: the 32 rays of a wave at one sample are 32 consecutive rows, so a
// half-wave stores whole 512-byte rows).  Arrays are allocated for Pp = N*S rounded up to 8 rows + RECORD_DUMP_ROWS; rows >= N*S are never read.
// The last RECORD_DUMP_ROWS rows of every array are a dump (round 6): the limb kernels write a tile's 32 rows through an LDS stage with whole
// cache lines per store instruction (record128_staged), and the padding points of a partial tile go there instead of being masked off -- every
// wave issues every store, which is what lets the ring waits count them (decode_limb.hip FIN_YOUNG).
//   Xd [Pp][64]   density-decoder input (mean of the 3 position features; columns 48..63 zero)     } written by the pass that
//   Hd [4][Pp][128]  post-ReLU output of density layer l;  Xr [Pp][192], Hr likewise for the rgb decoder } runs the FORWARD layers
//   Gd / Gr [4][Pp][128]  dL/d(pre-activation of layer l);  g4 [Pp][4]  dL/d raw (rgb, sigma)        } written by the backward
constexpr int RECORD_DUMP_ROWS = 32;
struct DecRecord {
    float *Xd, *Hd, *Gd, *Xr, *Hr, *Gr, *g4;
    long Pp;   // allocated rows = stride between the per-layer arrays
    long P;    // valid rows (N * S)
    long dump; // first of the RECORD_DUMP_ROWS dump rows (= Pp - RECORD_DUMP_ROWS)
};
constexpr long DEC_RECORD_FLOATS_PER_SLOT = 64 + 4 * HID + 4 * HID + 4 * C + 4 * HID + 4 * HID + 4;   // 2308
inline long record_rows(long N, int S) { return N * (long)S; }
inline long record_alloc_rows(long N, int S) { return (record_rows(N, S) + 7) / 8 * 8 + RECORD_DUMP_ROWS; }
inline DecRecord make_record(float* base, long N, int S) {
    DecRecord r;
    const long Pp = record_alloc_rows(N, S);
    r.Pp = Pp;
    r.P = record_rows(N, S);
    r.dump = Pp - RECORD_DUMP_ROWS;
    r.Xd = base;            base += 64 * Pp;
    r.Hd = base;            base += 4L * HID * Pp;
    r.Gd = base;            base += 4L * HID * Pp;
    r.Xr = base;            base += 4L * C * Pp;
    r.Hr = base;            base += 4L * HID * Pp;
    r.Gr = base;            base += 4L * HID * Pp;
    r.g4 = base;
    return r;
}

// row of point (ray, s) in the weight-gradient record.  Every producer (the forward and backward kernels of both arithmetic modes) goes
// through this one function; the consumers (decoder_wgrad.hip) sum over rows and do not care about their order.
#ifdef NVSR_RECORD_SAMPLE_MAJOR        // A/B switch (tools/ab_flags.sh): the order the 32-rays-per-wave f32 kernels were written for
__host__ __device__ inline long record_row(long ray, int s, long N, int S) { (void)S; return (long)s * N + ray; }
#else                                  // ray-major: the 32 samples of a limb-kernel tile are 32 consecutive rows
__host__ __device__ inline long record_row(long ray, int s, long N, int S) { (void)N; return ray * S + s; }
#endif

// ReLU gates of a layer (published by the training forwards, consumed by the gate-driven backwards): two words per lane and layer; the
// gate of accumulator element acc[ib][r] is bit gate_bit(ib, r) of word ib >> 1.  An element pair (r, r + 1), r even, sits at bits
// (p, p + 16): the two halves of a packed f16 pair -- the f16-limb forward takes a pair's two gates from the high limbs it has split
// anyway (v_pk_min_u16 + v_lshl_or_b32 per PAIR, decode_pair.hip).
__host__ __device__ constexpr int gate_bit(int ib, int r) { return (ib & 1) * 8 + (r >> 1) + 16 * (r & 1); }

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int64_t round4(int64_t n) { return (n + 3) / 4 * 4; }      // floats -> a multiple of 16 bytes

// torch.linspace(0, 1, n) element i in fp32 (symmetric two-sided form used by ATen's RangeFactories)
__host__ __device__ inline float linspace01(int i, int n) {
    if (n == 1) return 0.0f;
    const float step = 1.0f / (float)(n - 1);
    return (i < n / 2) ? (step * (float)i) : (1.0f - step * (float)(n - 1 - i));
}

// un-jittered coarse depth s of Nc between near and far (predict_and_render_radiance, train_utils.py:95-100)
__device__ __forceinline__ float coarse_depth(float nr, float fr, int s, int Nc, int lindisp) {
    const float t = linspace01(s, Nc);
    if (!lindisp) return __fadd_rn(__fmul_rn(nr, __fsub_rn(1.0f, t)), __fmul_rn(fr, t));
    return __fdiv_rn(1.0f, __fadd_rn(__fmul_rn(__fdiv_rn(1.0f, nr), __fsub_rn(1.0f, t)), __fmul_rn(__fdiv_rn(1.0f, fr), t)));
}

}  // namespace nvsr
