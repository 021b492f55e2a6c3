// Definitions shared by the backward kernels (render_bwd.hip, render_bwd2.hip) -- not part of the public ABI.
#pragma once
#include "decode_core.h"

namespace nvsr {

constexpr int TILE_FLOATS = 32 * C;       // per-wave transposition tile [32 points][48 channels]

// ---- transposed-weight blob ("packed_bwd"), in consumption order ------------------------------------------------------------
//   hidden^T layer (16384 floats): [kb][q][ib][lane][j] = W[32kb + 8q + 4h + j][32ib + (lane&31)]        (W = [out][in])
//   layer-0^T of one plane (8192 floats): [kb][q][ib 2][lane][j] = W0[32kb + 8q + 4h + j][48p + 32ib + (lane&31)], 0 for channel >= 48
constexpr int B_DEN_H = 0;                               // density L3^T, L2^T, L1^T
constexpr int B_DEN0 = B_DEN_H + 3 * P_HID_FLOATS;       // 49152
constexpr int B_RGB_H = B_DEN0 + 8192;                   // 57344: rgb L3^T, L2^T, L1^T
constexpr int B_RGB0 = B_RGB_H + 3 * P_HID_FLOATS;       // 106496: planes 0..3
constexpr int B_TOTAL = B_RGB0 + 4 * 8192;               // 139264
static_assert(B_TOTAL == NVSR_DECODER_PACKED_BWD_FLOATS, "backward blob size");

struct Masks { unsigned m[2]; };   // bit (ib&1)*16 + r of m[ib>>1]  <=>  post-ReLU activation acc[ib][r] > 0
struct GradPlanes { float* p[4]; };

}  // namespace nvsr
