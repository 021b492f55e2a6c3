// Wave-level scans and sums on the DPP data path (64 lanes) -- shared by aux.hip (compositor, resampler) and render_bwd.hip (the compositor's
// backward recomputes the forward's transmittances with the SAME scan: bit-identical T).  Not part of the public ABI.
#pragma once
#include "nvsr_common.h"

namespace nvsr {

template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_or(float identity, float v);
__device__ __forceinline__ float wave_scan_add(float v, int lane);
// sum over the wave, the same value in every lane: the DPP scan below, its last lane broadcast through a scalar register (rounds 1-4: a
// six-step __shfl_xor butterfly = six ds_bpermute round trips).
// PRECONDITION (unlike the butterfly, which read 0 from inactive lanes): EVERY lane of the wave is active -- lane 63 holds the total only if all
// 64 lanes took part in the scan, and v_readlane of an inactive lane returns whatever its register holds.  All callers are wave-uniform code;
// -DNVSR_DEBUG_WAVE_OPS=1 traps on a partial EXEC mask.
__device__ __forceinline__ float wave_sum(float v) {
#if defined(NVSR_DEBUG_WAVE_OPS) && NVSR_DEBUG_WAVE_OPS
    if (__builtin_amdgcn_read_exec() != ~0ull) __builtin_trap();
#endif
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_scan_add(v, 0)), 63));
}
// Inclusive scans across the wave on the DPP data path (round 5; rounds 1-4 ran six Hillis-Steele steps over __shfl_up, i.e. ds_bpermute: an
// LDS-hardware round trip + a select per step).  Four row_shr steps scan every row of 16 lanes (a lane whose source falls outside its row
// keeps the identity: bound_ctrl off, `old` = identity), row_bcast:15 adds a row's total to the odd rows behind it, row_bcast:31 the total of
// the first half to the second -- six v_add / v_mul with a DPP operand, no LDS.  Same set of operands per lane; the ORDER of the additions
// differs from the wave-wide Hillis-Steele order for lanes >= 16 (last-ulp differences in a cdf or a transmittance; integer-valued scans --
// the rank histogram of the resampler -- are exact either way).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_or(float identity, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118, DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;
__device__ __forceinline__ float wave_scan_add(float v, int) {
    v += dpp_or<DPP_ROW_SHR1, 0xf>(0.0f, v);
    v += dpp_or<DPP_ROW_SHR2, 0xf>(0.0f, v);
    v += dpp_or<DPP_ROW_SHR4, 0xf>(0.0f, v);
    v += dpp_or<DPP_ROW_SHR8, 0xf>(0.0f, v);
    v += dpp_or<DPP_ROW_BCAST15, 0xa>(0.0f, v);
    v += dpp_or<DPP_ROW_BCAST31, 0xc>(0.0f, v);
    return v;
}
__device__ __forceinline__ float wave_scan_mul(float v, int) {
    v *= dpp_or<DPP_ROW_SHR1, 0xf>(1.0f, v);
    v *= dpp_or<DPP_ROW_SHR2, 0xf>(1.0f, v);
    v *= dpp_or<DPP_ROW_SHR4, 0xf>(1.0f, v);
    v *= dpp_or<DPP_ROW_SHR8, 0xf>(1.0f, v);
    v *= dpp_or<DPP_ROW_BCAST15, 0xa>(1.0f, v);
    v *= dpp_or<DPP_ROW_BCAST31, 0xc>(1.0f, v);
    return v;
}


}  // namespace nvsr
