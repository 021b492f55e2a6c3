"""ctypes binding of the C ABI declared in include/nvsr.h (libnvsr_hip.so, built by build.py).

This is the only place that touches the native library.  The signatures carry raw device pointers, sizes and a
hipStream_t -- no torch types; torch tensors are turned into pointers here (`ptr`).  Loading fails loudly when the
library is missing: there is no CPU fallback for the product path.
"""
import ctypes as C
import os

import torch

from .build import LIB_PATH as _DEFAULT_LIB_PATH

# developer hook for kernel experiments (tools/): load an alternative build of the same ABI
LIB_PATH = os.environ.get("NVSR_HIP_LIB", _DEFAULT_LIB_PATH)

PLANE_CHANNELS = 48
DEC_CHANNELS = 128
DECODER_NATURAL_FLOATS = 130564
DECODER_PACKED_FLOATS = 453136
DECODER_PACKED_BWD_FLOATS = 487424
MSE_PAIR_MAX_ELEMS = 1 << 22          # NVSR_MSE_PAIR_MAX_ELEMS

_STATUS = {1: "NVSR_ERR_SHAPE (argument out of the supported range)", 2: "NVSR_ERR_LAUNCH (kernel launch failed)",
           3: "NVSR_ERR_NULL (required pointer is NULL)", 4: "NVSR_ERR_ALIGN (pointer not 16-byte aligned)"}


class NvsrError(RuntimeError):
    pass


class Scene(C.Structure):
    """struct nvsr_scene"""
    _fields_ = [("planes", C.c_void_p * 4), ("ph", C.c_int32 * 4), ("pw", C.c_int32 * 4), ("lo", C.c_float * 5),
                ("range", C.c_float * 5), ("proj", (C.c_float * 6) * 3)]


MAX_POSITION_PLANES = 15


class SceneExt(C.Structure):
    """struct nvsr_scene_ext: any number of position planes, grid_sample's align_corners (generic kernels only)"""
    _fields_ = [("num_position_planes", C.c_int32), ("align_corners", C.c_int32), ("plane_interp", C.c_int32),
                ("planes", C.c_void_p * (MAX_POSITION_PLANES + 1)),
                ("ph", C.c_int32 * (MAX_POSITION_PLANES + 1)), ("pw", C.c_int32 * (MAX_POSITION_PLANES + 1)), ("lo", C.c_float * 5),
                ("range", C.c_float * 5), ("proj", (C.c_float * 6) * MAX_POSITION_PLANES)]


class DecoderGeometry(C.Structure):
    """struct nvsr_decoder_geometry"""
    _fields_ = [(n, C.c_int32) for n in ("plane_channels", "viewdir_channels", "hidden", "density_layers", "rgb_layers", "skip_connect_every",
                                         "proj_combination", "viewdir_combination")]


_vp, _i, _i64, _d = C.c_void_p, C.c_int, C.c_int64, C.c_double
_PROTOS = {
    "nvsr_version": ([], C.c_int),
    "nvsr_fused_min_rays": ([], C.c_int64),
    "nvsr_limb_gemm_probe": ([_i, _i, _vp, _vp, _vp, _vp], _i),
    "nvsr_get_decoder_arithmetic": ([], C.c_int),
    "nvsr_set_decoder_arithmetic": ([_i], _i),
    "nvsr_set_range_flag": ([_vp], _i),
    "nvsr_get_range_flag": ([], _vp),
    "nvsr_get_conv_arithmetic": ([], C.c_int),
    "nvsr_set_conv_arithmetic": ([_i], _i),
    "nvsr_plane_to_channel_last": ([_vp, _vp, _i, _i, _i, _vp], _i),
    "nvsr_plane_from_channel_last": ([_vp, _vp, _i, _i, _i, _vp], _i),
    "nvsr_pack_decoder": ([_vp, _vp, _vp], _i),
    "nvsr_get_ray_bundle": ([_i, _i, _d, _d, _vp, _i, _d, _vp, _vp, _vp], _i),
    "nvsr_get_ray_bundle_at": ([_i, _i, _d, _d, _vp, _d, _i64, _vp, _vp, _vp, _vp], _i),
    "nvsr_sample_pixels": ([_i64, _i, _i, C.c_uint64, _i64, _i64, _vp, _i, _vp, _vp, _vp], _i),
    "nvsr_sample_key": ([C.c_uint64, C.c_uint64], C.c_uint64),
    "nvsr_sample_pixels_seq": ([_i64, _i, _i, _vp, _i64, _i64, _vp, _i, _vp, _vp, _vp], _i),
    "nvsr_mse_pair": ([_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp], _i),
    "nvsr_mse_pair_sum": ([_i64, _vp, _vp, _vp, _vp, _vp], _i),
    "nvsr_mse_pair_backward": ([_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp], _i),
    "nvsr_ndc_rays": ([_i, _i, _d, _d, _i64, _vp, _vp, _vp, _vp, _vp], _i),
    "nvsr_pack_rays": ([_i64, _vp, _vp, _vp, _d, _d, _vp, _vp], _i),
    "nvsr_coarse_z": ([_i64, _i, _vp, _i, _vp, _vp, _vp], _i),
    "nvsr_sample_pdf": ([_i64, _i, _i, _vp, _vp, _vp, _vp, _vp], _i),
    "nvsr_sort_rows": ([_i64, _i, _vp, _vp, _vp], _i),
    "nvsr_cumprod_exclusive": ([_i64, _i, _vp, _vp, _vp], _i),
    "nvsr_cumprod_exclusive_backward": ([_i64, _i, _vp, _vp, _vp, _vp, _vp], _i),
    "nvsr_importance_resample": ([_i64, _i, _i, _vp, _vp, _vp, _vp, _vp], _i),
    "nvsr_importance_resample_rays": ([_i64, _i, _i, _vp, _i, _vp, _vp, _vp, _vp], _i),
    "nvsr_triplane_decode": ([C.POINTER(Scene), _vp, _i64, _vp, _vp, _vp], _i),
    "nvsr_triplane_decode_arith": ([C.POINTER(Scene), _vp, _i64, _vp, _vp, _i, _vp], _i),
    "nvsr_composite": ([_i64, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp], _i),
    "nvsr_composite_rays": ([_i64, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp], _i),
    "nvsr_composite_mip": ([_i64, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp], _i),
    "nvsr_render_pass": ([C.POINTER(Scene), _vp, _i64, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp], _i),
    "nvsr_decode_rays": ([C.POINTER(Scene), _vp, _i64, _i, _vp, _vp, _vp, _vp], _i),
    "nvsr_render_workspace_floats": ([_i64, _i, _i], _i64),
    "nvsr_render_rays": ([C.POINTER(Scene), _vp, _vp, _i64, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                          _vp, _vp, _vp], _i),
}
PACK_ALL_ARITHMETICS = -3      # include/nvsr.h NVSR_PACK_ALL_ARITHMETICS
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)
_PROTOS_OPTIONAL = {   # feature-plane super-resolution (csrc/sr.hip)
    "nvsr_conv3x3_packed_floats": ([_i, _i], _i64),
    "nvsr_pack_conv3x3": ([_vp, _i, _i, _vp, _vp], _i),
    "nvsr_conv3x3": ([_vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp], _i),
    "nvsr_edsr_natural_floats": ([_i, _i, _i, _i, _i], _i64),
    "nvsr_edsr_packed_floats": ([_i, _i, _i, _i, _i], _i64),
    "nvsr_pack_edsr": ([_vp, _i, _i, _i, _i, _i, _vp, _vp], _i),
    "nvsr_pack_edsr_arith": ([_vp, _i, _i, _i, _i, _i, _vp, _i, _vp], _i),
    "nvsr_edsr_out_size": ([_i, _i, _i, _i, _ip, _ip], _i),
    "nvsr_edsr_workspace_floats": ([_i, _i, _i, _i, _i], _i64),
    "nvsr_edsr_forward": ([_vp, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp], _i),
    "nvsr_planes_sr_workspace_floats": ([_i, _i, _i, _i, _i, _i, _i, _fp], _i64),
    "nvsr_planes_sr": ([_vp, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _fp, _vp, _vp, _vp, _vp, _vp], _i),
    # training: plane gradients (csrc/render_bwd.hip)
    "nvsr_render_pass_ex": ([C.POINTER(Scene), _vp, _i64, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp], _i),
    "nvsr_pack_decoder_bwd": ([_vp, _vp, _vp], _i),
    "nvsr_composite_backward": ([_i64, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp], _i),
    "nvsr_composite_backward_mip": ([_i64, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp], _i),
    "nvsr_composite_backward_depth": ([_i64, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp], _i),
    "nvsr_composite_backward_rays": ([_i64, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp], _i),
    "nvsr_render_pass_backward": ([C.POINTER(Scene), _vp, _vp, _i64, _i, _vp, _vp, _vp, C.POINTER(C.c_void_p), _vp], _i),
    "nvsr_decode_rays_ex": ([C.POINTER(Scene), _vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp], _i),
    "nvsr_render_pass_backward_gates": ([C.POINTER(Scene), _vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, C.POINTER(C.c_void_p), _vp, _vp, _vp], _i),
    "nvsr_view_grad_workspace_floats": ([_i64, _i], _i64),
    "nvsr_decoder_record_floats": ([_i64, _i], _i64),
    "nvsr_render_pass_backward_ex": ([C.POINTER(Scene), _vp, _vp, _i64, _i, _vp, _vp, _vp, C.POINTER(C.c_void_p), _vp, _vp, _vp], _i),
    "nvsr_decoder_weight_grad": ([_i64, _i, _vp, _vp, _vp], _i),
    "nvsr_edsr_forward_batch": ([_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp], _i),
    "nvsr_planes_sr_batch": ([C.POINTER(C.c_void_p), _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _fp, _vp, _vp, C.POINTER(C.c_void_p), _vp, _vp], _i),
    # super-resolution CNN backward (csrc/sr_bwd.hip)
    "nvsr_pack_conv3x3_dgrad": ([_vp, _i, _i, _vp, _vp], _i),
    "nvsr_conv3x3_dgrad": ([_vp, _i, _i, _i, _vp, _i, _vp, _vp], _i),
    "nvsr_conv3x3_wgrad_workspace_floats": ([_i, _i, _i, _i], _i64),
    "nvsr_conv3x3_wgrad": ([_vp, _vp, _i, _i, _i, _i, C.c_float, _vp, _vp, _vp], _i),
    "nvsr_edsr_acts_floats": ([_i, _i, _i, _i, _i, _i, _i], _i64),
    "nvsr_edsr_forward_train": ([_vp, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp], _i),
    "nvsr_edsr_packed_dgrad_floats": ([_i, _i, _i, _i, _i], _i64),
    "nvsr_pack_edsr_dgrad": ([_vp, _i, _i, _i, _i, _i, _vp, _vp], _i),
    "nvsr_pack_edsr_dgrad_arith": ([_vp, _i, _i, _i, _i, _i, _vp, _i, _vp], _i),
    "nvsr_edsr_backward_workspace_floats": ([_i, _i, _i, _i, _i, _i, _i], _i64),
    "nvsr_edsr_backward": ([_vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp], _i),
    "nvsr_planes_sr_keep_floats": ([_i, _i, _i, _i, _i, _i, _i, _fp], _i64),
    "nvsr_planes_sr_train": ([_vp, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _fp, _vp, _vp, _vp, _vp, _vp, _vp], _i),
    "nvsr_planes_sr_backward_workspace_floats": ([_i, _i, _i, _i, _i, _i, _i, _fp], _i64),
    "nvsr_planes_sr_backward": ([_i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _fp, _vp, _vp, _vp, _vp, _vp, _vp], _i),
    # decoder geometries other than the shipped one (csrc/generic.hip)
    "nvsr_generic_decoder_natural_floats": ([C.POINTER(DecoderGeometry)], _i64),
    "nvsr_generic_decode_workspace_floats": ([C.POINTER(DecoderGeometry), _i64], _i64),
    "nvsr_generic_decode": ([C.POINTER(Scene), C.POINTER(DecoderGeometry), _vp, _i64, _vp, _vp, _vp, _vp], _i),
    "nvsr_generic_decode_backward_workspace_floats": ([C.POINTER(DecoderGeometry), _i64], _i64),
    "nvsr_generic_decode_backward": ([C.POINTER(Scene), C.POINTER(DecoderGeometry), _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp], _i),
    "nvsr_generic_decoder_natural_floats_ext": ([C.POINTER(DecoderGeometry), _i], _i64),
    "nvsr_generic_decode_workspace_floats_ext": ([C.POINTER(DecoderGeometry), _i, _i64], _i64),
    "nvsr_generic_decode_backward_workspace_floats_ext": ([C.POINTER(DecoderGeometry), _i, _i64], _i64),
    "nvsr_generic_decode_ext": ([C.POINTER(SceneExt), C.POINTER(DecoderGeometry), _vp, _i64, _vp, _vp, _vp, _vp, _vp], _i),
    "nvsr_generic_decode_backward_ext": ([C.POINTER(SceneExt), C.POINTER(DecoderGeometry), _vp, _i64, _vp, _vp, _vp, _vp, C.POINTER(C.c_void_p),
                                          _vp, _vp], _i),
    "nvsr_ray_points": ([_i64, _i, _vp, _vp, _vp, _vp], _i),
    "nvsr_render_shared_workspace_floats": ([_i64, _i, _i], _i64),
    "nvsr_render_rays_shared_arith": ([C.POINTER(Scene), _vp, _i64, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp], _i),
    "nvsr_set_sr_plane_interp": ([_i], _i),
    "nvsr_get_sr_plane_interp": ([], _i),
    "nvsr_set_sr_align_corners": ([_i], _i),
    "nvsr_get_sr_align_corners": ([], _i),
    # per-call arithmetic twins (include/nvsr.h, "per-call arithmetic")
    "nvsr_render_pass_arith": ([C.POINTER(Scene), _vp, _i64, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp], _i),
    "nvsr_decode_rays_arith": ([C.POINTER(Scene), _vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp], _i),
    "nvsr_render_rays_arith": ([C.POINTER(Scene), _vp, _vp, _i64, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                _vp, _vp, _i, _vp], _i),
    "nvsr_render_pass_backward_gates_arith": ([C.POINTER(Scene), _vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, C.POINTER(C.c_void_p), _vp, _vp, _i, _vp], _i),
    "nvsr_decoder_weight_grad_arith": ([_i64, _i, _vp, _vp, _i, _vp], _i),
    "nvsr_conv3x3_arith": ([_vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _i, _i, _vp], _i),
    "nvsr_conv3x3_dgrad_arith": ([_vp, _i, _i, _i, _vp, _i, _vp, _i, _i, _vp], _i),
    "nvsr_conv3x3_wgrad_arith": ([_vp, _vp, _i, _i, _i, _i, C.c_float, _vp, _vp, _i, _vp], _i),
    "nvsr_edsr_forward_batch_arith": ([_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp], _i),
    "nvsr_edsr_forward_train_arith": ([_vp, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp], _i),
    "nvsr_edsr_backward_arith": ([_vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp], _i),
    "nvsr_planes_sr_arith": ([_vp, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _fp, _vp, _vp, _vp, _vp, _i, _vp], _i),
    "nvsr_planes_sr_batch_arith": ([C.POINTER(C.c_void_p), _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _fp, _vp, _vp, C.POINTER(C.c_void_p), _vp, _i, _vp], _i),
    "nvsr_planes_sr_train_arith": ([_vp, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _fp, _vp, _vp, _vp, _vp, _vp, _i, _vp], _i),
    "nvsr_planes_sr_backward_arith": ([_i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _fp, _vp, _vp, _vp, _vp, _vp, _i, _vp], _i),
    # SR training on B regions of interest at once (csrc/sr.hip, csrc/sr_bwd.hip)
    "nvsr_planes_sr_batch_ex": ([C.POINTER(C.c_void_p), _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _fp, _vp, _vp, C.POINTER(C.c_void_p), _vp, _i, _i, _i, _vp], _i),
    "nvsr_planes_sr_batch_keep_floats": ([_i, _i, _i, _i, _i, _i, _i, _i, _fp], _i64),
    "nvsr_planes_sr_batch_workspace_floats": ([_i, _i, _i, _i, _i, _i, _i, _i, _fp], _i64),
    "nvsr_planes_sr_batch_backward_workspace_floats": ([_i, _i, _i, _i, _i, _i, _i, _i, _fp], _i64),
    "nvsr_planes_sr_train_batch_arith": ([C.POINTER(C.c_void_p), _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _fp, _vp, _vp, C.POINTER(C.c_void_p), _vp, _vp,
                                          _i, _i, _i, _vp], _i),
    "nvsr_planes_sr_backward_batch_arith": ([_i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _fp, _vp, C.POINTER(C.c_void_p), _vp, C.POINTER(C.c_void_p), _vp,
                                             _i, _i, _i, _vp], _i),
    "nvsr_planes_sr_backward_batch_marks": ([_i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _fp, _vp, C.POINTER(C.c_void_p), _vp, C.POINTER(C.c_void_p), _vp,
                                             _i, _i, _i, _i, C.POINTER(C.c_int32), C.POINTER(C.c_void_p), _vp], _i),
    # positional-encoding baseline (csrc/posenc.hip)
    "nvsr_positional_encoding": ([_i64, _i, _vp, _i, _i, _vp, _vp], _i),
    "nvsr_flexible_nerf_forward": ([_i64, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp], _i),
}

_lib = None


ARITHMETIC = {"f32": 0, "f16x2": 2, "bf16x3": 3}
SR_BATCH_MAX = 4          # planes per nvsr_planes_sr_*_batch_arith call (csrc/sr_core.h CONV_RAGGED_MAX)
ARITH_INHERIT = -1


def arith_code(mode):
    """'f32' | 'bf16x3' | 'f16x2' | None (= the process default) | an NVSR_ARITH_* code -> the int the *_arith entry points take"""
    if mode is None:
        return ARITH_INHERIT
    if isinstance(mode, str):
        return ARITHMETIC[mode]
    return int(mode)


def resolve_decoder_arithmetic(mode=None):
    """the concrete NVSR_ARITH_* code a call made now with `mode` runs in (what a forward stores for its backward)"""
    code = arith_code(mode)
    return lib().nvsr_get_decoder_arithmetic() if code == ARITH_INHERIT else code


def resolve_conv_arithmetic(mode=None):
    code = arith_code(mode)
    return lib().nvsr_get_conv_arithmetic() if code == ARITH_INHERIT else code


def set_decoder_arithmetic(mode):
    """Arithmetic of the decoder GEMMs inside the fused render pass: 'f32' | 'bf16x3' | 'f16x2' (include/nvsr.h, NVSR_ARITH_*)."""
    call("nvsr_set_decoder_arithmetic", ARITHMETIC[mode])


def get_decoder_arithmetic():
    code = lib().nvsr_get_decoder_arithmetic()
    return {v: k for k, v in ARITHMETIC.items()}[code]


def set_conv_arithmetic(mode):
    """Arithmetic of the wide 3x3 conv layers of the SR network: 'f32' | 'bf16x3' | 'f16x2' (forward convolutions, data and weight gradients; a
    gradient tensor is scaled by the power of two of its largest magnitude)."""
    call("nvsr_set_conv_arithmetic", ARITHMETIC[mode])


def get_conv_arithmetic():
    return {v: k for k, v in ARITHMETIC.items()}[lib().nvsr_get_conv_arithmetic()]


class RangeFlag:
    """The device word F16X2 launches OR a 1 into when they write a non-finite result (include/nvsr.h: nvsr_set_range_flag), with a pinned
    host mirror.  One per process (the registration is process-global like the default arithmetic); created on first use.
        reset()        zero the word on the current stream
        read_async()   enqueue the copy of the word to pinned memory -> a ticket
        raised(ticket, wait=True)   the word's value at the ticket: 0, or bits 1 (decoder kernels) | 2 (SR network) (waits for the copy;
                       wait=False: None while the copy is still in flight)"""

    def __init__(self, device):
        self.word = torch.zeros(1, dtype=torch.int32, device=device)
        call("nvsr_set_range_flag", ptr(self.word))

    def reset(self):
        self.word.zero_()

    def read_async(self):
        host = torch.empty(1, dtype=torch.int32, pin_memory=True)
        host.copy_(self.word, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return host, ev

    @staticmethod
    def raised(ticket, wait=True):
        host, ev = ticket
        if not wait and not ev.query():
            return None
        ev.synchronize()
        return int(host[0])


_range_flag = None


def range_flag(device=None):
    """the process's RangeFlag (created and registered with the library on first use)"""
    global _range_flag
    if _range_flag is None:
        _range_flag = RangeFlag(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    return _range_flag


def exported_symbols():
    """Names include/nvsr.h declares (used by the CPU test that checks the library exports them all)."""
    return sorted(list(_PROTOS) + list(_PROTOS_OPTIONAL))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NvsrError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        for name, (args, res) in {**_PROTOS, **_PROTOS_OPTIONAL}.items():
            fn = getattr(_lib, name)
            fn.argtypes, fn.restype = args, res
        for var, get in (("NVSR_DECODER_ARITHMETIC", _lib.nvsr_get_decoder_arithmetic), ("NVSR_CONV_ARITHMETIC", _lib.nvsr_get_conv_arithmetic)):
            if get() not in ARITHMETIC.values():          # (NVSR_ARITH_INVALID: an unknown string must not quietly select the default)
                bad, _lib = _lib, None
                raise NvsrError("%s=%r is not one of %s" % (var, os.environ.get(var), " | ".join(ARITHMETIC)))
    return _lib


_fused_min_rays = None


def fused_min_rays():
    """NVSR_FUSED_MIN_RAYS as the loaded library was built with (nvsr_fused_min_rays): from this many rays on a render pass runs fused, per
    ray; below it sample-parallel.  Host code that picks a kernel or a ray order per launch reads it here, never from a Python copy."""
    global _fused_min_rays
    if _fused_min_rays is None:
        _fused_min_rays = int(lib().nvsr_fused_min_rays())
    return _fused_min_rays


def call(name, *args):
    """Invoke an int-status entry point; raise NvsrError on a non-zero status."""
    st = getattr(lib(), name)(*args)
    if st != 0:
        raise NvsrError("%s failed: %s" % (name, _STATUS.get(st, "status %d" % st)))


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise NvsrError("nvsr_amd runs on the GPU only (got a %s tensor); there is no CPU fallback" % t.device)


def f32c(t):
    """float32 + contiguous view/copy of a CUDA tensor."""
    require_cuda(t)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
