"""`torch.ops.nvsr.*`: the hot path as PyTorch custom operators (torch.library) on top of the C ABI of include/nvsr.h.

Every operator is a thin shell: it allocates the outputs, turns tensors into raw device pointers and calls ONE entry point of
libnvsr_hip.so on the current stream (ctypes, capi.py) -- no arithmetic happens in Python.  Each has a fake (meta) implementation, so
the operators trace under FakeTensor / torch.compile / torch.export, and the differentiable ones carry their backward as another
operator of this library (`register_autograd`).  The host mirror (nerf_helpers / volume_rendering_utils / train_utils / models) is
written against these operators; they are the product path, not a side door.

Conventions
  * a scene is passed as `planes` = 4 CHANNEL-LAST [H,W,48] tensors (position planes D0..D2, view-direction plane) and `consts` = 28
    python floats: lo[5], range[5], proj[3][6] (struct nvsr_scene, include/nvsr.h);
  * `arithmetic` is the NVSR_ARITH_* code of the decoder GEMMs / SR convolutions for THIS call (-1 = the process default): a per-call
    argument, nothing global (capi.arith_code turns 'f32' / 'bf16x3' / 'f16x2' / None into it);
  * operators cannot return None: an output that was not asked for comes back as an empty tensor.

Reference functions behind the operators: models.py:381-421 (decoder), train_utils.py:71-182 (render passes), volume_rendering_utils.py:6-51
(compositing), nerf_helpers.py:668-702 (importance sampling), models.py:769-822,884-926 (EDSR / PlanesSR)."""
import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor
from torch.library import custom_op

from . import capi

PACK_ALL_ARITHMETICS = capi.PACK_ALL_ARITHMETICS

PC = capi.PLANE_CHANNELS
# decoder-gradient record: up to this many points of a pass (19 GB of record) the forward itself records the layer inputs and the backward
# never recomputes; beyond, the chunked recomputing path runs RECORD_RAYS rays at a time (train_utils re-exports both)
RECORD_FORWARD_MAX_POINTS = 1 << 21
RECORD_RAYS = 8192


def _f(*shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


def _c(t):
    """float32 + contiguous (None passes through): the kernels take dense row-major buffers"""
    return None if t is None else capi.f32c(t)


def _none_if_empty(t):
    return None if (t is None or t.numel() == 0) else t


def _scene(planes, consts, channels=None):
    """struct nvsr_scene from 4 channel-last planes + the 28 host constants (channels: (C, Cv) of a non-default decoder geometry)"""
    assert len(planes) == 4 and len(consts) == 28
    sc = capi.Scene()
    for d, p in enumerate(planes):
        capi.require_cuda(p)
        cc = PC if channels is None else channels[0 if d < 3 else 1]
        assert p.dtype == torch.float32 and p.dim() == 3 and p.shape[2] == cc and p.is_contiguous(), "planes must be channel-last [H,W,%d] f32" % cc
        sc.planes[d] = p.data_ptr()
        sc.ph[d], sc.pw[d] = p.shape[0], p.shape[1]
    for i in range(5):
        sc.lo[i] = consts[i]
        sc.range[i] = consts[5 + i]
    for d in range(3):
        for j in range(6):
            sc.proj[d][j] = consts[10 + 6 * d + j]
    return sc


def scene_consts(sc):
    """the 28 floats of a capi.Scene (lo, range, proj) as the operators take them"""
    return [float(sc.lo[i]) for i in range(5)] + [float(sc.range[i]) for i in range(5)] + [float(sc.proj[d][j]) for d in range(3) for j in range(6)]


# =====================================================================================================================================
# layout / packing
# =====================================================================================================================================
@custom_op("nvsr::plane_to_channel_last", mutates_args=(), device_types="cuda")
def plane_to_channel_last(plane: Tensor) -> Tensor:
    """[1,C,H,W] or [C,H,W] (reference layout, models.py:436-439) -> channel-last [H,W,C] copy"""
    p = capi.f32c(plane)
    Cc, H, W = p.shape[-3:]
    out = _f(H, W, Cc, like=p)
    capi.call("nvsr_plane_to_channel_last", capi.ptr(p), capi.ptr(out), Cc, H, W, capi.stream())
    return out


@plane_to_channel_last.register_fake
def _(plane):
    Cc, H, W = plane.shape[-3:]
    return plane.new_empty((H, W, Cc), dtype=torch.float32)


@custom_op("nvsr::plane_from_channel_last", mutates_args=(), device_types="cuda")
def plane_from_channel_last(plane_hwc: Tensor) -> Tensor:
    """channel-last [H,W,C] -> [1,C,H,W]"""
    p = capi.f32c(plane_hwc)
    H, W, Cc = p.shape
    out = _f(1, Cc, H, W, like=p)
    capi.call("nvsr_plane_from_channel_last", capi.ptr(p), capi.ptr(out), Cc, H, W, capi.stream())
    return out


@plane_from_channel_last.register_fake
def _(plane_hwc):
    H, W, Cc = plane_hwc.shape
    return plane_hwc.new_empty((1, Cc, H, W), dtype=torch.float32)


plane_to_channel_last.register_autograd(lambda ctx, g: torch.ops.nvsr.plane_from_channel_last(g).reshape(ctx.shape),
                                        setup_context=lambda ctx, inputs, output: setattr(ctx, "shape", inputs[0].shape))
plane_from_channel_last.register_autograd(lambda ctx, g: torch.ops.nvsr.plane_to_channel_last(g))


@custom_op("nvsr::pack_decoder", mutates_args=(), device_types="cuda")
def pack_decoder(natural: Tensor) -> Tensor:
    """decoder parameters in state-dict order (models.py:169-195) -> MFMA-fragment blob of the forward kernels"""
    nat = capi.f32c(natural)
    assert nat.numel() == capi.DECODER_NATURAL_FLOATS
    packed = _f(capi.DECODER_PACKED_FLOATS, like=nat)
    capi.call("nvsr_pack_decoder", capi.ptr(nat), capi.ptr(packed), capi.stream())
    return packed


@pack_decoder.register_fake
def _(natural):
    return natural.new_empty((capi.DECODER_PACKED_FLOATS,), dtype=torch.float32)


@custom_op("nvsr::pack_decoder_bwd", mutates_args=(), device_types="cuda")
def pack_decoder_bwd(natural: Tensor) -> Tensor:
    """-> fragments of the transposed layers (backward kernels)"""
    nat = capi.f32c(natural)
    assert nat.numel() == capi.DECODER_NATURAL_FLOATS
    packed = _f(capi.DECODER_PACKED_BWD_FLOATS, like=nat)
    capi.call("nvsr_pack_decoder_bwd", capi.ptr(nat), capi.ptr(packed), capi.stream())
    return packed


@pack_decoder_bwd.register_fake
def _(natural):
    return natural.new_empty((capi.DECODER_PACKED_BWD_FLOATS,), dtype=torch.float32)


# =====================================================================================================================================
# sampling helpers (train_utils.py:95-109,144-155; nerf_helpers.py:668-702)
# =====================================================================================================================================
@custom_op("nvsr::coarse_z", mutates_args=(), device_types="cuda")
def coarse_z(rays: Tensor, Nc: int, lindisp: bool, t_rand: Optional[Tensor]) -> Tensor:
    rays, t_rand = _c(rays), _c(t_rand)
    N = rays.shape[0]
    z = _f(N, Nc, like=rays)
    if N:
        capi.call("nvsr_coarse_z", N, Nc, capi.ptr(rays), int(lindisp), capi.ptr(t_rand), capi.ptr(z), capi.stream())
    return z


@coarse_z.register_fake
def _(rays, Nc, lindisp, t_rand):
    return rays.new_empty((rays.shape[0], Nc))


@custom_op("nvsr::importance_resample", mutates_args=(), device_types="cuda")
def importance_resample(z_coarse: Tensor, weights: Tensor, Nf: int, u: Optional[Tensor]) -> Tensor:
    """z_mid -> sample_pdf_2(z_mid, w[1:-1], Nf, det = (u is None)) -> sort(cat(z, samples)): [N, Nc+Nf]; no gradient (train_utils.py:153)"""
    z_coarse, weights, u = _c(z_coarse), _c(weights), _c(u)
    N, Nc = z_coarse.shape
    z_f = _f(N, Nc + Nf, like=z_coarse)
    if N:
        capi.call("nvsr_importance_resample", N, Nc, Nf, capi.ptr(z_coarse), capi.ptr(weights), capi.ptr(u), capi.ptr(z_f), capi.stream())
    return z_f


@importance_resample.register_fake
def _(z_coarse, weights, Nf, u):
    return z_coarse.new_empty((z_coarse.shape[0], z_coarse.shape[1] + Nf))


# =====================================================================================================================================
# tri-plane decoder (models.py:381-421) and the render passes (train_utils.py:71-182)
# =====================================================================================================================================
@custom_op("nvsr::triplane_decode", mutates_args=(), device_types="cuda")
def triplane_decode(planes: Sequence[Tensor], consts: Sequence[float], packed: Tensor, x: Tensor, arith: int = -1) -> Tensor:
    """TwoDimPlanesModel.forward on points: x [P,6] = [xyz, viewdir] -> [P,4] = [rgb_raw, sigma_raw].  arith: NVSR_ARITH_* (-1 = the process
    default): the limb modes run the training forward's kernel on tiles of 32 points, 0 the exact-f32 MFMA kernel"""
    x = _c(x)
    sc = _scene(planes, consts)
    P = x.shape[0]
    out = _f(P, 4, like=x)
    if P:
        capi.call("nvsr_triplane_decode_arith", C.byref(sc), capi.ptr(packed), P, capi.ptr(x), capi.ptr(out), int(arith), capi.stream())
    return out


@triplane_decode.register_fake
def _(planes, consts, packed, x, arith=-1):
    return x.new_empty((x.shape[0], 4))


def _scene_ext(planes, consts, channels, align_corners, bicubic=False):
    """struct nvsr_scene_ext from N + 1 channel-last planes (N position planes, then the view-direction plane) + 10 + 6 N host constants
    (lo, range, the N 3x2 projections)"""
    n_pos = len(planes) - 1
    assert 1 <= n_pos <= capi.MAX_POSITION_PLANES, "1 .. %d position planes" % capi.MAX_POSITION_PLANES
    assert len(consts) == 10 + 6 * n_pos, "scene constants: lo[5], range[5] and a 3x2 projection per position plane"
    sc = capi.SceneExt()
    sc.num_position_planes, sc.align_corners, sc.plane_interp = n_pos, int(bool(align_corners)), int(bool(bicubic))
    for d, p in enumerate(planes):
        capi.require_cuda(p)
        cc = channels[0 if d < n_pos else 1]
        assert p.dtype == torch.float32 and p.dim() == 3 and p.shape[2] == cc and p.is_contiguous(), "planes must be channel-last [H,W,%d] f32" % cc
        sc.planes[d] = p.data_ptr()
        sc.ph[d], sc.pw[d] = p.shape[0], p.shape[1]
    for i in range(5):
        sc.lo[i] = consts[i]
        sc.range[i] = consts[5 + i]
    for d in range(n_pos):
        for j in range(6):
            sc.proj[d][j] = consts[10 + 6 * d + j]
    return sc


def _generic_setup(planes, consts, natural, geometry, align_corners, coord_noise, P, bicubic=False):
    geo = capi.DecoderGeometry(*[int(v) for v in geometry])
    n_pos = len(planes) - 1
    n = capi.lib().nvsr_generic_decoder_natural_floats_ext(C.byref(geo), n_pos)
    if n < 0:
        raise capi.NvsrError("this decoder geometry is inconsistent (the reference's own layer sizes do not admit it)")
    assert natural.numel() == n, "natural blob: %d floats, the geometry needs %d" % (natural.numel(), n)
    if coord_noise is not None:
        assert tuple(coord_noise.shape) == (P, 3) and coord_noise.dtype == torch.float32, "coord_noise: [P,3] f32"
        capi.require_cuda(coord_noise)
    return geo, n_pos, n, _scene_ext(planes, consts, (geo.plane_channels, geo.viewdir_channels), align_corners, bicubic)


@custom_op("nvsr::triplane_decode_generic", mutates_args=(), device_types="cuda")
def triplane_decode_generic(planes: Sequence[Tensor], consts: Sequence[float], natural: Tensor, geometry: Sequence[int], x: Tensor,
                            align_corners: bool = True, coord_noise: Optional[Tensor] = None, bicubic: bool = False) -> Tensor:
    """TwoDimPlanesModel.forward for ANY decoder geometry the reference's layer sizes admit (csrc/generic.hip).
    geometry = [plane_channels, viewdir_channels, hidden, density_layers, rgb_layers, skip_connect_every (0 = None), proj_combination
    (0 sum, 1 avg, 2 concat), viewdir_combination (0 sum, 1 avg, 2 mult, 3 concat, 4 concat_pos)]; natural = the parameters in
    state-dict order; planes channel-last: N position planes [H,W,plane_channels], then [H,W,viewdir_channels]; consts = lo[5], range[5],
    N 3x2 projections; align_corners / bicubic: grid_sample's align_corners and mode; coord_noise [P,3]: added to the normalised positions
    (point_coords_noise, models.py:291-293)."""
    x, natural = _c(x), _c(natural)
    P = x.shape[0]
    coord_noise = None if coord_noise is None else _c(coord_noise)
    geo, n_pos, _, sc = _generic_setup(planes, consts, natural, geometry, align_corners, coord_noise, P, bicubic)
    out = _f(P, 4, like=x)
    if P:
        ws = _f(capi.lib().nvsr_generic_decode_workspace_floats_ext(C.byref(geo), n_pos, P), like=x)
        capi.call("nvsr_generic_decode_ext", C.byref(sc), C.byref(geo), capi.ptr(natural), P, capi.ptr(x),
                  None if coord_noise is None else capi.ptr(coord_noise), capi.ptr(out), capi.ptr(ws), capi.stream())
    return out


@triplane_decode_generic.register_fake
def _(planes, consts, natural, geometry, x, align_corners=True, coord_noise=None, bicubic=False):
    return x.new_empty((x.shape[0], 4))


@custom_op("nvsr::triplane_decode_generic_backward", mutates_args=(), device_types="cuda")
def triplane_decode_generic_backward(planes: Sequence[Tensor], consts: Sequence[float], natural: Tensor, geometry: Sequence[int], x: Tensor,
                                     g_out: Tensor, want_natural: bool, want_planes: Sequence[bool], align_corners: bool = True,
                                     coord_noise: Optional[Tensor] = None, bicubic: bool = False) -> List[Tensor]:
    """Backward of triplane_decode_generic (csrc/generic.hip; the reference: torch.autograd through models.py:381-421): g_out [P,4] ->
    [d_natural, d_plane0 .. d_planeN] (channel-last like the planes; a gradient that is not wanted comes back as an empty tensor)."""
    x, natural, g_out = _c(x), _c(natural), _c(g_out)
    P = x.shape[0]
    coord_noise = None if coord_noise is None else _c(coord_noise)
    geo, n_pos, n, sc = _generic_setup(planes, consts, natural, geometry, align_corners, coord_noise, P, bicubic)
    assert tuple(g_out.shape) == (P, 4)
    want_planes = [bool(w) for w in want_planes]
    assert len(want_planes) == n_pos + 1
    g_nat = torch.zeros(n if want_natural else 0, dtype=torch.float32, device=x.device)
    g_pl = [torch.zeros_like(pl) if w else _f(0, like=x) for pl, w in zip(planes, want_planes)]
    if P and (want_natural or any(want_planes)):
        ws = _f(capi.lib().nvsr_generic_decode_backward_workspace_floats_ext(C.byref(geo), n_pos, P), like=x)
        d_planes = (C.c_void_p * (n_pos + 1))(*[g.data_ptr() if w else None for g, w in zip(g_pl, want_planes)])
        capi.call("nvsr_generic_decode_backward_ext", C.byref(sc), C.byref(geo), capi.ptr(natural), P, capi.ptr(x),
                  None if coord_noise is None else capi.ptr(coord_noise), capi.ptr(g_out), capi.ptr(g_nat) if want_natural else None, d_planes,
                  capi.ptr(ws), capi.stream())
    return [g_nat] + g_pl


@triplane_decode_generic_backward.register_fake
def _(planes, consts, natural, geometry, x, g_out, want_natural, want_planes, align_corners=True, coord_noise=None, bicubic=False):
    return [natural.new_empty(natural.numel() if want_natural else 0)] + [pl.new_empty(pl.shape if w else (0,)) for pl, w in zip(planes, want_planes)]


@custom_op("nvsr::ray_points", mutates_args=(), device_types="cuda")
def ray_points(rays: Tensor, z: Tensor) -> Tensor:
    """run_network's model input (train_utils.py:15-64,111): [N*S,6] = [ro + rd * z, viewdir] from packed rays [N,11] and depths [N,S]"""
    rays, z = _c(rays), _c(z)
    N, S = z.shape
    x = _f(N * S, 6, like=rays)
    if N:
        capi.call("nvsr_ray_points", N, S, capi.ptr(rays), capi.ptr(z), capi.ptr(x), capi.stream())
    return x


@ray_points.register_fake
def _(rays, z):
    return rays.new_empty((z.shape[0] * z.shape[1], 6))


@custom_op("nvsr::render_pass", mutates_args=(), device_types="cuda")
def render_pass(planes: Sequence[Tensor], consts: Sequence[float], packed: Tensor, rays: Tensor, z: Tensor, noise: Optional[Tensor],
                white: bool, want_weights: bool, arithmetic: int) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """run_network + model + volume_render_radiance_field of one pass, fused (no [N,S,*] tensor reaches HBM):
    rays [N,11], z [N,S] -> rgb [N,3], disp [N], acc [N], weights [N,S] (empty unless want_weights)"""
    rays, z, noise = _c(rays), _c(z), _c(noise)
    sc = _scene(planes, consts)
    N, S = z.shape
    rgb, disp, acc = _f(N, 3, like=rays), _f(N, like=rays), _f(N, like=rays)
    w = _f(N, S, like=rays) if want_weights else _f(0, like=rays)
    if N:
        capi.call("nvsr_render_pass_arith", C.byref(sc), capi.ptr(packed), N, S, capi.ptr(rays), capi.ptr(z), capi.ptr(noise), int(white),
                  capi.ptr(rgb), capi.ptr(disp), capi.ptr(acc), capi.ptr(w) if want_weights else None, None, None, arithmetic, capi.stream())
    return rgb, disp, acc, w


@render_pass.register_fake
def _(planes, consts, packed, rays, z, noise, white, want_weights, arithmetic):
    N, S = z.shape
    return rays.new_empty((N, 3)), rays.new_empty((N,)), rays.new_empty((N,)), rays.new_empty((N, S) if want_weights else (0,))


@custom_op("nvsr::render_rays", mutates_args=(), device_types="cuda")
def render_rays(planes: Sequence[Tensor], consts: Sequence[float], packed_coarse: Tensor, packed_fine: Optional[Tensor], rays: Tensor,
                Nc: int, Nf: int, lindisp: bool, white: bool, t_rand: Optional[Tensor], u: Optional[Tensor], noise_coarse: Optional[Tensor],
                noise_fine: Optional[Tensor], arithmetic: int) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    """predict_and_render_radiance for a ray block (inference): coarse depths -> coarse pass -> importance resampling -> fine pass.
    -> rgb_c, disp_c, acc_c, rgb_f, disp_f, acc_f (the fine ones empty when Nf == 0)"""
    rays, t_rand, u, noise_coarse, noise_fine = _c(rays), _c(t_rand), _c(u), _c(noise_coarse), _c(noise_fine)
    sc = _scene(planes, consts)
    N = rays.shape[0]
    rgb_c, disp_c, acc_c = _f(N, 3, like=rays), _f(N, like=rays), _f(N, like=rays)
    if Nf > 0:
        rgb_f, disp_f, acc_f = _f(N, 3, like=rays), _f(N, like=rays), _f(N, like=rays)
    else:
        rgb_f, disp_f, acc_f = _f(0, 3, like=rays), _f(0, like=rays), _f(0, like=rays)
    # one decoder for both passes (models.fine.type == 'use_same'): the fine pass evaluates the importance samples only (nvsr.h)
    if N and Nf > 0 and packed_fine is not None and packed_fine.data_ptr() == packed_coarse.data_ptr():
        ws = _f(capi.lib().nvsr_render_shared_workspace_floats(N, Nc, Nf), like=rays)
        capi.call("nvsr_render_rays_shared_arith", C.byref(sc), capi.ptr(packed_coarse), N, Nc, Nf, capi.ptr(rays), int(lindisp), int(white),
                  capi.ptr(t_rand), capi.ptr(u), capi.ptr(noise_coarse), capi.ptr(noise_fine), capi.ptr(rgb_c), capi.ptr(disp_c), capi.ptr(acc_c),
                  capi.ptr(rgb_f), capi.ptr(disp_f), capi.ptr(acc_f), capi.ptr(ws), arithmetic, capi.stream())
    elif N:
        ws = _f(capi.lib().nvsr_render_workspace_floats(N, Nc, Nf), like=rays)
        capi.call("nvsr_render_rays_arith", C.byref(sc), capi.ptr(packed_coarse), capi.ptr(packed_fine), N, Nc, Nf, capi.ptr(rays), int(lindisp),
                  int(white), capi.ptr(t_rand), capi.ptr(u), capi.ptr(noise_coarse), capi.ptr(noise_fine), capi.ptr(rgb_c), capi.ptr(disp_c),
                  capi.ptr(acc_c), capi.ptr(rgb_f) if Nf > 0 else None, capi.ptr(disp_f) if Nf > 0 else None,
                  capi.ptr(acc_f) if Nf > 0 else None, capi.ptr(ws), arithmetic, capi.stream())
    return rgb_c, disp_c, acc_c, rgb_f, disp_f, acc_f


@render_rays.register_fake
def _(planes, consts, packed_coarse, packed_fine, rays, Nc, Nf, lindisp, white, t_rand, u, noise_coarse, noise_fine, arithmetic):
    N = rays.shape[0]
    M = N if Nf > 0 else 0
    e = rays.new_empty
    return e((N, 3)), e((N,)), e((N,)), e((M, 3)), e((M,)), e((M,))


@custom_op("nvsr::decode_rays", mutates_args=(), device_types="cuda")
def decode_rays(planes: Sequence[Tensor], consts: Sequence[float], packed: Tensor, rays: Tensor, z: Tensor, want_gates: bool,
                want_record: bool, arithmetic: int) -> Tuple[Tensor, Tensor, Tensor]:
    """decoder outputs of every sample of every ray, tiled over (ray block, sample): raw [N,S,4]; with want_gates the ReLU gates of every
    layer ([N,S,32] int32, 128 B per point) and with want_record the layer inputs (9.2 KB per point) the backward operators consume"""
    rays, z = _c(rays), _c(z)
    sc = _scene(planes, consts)
    N, S = z.shape
    raw = _f(N, S, 4, like=rays)
    gates = torch.empty((N, S, 32) if (want_gates or want_record) else (0,), dtype=torch.int32, device=rays.device)
    rec = _f(capi.lib().nvsr_decoder_record_floats(N, S) if want_record else 0, like=rays)
    if N:
        capi.call("nvsr_decode_rays_arith", C.byref(sc), capi.ptr(packed), N, S, capi.ptr(rays), capi.ptr(z), capi.ptr(raw),
                  capi.ptr(_none_if_empty(gates)), capi.ptr(_none_if_empty(rec)), arithmetic, capi.stream())
    return raw, gates, rec


@decode_rays.register_fake
def _(planes, consts, packed, rays, z, want_gates, want_record, arithmetic):
    N, S = z.shape
    nrec = capi.lib().nvsr_decoder_record_floats(int(N), int(S)) if want_record else 0      # (host-side size arithmetic of the library)
    return (rays.new_empty((N, S, 4)), rays.new_empty((N, S, 32) if (want_gates or want_record) else (0,), dtype=torch.int32),
            rays.new_empty((nrec,)))


def _decode_rays_backward_launch(planes, consts, packed, packed_bwd, rays, z, g_raw, gates, record, need, arithmetic, grads):
    """the gate-driven backward scattering into `grads` (channel-last gradient planes, ADDED to; entries of planes that need no gradient ignored)"""
    rays, z, g_raw = _c(rays), _c(z), _c(g_raw)
    sc = _scene(planes, consts)
    N, S = z.shape
    if N == 0 or (not any(need) and record is None):
        return
    gptrs = (C.c_void_p * 4)(*[g.data_ptr() if need[d] else None for d, g in enumerate(grads)]) if any(need) else None
    # per-tile rows of the view-direction plane's gradient (summed per ray before they touch the plane)
    view_ws = _f(N * S * PC, like=rays) if (gptrs is not None and need[3]) else None
    capi.call("nvsr_render_pass_backward_gates_arith", C.byref(sc), capi.ptr(packed), capi.ptr(packed_bwd), N, S, capi.ptr(rays), capi.ptr(z),
              capi.ptr(g_raw), capi.ptr(gates), gptrs, capi.ptr(view_ws), capi.ptr(record), arithmetic, capi.stream())


def zero_planes_like(planes, need, like):
    """zero gradient planes laid out like `planes` (empty tensors where need[d] is False) for decode_rays_backward_: contiguous planes are
    carved out of ONE zero-filled allocation (one fill kernel per step instead of four; every plane starts on a 16-byte boundary).  (An
    operator may not RETURN tensors that share a storage, so the functional decode_rays_backward fills its four planes one by one.)"""
    wanted = [p for d, p in enumerate(planes) if need[d]]
    if len(wanted) < 2 or not all(p.is_contiguous() and p.dtype == torch.float32 for p in wanted):
        return [torch.zeros_like(p) if need[d] else _f(0, like=like) for d, p in enumerate(planes)]
    sizes = [(p.numel() + 3) // 4 * 4 for p in wanted]
    flat = torch.zeros(sum(sizes), dtype=torch.float32, device=wanted[0].device)
    out, off = [], 0
    for d, p in enumerate(planes):
        if not need[d]:
            out.append(_f(0, like=like))
            continue
        out.append(flat[off: off + p.numel()].view(p.shape))
        off += (p.numel() + 3) // 4 * 4
    return out


@custom_op("nvsr::decode_rays_backward", mutates_args=("record",), device_types="cuda")
def decode_rays_backward(planes: Sequence[Tensor], consts: Sequence[float], packed: Tensor, packed_bwd: Tensor, rays: Tensor, z: Tensor,
                         g_raw: Tensor, gates: Tensor, record: Optional[Tensor], need: Sequence[bool], arithmetic: int) -> List[Tensor]:
    """gate-driven backward of decode_rays: g_raw [N,S,4] -> gradient planes (channel-last, zeros where need[d] is False -> empty tensor);
    with `record` (the forward's) the pre-activation gradients are added to it for decoder_weight_grad.  `arithmetic` must be the
    forward's."""
    grads = [torch.zeros_like(p) if need[d] else _f(0, like=rays) for d, p in enumerate(planes)]
    _decode_rays_backward_launch(planes, consts, packed, packed_bwd, rays, z, g_raw, gates, record, need, arithmetic, grads)
    return grads


@decode_rays_backward.register_fake
def _(planes, consts, packed, packed_bwd, rays, z, g_raw, gates, record, need, arithmetic):
    return [torch.empty_like(p) if need[d] else rays.new_empty((0,)) for d, p in enumerate(planes)]


@custom_op("nvsr::decode_rays_backward_", mutates_args=("record", "grads"), device_types="cuda")
def decode_rays_backward_(planes: Sequence[Tensor], consts: Sequence[float], packed: Tensor, packed_bwd: Tensor, rays: Tensor, z: Tensor,
                          g_raw: Tensor, gates: Tensor, record: Optional[Tensor], need: Sequence[bool], arithmetic: int,
                          grads: Sequence[Tensor]) -> None:
    """decode_rays_backward ACCUMULATING into existing gradient planes (`grads`: what the functional form returned for an earlier pass over
    the same planes): the fine pass of a training step scatters into the coarse pass's buffers instead of zero-filling a second set of
    full-size planes and adding the two (the reference's autograd accumulates into one .grad the same way)."""
    for d, (g, p) in enumerate(zip(grads, planes)):
        if need[d] and (g.shape != p.shape or g.stride() != p.stride() or g.dtype != torch.float32 or not g.is_cuda):
            raise ValueError("decode_rays_backward_: grads[%d] must be a float32 CUDA tensor laid out like planes[%d]" % (d, d))
    _decode_rays_backward_launch(planes, consts, packed, packed_bwd, rays, z, g_raw, gates, record, need, arithmetic, grads)


@decode_rays_backward_.register_fake
def _(planes, consts, packed, packed_bwd, rays, z, g_raw, gates, record, need, arithmetic, grads):
    return None


@custom_op("nvsr::decode_rays_backward_recompute", mutates_args=(), device_types="cuda")
def decode_rays_backward_recompute(planes: Sequence[Tensor], consts: Sequence[float], packed: Tensor, packed_bwd: Tensor, rays: Tensor,
                                   z: Tensor, g_raw: Tensor, need: Sequence[bool], want_decoder_grad: bool, arithmetic: int) -> List[Tensor]:
    """backward of a decode pass that recomputes the forward (exact-f32 kernel), RECORD_RAYS rays at a time: the memory-bounded path for
    decoder gradients of very large batches, and the path for passes that published no gates.
    -> [g_plane0..3 (empty where not needed), g_natural (empty unless want_decoder_grad)]"""
    rays, z, g_raw = _c(rays), _c(z), _c(g_raw)
    sc = _scene(planes, consts)
    N, S = z.shape
    grads = [torch.zeros_like(p) if need[d] else _f(0, like=rays) for d, p in enumerate(planes)]
    gnat = torch.zeros(capi.DECODER_NATURAL_FLOATS, dtype=torch.float32, device=rays.device) if want_decoder_grad else _f(0, like=rays)
    if N == 0 or (not any(need) and not want_decoder_grad):
        return grads + [gnat]
    gptrs = (C.c_void_p * 4)(*[g.data_ptr() if need[d] else None for d, g in enumerate(grads)]) if any(need) else None
    view_ws = _f(N * S * PC, like=rays) if (gptrs is not None and need[3]) else None
    st = capi.stream()
    if not want_decoder_grad:
        capi.call("nvsr_render_pass_backward_ex", C.byref(sc), capi.ptr(packed), capi.ptr(packed_bwd), N, S, capi.ptr(rays), capi.ptr(z),
                  capi.ptr(g_raw), gptrs, None, capi.ptr(view_ws), st)
        return grads + [gnat]
    step = min(N, RECORD_RAYS)
    record = _f(capi.lib().nvsr_decoder_record_floats(step, S), like=rays)
    for a in range(0, N, step):
        n = min(step, N - a)
        capi.call("nvsr_render_pass_backward_ex", C.byref(sc), capi.ptr(packed), capi.ptr(packed_bwd), n, S, capi.ptr(rays[a:]), capi.ptr(z[a:]),
                  capi.ptr(g_raw[a:]), gptrs, capi.ptr(record), capi.ptr(view_ws), st)
        # the recomputing kernel is the exact-f32 one; its record is contracted in the caller's arithmetic (both are held to 1e-5 of float64)
        capi.call("nvsr_decoder_weight_grad_arith", n, S, capi.ptr(record), capi.ptr(gnat), arithmetic, st)
    return grads + [gnat]


@decode_rays_backward_recompute.register_fake
def _(planes, consts, packed, packed_bwd, rays, z, g_raw, need, want_decoder_grad, arithmetic):
    return [torch.empty_like(p) if need[d] else rays.new_empty((0,)) for d, p in enumerate(planes)] + \
        [rays.new_empty((capi.DECODER_NATURAL_FLOATS if want_decoder_grad else 0,))]


@custom_op("nvsr::decoder_weight_grad", mutates_args=(), device_types="cuda")
def decoder_weight_grad(record: Tensor, N: int, S: int, arithmetic: int) -> Tensor:
    """one contraction over all points of the record -> weight / bias gradients in state-dict order"""
    gnat = torch.zeros(capi.DECODER_NATURAL_FLOATS, dtype=torch.float32, device=record.device)
    if N:
        capi.call("nvsr_decoder_weight_grad_arith", N, S, capi.ptr(record), capi.ptr(gnat), arithmetic, capi.stream())
    return gnat


@decoder_weight_grad.register_fake
def _(record, N, S, arithmetic):
    return record.new_empty((capi.DECODER_NATURAL_FLOATS,))


# =====================================================================================================================================
# compositing (volume_rendering_utils.py:6-51)
# =====================================================================================================================================
@custom_op("nvsr::composite", mutates_args=(), device_types="cuda")
def composite(raw: Tensor, z: Tensor, rd: Tensor, noise: Optional[Tensor], white: bool, mip: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """raw [N,S,4], z [N,S] (mip: the S+1 interval edges), rd [N,3] -> rgb [N,3], disp [N], acc [N], weights [N,S], depth [N]"""
    raw, z, rd, noise = _c(raw), _c(z), _c(rd), _c(noise)
    N = z.shape[0]
    S = z.shape[1] - (1 if mip else 0)
    rgb, disp, acc, depth, w = _f(N, 3, like=raw), _f(N, like=raw), _f(N, like=raw), _f(N, like=raw), _f(N, S, like=raw)
    if N:
        capi.call("nvsr_composite_mip" if mip else "nvsr_composite", N, S, capi.ptr(raw), capi.ptr(z), capi.ptr(rd), capi.ptr(noise), int(white),
                  capi.ptr(rgb), capi.ptr(disp), capi.ptr(acc), capi.ptr(w), capi.ptr(depth), capi.stream())
    return rgb, disp, acc, w, depth


@composite.register_fake
def _(raw, z, rd, noise, white, mip):
    N = z.shape[0]
    S = z.shape[1] - (1 if mip else 0)
    e = raw.new_empty
    return e((N, 3)), e((N,)), e((N,)), e((N, S)), e((N,))


@custom_op("nvsr::composite_rays", mutates_args=(), device_types="cuda")
def composite_rays(raw: Tensor, z: Tensor, rays: Tensor, noise: Optional[Tensor], white: bool, want_weights: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """composite with the ray directions taken from packed rays [N,11] -> rgb, disp, acc, weights (empty unless want_weights)"""
    raw, z, rays, noise = _c(raw), _c(z), _c(rays), _c(noise)
    N, S = z.shape
    rgb, disp, acc = _f(N, 3, like=raw), _f(N, like=raw), _f(N, like=raw)
    w = _f(N, S, like=raw) if want_weights else _f(0, like=raw)
    if N:
        capi.call("nvsr_composite_rays", N, S, capi.ptr(raw), capi.ptr(z), capi.ptr(rays), capi.ptr(noise), int(white), capi.ptr(rgb),
                  capi.ptr(disp), capi.ptr(acc), capi.ptr(w) if want_weights else None, None, capi.stream())
    return rgb, disp, acc, w


@composite_rays.register_fake
def _(raw, z, rays, noise, white, want_weights):
    N, S = z.shape
    e = raw.new_empty
    return e((N, 3)), e((N,)), e((N,)), e((N, S) if want_weights else (0,))


@custom_op("nvsr::composite_backward", mutates_args=(), device_types="cuda")
def composite_backward(raw: Tensor, z: Tensor, rd: Tensor, noise: Optional[Tensor], white: bool, mip: bool, g_rgb: Tensor,
                       g_acc: Optional[Tensor], g_depth: Optional[Tensor] = None) -> Tensor:
    """gradient of rgb_map / acc_map / depth_map with respect to the radiance field: -> g_raw [N,S,4]; S <= 512"""
    raw, z, rd, noise, g_rgb, g_acc, g_depth = _c(raw), _c(z), _c(rd), _c(noise), _c(g_rgb), _c(g_acc), _c(g_depth)
    N = z.shape[0]
    S = z.shape[1] - (1 if mip else 0)
    g_raw = torch.empty_like(raw)          # (the kernel writes all four channels of every sample)
    if N:
        if S > 512:
            raise NotImplementedError("nvsr_composite_backward handles up to 512 samples per ray")
        capi.call("nvsr_composite_backward_depth", N, S, capi.ptr(raw), capi.ptr(z), capi.ptr(rd), capi.ptr(noise), int(white), capi.ptr(g_rgb),
                  capi.ptr(g_acc), capi.ptr(g_depth), int(mip), capi.ptr(g_raw), capi.stream())
    return g_raw


@composite_backward.register_fake
def _(raw, z, rd, noise, white, mip, g_rgb, g_acc, g_depth=None):
    return torch.empty_like(raw)


@custom_op("nvsr::composite_backward_rays", mutates_args=(), device_types="cuda")
def composite_backward_rays(raw: Tensor, z: Tensor, rays: Tensor, noise: Optional[Tensor], white: bool, mip: bool, g_rgb: Tensor,
                            g_acc: Optional[Tensor], g_depth: Optional[Tensor] = None) -> Tensor:
    """composite_backward with the ray directions taken from packed rays [N,11] (the backward of composite_rays; no [N,3] copy)"""
    raw, z, rays, noise, g_rgb, g_acc, g_depth = _c(raw), _c(z), _c(rays), _c(noise), _c(g_rgb), _c(g_acc), _c(g_depth)
    N = z.shape[0]
    S = z.shape[1] - (1 if mip else 0)
    assert rays.shape == (N, 11)
    g_raw = torch.empty_like(raw)
    if N:
        if S > 512:
            raise NotImplementedError("nvsr_composite_backward handles up to 512 samples per ray")
        capi.call("nvsr_composite_backward_rays", N, S, capi.ptr(raw), capi.ptr(z), capi.ptr(rays), capi.ptr(noise), int(white), capi.ptr(g_rgb),
                  capi.ptr(g_acc), capi.ptr(g_depth), int(mip), capi.ptr(g_raw), capi.stream())
    return g_raw


@composite_backward_rays.register_fake
def _(raw, z, rays, noise, white, mip, g_rgb, g_acc, g_depth=None):
    return torch.empty_like(raw)


def fold_disp_grad(g_disp, q, acc, depth, g_acc, g_depth):
    """disp_map = 1 / max(1e-10, q), q = depth_map / acc_map (volume_rendering_utils.py:46): the incoming gradient of disp_map as additions to
    those of depth_map and acc_map (per-ray scalars; torch.max passes the gradient to the larger operand, q = NaN (acc = 0) passes none)
    -> (g_acc, g_depth)"""
    live = q > 1e-10
    dq = torch.where(live, -capi.f32c(g_disp) / (q * q), torch.zeros_like(q))
    gd = torch.where(live, dq / acc, torch.zeros_like(q))
    ga = torch.where(live, -dq * depth / (acc * acc), torch.zeros_like(q))
    return (ga if g_acc is None else g_acc + ga), (gd if g_depth is None else g_depth + gd)


def _composite_setup(ctx, inputs, output):
    raw, z, rd, noise, white, mip = inputs
    ctx.save_for_backward(raw, z, rd, noise, output[2], output[4])      # + acc_map, depth_map: disp_map's chain rule
    ctx.white, ctx.mip = white, mip
    ctx.mark_non_differentiable(output[3])


def _composite_bwd(ctx, g_rgb, g_disp, g_acc, g_w, g_depth):
    # rgb_map, acc_map, depth_map and disp_map = 1 / max(1e-10, depth_map / acc_map) are differentiable like the reference's
    # (volume_rendering_utils.py:38-46); the per-sample weights output is not (nothing downstream of the reference's renderer uses it).
    # disp's gradient is folded into those of depth and acc here -- per-ray scalars, plumbing; the kernel sums over the samples.
    raw, z, rd, noise, acc, depth = ctx.saved_tensors
    g_rgb = torch.zeros((z.shape[0], 3), dtype=torch.float32, device=raw.device) if g_rgb is None else capi.f32c(g_rgb)
    g_acc = None if g_acc is None else capi.f32c(g_acc)
    g_depth = None if g_depth is None else capi.f32c(g_depth)
    if g_disp is not None:
        g_acc, g_depth = fold_disp_grad(g_disp, depth / acc, acc, depth, g_acc, g_depth)
    return torch.ops.nvsr.composite_backward(raw, z, rd, noise, ctx.white, ctx.mip, g_rgb, g_acc, g_depth), None, None, None, None, None


composite.register_autograd(_composite_bwd, setup_context=_composite_setup)


# =====================================================================================================================================
# feature-plane super-resolution (models.py:769-822 EDSR, :884-926 PlanesSR).  geometry = [Cin, Cout, hidden, n_blocks, n_up]
# =====================================================================================================================================
def _edsr_out_size(H, W, nb, n_up):
    h, w = H - 2 - 4 * nb - 2, W - 2 - 4 * nb - 2
    for _ in range(n_up):
        h, w = (h - 2) * 2, (w - 2) * 2
    return h - 2, w - 2


@custom_op("nvsr::pack_edsr", mutates_args=(), device_types="cuda")
def pack_edsr(natural: Tensor, geometry: Sequence[int], dgrad: bool, arithmetic: int = PACK_ALL_ARITHMETICS) -> Tensor:
    """conv weights in state-dict order -> MFMA-fragment blob (dgrad: of the flipped, transposed kernels of the data gradient).
    arithmetic: PACK_ALL_ARITHMETICS = every fragment region; an NVSR_ARITH_* code = only the regions a launch in that arithmetic reads
    (nvsr_pack_edsr_arith: a fifth of the bytes, for weights that change every iteration); the other regions are uninitialised memory, so the
    blob serves launches in THAT arithmetic only (EDSR.packed_weights keys its cache on it)"""
    nat = capi.f32c(natural)
    lib = capi.lib()
    assert nat.numel() == lib.nvsr_edsr_natural_floats(*geometry)
    n = (lib.nvsr_edsr_packed_dgrad_floats if dgrad else lib.nvsr_edsr_packed_floats)(*geometry)
    assert n > 0
    packed = _f(n, like=nat)
    if os.environ.get("NVSR_PACK_POISON") == "1":          # (tests / tools: NaN words wherever the packer does not write)
        packed.fill_(float("nan"))
    capi.call("nvsr_pack_edsr_dgrad_arith" if dgrad else "nvsr_pack_edsr_arith", capi.ptr(nat), *geometry, capi.ptr(packed), int(arithmetic), capi.stream())
    return packed


@pack_edsr.register_fake
def _(natural, geometry, dgrad, arithmetic=PACK_ALL_ARITHMETICS):
    lib = capi.lib()
    return natural.new_empty(((lib.nvsr_edsr_packed_dgrad_floats if dgrad else lib.nvsr_edsr_packed_floats)(*[int(g) for g in geometry]),))


@custom_op("nvsr::edsr", mutates_args=(), device_types="cuda")
def edsr(x: Tensor, packed: Tensor, geometry: Sequence[int], arithmetic: int) -> Tensor:
    """EDSR.forward: x [B,Cin,H,W] -> [B,Cout,Ho,Wo] (un-padded 3x3 convolutions, fused ReLU / residual / PixelShuffle epilogues)"""
    x = _c(x)
    cin, cout, hid, nb, n_up = geometry
    B, Cin, H, W = x.shape
    assert Cin == cin
    Ho, Wo = _edsr_out_size(H, W, nb, n_up)
    if Ho < 1 or Wo < 1:
        raise capi.NvsrError("EDSR: input smaller than the receptive field")
    out = _f(B, cout, Ho, Wo, like=x)
    ws = _f(B * capi.lib().nvsr_edsr_workspace_floats(hid, nb, n_up, H, W), like=x)
    capi.call("nvsr_edsr_forward_batch_arith", capi.ptr(x), B, Cin, H, W, capi.ptr(packed), cout, hid, nb, n_up, capi.ptr(out), capi.ptr(ws),
              arithmetic, capi.stream())
    return out


@edsr.register_fake
def _(x, packed, geometry, arithmetic):
    cin, cout, hid, nb, n_up = geometry
    Ho, Wo = _edsr_out_size(x.shape[2], x.shape[3], nb, n_up)
    return x.new_empty((x.shape[0], cout, Ho, Wo))


@custom_op("nvsr::edsr_train", mutates_args=(), device_types="cuda")
def edsr_train(x: Tensor, natural: Tensor, packed: Tensor, packed_dgrad: Tensor, geometry: Sequence[int], arithmetic: int) -> Tuple[Tensor, Tensor]:
    """EDSR.forward that keeps every layer's input (`acts`) for the backward; differentiable in x and `natural` (the conv weights in
    state-dict order; `packed` / `packed_dgrad` are their fragment blobs).  x [1,Cin,H,W]"""
    x = _c(x)
    cin, cout, hid, nb, n_up = geometry
    assert x.shape[0] == 1 and x.shape[1] == cin, "the training forward runs one plane at a time"
    H, W = x.shape[2:]
    Ho, Wo = _edsr_out_size(H, W, nb, n_up)
    out = _f(1, cout, Ho, Wo, like=x)
    acts = _f(capi.lib().nvsr_edsr_acts_floats(cin, cout, hid, nb, n_up, H, W), like=x)
    capi.call("nvsr_edsr_forward_train_arith", capi.ptr(x), cin, H, W, capi.ptr(packed), cout, hid, nb, n_up, capi.ptr(out), capi.ptr(acts),
              arithmetic, capi.stream())
    return out, acts


@edsr_train.register_fake
def _(x, natural, packed, packed_dgrad, geometry, arithmetic):
    cin, cout, hid, nb, n_up = geometry
    Ho, Wo = _edsr_out_size(x.shape[2], x.shape[3], nb, n_up)
    return x.new_empty((1, cout, Ho, Wo)), x.new_empty((capi.lib().nvsr_edsr_acts_floats(cin, cout, hid, nb, n_up, int(x.shape[2]), int(x.shape[3])),))


@custom_op("nvsr::edsr_backward", mutates_args=(), device_types="cuda")
def edsr_backward(x: Tensor, acts: Tensor, packed_dgrad: Tensor, geometry: Sequence[int], d_out: Tensor, need_dx: bool, arithmetic: int) -> Tuple[Tensor, Tensor]:
    """-> (weight gradients in state-dict order, dx (empty unless need_dx)); deterministic"""
    x, d_out = _c(x), _c(d_out)
    cin, cout, hid, nb, n_up = geometry
    H, W = x.shape[2:]
    lib = capi.lib()
    gnat = torch.zeros(lib.nvsr_edsr_natural_floats(*geometry), dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x) if need_dx else _f(0, like=x)
    ws = _f(lib.nvsr_edsr_backward_workspace_floats(cin, cout, hid, nb, n_up, H, W), like=x)
    capi.call("nvsr_edsr_backward_arith", capi.ptr(x), cin, H, W, capi.ptr(acts), capi.ptr(packed_dgrad), cout, hid, nb, n_up, capi.ptr(d_out),
              capi.ptr(gnat), capi.ptr(dx) if need_dx else None, capi.ptr(ws), arithmetic, capi.stream())
    return gnat, dx


@edsr_backward.register_fake
def _(x, acts, packed_dgrad, geometry, d_out, need_dx, arithmetic):
    cin, cout, hid, nb, n_up = geometry
    n = 9 * (hid * cin + (2 * nb + 1) * hid * hid + n_up * 4 * hid * hid + cout * hid)
    return x.new_empty((n,)), (torch.empty_like(x) if need_dx else x.new_empty((0,)))


def _edsr_train_setup(ctx, inputs, output):
    x, natural, packed, packed_dgrad, geometry, arithmetic = inputs
    ctx.save_for_backward(x, output[1], packed_dgrad)
    ctx.geometry, ctx.arithmetic = list(geometry), arithmetic
    ctx.mark_non_differentiable(output[1])


def _edsr_train_bwd(ctx, d_out, d_acts):
    x, acts, packed_dgrad = ctx.saved_tensors
    gnat, dx = torch.ops.nvsr.edsr_backward(x, acts, packed_dgrad, ctx.geometry, capi.f32c(d_out), ctx.needs_input_grad[0], ctx.arithmetic)
    return (dx if ctx.needs_input_grad[0] else None), (gnat if ctx.needs_input_grad[1] else None), None, None, None, None


edsr_train.register_autograd(_edsr_train_bwd, setup_context=_edsr_train_setup)


def _roi_c(roi):
    return None if roi is None else (C.c_float * 4)(*[float(v) for v in roi])


@custom_op("nvsr::planes_sr", mutates_args=(), device_types="cuda")
def planes_sr(lr: Sequence[Tensor], packed: Tensor, geometry: Sequence[int], pad: int, over: int, roi: Optional[Sequence[float]],
              mean: Optional[Tensor], std: Optional[Tensor], arithmetic: int, align_corners: bool = True, bicubic: bool = False) -> List[Tensor]:
    """PlanesSR.forward for B equally sized LR planes [C,R0,R1] in ONE batched pass: crop + replicate pad -> EDSR -> crop over-padding
    -> + bilinear x sf of the LR plane (F.interpolate's align_corners as given), NaN outside the ROI.  roi: None (full plane) or
    [ymin, xmin, ymax, xmax] in [-1, 1].  -> B tensors [1,C,sf R0,sf R1]"""
    lr = [_c(t) for t in lr]
    cin, cout, hid, nb, n_up = geometry
    B = len(lr)
    Cc, R0, R1 = lr[0].shape[-3:]
    assert Cc == cin == cout and all(tuple(t.shape[-3:]) == (Cc, R0, R1) for t in lr)
    roi_c = _roi_c(roi)
    nws = capi.lib().nvsr_planes_sr_workspace_floats(Cc, R0, R1, hid, nb, n_up, pad, roi_c)
    if nws < 0:
        raise capi.NvsrError("PlanesSR: region of interest too small for the network")
    sf = 1 << n_up
    outs = [_f(1, Cc, R0 * sf, R1 * sf, like=lr[0]) for _ in lr]
    ws = _f(B * nws, like=lr[0])
    # (align_corners / interpolation of the residual are arguments of the call: no process-wide state between this thread and a backward thread)
    lr_ptrs = (C.c_void_p * B)(*[t.data_ptr() for t in lr])
    out_ptrs = (C.c_void_p * B)(*[t.data_ptr() for t in outs])
    capi.call("nvsr_planes_sr_batch_ex", lr_ptrs, B, Cc, R0, R1, capi.ptr(packed), hid, nb, n_up, pad, over, roi_c, capi.ptr(mean),
              capi.ptr(std), out_ptrs, capi.ptr(ws), arithmetic, int(bool(align_corners)), int(bool(bicubic)), capi.stream())
    return outs


@planes_sr.register_fake
def _(lr, packed, geometry, pad, over, roi, mean, std, arithmetic, align_corners=True, bicubic=False):
    sf = 1 << geometry[4]
    Cc, R0, R1 = lr[0].shape[-3:]
    return [t.new_empty((1, Cc, R0 * sf, R1 * sf)) for t in lr]


@custom_op("nvsr::planes_sr_train", mutates_args=(), device_types="cuda")
def planes_sr_train(lr: Tensor, natural: Tensor, packed: Tensor, packed_dgrad: Tensor, geometry: Sequence[int], pad: int, over: int,
                    roi: Optional[Sequence[float]], mean: Optional[Tensor], std: Optional[Tensor], arithmetic: int,
                    align_corners: bool = True, bicubic: bool = False) -> Tuple[Tensor, Tensor]:
    """planes_sr of one plane that keeps the prepared input + activation record (`keep`); differentiable in `lr` and `natural`"""
    lr = _c(lr)
    cin, cout, hid, nb, n_up = geometry
    Cc, R0, R1 = lr.shape[-3:]
    roi_c = _roi_c(roi)
    lib = capi.lib()
    nws = lib.nvsr_planes_sr_workspace_floats(Cc, R0, R1, hid, nb, n_up, pad, roi_c)
    nkeep = lib.nvsr_planes_sr_keep_floats(Cc, R0, R1, hid, nb, n_up, pad, roi_c)
    if nws < 0 or nkeep < 0:
        raise capi.NvsrError("PlanesSR: region of interest too small for the network")
    sf = 1 << n_up
    out, ws, keep = _f(1, Cc, R0 * sf, R1 * sf, like=lr), _f(nws, like=lr), _f(nkeep, like=lr)
    # (the B = 1 case of the batch entry point: same kernels plane by plane, the residual's settings are arguments of the call)
    lr_ptrs, out_ptrs = (C.c_void_p * 1)(lr.data_ptr()), (C.c_void_p * 1)(out.data_ptr())
    capi.call("nvsr_planes_sr_train_batch_arith", lr_ptrs, 1, Cc, R0, R1, capi.ptr(packed), hid, nb, n_up, pad, over, roi_c, capi.ptr(mean),
              capi.ptr(std), out_ptrs, capi.ptr(ws), capi.ptr(keep), arithmetic, int(bool(align_corners)), int(bool(bicubic)), capi.stream())
    return out, keep


@planes_sr_train.register_fake
def _(lr, natural, packed, packed_dgrad, geometry, pad, over, roi, mean, std, arithmetic, align_corners=True, bicubic=False):
    cin, cout, hid, nb, n_up = geometry
    sf = 1 << n_up
    Cc, R0, R1 = lr.shape[-3:]
    nkeep = capi.lib().nvsr_planes_sr_keep_floats(int(Cc), int(R0), int(R1), hid, nb, n_up, pad, _roi_c(roi))
    return lr.new_empty((1, Cc, R0 * sf, R1 * sf)), lr.new_empty((nkeep,))


@custom_op("nvsr::planes_sr_backward", mutates_args=(), device_types="cuda")
def planes_sr_backward(keep: Tensor, packed_dgrad: Tensor, plane_shape: Sequence[int], geometry: Sequence[int], pad: int, over: int,
                       roi: Optional[Sequence[float]], std: Optional[Tensor], d_out: Tensor, need_lr: bool, arithmetic: int,
                       align_corners: bool = True, bicubic: bool = False) -> Tuple[Tensor, Tensor]:
    """-> (EDSR weight gradients in state-dict order, d_lr [1,C,R0,R1] (empty unless need_lr))"""
    d_out = _c(d_out)
    cin, cout, hid, nb, n_up = geometry
    Cc, R0, R1 = plane_shape
    roi_c = _roi_c(roi)
    lib = capi.lib()
    gnat = torch.zeros(lib.nvsr_edsr_natural_floats(*geometry), dtype=torch.float32, device=keep.device)
    d_lr = torch.zeros((1, Cc, R0, R1), dtype=torch.float32, device=keep.device) if need_lr else _f(0, like=keep)
    ws = _f(lib.nvsr_planes_sr_batch_backward_workspace_floats(1, Cc, R0, R1, hid, nb, n_up, pad, roi_c), like=keep)
    d_out_ptrs = (C.c_void_p * 1)(d_out.data_ptr())
    d_lr_ptrs = (C.c_void_p * 1)(d_lr.data_ptr() if need_lr else None)
    capi.call("nvsr_planes_sr_backward_batch_arith", 1, Cc, R0, R1, capi.ptr(keep), capi.ptr(packed_dgrad), hid, nb, n_up, pad, over, roi_c, capi.ptr(std),
              d_out_ptrs, capi.ptr(gnat), d_lr_ptrs if need_lr else None, capi.ptr(ws), arithmetic, int(bool(align_corners)), int(bool(bicubic)),
              capi.stream())
    return gnat, d_lr


@planes_sr_backward.register_fake
def _(keep, packed_dgrad, plane_shape, geometry, pad, over, roi, std, d_out, need_lr, arithmetic, align_corners=True, bicubic=False):
    cin, cout, hid, nb, n_up = geometry
    n = 9 * (hid * cin + (2 * nb + 1) * hid * hid + n_up * 4 * hid * hid + cout * hid)
    Cc, R0, R1 = plane_shape
    return keep.new_empty((n,)), keep.new_empty((1, Cc, R0, R1) if need_lr else (0,))


def _planes_sr_train_setup(ctx, inputs, output):
    lr, natural, packed, packed_dgrad, geometry, pad, over, roi, mean, std, arithmetic, align_corners, bicubic = inputs
    ctx.save_for_backward(output[1], packed_dgrad, std)
    ctx.mark_non_differentiable(output[1])
    ctx.args = (list(lr.shape[-3:]), list(geometry), pad, over, None if roi is None else list(roi), arithmetic, lr.shape, bool(align_corners), bool(bicubic))


def _planes_sr_train_bwd(ctx, d_out, d_keep):
    keep, packed_dgrad, std = ctx.saved_tensors
    shape, geometry, pad, over, roi, arithmetic, lr_shape, align_corners, bicubic = ctx.args
    gnat, d_lr = torch.ops.nvsr.planes_sr_backward(keep, packed_dgrad, shape, geometry, pad, over, roi, std, capi.f32c(d_out),
                                                   ctx.needs_input_grad[0], arithmetic, align_corners, bicubic)
    return ((d_lr.reshape(lr_shape) if ctx.needs_input_grad[0] else None), (gnat if ctx.needs_input_grad[1] else None)) + (None,) * 11


planes_sr_train.register_autograd(_planes_sr_train_bwd, setup_context=_planes_sr_train_setup)



# ---- the regions of interest of B planes through the SR network at once (training; include/nvsr.h "SR training on the regions of interest of B
# planes at once").  Buffers are allocated at the capacity of FULL planes whatever the regions are: a training iteration's regions change with
# every batch of rays, and exact sizes would hand the caching allocator a new set of multi-GB block sizes per iteration (measured on the refine
# workload: the iteration waited ~40 ms for the allocator); with one size per buffer the blocks of the previous iteration are reused as they are.
def _rois_c(rois, B):
    if rois is None:
        return None
    assert len(rois) == 4 * B
    return (C.c_float * (4 * B))(*[float(v) for v in rois])


@custom_op("nvsr::planes_sr_train_batch", mutates_args=(), device_types="cuda")
def planes_sr_train_batch(lr: Sequence[Tensor], packed: Tensor, geometry: Sequence[int], pad: int, over: int, rois: Optional[Sequence[float]],
                          mean: Optional[Tensor], std: Optional[Tensor], arithmetic: int, align_corners: bool, bicubic: bool) -> List[Tensor]:
    """planes_sr_train of B equally sized planes [C,R0,R1], every one on its own region of interest (rois: 4 floats per plane, or None), one
    launch per layer for all of them.  -> [out_0 .. out_{B-1} ([1,C,sf R0,sf R1] each, NaN outside the region), keep]"""
    lr = [_c(t) for t in lr]
    cin, cout, hid, nb, n_up = geometry
    B = len(lr)
    Cc, R0, R1 = lr[0].shape[-3:]
    assert Cc == cin == cout and all(tuple(t.shape[-3:]) == (Cc, R0, R1) for t in lr)
    lib = capi.lib()
    rois_c = _rois_c(rois, B)
    if lib.nvsr_planes_sr_batch_keep_floats(B, Cc, R0, R1, hid, nb, n_up, pad, rois_c) < 0:
        raise capi.NvsrError("PlanesSR: region of interest too small for the network")
    nkeep = lib.nvsr_planes_sr_batch_keep_floats(B, Cc, R0, R1, hid, nb, n_up, pad, None)
    nws = lib.nvsr_planes_sr_batch_workspace_floats(B, Cc, R0, R1, hid, nb, n_up, pad, None)
    sf = 1 << n_up
    outs = [_f(1, Cc, R0 * sf, R1 * sf, like=lr[0]) for _ in lr]
    ws, keep = _f(nws, like=lr[0]), _f(nkeep, like=lr[0])
    lr_ptrs = (C.c_void_p * B)(*[t.data_ptr() for t in lr])
    out_ptrs = (C.c_void_p * B)(*[t.data_ptr() for t in outs])
    capi.call("nvsr_planes_sr_train_batch_arith", lr_ptrs, B, Cc, R0, R1, capi.ptr(packed), hid, nb, n_up, pad, over, rois_c, capi.ptr(mean), capi.ptr(std),
              out_ptrs, capi.ptr(ws), capi.ptr(keep), arithmetic, int(bool(align_corners)), int(bool(bicubic)), capi.stream())
    return outs + [keep]


@planes_sr_train_batch.register_fake
def _(lr, packed, geometry, pad, over, rois, mean, std, arithmetic, align_corners, bicubic):
    cin, cout, hid, nb, n_up = geometry
    sf = 1 << n_up
    Cc, R0, R1 = lr[0].shape[-3:]
    nkeep = capi.lib().nvsr_planes_sr_batch_keep_floats(len(lr), int(Cc), int(R0), int(R1), hid, nb, n_up, pad, None)
    return [t.new_empty((1, Cc, R0 * sf, R1 * sf)) for t in lr] + [lr[0].new_empty((nkeep,))]


@custom_op("nvsr::planes_sr_backward_batch", mutates_args=(), device_types="cuda")
def planes_sr_backward_batch(keep: Tensor, packed_dgrad: Tensor, plane_shape: Sequence[int], geometry: Sequence[int], pad: int, over: int,
                             rois: Optional[Sequence[float]], std: Optional[Tensor], d_out: Sequence[Tensor], need_lr: Sequence[bool], arithmetic: int,
                             align_corners: bool, bicubic: bool) -> List[Tensor]:
    """-> [EDSR weight gradients in state-dict order (summed over the planes), d_lr_0 .. d_lr_{B-1} ([1,C,R0,R1]; empty where not needed)]"""
    d_out = [_c(t) for t in d_out]
    cin, cout, hid, nb, n_up = geometry
    Cc, R0, R1 = plane_shape
    B = len(d_out)
    lib = capi.lib()
    rois_c = _rois_c(rois, B)
    gnat = torch.zeros(lib.nvsr_edsr_natural_floats(*geometry), dtype=torch.float32, device=keep.device)
    d_lr = [torch.zeros((1, Cc, R0, R1), dtype=torch.float32, device=keep.device) if n else _f(0, like=keep) for n in need_lr]
    ws = _f(lib.nvsr_planes_sr_batch_backward_workspace_floats(B, Cc, R0, R1, hid, nb, n_up, pad, None), like=keep)
    d_out_ptrs = (C.c_void_p * B)(*[t.data_ptr() for t in d_out])
    d_lr_ptrs = (C.c_void_p * B)(*[(t.data_ptr() if n else None) for t, n in zip(d_lr, need_lr)])
    capi.call("nvsr_planes_sr_backward_batch_arith", B, Cc, R0, R1, capi.ptr(keep), capi.ptr(packed_dgrad), hid, nb, n_up, pad, over, rois_c, capi.ptr(std),
              d_out_ptrs, capi.ptr(gnat), d_lr_ptrs if any(need_lr) else None, capi.ptr(ws), arithmetic, int(bool(align_corners)), int(bool(bicubic)),
              capi.stream())
    return [gnat] + d_lr


@planes_sr_backward_batch.register_fake
def _(keep, packed_dgrad, plane_shape, geometry, pad, over, rois, std, d_out, need_lr, arithmetic, align_corners, bicubic):
    cin, cout, hid, nb, n_up = geometry
    n = 9 * (hid * cin + (2 * nb + 1) * hid * hid + n_up * 4 * hid * hid + cout * hid)
    Cc, R0, R1 = plane_shape
    return [keep.new_empty((n,))] + [keep.new_empty((1, Cc, R0, R1) if k else (0,)) for k in need_lr]


def planes_sr_backward_batch_marked(keep, packed_dgrad, plane_shape, geometry, pad, over, rois, std, d_out, need_lr, arithmetic, align_corners, bicubic,
                                    bucket_sync):
    """planes_sr_backward_batch with the weight-gradient blob handed to `bucket_sync` (distributed.OverlappedSRGradSync) bucket by bucket WHILE the
    backward runs: the library records one event per bucket as soon as that part of the blob is final (nvsr_planes_sr_backward_batch_marks; the
    layers finish from the last to the first, the blob is in state-dict order: buckets are suffixes), bucket_sync starts the all-reduce of the
    bucket behind its event on the collective stream, and the iteration's stream waits for the collectives (and scales by 1 / world) before the
    blob goes back to autograd.  Not a registered operator: events and a process group are not tensors; called from PlanesSRBatchFn.backward only."""
    d_out = [_c(t) for t in d_out]
    cin, cout, hid, nb, n_up = geometry
    Cc, R0, R1 = plane_shape
    B = len(d_out)
    lib = capi.lib()
    rois_c = _rois_c(rois, B)
    gnat = torch.zeros(lib.nvsr_edsr_natural_floats(*geometry), dtype=torch.float32, device=keep.device)
    d_lr = [torch.zeros((1, Cc, R0, R1), dtype=torch.float32, device=keep.device) if n else _f(0, like=keep) for n in need_lr]
    ws = _f(lib.nvsr_planes_sr_batch_backward_workspace_floats(B, Cc, R0, R1, hid, nb, n_up, pad, None), like=keep)
    d_out_ptrs = (C.c_void_p * B)(*[t.data_ptr() for t in d_out])
    d_lr_ptrs = (C.c_void_p * B)(*[(t.data_ptr() if n else None) for t, n in zip(d_lr, need_lr)])
    sizes = [9 * hid * cin] + [9 * hid * hid] * (2 * nb + 1) + [9 * 4 * hid * hid] * n_up + [9 * cout * hid]       # floats per layer, state-dict order
    marks = bucket_sync.plan(sizes, gnat)            # [(first layer of the bucket, lo, hi, torch.cuda.Event)], last bucket of the blob first
    layers = (C.c_int32 * len(marks))(*[m[0] for m in marks])
    events = (C.c_void_p * len(marks))(*[m[3].cuda_event for m in marks])
    capi.call("nvsr_planes_sr_backward_batch_marks", B, Cc, R0, R1, capi.ptr(keep), capi.ptr(packed_dgrad), hid, nb, n_up, pad, over, rois_c, capi.ptr(std),
              d_out_ptrs, capi.ptr(gnat), d_lr_ptrs if any(need_lr) else None, capi.ptr(ws), arithmetic, int(bool(align_corners)), int(bool(bicubic)),
              len(marks), layers, events, capi.stream())
    bucket_sync.reduce_marked(gnat, marks)           # collectives behind their events; the current stream then waits for them and scales
    return [gnat] + d_lr


class PlanesSRBatchFn(torch.autograd.Function):
    """PlanesSR on the regions of interest of B planes as ONE autograd node: inputs (cfg, natural weights blob, lr_0 .. lr_{B-1}) -> B
    super-resolved planes; the backward returns ONE weight-gradient blob (torch's cat / reshape backward hands the slices to the
    convolutions' parameters once per iteration instead of once per plane) and the LR planes' gradients.
    cfg["bucket_sync"] (data-parallel training; distributed.OverlappedSRGradSync): the blob comes back ALREADY averaged over the ranks -- its
    buckets were all-reduced while the backward was still computing the layers in front of them."""

    @staticmethod
    def forward(ctx, cfg, natural, *lrs):
        res = direct.planes_sr_train_batch(list(lrs), cfg["packed"], cfg["geometry"], cfg["pad"], cfg["over"], cfg["rois"], cfg["mean"], cfg["std"],
                                           cfg["arithmetic"], cfg["align_corners"], cfg["bicubic"])
        outs, keep = res[:-1], res[-1]
        ctx.cfg = cfg
        ctx.shapes = [t.shape for t in lrs]
        ctx.save_for_backward(keep, cfg["packed_dgrad"])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *d_outs):
        keep, packed_dgrad = ctx.saved_tensors
        cfg = ctx.cfg
        need = ctx.needs_input_grad
        need_lr = [bool(n) for n in need[2:]]
        args = (keep, packed_dgrad, list(ctx.shapes[0][-3:]), cfg["geometry"], cfg["pad"], cfg["over"], cfg["rois"], cfg["std"],
                [capi.f32c(g) for g in d_outs], need_lr, cfg["arithmetic"], cfg["align_corners"], cfg["bicubic"])
        sync = cfg.get("bucket_sync")
        if sync is not None and need[1] and sync.active():
            res = planes_sr_backward_batch_marked(*args, sync)
        else:
            res = direct.planes_sr_backward_batch(*args)
        return (None, res[0] if need[1] else None) + tuple(g.reshape(sh) if n else None for g, sh, n in zip(res[1:], ctx.shapes, need_lr))


# operators whose forward is checked with torch.library.opcheck in the GPU tests
# =====================================================================================================================================
# cumprod_exclusive (nerf_helpers.py:409-430) -- differentiable like the reference's torch helper
# =====================================================================================================================================
@custom_op("nvsr::cumprod_exclusive", mutates_args=(), device_types="cuda")
def cumprod_exclusive(x: Tensor) -> Tensor:
    """out[..., 0] = 1, out[..., i] = x[..., 0] * ... * x[..., i-1]"""
    x = _c(x)
    out = torch.empty_like(x)
    if x.numel():
        capi.call("nvsr_cumprod_exclusive", x.numel() // x.shape[-1], x.shape[-1], capi.ptr(x), capi.ptr(out), capi.stream())
    return out


@cumprod_exclusive.register_fake
def _(x):
    return torch.empty_like(x, memory_format=torch.contiguous_format)


@custom_op("nvsr::cumprod_exclusive_backward", mutates_args=(), device_types="cuda")
def cumprod_exclusive_backward(x: Tensor, out: Tensor, g_out: Tensor) -> Tensor:
    x, out, g_out = _c(x), _c(out), _c(g_out)
    g = torch.empty_like(x)
    if x.numel():
        capi.call("nvsr_cumprod_exclusive_backward", x.numel() // x.shape[-1], x.shape[-1], capi.ptr(x), capi.ptr(out), capi.ptr(g_out), capi.ptr(g),
                  capi.stream())
    return g


@cumprod_exclusive_backward.register_fake
def _(x, out, g_out):
    return torch.empty_like(x, memory_format=torch.contiguous_format)


def _cumprod_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], output)


def _cumprod_bwd(ctx, g):
    x, out = ctx.saved_tensors
    return torch.ops.nvsr.cumprod_exclusive_backward(x, out, capi.f32c(g))


cumprod_exclusive.register_autograd(_cumprod_bwd, setup_context=_cumprod_setup)


FORWARD_OPS = ["plane_to_channel_last", "plane_from_channel_last", "pack_decoder", "coarse_z", "importance_resample", "triplane_decode",
               "triplane_decode_generic", "ray_points",
               "render_pass", "render_rays", "decode_rays", "composite", "composite_rays", "edsr", "planes_sr", "cumprod_exclusive"]


class _Direct:
    """The operators' Python bodies without the dispatcher round trip (torch.ops.nvsr.X(...) costs tens of microseconds of host time per call
    in schema matching and boxing; a 1.7 ms training step makes a dozen of them).  For callers that need none of what the dispatcher adds --
    train_utils._RenderRaysFn runs them inside an autograd.Function, where gradient mode is off and the tensors are already CUDA float32 --
    and only there: everything else goes through torch.ops.nvsr (fake implementations, registered autograd, opcheck)."""

    def __getattr__(self, name):
        op = globals()[name]
        fn = getattr(op, "_init_fn", op)          # (CustomOpDef keeps the decorated function; fall back to the operator itself)
        setattr(self, name, fn)
        return fn


direct = _Direct()
