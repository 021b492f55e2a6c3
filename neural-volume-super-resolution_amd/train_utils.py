"""Render orchestration -- mirror of the reference's train_utils.py (a3, a4, a5).

Same call surface as the reference (`run_one_iter_of_nerf` returns the 9-tuple, `eval_nerf` the 9-tuple of images); the
ray-chunk and point-chunk Python loops of the reference collapse into one `nvsr_render_rays` call per ray block: coarse
depths -> fused coarse pass -> inverse-CDF resampling + sort -> fused fine pass, all on the current stream."""
import ctypes as C

import torch

from . import capi
from . import models


def identity_encoding(x):
    return x


def _cfg(node, name, default=None):
    """attribute- or key-style access (CfgNode, SimpleNamespace, dict)"""
    if isinstance(node, dict):
        return node.get(name, default)
    return getattr(node, name, default)


MAX_RAYS_PER_LAUNCH = 1 << 22   # bounds the [N, 3*Nc+Nf] depth/weight workspace (5.4 GB at 64+128 samples)


def run_network(network_fn, pts, ray_batch, chunksize, embed_fn, embeddirs_fn, scene_id=None, **kw):
    """train_utils.py:15-64 for the planes model: points [..., 3] + the batch's view directions -> radiance field [..., 4].
    (chunksize is accepted for signature parity; the decode kernel walks the whole point list in one launch.)"""
    pts_shape = list(pts.shape)
    flat = embed_fn(pts.reshape((-1, pts_shape[-1])))
    if embeddirs_fn is not None:
        viewdirs = ray_batch[..., None, -3:]
        flat = torch.cat((flat, embeddirs_fn(viewdirs.expand(pts_shape).reshape((-1, 3)))), dim=-1)
    out = network_fn(flat)
    return out.reshape(pts_shape[:-1] + [out.shape[-1]])


def _reference_chunks(n_rays, options, mode, model_coarse, model_fine):
    """Ray-chunk partition of the reference (train_utils.py:228-235); only used to draw random numbers in its order."""
    chunk = _cfg(_cfg(options.nerf, mode), "chunksize")
    chunk = int(chunk / (model_coarse.num_density_planes / 3))
    if hasattr(model_fine, "SR_model"):
        chunk //= 10
    return [(i, min(i + chunk, n_rays)) for i in range(0, n_rays, chunk)]


def predict_and_render_radiance(ray_batch, model_coarse, model_fine, options, scene_id, mode="train", encode_position_fn=None,
                                encode_direction_fn=None, randoms=None):
    """train_utils.py:71-182 on packed rays [N,11].  `randoms` (extension) = dict(t_rand, u, noise_coarse, noise_fine) of
    explicit random inputs; when absent they are drawn on the CPU generator in the reference's order."""
    if _cfg(options.nerf, "encode_position_fn", None) == "mip":
        raise NotImplementedError("Mip-NeRF baseline is outside the tri-plane hot path (SURVEY.md 2)")
    m = _cfg(options.nerf, mode)
    Nc, Nf = int(m.num_coarse), int(m.num_fine)
    rays = capi.f32c(ray_batch)
    N = rays.shape[0]
    dev = rays.device
    std = float(m.radiance_field_noise_std)
    r = dict(randoms or {})
    if randoms is None:
        # same draws, same order as the reference: t_rand (train_utils.py:108), coarse noise (volume_rendering_utils.py:32),
        # u (nerf_helpers.py:683; only when perturb != 0), fine noise
        if m.perturb:
            r["t_rand"] = torch.rand([N, Nc])
        if std > 0.0:
            r["noise_coarse"] = torch.randn([N, Nc]) * std
        if Nf > 0:
            if m.perturb != 0.0:
                r["u"] = torch.rand([N, Nf])
            if std > 0.0:
                r["noise_fine"] = torch.randn([N, Nc + Nf]) * std
    t_rand, u, n_c, n_f = (None if r.get(k) is None else capi.f32c(r[k].to(dev)) for k in ("t_rand", "u", "noise_coarse", "noise_fine"))
    if not m.perturb:
        t_rand = None

    for mdl in (model_coarse, model_fine):
        mdl.set_cur_scene_id(scene_id)
    sc_c, keep_c = model_coarse.native_scene()
    packed_c = model_coarse.packed_decoder()
    packed_f = model_fine.packed_decoder() if Nf > 0 else None
    if Nf > 0:
        # both passes sample the same planes in the reference unless only the fine model super-resolves
        sc_f, keep_f = model_fine.native_scene()
        same = all(sc_c.planes[d] == sc_f.planes[d] for d in range(4))
    else:
        sc_f, keep_f, same = sc_c, keep_c, True

    rgb_c = torch.empty((N, 3), dtype=torch.float32, device=dev)
    disp_c, acc_c = (torch.empty(N, dtype=torch.float32, device=dev) for _ in range(2))
    rgb_f = disp_f = acc_f = None
    if Nf > 0:
        rgb_f = torch.empty((N, 3), dtype=torch.float32, device=dev)
        disp_f, acc_f = (torch.empty(N, dtype=torch.float32, device=dev) for _ in range(2))
    if N == 0:
        return rgb_c, disp_c, acc_c, rgb_f, disp_f, acc_f, None, None, None
    ws = torch.empty(capi.lib().nvsr_render_workspace_floats(N, Nc, Nf), dtype=torch.float32, device=dev)
    white = int(bool(m.white_background))
    lindisp = int(bool(m.lindisp))
    if same:
        capi.call("nvsr_render_rays", C.byref(sc_c), capi.ptr(packed_c), capi.ptr(packed_f), N, Nc, Nf, capi.ptr(rays), lindisp,
                  white, capi.ptr(t_rand), capi.ptr(u), capi.ptr(n_c), capi.ptr(n_f), capi.ptr(rgb_c), capi.ptr(disp_c),
                  capi.ptr(acc_c), capi.ptr(rgb_f), capi.ptr(disp_f), capi.ptr(acc_f), capi.ptr(ws), capi.stream())
    else:
        z_c, w_c, z_f = ws[:N * Nc], ws[N * Nc:2 * N * Nc], ws[2 * N * Nc:]
        st = capi.stream()
        capi.call("nvsr_coarse_z", N, Nc, capi.ptr(rays), lindisp, capi.ptr(t_rand), capi.ptr(z_c), st)
        capi.call("nvsr_render_pass", C.byref(sc_c), capi.ptr(packed_c), N, Nc, capi.ptr(rays), capi.ptr(z_c), capi.ptr(n_c), white,
                  capi.ptr(rgb_c), capi.ptr(disp_c), capi.ptr(acc_c), capi.ptr(w_c), None, st)
        capi.call("nvsr_importance_resample", N, Nc, Nf, capi.ptr(z_c), capi.ptr(w_c), capi.ptr(u), capi.ptr(z_f), st)
        capi.call("nvsr_render_pass", C.byref(sc_f), capi.ptr(packed_f), N, Nc + Nf, capi.ptr(rays), capi.ptr(z_f), capi.ptr(n_f),
                  white, capi.ptr(rgb_f), capi.ptr(disp_f), capi.ptr(acc_f), None, None, st)
    return rgb_c, disp_c, acc_c, rgb_f, disp_f, acc_f, None, None, None


def pack_rays(ray_origins, ray_directions, near, far, H=None, W=None, focal=None, no_ndc=True):
    """run_one_iter_of_nerf's ray packing (train_utils.py:207-226): rays [N,11] = [ro, rd, near, far, viewdir]."""
    ro, rd = capi.f32c(ray_origins).reshape(-1, 3), capi.f32c(ray_directions).reshape(-1, 3)
    N = ro.shape[0]
    view_src = rd
    if no_ndc is False:
        from .nerf_helpers import ndc_rays
        ro, rd = ndc_rays(H, W, focal, 1.0, ro, rd)
    rays = torch.empty((N, 11), dtype=torch.float32, device=ro.device)
    capi.call("nvsr_pack_rays", N, capi.ptr(ro), capi.ptr(rd), capi.ptr(view_src), float(near), float(far), capi.ptr(rays), capi.stream())
    return rays


def run_one_iter_of_nerf(H, W, focal, model_coarse, model_fine, batch_rays, options, scene_id, mode="train",
                         encode_position_fn=None, encode_direction_fn=None, scene_config={}, randoms=None):
    """train_utils.py:185-282 -> (rgb_coarse, disp_coarse, acc_coarse, rgb_fine, disp_fine, acc_fine, None, None, None)"""
    if not isinstance(model_coarse, models.TwoDimPlanesModel):
        raise NotImplementedError("only the tri-plane model is on the accelerated path")
    if not options.nerf.use_viewdirs:
        raise NotImplementedError("the decoder kernel expects use_viewdirs=True (all shipped configs)")
    rays = pack_rays(batch_rays[0], batch_rays[1], _cfg(scene_config, "near"), _cfg(scene_config, "far"), H, W, focal,
                     no_ndc=_cfg(scene_config, "no_ndc"))
    N = rays.shape[0]
    m = _cfg(options.nerf, mode)
    if randoms is None and (m.perturb or float(m.radiance_field_noise_std) > 0.0):
        # draw per reference ray chunk so that the CPU generator is consumed in the reference's order
        parts = []
        Nc, Nf, std = int(m.num_coarse), int(m.num_fine), float(m.radiance_field_noise_std)
        for a, b in _reference_chunks(N, options, mode, model_coarse, model_fine):
            n = b - a
            p = {}
            if m.perturb:
                p["t_rand"] = torch.rand([n, Nc])
            if std > 0.0:
                p["noise_coarse"] = torch.randn([n, Nc]) * std
            if Nf > 0 and m.perturb != 0.0:
                p["u"] = torch.rand([n, Nf])
            if Nf > 0 and std > 0.0:
                p["noise_fine"] = torch.randn([n, Nc + Nf]) * std
            parts.append(p)
        randoms = {k: torch.cat([p[k] for p in parts], 0) for k in parts[0]} if parts else {}
    outs = []
    for a in range(0, max(N, 1), MAX_RAYS_PER_LAUNCH):
        b = min(a + MAX_RAYS_PER_LAUNCH, N)
        sub = None if randoms is None else {k: v[a:b] for k, v in randoms.items()}
        outs.append(predict_and_render_radiance(rays[a:b], model_coarse, model_fine, options, scene_id, mode=mode, randoms=sub if sub is not None else {}))
    if len(outs) == 1:
        return outs[0]
    return tuple(None if outs[0][i] is None else torch.cat([o[i] for o in outs], 0) for i in range(9))


def eval_nerf(height, width, focal_length, model_coarse, model_fine, ray_origins, ray_directions, options, scene_id,
              mode="validation", encode_position_fn=None, encode_direction_fn=None, scene_config={}):
    """train_utils.py:285-331 -> (rgb_coarse[H,W,3], None, None, rgb_fine[H,W,3] | None, None, None, None, None, None)"""
    ray_origins = ray_origins.reshape((1, -1, 3))
    ray_directions = ray_directions.reshape((1, -1, 3))
    batch_rays = torch.cat((ray_origins, ray_directions), dim=0)
    rgb_coarse, _, _, rgb_fine, _, _, rgb_SR, _, _ = run_one_iter_of_nerf(
        height, width, focal_length, model_coarse, model_fine, batch_rays, options, mode="validation",
        encode_position_fn=encode_position_fn, encode_direction_fn=encode_direction_fn, scene_id=scene_id, scene_config=scene_config)
    rgb_coarse = rgb_coarse.reshape([height, width, -1])
    if rgb_fine is not None:
        rgb_fine = rgb_fine.reshape([height, width, -1])
    return rgb_coarse, None, None, rgb_fine, None, None, rgb_SR, None, None
