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


RECORD_RAYS = 8192   # rays per backward launch of the RECOMPUTING decoder-gradient path (bounds its record: 9.2 KB per point)
RECORD_FORWARD_MAX_POINTS = 1 << 21   # up to this many points of a pass (19 GB of record) the forward itself records the layer
                                      # inputs and the backward never recomputes; beyond, the chunked recomputing path runs


class _RenderRaysFn(torch.autograd.Function):
    """Differentiable predict_and_render_radiance.  Leaves: the four planes of the current scene and the decoder parameters of
    the coarse / fine model, the latter as flat blobs in state-dict order (`TwoDimPlanesModel.natural_blob(differentiable=True)`;
    torch's own cat/reshape backward hands the slices to the parameters).

    forward  = coarse z -> decode + composite (keeps raw + weights) -> importance resampling (no gradient, train_utils.py:153)
               -> fine decode + composite (keeps raw);
    backward = per pass: composite backward (wave per ray) -> decoder backward + plane scatter-add (MFMA + float atomics)
               [+ record of layer inputs / deltas -> weight-gradient contraction]."""

    @staticmethod
    def forward(ctx, cfg, p0, p1, p2, pv, nat_c, nat_f):
        capi_ = capi
        N, Nc, Nf, dev = cfg["N"], cfg["Nc"], cfg["Nf"], cfg["rays"].device
        rays, st = cfg["rays"], capi.stream()
        f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        z_c, w_c, raw_c = f(N, Nc), f(N, Nc), f(N, Nc, 4)
        rgb_c, disp_c, acc_c = f(N, 3), f(N), f(N)
        capi_.call("nvsr_coarse_z", N, Nc, capi.ptr(rays), cfg["lindisp"], capi.ptr(cfg["t_rand"]), capi.ptr(z_c), st)
        # training batches are a few thousand rays: the sample-parallel decoder + the wave-per-ray compositor fill the chip,
        # the fused per-ray kernel would run 32 workgroups; raw is needed by the backward anyway
        # the forward publishes its ReLU gates (128 B per point) so that the backward does not recompute it; when decoder gradients
        # are wanted it also records every layer's input (9.2 KB per point) unless the pass is too large for that
        def aux(S, dec_grad):
            fwd_rec = dec_grad and N * S <= RECORD_FORWARD_MAX_POINTS
            gates = torch.empty((N, S, 32), dtype=torch.int32, device=dev) if (fwd_rec or not dec_grad) else None
            rec = torch.empty(capi.lib().nvsr_decoder_record_floats(N, S), dtype=torch.float32, device=dev) if fwd_rec else None
            return gates, rec

        gates_c, rec_c = aux(Nc, cfg["dec_c_grad"] and cfg["coarse_grad"])
        capi_.call("nvsr_decode_rays_ex", C.byref(cfg["scene_c"]), capi.ptr(cfg["packed_c"]), N, Nc, capi.ptr(rays), capi.ptr(z_c), capi.ptr(raw_c),
                   capi.ptr(gates_c), capi.ptr(rec_c), st)
        capi_.call("nvsr_composite_rays", N, Nc, capi.ptr(raw_c), capi.ptr(z_c), capi.ptr(rays), capi.ptr(cfg["noise_c"]), cfg["white"],
                   capi.ptr(rgb_c), capi.ptr(disp_c), capi.ptr(acc_c), capi.ptr(w_c), None, st)
        outs = [rgb_c, disp_c, acc_c]
        saved = dict(z_c=z_c, raw_c=raw_c, gates_c=gates_c, rec_c=rec_c)
        if Nf > 0:
            z_f, raw_f = f(N, Nc + Nf), f(N, Nc + Nf, 4)
            rgb_f, disp_f, acc_f = f(N, 3), f(N), f(N)
            capi_.call("nvsr_importance_resample", N, Nc, Nf, capi.ptr(z_c), capi.ptr(w_c), capi.ptr(cfg["u"]), capi.ptr(z_f), st)
            gates_f, rec_f = aux(Nc + Nf, cfg["dec_f_grad"])
            capi_.call("nvsr_decode_rays_ex", C.byref(cfg["scene_f"]), capi.ptr(cfg["packed_f"]), N, Nc + Nf, capi.ptr(rays), capi.ptr(z_f),
                       capi.ptr(raw_f), capi.ptr(gates_f), capi.ptr(rec_f), st)
            capi_.call("nvsr_composite_rays", N, Nc + Nf, capi.ptr(raw_f), capi.ptr(z_f), capi.ptr(rays), capi.ptr(cfg["noise_f"]), cfg["white"],
                       capi.ptr(rgb_f), capi.ptr(disp_f), capi.ptr(acc_f), None, None, st)
            outs += [rgb_f, disp_f, acc_f]
            saved.update(z_f=z_f, raw_f=raw_f, gates_f=gates_f, rec_f=rec_f)
        ctx.cfg, ctx.saved = cfg, saved          # (ctx.saved is also what the parity tests read the fine depths from)
        ctx.mark_non_differentiable(*[o for i, o in enumerate(outs) if i % 3 == 1])    # disparity: no gradient path implemented
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        cfg, sv = ctx.cfg, ctx.saved
        N, Nc, Nf, rays, st = cfg["N"], cfg["Nc"], cfg["Nf"], cfg["rays"], capi.stream()
        dev = rays.device
        rd = rays[:, 3:6].contiguous()
        shapes = cfg["plane_shapes"]                                    # channel-last (H, W, C)
        need = ctx.needs_input_grad
        gplanes = [torch.zeros(sh, dtype=torch.float32, device=dev) if need[1 + d] else None for d, sh in enumerate(shapes)]
        gptrs = (C.c_void_p * 4)(*[None if g is None else g.data_ptr() for g in gplanes]) if any(need[1:5]) else None
        gdec_c = torch.zeros(capi.DECODER_NATURAL_FLOATS, dtype=torch.float32, device=dev) if need[5] else None
        gdec_f = torch.zeros(capi.DECODER_NATURAL_FLOATS, dtype=torch.float32, device=dev) if (need[6] and Nf > 0) else None

        def one_pass(S, z, raw, noise, scene, packed, packed_bwd, g_rgb, g_acc, gdec, gates, fwd_rec):
            if (g_rgb is None and g_acc is None) or (gptrs is None and gdec is None):
                return
            g_rgb = torch.zeros((N, 3), dtype=torch.float32, device=dev) if g_rgb is None else capi.f32c(g_rgb)
            g_acc = None if g_acc is None else capi.f32c(g_acc)
            g_raw = torch.empty((N, S, 4), dtype=torch.float32, device=dev)
            capi.call("nvsr_composite_backward", N, S, capi.ptr(raw), capi.ptr(z), capi.ptr(rd), capi.ptr(noise), cfg["white"],
                      capi.ptr(g_rgb), capi.ptr(g_acc), capi.ptr(g_raw), st)
            # per-point rows of the view-direction plane's gradient (summed per ray before they touch the plane)
            view_ws = torch.empty(N * S * capi.PLANE_CHANNELS, dtype=torch.float32, device=dev) if (gptrs is not None and need[4]) else None
            if gdec is not None and fwd_rec is not None:
                # the forward recorded the layer inputs: gate-driven backward adds the gradient half, then the contraction
                capi.call("nvsr_render_pass_backward_gates", C.byref(scene), capi.ptr(packed), capi.ptr(packed_bwd), N, S, capi.ptr(rays),
                          capi.ptr(z), capi.ptr(g_raw), capi.ptr(gates), gptrs, capi.ptr(view_ws), capi.ptr(fwd_rec), st)
                capi.call("nvsr_decoder_weight_grad", N, S, capi.ptr(fwd_rec), capi.ptr(gdec), st)
                return
            if gdec is None:
                if gates is not None:
                    capi.call("nvsr_render_pass_backward_gates", C.byref(scene), capi.ptr(packed), capi.ptr(packed_bwd), N, S, capi.ptr(rays),
                              capi.ptr(z), capi.ptr(g_raw), capi.ptr(gates), gptrs, capi.ptr(view_ws), None, st)
                else:
                    capi.call("nvsr_render_pass_backward_ex", C.byref(scene), capi.ptr(packed), capi.ptr(packed_bwd), N, S, capi.ptr(rays),
                              capi.ptr(z), capi.ptr(g_raw), gptrs, None, capi.ptr(view_ws), st)
                return
            step = min(N, RECORD_RAYS)
            record = torch.empty(capi.lib().nvsr_decoder_record_floats(step, S), dtype=torch.float32, device=dev)
            for a in range(0, N, step):
                n = min(step, N - a)
                capi.call("nvsr_render_pass_backward_ex", C.byref(scene), capi.ptr(packed), capi.ptr(packed_bwd), n, S, capi.ptr(rays[a:]),
                          capi.ptr(z[a:]), capi.ptr(g_raw[a:]), gptrs, capi.ptr(record), capi.ptr(view_ws), st)
                capi.call("nvsr_decoder_weight_grad", n, S, capi.ptr(record), capi.ptr(gdec), st)

        if cfg["coarse_grad"]:
            one_pass(Nc, sv["z_c"], sv["raw_c"], cfg["noise_c"], cfg["scene_c"], cfg["packed_c"], cfg["packed_bwd_c"], grads[0], grads[2], gdec_c,
                     sv["gates_c"], sv["rec_c"])
        if Nf > 0:
            one_pass(Nc + Nf, sv["z_f"], sv["raw_f"], cfg["noise_f"], cfg["scene_f"], cfg["packed_f"], cfg["packed_bwd_f"], grads[3], grads[5],
                     gdec_f, sv["gates_f"], sv["rec_f"])
        out = [None]
        for g, src in zip(gplanes, cfg["plane_leaves"]):
            out.append(None if g is None else models.from_channel_last(g, like=src))    # back to the reference's [1,C,H,W]
        return tuple(out) + (gdec_c, gdec_f)


def _planes_need_grad(model):
    if not torch.is_grad_enabled() or getattr(model, "planes_", None) is None:
        return False
    names = [models.get_plane_name(model.cur_id, d) for d in range(model.num_density_planes + 1)]
    return any(n in model.planes_ and model.planes_[n].requires_grad for n in names)


def _decoder_needs_grad(model):
    return torch.is_grad_enabled() and any(p.requires_grad for p in model.decoder_parameters())


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
    packed_c = model_coarse.packed_decoder()
    packed_f = model_fine.packed_decoder() if Nf > 0 else None
    top = model_fine if Nf > 0 else model_coarse
    dec_c_grad = _decoder_needs_grad(model_coarse)
    dec_f_grad = Nf > 0 and _decoder_needs_grad(model_fine)
    sr_on = hasattr(top, "SR_model") and not top.skip_SR_
    sr_grad = sr_on and torch.is_grad_enabled() and top.SR_model.training and \
        top.SR_model.inner_model.wants_grad(*top.SR_model.LR_planes.values())
    train_path = mode == "train" and N > 0 and (_planes_need_grad(top) or dec_c_grad or dec_f_grad or sr_grad)
    if not (train_path and sr_on):       # (the SR training path builds its scene from the ROI planes below)
        sc_c, keep_c = model_coarse.native_scene()
        if Nf > 0:
            # both passes sample the same planes in the reference unless only the fine model super-resolves
            sc_f, keep_f = model_fine.native_scene()
            same = all(sc_c.planes[d] == sc_f.planes[d] for d in range(4))
        else:
            sc_f, keep_f, same = sc_c, keep_c, True

    if train_path:
        # training path (mode == "train" only; evaluation never builds a graph): gradients flow to whatever requires grad among the
        # planes of the current scene, the decoder parameters of the two models and -- through the super-resolved planes -- the SR
        # network and its LR planes
        if sr_on:
            if Nf > 0 and not (hasattr(model_coarse, "SR_model") and not model_coarse.skip_SR_):
                raise NotImplementedError("training with only one of the two models super-resolving")
            leaves = top.training_planes(rays)
            keep_f = [models.to_channel_last(p.detach()) for p in leaves]
            sc_f, keep_f = top.native_scene(planes=keep_f)
            sc_c, keep_c = sc_f, keep_f
        else:
            names = [models.get_plane_name(scene_id, d) for d in range(4)]
            leaves = [top.planes_[n] for n in names]
        leaves += [model_coarse.natural_blob(differentiable=True) if dec_c_grad else None,
                   model_fine.natural_blob(differentiable=True) if dec_f_grad else None]
        coarse_grad = not isinstance(model_coarse.optional_no_grad(), torch.no_grad) if hasattr(model_coarse, "optional_no_grad") else True
        cfg = dict(N=N, Nc=Nc, Nf=Nf, rays=rays, lindisp=int(bool(m.lindisp)), white=int(bool(m.white_background)), t_rand=t_rand, u=u,
                   noise_c=n_c, noise_f=n_f, scene_c=sc_c, scene_f=sc_f, keep=(keep_c, keep_f), packed_c=packed_c, packed_f=packed_f,
                   packed_bwd_c=model_coarse.packed_decoder_bwd(), packed_bwd_f=model_fine.packed_decoder_bwd() if Nf > 0 else None,
                   plane_shapes=[tuple(k.shape) for k in keep_f], plane_leaves=leaves[:4], coarse_grad=coarse_grad, dec_c_grad=dec_c_grad,
                   dec_f_grad=dec_f_grad)
        outs = _RenderRaysFn.apply(cfg, *leaves)
        if Nf > 0:
            return outs[0], outs[1], outs[2], outs[3], outs[4], outs[5], None, None, None
        return outs[0], outs[1], outs[2], None, None, None, None, None, None

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
    if N == 0:
        return rays
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
