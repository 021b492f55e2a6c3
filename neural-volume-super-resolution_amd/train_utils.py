"""Render orchestration -- mirror of the reference's train_utils.py (a3, a4, a5).

Same call surface as the reference (`run_one_iter_of_nerf` returns the 9-tuple, `eval_nerf` the 9-tuple of images); the
ray-chunk and point-chunk Python loops of the reference collapse into one `nvsr_render_rays` call per ray block: coarse
depths -> fused coarse pass -> inverse-CDF resampling + sort -> fused fine pass, all on the current stream."""
import ctypes as C
import os

import torch

from . import capi
from . import models
from . import ops


def identity_encoding(x):
    return x


def _cfg(node, name, default=None):
    """attribute- or key-style access (CfgNode, SimpleNamespace, dict)"""
    if isinstance(node, dict):
        return node.get(name, default)
    return getattr(node, name, default)


MAX_RAYS_PER_LAUNCH = 1 << 22   # bounds the [N, 3*Nc+Nf] depth/weight workspace (5.4 GB at 64+128 samples)
MAX_RAYS_PER_LAUNCH_GENERIC = 1 << 17   # generic geometries materialise [N*S,6] points and [N*S,4] outputs per pass (0.25 GB at 192 samples)


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


def _draw_point_jitter(model, n_rays, S, chunksize):
    """The point_coords_noise draws of one pass over a ray chunk, in the reference's order: run_network cuts the chunk's n_rays * S points
    into network batches of `chunksize` points (train_utils.py:47-58) and every model call draws torch.normal of its batch's shape from the
    CPU generator (models.py:291-293).  -> [n_rays, S, 3] or None when the model does not jitter right now."""
    std = model.jitter_std()
    if not std:
        return None
    n = n_rays * S
    step = max(1, n if chunksize is None else int(chunksize))
    parts = [torch.normal(mean=0, std=std, size=[min(step, n - i), 3]) for i in range(0, n, step)]
    return torch.cat(parts, 0).reshape(n_rays, S, 3) if parts else torch.empty(0, S, 3)


def _reference_chunks(n_rays, options, mode, model_coarse, model_fine):
    """Ray-chunk partition of the reference (train_utils.py:228-235); only used to draw random numbers in its order."""
    chunk = _cfg(_cfg(options.nerf, mode), "chunksize")
    chunk = int(chunk / (model_coarse.num_density_planes / 3))
    if hasattr(model_fine, "SR_model"):
        chunk //= 10
    return [(i, min(i + chunk, n_rays)) for i in range(0, n_rays, chunk)]


from .ops import RECORD_RAYS, RECORD_FORWARD_MAX_POINTS  # noqa: E402,F401  (tests and tools tune them through this module)


def _record_limits():
    """(RECORD_RAYS, RECORD_FORWARD_MAX_POINTS) as currently set on THIS module (tests lower them to force the recomputing path)"""
    import sys
    me = sys.modules[__name__]
    ops.RECORD_RAYS = me.RECORD_RAYS
    return me.RECORD_RAYS, me.RECORD_FORWARD_MAX_POINTS


def _NV():
    """the operator namespace _RenderRaysFn calls: the operators' bodies (ops.direct: no dispatcher round trip inside the Function);
    NVSR_OPS_DISPATCH=1 routes the calls through torch.ops.nvsr instead (host-time A/B)"""
    return torch.ops.nvsr if os.environ.get("NVSR_OPS_DISPATCH") == "1" else ops.direct


class _RenderRaysFn(torch.autograd.Function):
    """Differentiable predict_and_render_radiance, written against torch.ops.nvsr.*.  Leaves: the four planes of the current scene and
    the decoder parameters of the coarse / fine model, the latter as flat blobs in state-dict order
    (`TwoDimPlanesModel.natural_blob(differentiable=True)`; torch's own cat/reshape backward hands the slices to the parameters).

    forward  = coarse_z -> decode_rays + composite_rays (keeps raw + weights) -> importance_resample (no gradient, train_utils.py:153)
               -> fine decode_rays + composite_rays (keeps raw);
    backward = per pass: composite_backward (wave per ray) -> decode_rays_backward (MFMA + plane scatter)
               [+ record of layer inputs / deltas -> decoder_weight_grad].
    The arithmetic the forward ran in is stored with the context and handed to every backward operator."""

    @staticmethod
    def forward(ctx, cfg, p0, p1, p2, pv, nat_c, nat_f, *coarse_leaves):
        # coarse_leaves: the four planes the COARSE pass samples when they are other tensors than the fine pass's p0..pv -- the reference's default
        # SR training (train_nerf.py:554-561, apply_2_coarse False): only the fine model super-resolves, the coarse model samples the LR planes
        N, Nc, Nf, rays = cfg["N"], cfg["Nc"], cfg["Nf"], cfg["rays"]
        nv = _NV()
        arith_c, arith_f = cfg["arith_c"], cfg["arith_f"]
        # an output the loss does not use hands None to backward() instead of a zero tensor: the disparity / opacity chain rules are
        # skipped for them (a dozen per-ray kernels per pass otherwise)
        ctx.set_materialize_grads(False)
        _, fwd_max = _record_limits()
        z_c = nv.coarse_z(rays, Nc, bool(cfg["lindisp"]), cfg["t_rand"])
        # training batches are a few thousand rays: the sample-parallel decoder + the wave-per-ray compositor fill the chip, the fused
        # per-ray kernel would run 32 workgroups; raw is needed by the backward anyway.  The forward publishes its ReLU gates (128 B per
        # point) so that the backward does not recompute it; when decoder gradients are wanted it also records every layer's input
        # (9.2 KB per point) unless the pass is too large for that
        def flags(S, dec_grad):
            fwd_rec = dec_grad and N * S <= fwd_max
            return (fwd_rec or not dec_grad), fwd_rec

        def none_if_empty(t):
            return t if t.numel() else None

        want_g, want_r = flags(Nc, cfg["dec_c_grad"] and cfg["coarse_grad"])
        raw_c, gates_c, rec_c = nv.decode_rays(cfg["planes_c"], cfg["consts"], cfg["packed_c"], rays, z_c, want_g, want_r, arith_c)
        rgb_c, disp_c, acc_c, w_c = nv.composite_rays(raw_c, z_c, rays, cfg["noise_c"], bool(cfg["white"]), True)
        outs = [rgb_c, disp_c, acc_c]
        saved = dict(z_c=z_c, raw_c=raw_c, gates_c=none_if_empty(gates_c), rec_c=none_if_empty(rec_c))
        if Nf > 0:
            z_f = nv.importance_resample(z_c, w_c, Nf, cfg["u"])
            want_g, want_r = flags(Nc + Nf, cfg["dec_f_grad"])
            raw_f, gates_f, rec_f = nv.decode_rays(cfg["planes_f"], cfg["consts"], cfg["packed_f"], rays, z_f, want_g, want_r, arith_f)
            rgb_f, disp_f, acc_f, _ = nv.composite_rays(raw_f, z_f, rays, cfg["noise_f"], bool(cfg["white"]), False)
            outs += [rgb_f, disp_f, acc_f]
            saved.update(z_f=z_f, raw_f=raw_f, gates_f=none_if_empty(gates_f), rec_f=none_if_empty(rec_f))
        ctx.cfg, ctx.saved = cfg, saved          # (ctx.saved is also what the parity tests read the fine depths from)
        # disp_map is differentiable like the reference's (volume_rendering_utils.py:46): its gradient is folded into those of depth / acc
        # (detached views: the outputs themselves would close a reference cycle ctx -> saved -> output -> grad_fn -> ctx, and every step's
        #  7 GB of forward record would wait for the cyclic garbage collector)
        saved.update(disp_c=disp_c.detach(), acc_c=acc_c.detach())
        if Nf > 0:
            saved.update(disp_f=disp_f.detach(), acc_f=acc_f.detach())
        # autograd's version check: everything the backward reads -- the forward's intermediates and the channel-last planes (views of the
        # plane parameters, sharing their version counters) -- also goes through save_for_backward, so an in-place update of a plane or of a
        # saved tensor between forward and backward raises instead of producing the gradient of another function
        guarded = [t for t in list(saved.values()) + list(cfg["planes_c"]) + list(cfg["planes_f"]) + [rays, cfg["packed_c"], cfg["packed_f"]]
                   if isinstance(t, torch.Tensor)]
        ctx.save_for_backward(*guarded)
        if not cfg["coarse_grad"] and Nf > 0:
            ctx.mark_non_differentiable(rgb_c, disp_c, acc_c)     # the coarse pass ran under the model's optional_no_grad (train_nerf.py:560)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        ctx.saved_tensors          # (raises if a tensor saved by the forward was modified in place since)
        cfg, sv = ctx.cfg, ctx.saved
        N, Nc, Nf, rays = cfg["N"], cfg["Nc"], cfg["Nf"], cfg["rays"]
        nv = _NV()
        dev = rays.device
        need = ctx.needs_input_grad
        sep = bool(cfg.get("separate_coarse"))                 # the coarse pass has plane leaves of its own (inputs 7..10)
        need_planes_f = [bool(n) for n in need[1:5]]
        need_planes_c = [bool(n) for n in need[7:11]] if sep else need_planes_f
        gplanes_f = [None, None, None, None]
        gplanes_c = [None, None, None, None] if sep else gplanes_f

        def one_pass(S, z, raw, noise, planes, packed, packed_bwd, g_rgb, g_disp, g_acc, disp, acc, want_dec, gates, fwd_rec, arith, need_planes, gplanes):
            """-> decoder gradient of this pass (state-dict order) or None; plane gradients are added into `gplanes`"""
            def add_planes(gs):
                for d, g in enumerate(gs):
                    if need_planes[d]:
                        gplanes[d] = g if gplanes[d] is None else gplanes[d].add_(g)

            if (g_rgb is None and g_acc is None and g_disp is None) or (not any(need_planes) and not want_dec):
                return None
            g_rgb = torch.zeros((N, 3), dtype=torch.float32, device=dev) if g_rgb is None else capi.f32c(g_rgb)
            g_acc = None if g_acc is None else capi.f32c(g_acc)
            g_depth = None
            if g_disp is not None:
                q = 1.0 / disp                      # disp = 1 / max(1e-10, q): q itself wherever the gradient is not zero (NaN stays NaN)
                g_acc, g_depth = ops.fold_disp_grad(g_disp, q, acc, q * acc, g_acc, None)
            g_raw = nv.composite_backward_rays(raw, z, rays, noise, bool(cfg["white"]), False, g_rgb, g_acc, g_depth)     # (directions read from the packed rays)
            if gates is not None and (not want_dec or fwd_rec is not None):
                # gate-driven backward (no recomputation); with the forward's record it adds the gradient half, then ONE contraction
                rec = fwd_rec if want_dec else None
                have = [gplanes[d] is not None for d in range(4) if need_planes[d]]
                if have and not any(have):
                    # first pass of the step: one zero-filled allocation holds all gradient planes
                    for d, g in enumerate(ops.zero_planes_like(planes, need_planes, rays)):
                        if need_planes[d]:
                            gplanes[d] = g
                    have = [True] * len(have)
                if have and all(have) and all(gplanes[d].shape == planes[d].shape and gplanes[d].stride() == planes[d].stride()
                                              for d in range(4) if need_planes[d]):
                    # a second pass over the same planes scatters into the first pass's gradient planes (no second zero-fill, no add)
                    nv.decode_rays_backward_(planes, cfg["consts"], packed, packed_bwd, rays, z, g_raw, gates, rec, need_planes, arith,
                                             [g if g is not None else rays.new_empty((0,)) for g in gplanes])
                else:
                    add_planes(nv.decode_rays_backward(planes, cfg["consts"], packed, packed_bwd, rays, z, g_raw, gates, rec, need_planes, arith))
                return nv.decoder_weight_grad(fwd_rec, N, S, arith) if want_dec else None
            # no gates published (pass too large for a forward record): recompute the forward in the backward, RECORD_RAYS rays at a time
            _record_limits()
            out = nv.decode_rays_backward_recompute(planes, cfg["consts"], packed, packed_bwd, rays, z, g_raw, need_planes, want_dec, arith)
            add_planes(out[:4])
            return out[4] if want_dec else None

        gdec_c = gdec_f = None
        coarse = lambda: one_pass(Nc, sv["z_c"], sv["raw_c"], cfg["noise_c"], cfg["planes_c"], cfg["packed_c"], cfg["packed_bwd_c"], grads[0], grads[1],
                                  grads[2], sv["disp_c"], sv["acc_c"], bool(need[5]), sv["gates_c"], sv["rec_c"], cfg["arith_c"], need_planes_c, gplanes_c)
        fine = lambda: one_pass(Nc + Nf, sv["z_f"], sv["raw_f"], cfg["noise_f"], cfg["planes_f"], cfg["packed_f"], cfg["packed_bwd_f"], grads[3], grads[4],
                                grads[5], sv["disp_f"], sv["acc_f"], bool(need[6]), sv["gates_f"], sv["rec_f"], cfg["arith_f"], need_planes_f, gplanes_f)
        # The two passes' backward kernels are independent (both ADD into the gradient planes with float atomics): the coarse pass runs on a
        # second stream, so that its workgroups fill the fine pass's partly empty rounds (same-box A/B of the planes-only iteration: eager
        # 1.683 -> 1.631 ms, replayed from a graph 1.728 -> 1.709 ms; NVSR_BWD_STREAMS=0 keeps one stream).  Planes-only passes: nothing
        # allocated on the second stream outlives the join.
        need_planes, gplanes = need_planes_f, gplanes_f
        two = (os.environ.get("NVSR_BWD_STREAMS", "1") == "1" and not sep and cfg["coarse_grad"] and Nf > 0 and dev.type == "cuda" and any(need_planes)
               and not need[5] and not need[6] and sv["gates_c"] is not None and sv["gates_f"] is not None
               and all(a.shape == b.shape and a.stride() == b.stride() for a, b in zip(cfg["planes_c"], cfg["planes_f"])))
        if two:
            for d, g in enumerate(ops.zero_planes_like(cfg["planes_f"], need_planes, rays)):      # zero-filled before the fork
                if need_planes[d]:
                    gplanes[d] = g
            cur, side = torch.cuda.current_stream(dev), _side_stream(dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                gdec_c = coarse()
            gdec_f = fine()
            cur.wait_stream(side)
        else:
            if cfg["coarse_grad"]:
                gdec_c = coarse()
            if Nf > 0:
                gdec_f = fine()
        out = [None]
        for d, src in enumerate(cfg["plane_leaves"]):
            g = gplanes[d]
            if need_planes[d] and g is None:
                g = torch.zeros(cfg["plane_shapes"][d], dtype=torch.float32, device=dev)
            out.append(None if g is None else models.from_channel_last(g, like=src))    # back to the reference's [1,C,H,W]
        if need[5] and gdec_c is None:
            gdec_c = torch.zeros(capi.DECODER_NATURAL_FLOATS, dtype=torch.float32, device=dev)
        if need[6] and gdec_f is None and Nf > 0:
            gdec_f = torch.zeros(capi.DECODER_NATURAL_FLOATS, dtype=torch.float32, device=dev)
        out_c = []
        if sep:
            for d, src in enumerate(cfg["plane_leaves_c"]):
                g = gplanes_c[d]
                if need_planes_c[d] and g is None:
                    g = torch.zeros(cfg["plane_shapes_c"][d], dtype=torch.float32, device=dev)
                out_c.append(None if g is None else models.from_channel_last(g, like=src))
        return tuple(out) + (gdec_c, gdec_f) + tuple(out_c)


_SIDE_STREAMS = {}


def _side_stream(dev):
    k = (dev.type, dev.index)
    if k not in _SIDE_STREAMS:
        _SIDE_STREAMS[k] = torch.cuda.Stream(device=dev)
    return _SIDE_STREAMS[k]


def _planes_need_grad(model):
    if not torch.is_grad_enabled() or getattr(model, "planes_", None) is None:
        return False
    names = [models.get_plane_name(model.cur_id, d) for d in range(model.num_density_planes + 1)]
    return any(n in model.planes_ and model.planes_[n].requires_grad for n in names)


def _decoder_needs_grad(model):
    return torch.is_grad_enabled() and any(p.requires_grad for p in model.decoder_parameters())


def predict_and_render_radiance(ray_batch, model_coarse, model_fine, options, scene_id, mode="train", encode_position_fn=None,
                                encode_direction_fn=None, randoms=None, force_arith=None):
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
        for mdl in (model_coarse, model_fine):
            mdl.set_cur_scene_id(scene_id)
        if m.perturb:
            r["t_rand"] = torch.rand([N, Nc])
        r["jitter_coarse"] = _draw_point_jitter(model_coarse, N, Nc, m.chunksize)
        if std > 0.0:
            r["noise_coarse"] = torch.randn([N, Nc]) * std
        if Nf > 0:
            if m.perturb != 0.0:
                r["u"] = torch.rand([N, Nf])
            r["jitter_fine"] = _draw_point_jitter(model_fine, N, Nc + Nf, m.chunksize)
            if std > 0.0:
                r["noise_fine"] = torch.randn([N, Nc + Nf]) * std
    t_rand, u, n_c, n_f = (None if r.get(k) is None else capi.f32c(r[k].to(dev)) for k in ("t_rand", "u", "noise_coarse", "noise_fine"))
    jit_c, jit_f = (None if r.get(k) is None else capi.f32c(r[k].to(dev)).reshape(-1, 3) for k in ("jitter_coarse", "jitter_fine"))
    if not m.perturb:
        t_rand = None

    for mdl in (model_coarse, model_fine):
        mdl.set_cur_scene_id(scene_id)
    if not (model_coarse.is_native_geometry() and (Nf <= 0 or model_fine.is_native_geometry())):
        return _render_generic(rays, model_coarse, model_fine, m, Nc, Nf, t_rand, u, n_c, n_f, jit_c, jit_f)
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
        planes_c, consts = model_coarse.scene_args()
        if Nf > 0:
            # both passes sample the same planes in the reference unless only the fine model super-resolves
            planes_f, consts_f = model_fine.scene_args()
            same = all(a.data_ptr() == b.data_ptr() for a, b in zip(planes_c, planes_f))
        else:
            planes_f, same = planes_c, True
    # (an evaluation render whose operands leave the f16 limbs' range is found by the library's range flag and rendered again with
    #  force_arith = bf16x3: run_one_iter_of_nerf; a training step raises: training.TrainStep)
    arith_c = capi.resolve_decoder_arithmetic(model_coarse.arithmetic if force_arith is None else force_arith)
    arith_f = capi.resolve_decoder_arithmetic(model_fine.arithmetic if force_arith is None else force_arith) if Nf > 0 else arith_c
    white, lindisp = bool(m.white_background), bool(m.lindisp)

    if train_path:
        # training path (mode == "train" only; evaluation never builds a graph): gradients flow to whatever requires grad among the
        # planes of the current scene, the decoder parameters of the two models and -- through the super-resolved planes -- the SR
        # network and its LR planes
        leaves_c = []
        if sr_on:
            leaves = top.training_planes(rays)
            planes_f, consts = top.scene_args(planes=[models.to_channel_last(p.detach()) for p in leaves])
            planes_c = planes_f
            if Nf > 0 and not (hasattr(model_coarse, "SR_model") and not model_coarse.skip_SR_):
                # the reference's default (train_nerf.py:554-561, super_resolution.apply_2_coarse False): only the fine model super-resolves; the
                # coarse pass samples the LR planes -- plane leaves of its own
                leaves_c = model_coarse.training_planes(rays)
                planes_c, _ = model_coarse.scene_args(planes=[models.to_channel_last(p.detach()) for p in leaves_c])
        else:
            names = [models.get_plane_name(scene_id, d) for d in range(4)]
            leaves = [top.planes_[n] for n in names]
        leaves += [model_coarse.natural_blob(differentiable=True) if dec_c_grad else None,
                   model_fine.natural_blob(differentiable=True) if dec_f_grad else None]
        leaves += leaves_c
        coarse_grad = not isinstance(model_coarse.optional_no_grad(), torch.no_grad) if hasattr(model_coarse, "optional_no_grad") else True
        # 'f16x2': the library runs the forward of a pass whose decoder is not trained (no weight-gradient record) and every gate-driven
        # backward on 2 f16 limbs, the recording forward and the weight-gradient contraction on 3 bf16 limbs (include/nvsr.h); the gates a
        # forward publishes are signs, valid for either backward
        cfg = dict(N=N, Nc=Nc, Nf=Nf, rays=rays, lindisp=lindisp, white=white, t_rand=t_rand, u=u, noise_c=n_c, noise_f=n_f,
                   planes_c=planes_c, planes_f=planes_f, consts=consts, packed_c=packed_c, packed_f=packed_f,
                   packed_bwd_c=model_coarse.packed_decoder_bwd(), packed_bwd_f=model_fine.packed_decoder_bwd() if Nf > 0 else None,
                   plane_shapes=[tuple(k.shape) for k in planes_f], plane_leaves=leaves[:4], separate_coarse=bool(leaves_c),
                   plane_shapes_c=[tuple(k.shape) for k in planes_c], plane_leaves_c=leaves_c, coarse_grad=coarse_grad, dec_c_grad=dec_c_grad,
                   dec_f_grad=dec_f_grad, arith_c=arith_c, arith_f=arith_f)
        outs = _RenderRaysFn.apply(cfg, *leaves)
        if Nf > 0:
            return outs[0], outs[1], outs[2], outs[3], outs[4], outs[5], None, None, None
        return outs[0], outs[1], outs[2], None, None, None, None, None, None

    nv = torch.ops.nvsr
    if N == 0:
        e = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
        return (e(0, 3), e(0), e(0)) + ((e(0, 3), e(0), e(0)) if Nf > 0 else (None, None, None)) + (None, None, None)
    if same and arith_c == arith_f:
        rgb_c, disp_c, acc_c, rgb_f, disp_f, acc_f = nv.render_rays(planes_c, consts, packed_c, packed_f, rays, Nc, Nf, lindisp, white, t_rand, u,
                                                                   n_c, n_f, arith_c)
        if Nf <= 0:
            rgb_f = disp_f = acc_f = None
    else:
        # the two passes sample different planes (only the fine model super-resolves) or run in different arithmetic: pass by pass
        fused = N >= capi.fused_min_rays()     # below it the sample-parallel decoder + the wave-per-ray compositor fill the chip

        def one_pass(planes, packed, z, noise, want_w, arith):
            if fused:
                return nv.render_pass(planes, consts, packed, rays, z, noise, white, want_w, arith)
            raw, _, _ = nv.decode_rays(planes, consts, packed, rays, z, False, False, arith)
            return nv.composite_rays(raw, z, rays, noise, white, want_w)

        z_c = nv.coarse_z(rays, Nc, lindisp, t_rand)
        rgb_c, disp_c, acc_c, w_c = one_pass(planes_c, packed_c, z_c, n_c, Nf > 0, arith_c)
        rgb_f = disp_f = acc_f = None
        if Nf > 0:
            z_f = nv.importance_resample(z_c, w_c, Nf, u)
            rgb_f, disp_f, acc_f, _ = one_pass(planes_f, packed_f, z_f, n_f, False, arith_f)
    return rgb_c, disp_c, acc_c, rgb_f, disp_f, acc_f, None, None, None


def _render_generic(rays, model_coarse, model_fine, m, Nc, Nf, t_rand, u, n_c, n_f, jit_c=None, jit_f=None):
    """predict_and_render_radiance for decoder geometries other than the shipped one, pass by pass like the reference (train_utils.py:95-180):
    depths -> run_network (the model's generic kernels on the [N*S,6] point list) -> compositing -> importance resampling -> again.
    With gradients enabled the model call and the compositing are the differentiable operators (the importance samples are detached,
    train_utils.py:153)."""
    nv = torch.ops.nvsr
    N = rays.shape[0]
    white, lindisp = bool(m.white_background), bool(m.lindisp)
    if N == 0:
        e = lambda *sh: torch.empty(sh, dtype=torch.float32, device=rays.device)
        return (e(0, 3), e(0), e(0)) + ((e(0, 3), e(0), e(0)) if Nf > 0 else (None, None, None)) + (None, None, None)
    def composite(raw, z, noise, want_weights):
        if raw.requires_grad:
            rgb, disp, acc, w, _ = nv.composite(raw, z, rays[:, 3:6].contiguous(), noise, white, False)
            return rgb, disp, acc, w.detach()
        return nv.composite_rays(raw, z, rays, noise, white, want_weights)

    z_c = nv.coarse_z(rays, Nc, lindisp, t_rand)
    raw = model_coarse(nv.ray_points(rays, z_c), coord_noise=jit_c).reshape(N, Nc, 4)
    rgb_c, disp_c, acc_c, w_c = composite(raw, z_c, n_c, Nf > 0)
    rgb_f = disp_f = acc_f = None
    if Nf > 0:
        z_f = nv.importance_resample(z_c, w_c, Nf, u)
        raw_f = model_fine(nv.ray_points(rays, z_f), coord_noise=jit_f).reshape(N, Nc + Nf, 4)
        rgb_f, disp_f, acc_f, _ = composite(raw_f, z_f, n_f, False)
    return rgb_c, disp_c, acc_c, rgb_f, disp_f, acc_f, None, None, None


def _sr_of(model):
    return model.SR_model if hasattr(model, "SR_model") and not model.skip_SR_ else None


def _f16_in_play(*ms):
    """does an evaluation of these models launch anything in NVSR_ARITH_F16X2?"""
    f16 = capi.ARITHMETIC["f16x2"]
    for m in ms:
        if capi.resolve_decoder_arithmetic(m.arithmetic) == f16:
            return True
        sr = _sr_of(m)
        if sr is not None and capi.resolve_conv_arithmetic(sr.inner_model.arithmetic) == f16:
            return True
    return False


def _range_key(*ms):
    """(storage, version) of everything an evaluation reads: decoder parameters, planes, SR weights"""
    key = []
    for m in ms:
        ts = list(m.decoder_parameters()) + list((getattr(m, "planes_", None) or {}).values())
        sr = _sr_of(m)
        if sr is not None:
            ts += list(sr.inner_model.parameters())
        key += [(t.data_ptr(), t._version) for t in ts]
    return tuple(key)


def _sr_fallback(*ms, redo=False):
    """the SR networks of these models super-resolve in 'bf16x3' for the evaluation frame being rendered.  Cached SR planes are dropped when `redo`
    says so (they come from an F16X2 pass that raised the range flag: NaN inside) or when they were NOT made in 'bf16x3' (`_planes_arith`, set by
    _sr_restore behind a fallback frame, forgotten by clear_SR_planes): planes a fallback frame has super-resolved stay cached for the next frames of
    the same parameters (the arithmetic was put back after every frame, so every later frame found 'f16x2', dropped the planes and ran the
    SR stage again, ~80 ms per scene and frame).  -> [(PlanesSR, its network's arithmetic before)]: the caller puts it back after the frame
    (_sr_restore, in a `finally`), so that a later training iteration of the same SR model runs in the arithmetic it was configured with"""
    changed, seen = [], set()
    for m in ms:
        sr = _sr_of(m)
        if sr is None or id(sr) in seen:
            continue
        seen.add(id(sr))
        if capi.resolve_conv_arithmetic(sr.inner_model.arithmetic) == capi.ARITHMETIC["f16x2"]:
            changed.append((sr, sr.inner_model.arithmetic))
            sr.inner_model.arithmetic = "bf16x3"
            if redo or sr.__dict__.get("_planes_arith") != "bf16x3":
                sr.clear_SR_planes()
        elif redo:
            sr.clear_SR_planes()
    return changed


def _sr_restore(changed, rendered=True):
    """puts the SR networks' arithmetic back; rendered: the frame completed, the planes now cached were made in 'bf16x3'"""
    for sr, before in changed:
        sr.inner_model.arithmetic = before
        if rendered:
            sr.__dict__["_planes_arith"] = "bf16x3"


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


_PATCH_ORDER = {}
PATCH_W, PATCH_H = 16, 2                 # 32 rays = one wave tile of the fused pass
SUPER_W, SUPER_H = 8, 16                 # patches per super-block: 128 x 32 pixels, visited block by block


def __getattr__(name):
    # PATCH_ORDER_MIN_RAYS = the library's fused-path threshold (the fused passes tile by ray; below it the sample-parallel kernels do not),
    # read from the library when first used so that the two can never disagree
    if name == "PATCH_ORDER_MIN_RAYS":
        return capi.fused_min_rays()
    raise AttributeError("module %r has no attribute %r" % (__name__, name))



def patch_order(n_rays, grid_width, device):
    """(perm, inv) that reorder the rays of a row-major pixel grid [n_rays / grid_width, grid_width] into 16 x 2 pixel patches (row-major
    inside a patch), the patches visited super-block by super-block (8 x 16 patches = 128 x 32 pixels, row-major inside and between
    blocks).  A wave tile of the fused render pass is 32 consecutive rays: as a patch they are half as far apart as 32 pixels of a row,
    their samples share more texels (fewer distinct cache lines per gather instruction, more L1 / L2 hits), and the workgroups that run
    side by side stay inside one compact block; the frame renders 3-4 % faster than in row order -- the pixels are the same bits, every
    ray is independent of its neighbours (tools/ray_order_time.py: 8 x 4 and 16 x 2 patches are within 1 % of each other, the blocks add
    0.5 %).  Ragged edges are fine: any permutation is valid."""
    key = (int(n_rays), int(grid_width), str(device))
    hit = _PATCH_ORDER.get(key)
    if hit is None:
        rows = n_rays // grid_width
        ys = torch.arange(rows, device=device)[:, None]
        xs = torch.arange(grid_width, device=device)[None, :]
        bw, bh = PATCH_W * SUPER_W, PATCH_H * SUPER_H
        k = ((((ys // bh) * ((grid_width + bw - 1) // bw) + xs // bw) * SUPER_H + (ys // PATCH_H) % SUPER_H) * SUPER_W + (xs // PATCH_W) % SUPER_W) \
            * (PATCH_W * PATCH_H) + (ys % PATCH_H) * PATCH_W + xs % PATCH_W
        perm = torch.argsort(k.reshape(-1))
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(perm.numel(), device=device)
        if len(_PATCH_ORDER) > 16:
            _PATCH_ORDER.clear()
        hit = _PATCH_ORDER[key] = (perm, inv)
    return hit


def run_one_iter_of_nerf(H, W, focal, model_coarse, model_fine, batch_rays, options, scene_id, mode="train",
                         encode_position_fn=None, encode_direction_fn=None, scene_config={}, randoms=None, ray_grid_width=None):
    """train_utils.py:185-282 -> (rgb_coarse, disp_coarse, acc_coarse, rgb_fine, disp_fine, acc_fine, None, None, None)
    ray_grid_width (not in the reference): the rays are whole rows of a row-major pixel grid of this width (eval_nerf, the row-sharded
    renders) -- an evaluation pass then renders them in patch order (patch_order) and returns the results in the caller's order."""
    if not isinstance(model_coarse, models.TwoDimPlanesModel):
        raise NotImplementedError("only the tri-plane model is on the accelerated path")
    if not options.nerf.use_viewdirs:
        raise NotImplementedError("the decoder kernel expects use_viewdirs=True (all shipped configs)")
    rays = pack_rays(batch_rays[0], batch_rays[1], _cfg(scene_config, "near"), _cfg(scene_config, "far"), H, W, focal,
                     no_ndc=_cfg(scene_config, "no_ndc"))
    N = rays.shape[0]
    m = _cfg(options.nerf, mode)
    for mdl in (model_coarse, model_fine):
        mdl.set_cur_scene_id(scene_id)
    jitters = bool(model_coarse.jitter_std() or (int(m.num_fine) > 0 and model_fine.jitter_std()))
    if randoms is None and (m.perturb or float(m.radiance_field_noise_std) > 0.0 or jitters):
        # draw per reference ray chunk so that the CPU generator is consumed in the reference's order
        parts = []
        Nc, Nf, std = int(m.num_coarse), int(m.num_fine), float(m.radiance_field_noise_std)
        for a, b in _reference_chunks(N, options, mode, model_coarse, model_fine):
            n = b - a
            p = {}
            if m.perturb:
                p["t_rand"] = torch.rand([n, Nc])
            if model_coarse.jitter_std():
                p["jitter_coarse"] = _draw_point_jitter(model_coarse, n, Nc, m.chunksize)
            if std > 0.0:
                p["noise_coarse"] = torch.randn([n, Nc]) * std
            if Nf > 0 and m.perturb != 0.0:
                p["u"] = torch.rand([n, Nf])
            if Nf > 0 and model_fine.jitter_std():
                p["jitter_fine"] = _draw_point_jitter(model_fine, n, Nc + Nf, m.chunksize)
            if Nf > 0 and std > 0.0:
                p["noise_fine"] = torch.randn([n, Nc + Nf]) * std
            parts.append(p)
        randoms = {k: torch.cat([p[k] for p in parts], 0) for k in parts[0]} if parts else {}
    outs = []
    native = model_coarse.is_native_geometry() and model_fine.is_native_geometry()
    inv = None
    if (ray_grid_width and native and mode != "train" and not randoms and N >= capi.fused_min_rays()
            and N % int(ray_grid_width) == 0 and PATCH_H * SUPER_H * int(ray_grid_width) <= MAX_RAYS_PER_LAUNCH and not os.environ.get("NVSR_ROW_ORDER")):
        perm, inv = patch_order(N, ray_grid_width, rays.device)
        rays = rays.index_select(0, perm)
    step = MAX_RAYS_PER_LAUNCH if native else MAX_RAYS_PER_LAUNCH_GENERIC
    if inv is not None:
        step -= step % (PATCH_H * SUPER_H * int(ray_grid_width))       # launches split between rows of super-blocks
    def launch_all(force_arith=None):
        outs = []
        for a in range(0, max(N, 1), step):
            b = min(a + step, N)
            sub = None if randoms is None else {k: v[a:b] for k, v in randoms.items()}
            outs.append(predict_and_render_radiance(rays[a:b], model_coarse, model_fine, options, scene_id, mode=mode, randoms=sub if sub is not None else {},
                                                    force_arith=force_arith))
        return outs[0] if len(outs) == 1 else tuple(None if outs[0][i] is None else torch.cat([o[i] for o in outs], 0) for i in range(9))

    # Evaluation in NVSR_ARITH_F16X2 heals itself: the reference renders finite pixels for any f32 model (models.py:395-421); the f16 limbs
    # have ranges (|W| < 255, |feature / activation| < 4094), beyond which the kernels write NaN and raise the library's range flag.  One
    # device word, zeroed before the frame and read back after it (the only host wait of an evaluation frame; a frame is >= 100 ms of
    # kernels at 800 x 800): raised -> the frame is rendered again in the 3-bf16-limb arithmetic (warned once), and later frames of the same
    # parameters go there directly.  Covers hidden activations, which no check of the operands could see beforehand.
    range_checked = mode != "train" and native and N > 0 and _f16_in_play(model_coarse, model_fine)
    force, sr_changed = None, []
    if range_checked:
        key = _range_key(model_coarse, model_fine)
        if model_fine.__dict__.get("_f16_unfit") == key:
            force = "bf16x3"
            # (only when it was the SR stage that left the range -- bit 2: a decoder out of range says nothing about planes super-resolved in f16x2)
            if model_fine.__dict__.get("_f16_unfit_bits", 3) & 2:
                sr_changed = _sr_fallback(model_coarse, model_fine)
        else:
            flag = capi.range_flag(rays.device)
            flag.reset()
    rendered = False
    try:
        out = launch_all(force)
        if range_checked and force is None:
            bits = capi.RangeFlag.raised(flag.read_async())
            if bits:
                import warnings
                if not model_fine.__dict__.get("_f16_unfit_warned"):
                    warnings.warn("weights, plane values or activations beyond NVSR_ARITH_F16X2's range (|W| < 255, |feature / activation| < 4094): this "
                                  "model renders in 'bf16x3' while its parameters stay as they are (set model.arithmetic = 'bf16x3' to render there in "
                                  "the first place)")
                    model_fine.__dict__["_f16_unfit_warned"] = True
                model_fine.__dict__["_f16_unfit"] = key
                model_fine.__dict__["_f16_unfit_bits"] = int(bits)
                if bits & 2:
                    sr_changed = _sr_fallback(model_coarse, model_fine, redo=True)
                out = launch_all("bf16x3")
                flag.reset()
        rendered = True
    finally:
        _sr_restore(sr_changed, rendered)          # (also when a launch raised: the model must not stay in 'bf16x3')
    if inv is not None:
        out = tuple(None if t is None else t.index_select(0, inv) for t in out)
    return out


def eval_nerf(height, width, focal_length, model_coarse, model_fine, ray_origins, ray_directions, options, scene_id,
              mode="validation", encode_position_fn=None, encode_direction_fn=None, scene_config={}):
    """train_utils.py:285-331 -> (rgb_coarse[H,W,3], None, None, rgb_fine[H,W,3] | None, None, None, None, None, None)"""
    ray_origins = ray_origins.reshape((1, -1, 3))
    ray_directions = ray_directions.reshape((1, -1, 3))
    batch_rays = torch.cat((ray_origins, ray_directions), dim=0)
    rgb_coarse, _, _, rgb_fine, _, _, rgb_SR, _, _ = run_one_iter_of_nerf(
        height, width, focal_length, model_coarse, model_fine, batch_rays, options, mode="validation",
        encode_position_fn=encode_position_fn, encode_direction_fn=encode_direction_fn, scene_id=scene_id, scene_config=scene_config,
        ray_grid_width=width)
    rgb_coarse = rgb_coarse.reshape([height, width, -1])
    if rgb_fine is not None:
        rgb_fine = rgb_fine.reshape([height, width, -1])
    return rgb_coarse, None, None, rgb_fine, None, None, rgb_SR, None, None


def find_latest_checkpoint(ckpt_path, sr, find_best=False):
    """train_utils.py:333-345 (implemented with the rest of the store protocol in plane_store.py)"""
    from .plane_store import find_latest_checkpoint as impl
    return impl(ckpt_path, sr, find_best)
