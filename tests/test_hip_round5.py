"""GPU parity tests added in round 5 (-m gpu): the tile-pair training forward against the ORACLE, derived copies after HIP-graph replays, the
batched SR-training path (three regions of interest through one launch sequence) against the plane-by-plane path and the oracle."""
import ctypes as C

import numpy as np
import pytest
import torch

from test_hip_parity import DEV, N_

pytestmark = pytest.mark.gpu


def _oracle_scene_of(oracle, hip, m, sid):
    from oracle.oracle import decoder_blob
    planes = [N_(m.planes_[hip.models.get_plane_name(sid, d)]) for d in range(4)]
    sc = oracle.scene(planes, m.box_coords[sid].numpy())
    dec = oracle.decoder(decoder_blob({k: N_(v) for k, v in m.state_dict().items()}))
    return sc, dec


# ---------------------------------------------------------------------------------------------------------------------------------
# decode_rays_pair_kernel (the default training forward from S = 128 on) directly against the oracle
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,S", [(1, 33), (5, 37), (3, 64), (7, 65), (9, 97), (64, 128), (33, 192)])
def test_pair_forward_against_the_oracle(hip, oracle, N, S):
    """train_utils.py:15-64 + models.py:381-421 at given depths: the raw decoder outputs [rgb, sigma] of decode_rays_pair_kernel (two 32-sample
    tiles per wave, 2 f16 limbs) against oracle.render_given_z(..., want_raw=True) (C restatement, float64 accumulation) on the ragged shapes of
    test_pair_forward_matches_the_one_tile_forward.  Tolerance: 2e-5 of the output range (the stated tolerance of every decoder kernel)."""
    from bench import make_synthetic_scene
    capi = hip.capi
    lib = capi.lib()
    seed = N + S
    mc, mf, sid, pose = make_synthetic_scene(DEV, plane_res=48, view_res=16, seed=seed)
    g = torch.Generator(device=DEV).manual_seed(seed)
    H = W = 64
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    sel = torch.randint(0, H, (N, 2), device=DEV, generator=g)
    ro, rd = hip.training.get_ray_bundle_at(H, W, focal, pose, sel)
    rays = hip.train_utils.pack_rays(ro, rd, 2.0, 6.0)
    z = torch.sort(torch.rand(N, S, device=DEV, generator=g) * 4 + 2, -1)[0].contiguous()
    sc, keep = mf.native_scene()
    raw = torch.full((N, S, 4), float("nan"), device=DEV)
    f = lib.nvsr_decode_rays_pair_launch
    f.restype = C.c_int
    assert f(C.byref(sc), capi.ptr(mf.packed_decoder()), C.c_int64(N), C.c_int(S), capi.ptr(rays), capi.ptr(z), capi.ptr(raw), None, capi.stream()) == 0
    osc, dec = _oracle_scene_of(oracle, hip, mf, sid)
    want = oracle.render_given_z(osc, dec, N_(rays), N_(z), want_raw=True)["raw"]
    got = N_(raw)
    assert np.isfinite(got).all()
    scale = np.abs(want).max()
    err = np.abs(got.astype(np.float64) - want).max() / scale
    assert err <= 2e-5, err


# ---------------------------------------------------------------------------------------------------------------------------------
# a HIP-graph replay updates the parameters in place without running Python -- every derived copy must follow
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("what", [("LR_planes",), ("LR_planes", "decoder")])
@pytest.mark.parametrize("channels_last", [True, False])
def test_eager_renders_between_graph_replays_see_the_updated_parameters(hip, what, channels_last):
    """training.GraphedTrainStep replays, then evaluate_view / eval_nerf (train_nerf.py:625-788 runs between iterations): the frame must be the
    frame of the parameters AS THEY ARE NOW -- equal to a render after model.invalidate() dropped every derived copy (packed decoder blobs,
    channel-last copies of NCHW planes, the f16 range cache), not the frame of the parameters as they stood after the capture."""
    from bench import make_synthetic_scene, render_options
    what = set(what)
    mc, mf, sid, pose = make_synthetic_scene(DEV, plane_res=64, view_res=16, seed=21, channels_last=channels_last)
    for m in (mc, mf):
        for n, p in m.named_parameters():
            p.requires_grad_("rot_mats" not in n and ("planes_" in n or "decoder" in what))
        m.train()
    opts, scfg = render_options(32, 32, perturb=True, noise=0.2)
    planes = list(mc.planes_.values())
    dec = list({id(p): p for m in (mc, mf) for p in m.decoder_parameters()}.values())
    popt = torch.optim.Adam(planes, lr=5e-2, fused=True, capturable=True)
    opt = torch.optim.Adam(dec, lr=5e-3, fused=True, capturable=True) if "decoder" in what else None
    step = hip.training.TrainStep(mc, mf, opts, what, optimizer=opt, planes_optimizer=popt, pixel_sampler=hip.training.DevicePixelSampler(seed=5))
    H = W = 48
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    g = torch.Generator(device=DEV).manual_seed(3)
    img = torch.rand(H, W, 3, device=DEV, generator=g)
    N, Nc, Nf = 512, 32, 32
    rnd = dict(t_rand=torch.rand(N, Nc, device=DEV, generator=g), u=torch.rand(N, Nf, device=DEV, generator=g),
               noise_coarse=0.2 * torch.randn(N, Nc, device=DEV, generator=g), noise_fine=0.2 * torch.randn(N, Nc + Nf, device=DEV, generator=g))
    graphed = hip.training.GraphedTrainStep(step, img, pose, H, W, focal, 1, sid, scfg, N, randoms_fn=rnd, warmup=2)
    eopts, _ = render_options(32, 32)

    def frame():
        for m in (mc, mf):
            m.eval()
        with torch.no_grad():
            ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, pose)
            out = hip.train_utils.eval_nerf(H, W, focal, mc, mf, ro, rd, eopts, scene_id=sid, scene_config=scfg)
        return out[0].clone(), out[3].clone()

    first = frame()                                   # (fills every cache with copies of the parameters as they stand after the capture)
    for _ in range(4):
        graphed()
    torch.cuda.synchronize()
    got = frame()
    for m in (mc, mf):
        m.invalidate()
    want = frame()
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    assert float((want[1] - first[1]).abs().max()) > 1e-3          # the replays did move the scene (lr is large on purpose)
    # ... and an eager TrainStep after the replays trains the current parameters: its loss equals the next replay's on the same draw
    m_e = None
    if "decoder" not in what:      # (planes only: deterministic up to the scatter's ordering noise)
        sampler2 = hip.training.DevicePixelSampler(seed=5)
        sampler2.calls = graphed.sampler.calls
        eager = hip.training.TrainStep(mc, mf, opts, what, planes_optimizer=torch.optim.SGD(planes, lr=0.0), pixel_sampler=sampler2)
        m_e = eager(0, img, pose, H, W, focal, 1, sid, scfg, N, randoms=rnd)
        loss_e = m_e["loss"]
        graphed()
        assert abs(graphed.metrics()["loss"] - loss_e) <= 1e-5 * max(1.0, abs(loss_e)), (graphed.metrics()["loss"], loss_e)


# ---------------------------------------------------------------------------------------------------------------------------------
# the regions of interest of a scene's planes through the SR network in ONE launch sequence
# ---------------------------------------------------------------------------------------------------------------------------------
def _sr_setup(hip, hid, nb, R, seed, scale=10.0):
    torch.manual_seed(seed)
    sr = hip.models.PlanesSR(hip.models.EDSR, 4, 48, 48, {"model": {"hidden_size": hid, "n_blocks": nb}}, "bilinear").to(DEV)
    with torch.no_grad():
        for p_ in sr.parameters():
            p_.mul_(scale)
    sr.train()
    g = torch.Generator(device=DEV).manual_seed(seed + 1)
    lrs = [torch.nn.Parameter(torch.randn(1, 48, R, R, device=DEV, generator=g) * 0.5) for _ in range(3)]
    for k, t in enumerate(lrs):
        sr.set_LR_plane(t, id="p%d" % k, save_interpolated=False)
    return sr, lrs


def _blob(sr):
    return torch.cat([(w.grad if w.grad is not None else torch.zeros_like(w)).reshape(-1) for w in sr.inner_model.conv_parameters()])


@pytest.mark.parametrize("hid,nb", [(16, 2), (128, 1)])
@pytest.mark.parametrize("mode", ["f16x2", "bf16x3", "f32"])
def test_batched_sr_training_equals_the_plane_by_plane_path(hip, hid, nb, mode):
    """PlanesSR.forward_many (ops.PlanesSRBatchFn -> nvsr_planes_sr_train_batch_arith / _backward_batch_arith: ragged launches, one
    weight-gradient pass per layer over all planes) against three PlanesSR.forward calls (models.py:884-926 per plane) on three DIFFERENT
    regions of interest (one of them touching the plane's border: replicate padding), hidden 16 (the narrow limb kernels) and 128 (the
    16x16x32 kernels): the planes are bit-identical (an output element's accumulation order does not depend on the tile it sits in; exact f32
    runs plane by plane inside the batch call); gradients: relative L2 <= 1e-5 (the order of the weight-gradient sums; f16 limbs: one
    power-of-two gradient scale for all planes instead of one per plane)."""
    R = 24
    rois = [[-0.9, -0.35, 0.1, 0.8], [-1.0, -1.0, 0.2, 0.3], [-0.2, -0.6, 0.95, 0.4]]
    res = {}
    for path in ("batched", "single"):
        sr, lrs = _sr_setup(hip, hid, nb, R, seed=31)
        sr.inner_model.arithmetic = mode
        if path == "batched":
            outs = sr.forward_many([("p%d" % k, rois[k]) for k in range(3)])
            assert outs[0].grad_fn is outs[1].grad_fn or type(outs[0].grad_fn).__name__ == type(outs[1].grad_fn).__name__
        else:
            outs = [sr(("p%d" % k, torch.tensor(rois[k]).reshape(2, 2))) for k in range(3)]
        gen = torch.Generator(device=DEV).manual_seed(77)
        loss = 0.0
        for o in outs:
            w = torch.randn(o.shape, device=DEV, generator=gen)
            loss = loss + (torch.nan_to_num(o) * w).sum()
        loss.backward()
        res[path] = ([o.detach().clone() for o in outs], _blob(sr).clone(), [t.grad.clone() for t in lrs])
    for a, b in zip(res["batched"][0], res["single"][0]):
        assert torch.equal(torch.isnan(a), torch.isnan(b))
        assert torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
    assert float(res["single"][1].norm()) > 0
    assert rel(res["batched"][1], res["single"][1]) <= 1e-5, rel(res["batched"][1], res["single"][1])
    for a, b in zip(res["batched"][2], res["single"][2]):
        assert float(b.norm()) > 0 and rel(a, b) <= 1e-5, rel(a, b)


def test_batched_sr_training_with_detached_and_full_planes(hip):
    """the batch call with full planes (rois None) and with LR planes that want no gradient (detach_LR_planes, models.py:272): only the planes
    that require it receive one, the weight gradient is the sum over all planes"""
    R = 20
    sr, lrs = _sr_setup(hip, 16, 2, R, seed=5)
    lrs[1].requires_grad_(False)
    outs = sr.forward_many(["p0", "p1", "p2"])
    assert all(o.shape == (1, 48, 4 * R, 4 * R) and bool(torch.isfinite(o).all()) for o in outs)
    sum((o * o).sum() for o in outs).backward()
    gw = _blob(sr).clone()
    g0, g2 = lrs[0].grad.clone(), lrs[2].grad.clone()
    assert lrs[1].grad is None
    sr2, lrs2 = _sr_setup(hip, 16, 2, R, seed=5)
    outs2 = [sr2("p%d" % k) for k in range(3)]
    sum((o * o).sum() for o in outs2).backward()
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
    assert all(torch.equal(a, b) for a, b in zip(outs, outs2))
    assert rel(gw, _blob(sr2)) <= 1e-5 and rel(g0, lrs2[0].grad) <= 1e-5 and rel(g2, lrs2[2].grad) <= 1e-5


def test_residual_gradient_gather_equals_the_scatter(hip, oracle):
    """the bilinear residual's share of the LR plane's gradient (models.py:858-868 transposed), now a gather per LR texel
    (sr_residual_backward_gather_kernel), against the oracle's planes_sr_backward on a network with ZERO weights (the residual is then the
    whole function), ROI touching two borders, align_corners True and False"""
    R = 18
    for align in (True, False):
        torch.manual_seed(3)
        sr = hip.models.PlanesSR(hip.models.EDSR, 4, 48, 48, {"model": {"hidden_size": 16, "n_blocks": 1}}, "bilinear").to(DEV)
        sr.align_corners = align
        with torch.no_grad():
            for p_ in sr.parameters():
                p_.zero_()
        sr.train()
        lr = torch.nn.Parameter(torch.randn(1, 48, R, R, device=DEV))
        sr.set_LR_plane(lr, id="p", save_interpolated=False)
        roi = [-1.0, -0.3, 0.4, 1.0]
        out = sr(("p", torch.tensor(roi).reshape(2, 2)))
        w = torch.randn(out.shape, device=DEV)
        (torch.nan_to_num(out) * w).sum().backward()
        # reference: autograd through F.interpolate on the region's pixels
        lr2 = lr.detach().clone().requires_grad_(True)
        up = torch.nn.functional.interpolate(lr2, scale_factor=4, mode="bilinear", align_corners=align)
        inside = ~torch.isnan(out.detach())
        (torch.where(inside, up, torch.zeros_like(up)) * w).sum().backward()
        err = float((lr.grad - lr2.grad).abs().max()) / float(lr2.grad.abs().max())
        assert err <= 2e-6, (align, err)


@pytest.mark.parametrize("ndc", [False, True])
def test_refine_iteration_with_regions_drawn_ahead_equals_the_in_stream_iteration(hip, ndc):
    """training.TrainStep._draw_rays: pixels, rays and regions of interest produced on a side stream ahead of the iteration (`_roi_hint`) against
    the same iteration with everything on the iteration's stream -- same pixels (device sampler, same key), same regions, same loss and
    gradients (to the ordering noise of float atomics)."""
    from conftest import load_golden
    from test_hip_parity import T, _grad_models, _gt_and_student, make_options
    g = load_golden("g11_grads.npz")
    sid = "lego_DS8_PlRes20_8"
    H = W = 20
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    pose = T(load_golden("g08_render.npz")["pose"])
    opts, scfg = make_options(24, 24)
    if ndc:                                          # an LLFF-style scene (BASELINE configs[4]): NDC rays, near 0, far 1 (train_utils.py:215-218)
        from test_hip_parity import Opt
        scfg = Opt(near=0.0, far=1.0, no_ndc=False)
    res = {}
    for ahead in (True, False):
        _, noisy = _gt_and_student(hip, g, sid, seed=84)
        mc, mf = _grad_models(hip, g, noisy, sid, what=("planes",))
        torch.manual_seed(8)
        sr = hip.models.PlanesSR(hip.models.EDSR, 4, 48, 48, {"model": {"hidden_size": 16, "n_blocks": 2}}, "bilinear").to(DEV)
        with torch.no_grad():
            for p_ in sr.parameters():
                p_.mul_(10.0)
        mf.assign_SR_model(sr, SR_viewdir=False)
        mf.assign_LR_planes()
        img = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(5)).to(DEV)
        step = hip.training.TrainStep(mc, mf, opts, {"SR", "LR_planes"}, SR_optimizer=torch.optim.SGD(sr.parameters(), lr=0.0), SR_model=sr, sr_loss="fine",
                                      planes_optimizer=torch.optim.SGD(list(mc.planes_.values()), lr=0.0),
                                      pixel_sampler=hip.training.DevicePixelSampler(seed=9))
        step.prologue_ahead = ahead
        torch.manual_seed(12)                        # (the iteration's random inputs come from the CPU generator)
        m = step(0, img, pose, H, W, focal, 1, sid, scfg, 150, sr_iter=True)
        assert ("_prologue_stream" in step.__dict__) == ahead and "_roi_hint" not in mf.__dict__        # (the hint was consumed)
        res[ahead] = (m["loss"], _blob(sr).clone(), [p_.grad.clone() for p_ in mc.planes_.values()])
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
    assert abs(res[True][0] - res[False][0]) <= 1e-6 * max(1.0, abs(res[False][0]))
    assert float(res[False][1].norm()) > 0 and rel(res[True][1], res[False][1]) <= 1e-5
    for a, b in zip(res[True][2], res[False][2]):
        assert rel(a, b) <= 1e-5


def test_default_sr_training_on_a_coupled_scene(hip):
    """the reference's real SR setup couples an HR scene to the LR scene whose planes it samples (SceneCoupler, models.py:936-1011:
    every plane name goes through scene_with_saved_plane, :273).  With only the fine model super-resolving (apply_2_coarse False) the COARSE
    model samples the saved LR planes raw -- training_planes must map its plane names through the coupler too (it raised KeyError).  The
    coupled iteration equals the iteration of an un-coupled twin that holds the same planes under the HR scene's own names."""
    from conftest import load_golden
    from test_hip_parity import T, _grad_models, _gt_and_student, make_options

    class Coupler:                       # what models.SceneCoupler answers for one HR scene coupled to one LR scene
        def __init__(self, hr, lr):
            self.scene2saved, self.hr = {hr: lr, lr: lr}, hr
        def scene_with_saved_plane(self, name, plane_not_scene=False):
            return name.replace(self.hr, self.scene2saved[self.hr]) if plane_not_scene else self.scene2saved[name]
        def should_SR(self, name, plane_not_scene=False):
            return self.hr in name
        def should_downsample(self, plane_name, for_LR_loading=False):
            return False

    g = load_golden("g11_grads.npz")
    lr_id, hr_id = "lego_DS8_PlRes20_8", "lego_DS2_PlRes80_8"
    H = W = 20
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    pose = T(load_golden("g08_render.npz")["pose"])
    opts, scfg = make_options(24, 24)
    res = {}
    for coupled in (True, False):
        _, noisy = _gt_and_student(hip, g, lr_id, seed=85)
        saved_id = lr_id if coupled else hr_id
        mc, mf = _grad_models(hip, g, noisy, saved_id, what=("planes",))
        if coupled:
            for m in (mc, mf):
                m.scene_coupler = Coupler(hr_id, lr_id)
                m.box_coords = {hr_id: m.box_coords[lr_id]}
                m.__dict__.pop("_scene_consts", None)
        torch.manual_seed(8)
        sr = hip.models.PlanesSR(hip.models.EDSR, 4, 48, 48, {"model": {"hidden_size": 16, "n_blocks": 2}}, "bilinear").to(DEV)
        with torch.no_grad():
            for p_ in sr.parameters():
                p_.mul_(10.0)
        sr.train()
        mf.assign_SR_model(sr, SR_viewdir=False)
        mf.assign_LR_planes()
        ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, pose)
        sel = torch.arange(0, H * W, 3, device=DEV)
        batch = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0)
        out = hip.train_utils.run_one_iter_of_nerf(H, W, focal, mc, mf, batch, opts, hr_id, mode="train", scene_config=scfg, randoms={})
        (out[0].sum() + out[3].sum()).backward()
        planes = [mc.planes_[hip.models.get_plane_name(saved_id, d)] for d in range(4)]
        assert all(p_.grad is not None and float(p_.grad.abs().sum()) > 0 for p_ in planes)
        res[coupled] = (out[0].detach().clone(), out[3].detach().clone(), _blob(sr).clone(), [p_.grad.clone() for p_ in planes])
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    assert rel(res[True][2], res[False][2]) <= 1e-5
    for a, b in zip(res[True][3], res[False][3]):
        assert rel(a, b) <= 1e-5


@pytest.mark.parametrize("variant", ["align_corners_false", "bicubic", "normalized", "rf_bound"])
def test_batched_sr_training_with_the_options_of_planes_sr(hip, variant):
    """the batched path under the options PlanesSR.forward is pinned for by g20 / g22 (models.py:858-859 align_corners / plane_interp of the
    residual, :899-901 input normalisation, :793-798 receptive_field_bound): forward_many on three regions equals three forward calls"""
    R, hid, nb = 22, 16, 2
    rois = [[-0.8, -0.5, 0.3, 0.9], [-1.0, -0.2, 0.1, 1.0], [-0.3, -1.0, 1.0, 0.2]]
    res = {}
    for path in ("batched", "single"):
        torch.manual_seed(41)
        model_cfg = {"hidden_size": hid, "n_blocks": nb}
        if variant == "rf_bound":
            model_cfg["receptive_field_bound"] = 7
        cfg = {"model": model_cfg, "input_normalization": variant == "normalized"}
        sr = hip.models.PlanesSR(hip.models.EDSR, 4, 48, 48, cfg, "bicubic" if variant == "bicubic" else "bilinear").to(DEV)
        with torch.no_grad():
            for n_, p_ in sr.named_parameters():
                if "NON_LEARNED" not in n_:
                    p_.mul_(10.0)
        if variant == "normalized":
            g_ = torch.Generator().manual_seed(3)
            sr.normalization_params({"mean": torch.randn(48, generator=g_) * 0.1, "std": 0.5 + torch.rand(48, generator=g_)})
            sr = sr.to(DEV)
        sr.align_corners = variant != "align_corners_false"
        sr.train()
        g = torch.Generator(device=DEV).manual_seed(42)
        lrs = [torch.nn.Parameter(torch.randn(1, 48, R, R, device=DEV, generator=g) * 0.5) for _ in range(3)]
        for k, t in enumerate(lrs):
            sr.set_LR_plane(t, id="p%d" % k, save_interpolated=False)
        if path == "batched":
            outs = sr.forward_many([("p%d" % k, rois[k]) for k in range(3)])
        else:
            outs = [sr(("p%d" % k, torch.tensor(rois[k]).reshape(2, 2))) for k in range(3)]
        gen = torch.Generator(device=DEV).manual_seed(7)
        sum((torch.nan_to_num(o) * torch.randn(o.shape, device=DEV, generator=gen)).sum() for o in outs).backward()
        res[path] = ([o.detach().clone() for o in outs], _blob(sr).clone(), [t.grad.clone() for t in lrs])
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
    for a, b in zip(res["batched"][0], res["single"][0]):
        assert torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))
    assert float(res["single"][1].norm()) > 0 and rel(res["batched"][1], res["single"][1]) <= 1e-5
    for a, b in zip(res["batched"][2], res["single"][2]):
        assert float(b.norm()) > 0 and rel(a, b) <= 1e-5


def test_refine_iteration_on_two_ranks():
    """SURVEY 8e training partition for BASELINE configs[4]: `bench.py --workload refine --gpus 2` (two ranks on cuda:0 over gloo, rehearsal): every
    rank draws its own rays and regions of interest; round 6: the SR network's gradient (173 MB) is all-reduced bucket by bucket inside the SR backward
    (distributed.OverlappedSRGradSync through nvsr_planes_sr_backward_batch_marks; host-staged over gloo here), the planes' and both decoders' gradients go
    through distributed.allreduce_gradients behind it, then the three optimizers step; the run asserts that a sample of every parameter group is
    bit-identical on both ranks after its iterations and finishes with one line, finite, weak scaling."""
    from test_hip_round3 import _bench_rehearsal
    r, err = _bench_rehearsal(["--workload", "refine", "--refine-what", "joint", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    for rank in (0, 1):
        assert "REFINE_PARAMS_IDENTICAL rank %d of 2 (overlapped gradient sync, 4 buckets)" % rank in err, err[-3000:]
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["config"]["rays_per_step_per_gpu"] == 4096
    assert np.isfinite(r["value"]) and r["value"] > 0 and r["roofline"]["frac"] > 0
    assert r["collectives"]["backend"] == "gloo" and r["collectives"]["world_size"] == 2


def test_rccl_backend_executes_the_collective_branch_on_one_rank():
    """"the RCCL branch has still never executed against RCCL".  No second GPU is reachable from a test box, but RCCL itself is: a
    process group of ONE rank on backend "nccl" (= RCCL on ROCm) runs the collectives of distributed.py on device buffers through the library
    -- initialisation, `all_reduce(async_op=True)` on a row-major tensor and on the row-major [N,H,W,C] view of a channels_last plane gradient
    (RCCL refuses non-contiguous tensors: the gloo rehearsals could not tell), `all_gather_into_tensor` into a caller's buffer, the averaging of
    allreduce_gradients' RCCL path (forced: world 1 short-circuits it) and a row-sharded frame through the real renderer.  Sums over one rank
    are the inputs: values are checked, not only that nothing raises."""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import os, sys, socket
        sys.path.insert(0, %r)
        import numpy as np, torch, torch.distributed as dist
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        assert dist.get_backend() == "nccl"
        import nvsr_amd
        D = nvsr_amd.distributed
        base = torch.arange(2 * 48 * 5 * 7, dtype=torch.float32, device=dev).reshape(2, 48, 5, 7)
        cl = base.clone().contiguous(memory_format=torch.channels_last)
        rm = torch.arange(1000, dtype=torch.float32, device=dev)
        ptrs = (cl.data_ptr(), rm.data_ptr())
        D.allreduce_in_place([cl, rm], scale=0.5)                      # the RCCL branch of allreduce_gradients: async all-reduces, one wait, one fused scale
        torch.cuda.synchronize()
        assert torch.equal(cl, base * 0.5) and torch.equal(rm, torch.arange(1000, dtype=torch.float32, device=dev) * 0.5)
        assert (cl.data_ptr(), rm.data_ptr()) == ptrs and cl.is_contiguous(memory_format=torch.channels_last)
        # an SR network's many small weight gradients: flat buckets through RCCL (allreduce_coalesced), layouts and storage kept
        many = [torch.arange(n, dtype=torch.float32, device=dev) + k for k, n in enumerate([300, 7, 1024, 513, 2, 4096, 33])] + [cl]
        mptrs = [t.data_ptr() for t in many]
        D.allreduce_coalesced(many, scale=2.0, bucket_bytes=4096)
        torch.cuda.synchronize()
        for k, n in enumerate([300, 7, 1024, 513, 2, 4096, 33]):
            assert torch.equal(many[k], (torch.arange(n, dtype=torch.float32, device=dev) + k) * 2.0)
        assert torch.equal(cl, base) and [t.data_ptr() for t in many] == mptrs and cl.is_contiguous(memory_format=torch.channels_last)
        out = torch.empty(6, 3, device=dev)
        t = torch.rand(6, 3, device=dev)
        dist.all_gather_into_tensor(out, t)                            # what gather_row_blocks issues on an even split
        assert torch.equal(out, t)
        w = torch.ones(3, device=dev)
        dist.all_reduce(w, op=dist.ReduceOp.MAX); dist.barrier()
        # the row-sharded frame of the bench's N > 1 default partition, through the fused passes
        from bench import make_synthetic_scene, render_options
        mc, mf, sid, pose = make_synthetic_scene(dev, plane_res=48, view_res=16, seed=2)
        H = W = 32; focal = 0.5 * W / np.tan(0.5 * 0.6911112)
        opts, scfg = render_options(16, 16)
        c, f = D.render_views_sharded(H, W, focal, mc, mf, [pose], opts, sid, scfg)
        ro, rd = nvsr_amd.nerf_helpers.get_ray_bundle(H, W, focal, pose)
        alone = nvsr_amd.train_utils.eval_nerf(H, W, focal, mc, mf, ro, rd, opts, scene_id=sid, scene_config=scfg)
        assert torch.equal(c[0], alone[0]) and torch.equal(f[0], alone[3])
        dist.barrier(); dist.destroy_process_group()
        print("RCCL_ONE_RANK_OK")
    """ % root)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL_ONE_RANK_OK" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])


def test_batched_sr_training_at_the_benchmark_size(hip):
    """BASELINE configs[4] at its real size: PlanesSR(EDSR hidden 256, 32 blocks, x4) on the regions of interest of three 48 x 200^2 planes (the crops of
    the refine bench: 115 x 143, 115 x 125, 143 x 125 LR texels + 68 of context per side) -- the batched path (69 ragged launches per pass, one
    weight-gradient pass per layer over the three crops, gradient magnitudes from the epilogues) against three PlanesSR.forward calls: planes bit
    for bit, the 43.3 M weight gradients and the LR gradients within 1e-5 relative L2."""
    R = 200
    rois = [[-0.73, -0.88, 0.42, 0.55], [-0.73, -0.82, 0.42, 0.43], [-0.88, -0.82, 0.55, 0.43]]
    res = {}
    for path in ("batched", "single"):
        torch.manual_seed(51)
        sr = hip.models.PlanesSR(hip.models.EDSR, 4, 48, 48, {"model": {"hidden_size": 256, "n_blocks": 32}}, "bilinear").to(DEV)
        with torch.no_grad():
            for p_ in sr.parameters():
                p_.mul_(3.0)
        sr.train()
        g = torch.Generator(device=DEV).manual_seed(52)
        lrs = [torch.nn.Parameter(torch.randn(1, 48, R, R, device=DEV, generator=g) * 0.5) for _ in range(3)]
        for k, t in enumerate(lrs):
            sr.set_LR_plane(t, id="p%d" % k, save_interpolated=False)
        if path == "batched":
            outs = sr.forward_many([("p%d" % k, rois[k]) for k in range(3)])
        else:
            outs = [sr(("p%d" % k, torch.tensor(rois[k]).reshape(2, 2))) for k in range(3)]
        gen = torch.Generator(device=DEV).manual_seed(53)
        sum((torch.nan_to_num(o) * torch.randn(o.shape, device=DEV, generator=gen) * 1e-3).sum() for o in outs).backward()
        res[path] = ([o.detach().clone() for o in outs], _blob(sr).clone(), [t.grad.clone() for t in lrs])
        del sr, lrs, outs
        torch.cuda.empty_cache()
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
    for a, b in zip(res["batched"][0], res["single"][0]):
        assert bool(torch.isnan(a).any()) and torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))
        assert bool(torch.isfinite(torch.nan_to_num(a)).all())
    assert float(res["single"][1].norm()) > 0 and rel(res["batched"][1], res["single"][1]) <= 1e-5, rel(res["batched"][1], res["single"][1])
    for a, b in zip(res["batched"][2], res["single"][2]):
        assert float(b.norm()) > 0 and rel(a, b) <= 1e-5, rel(a, b)


# ---------------------------------------------------------------------------------------------------------------------------------
# one launch per fragment kind for a whole network's weights (pack_layers) against the layer-by-layer packing it replaced
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("geometry", [(48, 48, 256, 32, 2), (48, 48, 64, 3, 2), (48, 48, 128, 40, 1), (16, 16, 32, 1, 0)])
@pytest.mark.parametrize("dgrad", [False, True])
def test_network_pack_is_the_layer_by_layer_pack(hip, geometry, dgrad):
    """nvsr_pack_edsr / nvsr_pack_edsr_dgrad (a table of up to 36 layers per launch; 82 layers for the 40-block case = three tables) give bit for
    bit the blobs nvsr_pack_conv3x3 / nvsr_pack_conv3x3_dgrad write layer by layer (models.py:803-872 EDSR: head, 2 convs per block, body tail,
    upsampler convs, tail)."""
    capi = hip.capi
    lib = capi.lib()
    Cin, Cout, hid, nb, n_up = geometry
    shapes = [(Cin, hid)] + [(hid, hid)] * (2 * nb + 1) + [(hid, 4 * hid)] * n_up + [(hid, Cout)]
    g = torch.Generator(device=DEV).manual_seed(sum(geometry))
    nat = torch.randn(sum(9 * a * b for a, b in shapes), device=DEV, generator=g)
    got = torch.ops.nvsr.pack_edsr(nat, list(geometry), dgrad)
    want = torch.full_like(got, float("nan"))
    o_nat = o_pk = 0
    for ci, co in shapes:
        n_pk = lib.nvsr_conv3x3_packed_floats(co, ci) if dgrad else lib.nvsr_conv3x3_packed_floats(ci, co)
        capi.call("nvsr_pack_conv3x3_dgrad" if dgrad else "nvsr_pack_conv3x3", capi.ptr(nat[o_nat:]), ci, co, capi.ptr(want[o_pk:]), capi.stream())
        o_nat += 9 * ci * co
        o_pk += n_pk
    assert o_nat == nat.numel() and o_pk == got.numel()
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))


# ---------------------------------------------------------------------------------------------------------------------------------
# nvsr_pack_edsr_arith: a blob with only the fragment regions ONE arithmetic reads (what a training iteration re-packs)
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("hid,nb", [(16, 2), (64, 1), (128, 1), (256, 1)])
@pytest.mark.parametrize("mode", ["f16x2", "bf16x3", "f32"])
def test_arithmetic_packed_blob_runs_its_arithmetic(hip, hid, nb, mode):
    """nvsr_pack_edsr_arith / nvsr_pack_edsr_dgrad_arith into NaN-filled buffers: every word they write is the word nvsr_pack_edsr writes, they
    write fewer words than it, and the EDSR training forward + backward (models.py:769-822 under autograd; edsr_train / edsr_backward) in THAT
    arithmetic give bit for bit the outputs, input gradient and weight gradients of the full blobs without a single NaN -- i.e. launch_conv reads
    no region that conv_kinds_for (sr_core.h) left out, for the narrow (16, 64), the 128- and the 256-channel layer families."""
    capi, nv = hip.capi, torch.ops.nvsr
    lib = capi.lib()
    torch.manual_seed(hid + nb)
    net = hip.models.PlanesSR(hip.models.EDSR, 4, 48, 48, {"model": {"hidden_size": hid, "n_blocks": nb}}, "bilinear").to(DEV).inner_model
    geom = list(net.geometry)
    code = capi.ARITHMETIC[mode]
    nat = net.natural_blob()
    blobs = {}
    for dgrad in (False, True):
        full = nv.pack_edsr(nat, geom, dgrad)
        part = torch.full_like(full, float("nan"))
        fill = part.view(torch.int32)[0].item()
        capi.call("nvsr_pack_edsr_dgrad_arith" if dgrad else "nvsr_pack_edsr_arith", capi.ptr(nat), *geom, capi.ptr(part), code, capi.stream())
        pi, fi = part.view(torch.int32), full.view(torch.int32)
        written = pi != fill
        assert torch.equal(pi[written], fi[written])
        n_written = int(written.sum())
        assert 0 < n_written < 0.75 * full.numel(), (n_written, full.numel())
        blobs[dgrad] = (full, part)
        # the operator's own path (uninitialised remainder) writes the same words
        op = nv.pack_edsr(nat, geom, dgrad, code).view(torch.int32)
        assert torch.equal(op[written], fi[written])
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn((1, 48, 4 * nb + 14, 4 * nb + 17), device=DEV, generator=g)
    res = []
    for which in (0, 1):
        out, acts = nv.edsr_train(x, nat, blobs[False][which], blobs[True][which], geom, code)
        gy = torch.randn(out.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(6))
        gnat, dx = nv.edsr_backward(x, acts, blobs[True][which], geom, gy, True, code)
        res.append((out, gnat, dx))
    for a, b in zip(res[0], res[1]):
        assert bool(torch.isfinite(b).all())
        assert torch.equal(a, b)
    # the model's cache: a blob of one arithmetic is not served to another arithmetic or to a caller that asks for every region
    net.invalidate()
    p_a = net.packed_weights(code)
    assert net.packed_weights(code) is p_a
    p_all = net.packed_weights()
    assert p_all is not p_a and torch.equal(p_all.view(torch.int32), blobs[False][0].view(torch.int32))
    assert net.packed_weights(code) is p_all                  # (a blob with every region serves any arithmetic)


# ---------------------------------------------------------------------------------------------------------------------------------
# fused optimizers do not bump tensor version counters: the derived copies of the parameters must follow the step all the same
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fused", [True, False])
def test_derived_copies_follow_a_fused_optimizer_step(hip, fused):
    """A joint SR-refinement iteration (what = LR_planes + decoder + SR) stepped by Adam(fused=True) -- the optimizers of bench.py -- and by plain
    Adam: after the iteration every derived copy the NEXT forward would use is the copy of the CURRENT parameters: the EDSR fragment blobs
    (forward and data gradient), the packed decoder blobs of both models.  (torch's fused Adam leaves `_version` alone: before round 5's
    TrainStep._step the blobs of the next iteration were the ones packed before the step.)"""
    from conftest import load_golden
    from test_hip_parity import T, _grad_models, _gt_and_student, make_options
    g = load_golden("g11_grads.npz")
    sid = "lego_DS8_PlRes20_8"
    H = W = 20
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    pose = T(load_golden("g08_render.npz")["pose"])
    opts, scfg = make_options(24, 24)
    _, noisy = _gt_and_student(hip, g, sid, seed=84)
    mc, mf = _grad_models(hip, g, noisy, sid, what=("planes", "decoder"))
    torch.manual_seed(8)
    sr = hip.models.PlanesSR(hip.models.EDSR, 4, 48, 48, {"model": {"hidden_size": 16, "n_blocks": 2}}, "bilinear").to(DEV)
    with torch.no_grad():
        for p_ in sr.parameters():
            p_.mul_(10.0)
    mf.assign_SR_model(sr, SR_viewdir=False)
    mf.assign_LR_planes()
    img = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(5)).to(DEV)
    dec = [p_ for m in (mc, mf) for p_ in m.decoder_parameters()]
    kw = {"fused": True} if fused else {}
    step = hip.training.TrainStep(mc, mf, opts, {"SR", "LR_planes", "decoder"}, optimizer=torch.optim.Adam(dec, lr=1e-3, **kw),
                                  SR_optimizer=torch.optim.Adam(sr.parameters(), lr=1e-3, **kw), SR_model=sr, sr_loss="fine",
                                  planes_optimizer=torch.optim.Adam(list(mc.planes_.values()), lr=1e-3, **kw),
                                  pixel_sampler=hip.training.DevicePixelSampler(seed=9))
    net = sr.inner_model
    torch.manual_seed(12)
    step(0, img, pose, H, W, focal, 1, sid, scfg, 150, sr_iter=True)
    used = (net.packed_weights().clone(), net.packed_dgrad_weights().clone(), mc.packed_decoder().clone(), mf.packed_decoder().clone())
    torch.manual_seed(13)
    step(1, img, pose, H, W, focal, 1, sid, scfg, 150, sr_iter=True)
    torch.cuda.synchronize()
    fresh = lambda dgrad: torch.ops.nvsr.pack_edsr(net.natural_blob(), list(net.geometry), dgrad)
    now = (net.packed_weights(), net.packed_dgrad_weights(), mc.packed_decoder(), mf.packed_decoder())
    assert torch.equal(now[0].view(torch.int32), fresh(False).view(torch.int32)) and torch.equal(now[1].view(torch.int32), fresh(True).view(torch.int32))
    for m, blob in ((mc, now[2]), (mf, now[3])):
        m._packed_cache = None
        assert torch.equal(blob.view(torch.int32), m.packed_decoder().view(torch.int32))
    for a, b in zip(used, now):                      # ... and the step did move every one of them
        assert not torch.equal(a.view(torch.int32), b.view(torch.int32))


# ---------------------------------------------------------------------------------------------------------------------------------
# the launches between the render passes of a training iteration (round 5, second half): loss sum + one-launch loss backward, the
# compositor's backward on packed rays, the iteration's scalars as one device tensor
# ---------------------------------------------------------------------------------------------------------------------------------
def test_loss_pair_sum_and_its_backward_equal_torch(hip):
    """training.mse_loss_pair_sum (nvsr_mse_pair_sum / nvsr_mse_pair_backward) against F.mse_loss + torch's addition and autograd
    (train_nerf.py:893-905): the three losses within 1e-6 relative, the gradients of every way the losses can be used -- the sum alone (the
    iteration), a weighted sum, one loss alone, the sum plus one loss -- within 1e-6 of torch's, and bit-identical to the launch they replace
    (nvsr_mse_pair's gradients times the incoming scalar)."""
    T = hip.training
    g = torch.Generator(device=DEV).manual_seed(3)
    for n in (1, 7, 4096, 5000):
        a0, b0, t = (torch.rand(n, 3, device=DEV, generator=g) for _ in range(3))
        for use in ("sum", "weighted", "coarse", "sum+fine"):
            a, b = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
            ar, br = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
            lc, lf, both, packed = T.mse_loss_pair_sum(a, b, t)
            rc, rf = torch.nn.functional.mse_loss(ar, t), torch.nn.functional.mse_loss(br, t)
            assert packed.shape == (3,) and torch.equal(packed, torch.stack([lc, lf, both]).detach())
            torch.testing.assert_close(torch.stack([lc, lf, both]), torch.stack([rc, rf, rc + rf]), rtol=1e-6, atol=0)
            assert torch.equal(both.detach(), (lc + lf).detach())
            if use == "sum":
                both.backward(); (rc + rf).backward()
            elif use == "weighted":
                (both * 0.37).backward(); ((rc + rf) * 0.37).backward()
            elif use == "coarse":
                lc.backward(); rc.backward()
            else:
                (both + 2.0 * lf).backward(); (rc + 3.0 * rf).backward()
            for x, r in ((a, ar), (b, br)):
                if r.grad is None:
                    assert x.grad is None
                else:
                    torch.testing.assert_close(x.grad, r.grad, rtol=2e-6, atol=1e-12)
        # bit for bit what the pair operator's stored gradients times the seed give
        a, b = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        a2, b2 = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        T.mse_loss_pair_sum(a, b, t)[2].backward()
        l0, l1 = T.mse_loss_pair(a2, b2, t)
        (l0 + l1).backward()
        assert torch.equal(a.grad, a2.grad) and torch.equal(b.grad, b2.grad)


def test_composite_backward_reads_the_directions_from_packed_rays(hip):
    """nvsr_composite_backward_rays (directions = columns 3..5 of the packed rays) is nvsr_composite_backward_depth on their contiguous copy, bit
    for bit, with and without noise / opacity / depth gradients (volume_rendering_utils.py:18-49 under autograd)."""
    nv = torch.ops.nvsr
    g = torch.Generator(device=DEV).manual_seed(11)
    N, S = 37, 65
    rays = torch.randn(N, 11, device=DEV, generator=g)
    raw = torch.randn(N, S, 4, device=DEV, generator=g)
    z = torch.sort(torch.rand(N, S, device=DEV, generator=g) * 4 + 2, -1)[0].contiguous()
    noise = torch.randn(N, S, device=DEV, generator=g) * 0.2
    g_rgb, g_acc, g_dep = torch.randn(N, 3, device=DEV, generator=g), torch.randn(N, device=DEV, generator=g), torch.randn(N, device=DEV, generator=g)
    rd = rays[:, 3:6].contiguous()
    for nz, ga, gd, white in ((None, None, None, False), (noise, g_acc, None, True), (noise, g_acc, g_dep, False)):
        want = nv.composite_backward(raw, z, rd, nz, white, False, g_rgb, ga, gd)
        got = nv.composite_backward_rays(raw, z, rays, nz, white, False, g_rgb, ga, gd)
        assert torch.equal(got, want) and bool(torch.isfinite(got).all())


def test_step_metrics_from_the_packed_scalars(hip):
    """StepMetrics of a planes-only iteration (the loss kernel's [coarse, fine, sum] tensor + the range flag word, two asynchronous copies) reports
    what the five-scalar gather reports: loss = rendering loss = coarse + fine, psnr = mse2psnr(loss) (train_nerf.py:893-921)."""
    from bench import make_synthetic_scene, render_options
    T = hip.training
    mc, mf, sid, pose = make_synthetic_scene(DEV, 48, 16, seed=2, channels_last=True)
    opts, scfg = render_options(16, 16, perturb=True, noise=0.2)
    for m_ in (mc, mf):
        for n_, p_ in m_.named_parameters():
            p_.requires_grad_("rot_mats" not in n_ and "planes_" in n_)
        m_.train()
    popt = torch.optim.Adam(list(mc.planes_.values()), lr=1e-3)
    step = T.TrainStep(mc, mf, opts, ["LR_planes"], planes_optimizer=popt, pixel_sampler=T.DevicePixelSampler(seed=5))
    H = W = 64
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    target = torch.rand(H, W, 3, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    m = step(0, target, pose, H, W, focal, 1, sid, scfg, 256)
    assert m._packed_host is not None                      # (the device path was taken)
    vals = dict(m)
    assert abs(vals["loss"] - (vals["coarse_loss"] + vals["fine_loss"])) <= 1e-6 * vals["loss"]
    assert abs(vals["psnr"] - hip.nerf_helpers.mse2psnr(vals["loss"])) < 1e-9
    assert all(np.isfinite(v) for v in vals.values())


def test_kernels_give_the_same_results_while_another_process_shares_the_gpu():
    """Two processes time-slicing the GPU (tools/shared_gpu_check.py, 25 s each, started together): the training forward (S = 64 / 128), a fused
    320 x 320 frame, the EDSR forward and the EDSR training forward + backward repeat their first result BIT FOR BIT, the gate-driven backward and a
    whole training iteration's plane + decoder gradients (float atomics) within a relative L2 of 1e-4, launch after launch.  Round 5: a workgroup barrier was missing in front of the backward kernels' transposed alpha head (train_nerf.py:860-906
    through models.py:381-421's autograd); a process that had the GPU to itself never hit the race, two sharing it did in ~0.1 % of the launches
    (profiles/r05_backward_prologue_race.txt) -- which is what this test would see again."""
    import os, re, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "tools", "shared_gpu_check.py"), "25"]
    procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for _ in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, out in zip(procs, outs):
        assert p.returncode == 0, out[-3000:]
        counts = re.findall(r"(\d+) mismatches of (\d+)", out)
        assert len(counts) == 8, out[-3000:]
        assert all(int(bad) == 0 and int(n) >= 10 for bad, n in counts), out[-3000:]
