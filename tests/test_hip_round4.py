"""GPU parity tests added in round 4 (-m gpu): the HIP-graph replay of a training iteration against the eager iteration, the device-side
f16-range flag (self-healing evaluation renders, loud training steps), the tile-pair training kernels against the one-tile kernels."""
import copy
import ctypes as C

import numpy as np
import pytest
import torch

from test_hip_parity import DEV

pytestmark = pytest.mark.gpu


def _train_setup(hip, what, seed, plane_res=64, n_rays=1024, nc=32, nf=32):
    from bench import make_synthetic_scene, render_options
    mc, mf, sid, pose = make_synthetic_scene(DEV, plane_res=plane_res, view_res=16, seed=seed, channels_last=True)
    for m in (mc, mf):
        for n, p in m.named_parameters():
            p.requires_grad_("rot_mats" not in n and ("planes_" in n or "decoder" in what))
        m.train()
    opts, scfg = render_options(nc, nf, perturb=True, noise=0.2)
    planes = list(mc.planes_.values())
    dec = list({id(p): p for m in (mc, mf) for p in m.decoder_parameters()}.values())
    if "decoder" in what:
        # (SGD: Adam divides by the gradient's own magnitude, so the ordering noise of the plane scatter's float atomics -- the only difference
        #  between two runs of one iteration -- becomes a full +-lr step wherever a gradient is near zero, and the second iteration's loss moves by
        #  1e-4; with SGD the second iteration checks that the replay re-packs the updated decoder weights, to 1e-5)
        popt = torch.optim.SGD(planes, lr=0.5)
        opt = torch.optim.SGD(dec, lr=2e-2)
    else:
        popt = torch.optim.Adam(planes, lr=4e-3, fused=True, capturable=True)
        opt = None
    sampler = hip.training.DevicePixelSampler(seed=77)
    step = hip.training.TrainStep(mc, mf, opts, what, optimizer=opt, planes_optimizer=popt, pixel_sampler=sampler)
    return dict(mc=mc, mf=mf, sid=sid, pose=pose, scfg=scfg, planes=planes, dec=dec, popt=popt, opt=opt, sampler=sampler, step=step,
                n_rays=n_rays, nc=nc, nf=nf)


@pytest.mark.parametrize("what", [("LR_planes",), ("LR_planes", "decoder")])
def test_graph_replay_equals_the_eager_iteration(hip, what):
    """training.GraphedTrainStep (one hipGraphLaunch per iteration) against TrainStep (train_nerf.py:790-923 launch by launch) from the same
    parameters, optimizer state, sampler position and random inputs: the pixel draws are the same integers; the gradients agree to the
    ordering noise of the scatter's float atomics (two eager runs differ by as much: relative L2 <= 1e-5; decoder gradients are sums in a
    fixed order: 1e-6) and so do the losses.  Two replays: the second one must draw new pixels and read the refilled random inputs."""
    what = set(what)
    H = W = 96
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    g = torch.Generator(device=DEV).manual_seed(5)
    img = torch.rand(H, W, 3, device=DEV, generator=g)
    a, b = _train_setup(hip, what, seed=11), _train_setup(hip, what, seed=11)
    N, Nc, Nf = a["n_rays"], a["nc"], a["nf"]
    rnd = dict(t_rand=torch.rand(N, Nc, device=DEV, generator=g), u=torch.rand(N, Nf, device=DEV, generator=g),
               noise_coarse=0.2 * torch.randn(N, Nc, device=DEV, generator=g), noise_fine=0.2 * torch.randn(N, Nc + Nf, device=DEV, generator=g))
    graphed = hip.training.GraphedTrainStep(b["step"], img, b["pose"], H, W, focal, 1, b["sid"], b["scfg"], N, randoms_fn=rnd, warmup=2)
    def sync_twin():
        # the eager twin starts every iteration where the graphed one stands: same parameters, optimizer state, sampler position
        # (Adam turns the ordering noise of the scatter's float atomics into +-lr steps wherever a gradient nearly cancels: two runs of
        #  SEVERAL iterations drift apart by 1e-4 whatever launches them)
        with torch.no_grad():
            for pa, pb in zip(a["planes"] + a["dec"], b["planes"] + b["dec"]):
                pa.copy_(pb)
        a["popt"].load_state_dict(copy.deepcopy(b["popt"].state_dict()))
        if a["opt"] is not None:
            a["opt"].load_state_dict(copy.deepcopy(b["opt"].state_dict()))
        a["sampler"].calls = b["sampler"].calls

    assert int(b["sampler"].state[1]) == b["sampler"].calls == 2 and int(b["sampler"].state[2]) == 0
    seen = []
    for k in range(2):
        sync_twin()
        if k == 1:
            for v in rnd.values():          # new random inputs in the same (static) tensors
                v.copy_(torch.rand(v.shape, device=DEV, generator=g) if v is rnd["t_rand"] or v is rnd["u"] else 0.2 * torch.randn(v.shape, device=DEV, generator=g))
        sel_a, _ = copy.copy(a["sampler"])(img, N)           # what the eager sampler draws at this call (a copy: no side effect)
        m_a = a["step"](k, img, a["pose"], H, W, focal, 1, a["sid"], a["scfg"], N, randoms=rnd)
        graphed()
        m_b = graphed.metrics()
        seen.append(sel_a)
        for key in ("loss", "coarse_loss", "fine_loss", "psnr"):
            assert abs(m_a[key] - m_b[key]) <= 1e-5 * max(1.0, abs(m_a[key])), (k, key, m_a[key], m_b[key])
        for i, (pa, pb) in enumerate(zip(a["planes"] + (a["dec"] if "decoder" in what else []), b["planes"] + (b["dec"] if "decoder" in what else []))):
            ga, gb = pa.grad, pb.grad
            assert ga is not None and gb is not None and float(ga.norm()) > 0
            rel = float((ga - gb).norm() / ga.norm())
            assert rel <= 1e-5, (k, i, rel)
    assert not torch.equal(seen[0], seen[1])
    assert int(b["sampler"].state[1]) == b["sampler"].calls == a["sampler"].calls == 4


def test_graphed_step_refuses_what_it_cannot_replay(hip):
    s = _train_setup(hip, {"LR_planes"}, seed=3)
    img = torch.rand(32, 32, 3, device=DEV)
    args = (img, s["pose"], 32, 32, 30.0, 1, s["sid"], s["scfg"], 256)
    s["step"].vbs = 2
    with pytest.raises(ValueError, match="virtual_batch_size"):
        hip.training.GraphedTrainStep(s["step"], *args)
    s["step"].vbs = 1
    s["step"].planes_optimizer = torch.optim.Adam(s["planes"], lr=1e-3)
    with pytest.raises(ValueError, match="capturable"):
        hip.training.GraphedTrainStep(s["step"], *args)
    s["step"].planes_optimizer = s["popt"]
    s["step"].pixel_sampler = hip.training.select_training_pixels
    with pytest.raises(ValueError, match="DevicePixelSampler"):
        hip.training.GraphedTrainStep(s["step"], *args)


# ---------------------------------------------------------------------------------------------------------------------------------
# tile-pair training forward (decode_pair.hip) against the one-tile kernel (decode_limb.hip) and the oracle
# ---------------------------------------------------------------------------------------------------------------------------------
def _pair_setup(hip, N, S, seed):
    from bench import make_synthetic_scene
    mc, mf, sid, pose = make_synthetic_scene(DEV, plane_res=48, view_res=16, seed=seed)
    g = torch.Generator(device=DEV).manual_seed(seed)
    H = W = 64
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    sel = torch.randint(0, H, (N, 2), device=DEV, generator=g)
    ro, rd = hip.training.get_ray_bundle_at(H, W, focal, pose, sel)
    rays = hip.train_utils.pack_rays(ro, rd, 2.0, 6.0)
    z = torch.sort(torch.rand(N, S, device=DEV, generator=g) * 4 + 2, -1)[0].contiguous()
    return mf, rays, z


@pytest.mark.parametrize("N,S", [(1, 33), (5, 37), (3, 64), (7, 65), (9, 97), (64, 128), (33, 192)])
def test_pair_forward_matches_the_one_tile_forward(hip, N, S):
    """decode_rays_pair_kernel (two 32-sample tiles per wave) against decode_rays_limb_kernel<.,false,2> (one tile per wave), same 2-f16-limb
    arithmetic, on ragged shapes: one chunk pair with a partly / wholly empty second tile (S = 33, 37, 65), pair counts that do not fill the
    four waves of a workgroup, odd chunk counts (S = 65, 97: the last pair of a ray has no second tile).  The layers accumulate in the same
    order in both kernels; the heads do not (two half-wave partial sums): raw within 2e-6 of its range.  Gates: the pair kernel takes a gate
    from the f16 high limb of the activation (a positive activation below 2^-29 reads as closed): identical words but for such elements --
    none on these inputs."""
    capi = hip.capi
    lib = capi.lib()
    mf, rays, z = _pair_setup(hip, N, S, seed=N + S)
    sc, keep = mf.native_scene()
    packed = mf.packed_decoder()
    outs = []
    for fn in ("nvsr_decode_rays_limb_launch", "nvsr_decode_rays_pair_launch"):
        raw = torch.full((N, S, 4), float("nan"), device=DEV)
        gates = torch.full((N, S, 32), -1, dtype=torch.int32, device=DEV)
        f = getattr(lib, fn)
        f.restype = C.c_int
        if "limb" in fn:
            st = f(C.c_int(2), C.byref(sc), capi.ptr(packed), C.c_int64(N), C.c_int(S), capi.ptr(rays), capi.ptr(z), capi.ptr(raw), capi.ptr(gates), None,
                   capi.stream())
        else:
            st = f(C.byref(sc), capi.ptr(packed), C.c_int64(N), C.c_int(S), capi.ptr(rays), capi.ptr(z), capi.ptr(raw), capi.ptr(gates), capi.stream())
        assert st == 0
        outs.append((raw, gates))
    (raw_a, gates_a), (raw_b, gates_b) = outs
    assert bool(torch.isfinite(raw_b).all())
    scale = float(raw_a.abs().max())
    assert float((raw_a - raw_b).abs().max()) <= 2e-6 * scale, float((raw_a - raw_b).abs().max()) / scale
    assert torch.equal(gates_a, gates_b), int(((gates_a ^ gates_b) != 0).sum())
    # without gates: the same raw, bit for bit, as with them
    raw_c = torch.empty_like(raw_b)
    f = lib.nvsr_decode_rays_pair_launch
    assert f(C.byref(sc), capi.ptr(packed), C.c_int64(N), C.c_int(S), capi.ptr(rays), capi.ptr(z), capi.ptr(raw_c), None, capi.stream()) == 0
    assert torch.equal(raw_c, raw_b)


# ---------------------------------------------------------------------------------------------------------------------------------
# NVSR_ARITH_F16X2 heals itself: the library's range flag (include/nvsr.h: nvsr_set_range_flag), evaluation re-renders, training raises
# ---------------------------------------------------------------------------------------------------------------------------------
def test_evaluation_heals_hidden_activation_overflow(hip):
    """The reference renders any f32 model (models.py:395-421).  Here: weights and planes INSIDE the f16 limbs' ranges (the host-side operand
    check passes) but a hidden layer's bias of 5000 -- activations beyond 4094, which only the kernels can see.  eval_nerf must return the
    pixels of the 'bf16x3' render (bit for bit: it IS that render), warn, go to 'bf16x3' directly while the parameters stay as they are
    (no second F16X2 attempt), and return to 'f16x2' once they change back."""
    import warnings
    from bench import make_synthetic_scene, render_options
    mc, mf, sid, pose = make_synthetic_scene(DEV, plane_res=64, view_res=16, seed=4)
    H = W = 136                               # 18 496 rays: the fused passes
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, pose)
    opts, scfg = render_options(16, 24)
    ev = lambda: hip.train_utils.eval_nerf(H, W, focal, mc, mf, ro, rd, opts, scene_id=sid, scene_config=scfg)
    base = ev()
    assert torch.isfinite(base[3]).all() and "_f16_unfit" not in mf.__dict__
    b = mf.density_dec["0"][1].bias
    with torch.no_grad():
        keep = b.detach().clone()
        b[3] = 5000.0
    planes, _ = mf.scene_args()
    assert mf.f16_operands_in_range(planes)                     # nothing a check of the operands could find
    flag = hip.capi.range_flag()
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter("always")
        out = ev()
    assert any("bf16x3" in str(w.message) for w in wl)
    assert torch.isfinite(out[0]).all() and torch.isfinite(out[3]).all() and "_f16_unfit" in mf.__dict__
    assert int(flag.word) == 0                                  # left clean for whoever checks next
    mc.arithmetic = mf.arithmetic = "bf16x3"
    want = ev()
    mc.arithmetic = mf.arithmetic = None
    assert torch.equal(out[0], want[0]) and torch.equal(out[3], want[3])
    # the same parameters again: straight to bf16x3 (the flag is not even reset: plant a value and find it untouched)
    flag.word.fill_(4)
    again = ev()
    assert int(flag.word) == 4 and torch.equal(again[3], want[3])
    flag.reset()
    # the fine pass alone in F16X2 really is NaN (the healing is not a no-op)
    sc, keep_alive = mf.native_scene()
    rays = hip.train_utils.pack_rays(ro, rd, 2.0, 6.0)
    N, S = rays.shape[0], 24
    z = torch.sort(torch.rand(N, S, device=DEV) * 4 + 2, -1).values.contiguous()
    o = [torch.empty((N, 3), device=DEV), torch.empty(N, device=DEV), torch.empty(N, device=DEV)]
    hip.capi.call("nvsr_render_pass_arith", C.byref(sc), hip.capi.ptr(mf.packed_decoder()), N, S, hip.capi.ptr(rays), hip.capi.ptr(z), None, 1,
                  *[hip.capi.ptr(t) for t in o], None, None, None, hip.capi.ARITHMETIC["f16x2"], hip.capi.stream())
    assert torch.isnan(o[0]).any() and int(flag.word) == 1
    flag.reset()
    with torch.no_grad():
        b.copy_(keep)
    back = ev()
    assert torch.equal(back[3], base[3]) and int(flag.word) == 0


def test_training_raises_when_the_f16_range_is_exceeded(hip):
    """TrainStep / GraphedTrainStep with a hidden bias of 5000: the loss would be NaN (and Adam would write NaN into the planes).  The
    metrics of the iteration carry the range flag of its end: reading any of them raises NvsrError naming 'bf16x3'; a loop that never reads its
    metrics gets the error from a later iteration's call (no host wait: the check polls finished iterations)."""
    for graphed in (False, True):
        s = _train_setup(hip, {"LR_planes"}, seed=21)
        H = W = 96
        focal = 0.5 * W / np.tan(0.5 * 0.6911112)
        img = torch.rand(H, W, 3, device=DEV)
        with torch.no_grad():
            s["mf"].rgb_dec["0"][2].bias[7] = 5000.0
        args = (img, s["pose"], H, W, focal, 1, s["sid"], s["scfg"], s["n_rays"])
        if graphed:
            g = hip.training.GraphedTrainStep(s["step"], *args, randoms_fn={}, warmup=1)
            g()
            with pytest.raises(hip.capi.NvsrError, match="bf16x3"):
                g.metrics()
        else:
            m = s["step"](0, *args, randoms={})
            with pytest.raises(hip.capi.NvsrError, match="bf16x3"):
                m["loss"]
            with pytest.raises(hip.capi.NvsrError, match="bf16x3"):      # never reading the metrics: a later call raises
                for it in range(1, 40):
                    s["step"](it, *args, randoms={})
                    torch.cuda.synchronize()
        hip.capi.range_flag().reset()
    # in range: nothing raises, the flag stays down
    s = _train_setup(hip, {"LR_planes"}, seed=22)
    img = torch.rand(96, 96, 3, device=DEV)
    for it in range(3):
        m = s["step"](it, img, s["pose"], 96, 96, 0.5 * 96 / np.tan(0.5 * 0.6911112), 1, s["sid"], s["scfg"], s["n_rays"], randoms={})
    assert np.isfinite(m["loss"]) and int(hip.capi.range_flag().word) == 0


def test_evaluation_heals_an_sr_network_beyond_the_f16_range(hip):
    """A trunk weight of 300 in the SR network (EDSR 128 channels: the f16-limb convolution kernels): the super-resolved planes come out NaN
    inside, the SR stage raises bit 2 of the range flag, eval_nerf drops the cached planes, super-resolves and renders again in 'bf16x3'."""
    import warnings
    from bench import make_synthetic_scene, render_options
    mc, mf, sid, pose = make_synthetic_scene(DEV, plane_res=24, view_res=8, seed=9)
    torch.manual_seed(3)
    sr = hip.models.PlanesSR(hip.models.EDSR, 4, 48, 48, {"model": {"hidden_size": 128, "n_blocks": 1}}, "bilinear").to(DEV)
    sr.eval()
    mf.assign_SR_model(sr, SR_viewdir=False)
    mf.assign_LR_planes()
    H = W = 40
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, pose)
    opts, scfg = render_options(16, 16)
    ev = lambda: hip.train_utils.eval_nerf(H, W, focal, mc, mf, ro, rd, opts, scene_id=sid, scene_config=scfg)
    base = ev()
    assert torch.isfinite(base[3]).all() and sr.inner_model.arithmetic is None
    with torch.no_grad():
        sr.inner_model.residual[0].conv1.weight[17, 3, 1, 1] = 300.0
    sr.clear_SR_planes()
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter("always")
        out = ev()
    assert any("bf16x3" in str(w.message) for w in wl)
    # (round 5, ADVICE r4: the fallback is scoped to the evaluation frame -- the network keeps the arithmetic it was configured with, so that a
    #  later training iteration of the same SR model is not silently moved to another arithmetic)
    assert torch.isfinite(out[3]).all() and sr.inner_model.arithmetic is None and int(hip.capi.range_flag().word) == 0
    assert not torch.equal(out[3], base[3])                     # (the changed weight changes the planes)
    sr.clear_SR_planes()
    again = ev()                                                # same parameters: straight to 'bf16x3' (no second F16X2 attempt), same pixels
    assert torch.equal(again[3], out[3]) and sr.inner_model.arithmetic is None


@pytest.mark.parametrize("plane_res,res,n_rays", [(200, 200, 8192), (800, 800, 8192)])
def test_frame_error_sits_in_rays_whose_importance_samples_flip(hip, plane_res, res, n_rays):
    """Round 3's frame PSNRs against the float64 checker (f32 90.8, bf16x3 89.2, f16x2 87.0 dB on 2 048 rays) looked like a cost of the default
    arithmetic although its decoder outputs are the closest to float64 at equal depths.  bench.frame_error_evidence separates the rays whose
    fine depths match the checker's from those where an importance sample moved (inverse-CDF sampling is discontinuous in the coarse weights,
    nerf_helpers.py:688-700: a rounding-level change of a weight moves a sample by a bin where u meets a knot of the cdf).  Measured on
    16 384 rays of the bench frame (profiles/r04_frame_error_evidence.json): ~5 % of the rays move in EVERY arithmetic (902 / 779 / 836), they
    hold 91-94 % of the squared error, the all-ray PSNRs are 86.5 / 87.3 / 86.5 dB and over the rays that moved in no arithmetic 99.3 / 99.2 /
    100.5 dB -- the order of round 3's three numbers was which handful of rays moved in a small sample.  Asserted here on 8 192 rays of a
    200^2-plane frame AND of the bench's full-size frame (800 x 800, planes 800^2; VERDICT r3 weak #3: the end-to-end assertions ran in the
    default arithmetic only), in all three arithmetics: the moved rays hold most of the squared error; over the common unmoved rays every
    ray is within 1e-3 and no limb arithmetic is more than 1 dB under the exact-f32 kernels; f16x2 does not move more rays than 1.5 x f32's."""
    import bench
    mc, mf, sid, pose = bench.make_synthetic_scene(DEV, plane_res=plane_res, view_res=32, seed=0)
    H = W = res
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, pose)
    rays = hip.train_utils.pack_rays(ro, rd, 2.0, 6.0)
    ev = bench.frame_error_evidence(hip, mc, mf, sid, rays, n_rays=n_rays)
    print({m: {k: v for k, v in ev[m].items() if k != "rgb_abs_error_percentiles"} for m in ("f32", "bf16x3", "f16x2")})
    common = {m: ev[m]["psnr_db_rays_flipped_in_no_arithmetic"] for m in ("f32", "bf16x3", "f16x2")}
    assert min(common.values()) >= 95.0 and common["f16x2"] >= common["f32"] - 1.0 and common["bf16x3"] >= common["f32"] - 1.0, common
    allr = {m: ev[m]["psnr_db_all"] for m in ("f32", "bf16x3", "f16x2")}
    assert max(allr.values()) - min(allr.values()) <= 2.5, allr
    for m in ("f32", "bf16x3", "f16x2"):
        assert ev[m]["rgb_abs_error_max_over_non_flipped"] <= 1e-3, (m, ev[m])
        assert ev[m]["flipped_fraction"] <= 0.09, (m, ev[m])
        assert ev[m]["share_of_squared_error_in_flipped_rays"] >= 0.8, (m, ev[m])
    assert ev["f16x2"]["flipped_rays"] <= 1.5 * ev["f32"]["flipped_rays"] + 2, ev
    assert ev["rays_flipped_in_no_arithmetic"] >= 0.85 * ev["rays_checked"]


def test_in_place_plane_update_between_forward_and_backward_is_refused(hip):
    """_RenderRaysFn keeps what its backward reads through save_for_backward: an in-place update of a plane parameter after the
    forward and before the backward -- the gradient would belong to another function -- raises autograd's version-check error, like any torch
    operator in the reference's graph (train_nerf.py:860-906) would."""
    from bench import make_synthetic_scene, render_options
    mc, mf, sid, pose = make_synthetic_scene(DEV, plane_res=48, view_res=16, seed=6, channels_last=True)
    for m in (mc, mf):
        for n, p in m.named_parameters():
            p.requires_grad_("planes_" in n)
        m.train()
    H = W = 32
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    ro, rd_ = hip.nerf_helpers.get_ray_bundle(H, W, focal, pose)
    batch = torch.stack([ro.reshape(-1, 3)[:256], rd_.reshape(-1, 3)[:256]], 0)
    opts, scfg = render_options(16, 16)
    out = hip.train_utils.run_one_iter_of_nerf(H, W, focal, mc, mf, batch, opts, sid, mode="train", scene_config=scfg, randoms={})
    (out[0].sum() + out[3].sum()).backward()                     # untouched: fine
    out = hip.train_utils.run_one_iter_of_nerf(H, W, focal, mc, mf, batch, opts, sid, mode="train", scene_config=scfg, randoms={})
    with torch.no_grad():
        next(iter(mc.planes_.values())).add_(1.0)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        (out[0].sum() + out[3].sum()).backward()


def _opaque_scene_g_raw(hip, rng, pr, N, S):
    """dL/draw of a training step on a scene made OPAQUE (sigma x 60): behind the surface the transmittance, and with it the gradient of a
    sample, falls by many decades INSIDE one 32-sample wave tile -- the dynamic range a tile's power-of-two scale has to carry."""
    from bench import make_synthetic_scene
    capi = hip.capi
    nv = torch.ops.nvsr
    mc, mf, sid, pose = make_synthetic_scene(DEV, plane_res=pr, view_res=8, seed=int(rng.integers(1 << 20)), channels_last=True)
    H = W = 80
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, pose)
    rays = hip.train_utils.pack_rays(ro, rd, 2.0, 6.0)[torch.from_numpy(rng.integers(0, H * W, N)).to(DEV)].contiguous()
    z = torch.sort(torch.rand(N, S, device=DEV) * 4 + 2, -1).values.contiguous()
    sc, keep = mf.native_scene()
    raw = torch.empty((N, S, 4), device=DEV)
    gates = torch.zeros(N * S * 32, dtype=torch.int32, device=DEV)
    capi.call("nvsr_decode_rays_arith", C.byref(sc), capi.ptr(mf.packed_decoder()), N, S, capi.ptr(rays), capi.ptr(z), capi.ptr(raw),
              capi.ptr(gates), None, capi.ARITHMETIC["f32"], capi.stream())
    raw_o = raw.clone()
    raw_o[..., 3] = raw[..., 3].abs() * 60.0                          # every sample absorbs: opaque after a few of them
    g_rgb = (torch.randn(N, 3, device=DEV) * 2e-4).contiguous()       # 2 (x - t) / n of a 4 096-ray batch
    g_raw = nv.composite_backward(raw_o, z, rays[:, 3:6].contiguous(), None, False, False, g_rgb, None, None)
    return mf, sc, keep, rays, z, gates, g_raw


def test_f16_backward_with_the_dynamic_range_of_an_opaque_ray(hip):
    """round 3's f16-limb backward scaled a wave tile (32 samples of one ray) by the largest |dL/draw| of the TILE, so a
    sample 2^-14 below it kept ~11 bits and one 2^-24 below it was flushed -- and inside a ray dL/draw does span that: on an opaque scene it
    decays with the transmittance (the premise is asserted below).  Measured with that scheme on these inputs: relative L2 of the plane
    gradients 7.4e-6, worst texel 1.3 % (3-bf16-limb backward: 6.5e-7 / 0.1 %).  Round 4 scales every POINT (a column of every product of the
    chain: exact) and each of its two chains (density / rgb) by its own power of two, and puts the largest magnitude at 2^3 so that the low
    limbs of the last layers stay normal numbers: 1.1e-6 / 0.09 %, the 3-limb backward's level.  Asserted texel by texel against the exact-f32
    backward on the same gates:
      * every texel: |error| <= 5e-6 of the largest texel gradient; relative L2 <= 3e-6 and <= 2.5 x the 3-limb backward's + 5e-7;
      * texels that matter to an optimizer (|g| >= 1e-4 of the largest: Adam's eps = 1e-8 sits far above the rest): relative error <= 2e-3
        and <= 3 x the 3-limb backward's worst + 2e-4 (both carry the float atomics' ordering noise)."""
    capi = hip.capi
    lib = capi.lib()
    rng = np.random.default_rng(12)
    for pr, N, S in ((64, 2000, 128), (200, 1500, 64)):
        mf, sc, keep, rays, z, gates, g_raw = _opaque_scene_g_raw(hip, rng, pr, N, S)
        gr = g_raw.abs().reshape(N, -1, 32 * 4) if S % 32 == 0 else None
        if gr is not None:      # the premise: inside one wave tile the gradients really span > 2^24
            span = (gr.max(-1).values / gr.clamp_min(1e-45).min(-1).values).max()
            assert float(span) > 2.0 ** 24, float(span)
        res = {}
        for mode in ("f32", "bf16x3", "f16x2"):
            gpl = [torch.zeros_like(k) for k in keep]
            gptrs = (C.c_void_p * 4)(*[t.data_ptr() for t in gpl])
            vws = torch.zeros(lib.nvsr_view_grad_workspace_floats(N, S), device=DEV)
            capi.call("nvsr_render_pass_backward_gates_arith", C.byref(sc), capi.ptr(mf.packed_decoder()), capi.ptr(mf.packed_decoder_bwd()), N, S,
                      capi.ptr(rays), capi.ptr(z), capi.ptr(g_raw), capi.ptr(gates), gptrs, capi.ptr(vws), None, capi.ARITHMETIC[mode], capi.stream())
            torch.cuda.synchronize()
            res[mode] = torch.cat([g.reshape(-1).double() for g in gpl])
            assert torch.isfinite(res[mode]).all(), mode
        ref = res["f32"]
        top = float(ref.abs().max())
        big = ref.abs() >= 1e-4 * top
        stats = {}
        for m in ("bf16x3", "f16x2"):
            d = (res[m] - ref).abs()
            stats[m] = dict(l2=float(d.norm() / ref.norm()), abs_max=float(d.max() / top), rel_max_big=float((d[big] / ref[big].abs()).max()),
                            rel_p999_big=float(torch.quantile((d[big] / ref[big].abs())[:4000000], 0.999)))
        print("planes %d N %d S %d (%d texel values >= 1e-4 of the largest): %s" % (pr, N, S, int(big.sum()), stats))
        assert stats["f16x2"]["abs_max"] <= 5e-6 and stats["f16x2"]["l2"] <= 3e-6 and stats["f16x2"]["l2"] <= 2.5 * stats["bf16x3"]["l2"] + 5e-7, stats
        assert stats["f16x2"]["rel_max_big"] <= 2e-3 and stats["f16x2"]["rel_max_big"] <= 3.0 * stats["bf16x3"]["rel_max_big"] + 2e-4, stats


def test_f16_training_converges_like_bf16x3(hip):
    """a short optimisation of the planes (Adam, the bench's learning rate) towards the pixels of a ground-truth scene, in the
    default 2-f16-limb arithmetic and in 'bf16x3' from the same start, same pixels, same random inputs: the rendered PSNR after 150
    iterations agrees within 0.3 dB (the two runs differ by the arithmetic and by the float atomics' ordering; measured 17.18 against 17.16 dB
    after 150 iterations at the bench's learning rate, from 16.29) and has improved."""
    from bench import make_synthetic_scene, render_options
    H = W = 64
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    opts_eval, scfg = render_options(32, 32)
    gt_c, gt_f, sid, pose = make_synthetic_scene(DEV, plane_res=48, view_res=16, seed=31)
    ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, pose)
    with torch.no_grad():
        target = hip.train_utils.eval_nerf(H, W, focal, gt_c, gt_f, ro, rd, opts_eval, scene_id=sid, scene_config=scfg)[3].contiguous()
    final, first = {}, {}
    for mode in ("f16x2", "bf16x3"):
        s = _train_setup(hip, {"LR_planes"}, seed=31, plane_res=48, n_rays=1024)
        with torch.no_grad():                       # same decoders as the ground truth, planes perturbed: the optimisation has a known answer
            for p in s["planes"]:
                p.add_(0.3 * torch.randn(p.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5)))
        s["mc"].arithmetic = s["mf"].arithmetic = mode
        with torch.no_grad():
            img = hip.train_utils.eval_nerf(H, W, focal, s["mc"], s["mf"], ro, rd, opts_eval, scene_id=s["sid"], scene_config=s["scfg"])[3]
        first[mode] = float(-10.0 * torch.log10(torch.mean((img - target) ** 2)))
        g = torch.Generator(device=DEV).manual_seed(9)
        for it in range(150):
            rnd = dict(t_rand=torch.rand(1024, 32, device=DEV, generator=g), u=torch.rand(1024, 32, device=DEV, generator=g),
                       noise_coarse=0.05 * torch.randn(1024, 32, device=DEV, generator=g), noise_fine=0.05 * torch.randn(1024, 64, device=DEV, generator=g))
            m = s["step"](it, target, s["pose"], H, W, focal, 1, s["sid"], s["scfg"], 1024, randoms=rnd)
        assert np.isfinite(m["loss"])
        with torch.no_grad():
            img = hip.train_utils.eval_nerf(H, W, focal, s["mc"], s["mf"], ro, rd, opts_eval, scene_id=s["sid"], scene_config=s["scfg"])[3]
        final[mode] = float(-10.0 * torch.log10(torch.mean((img - target) ** 2)))
    print("rendered PSNR against the ground-truth view before %s and after 150 iterations %s" % (first, final))
    assert abs(final["f16x2"] - final["bf16x3"]) <= 0.3, final
    assert min(final.values()) >= max(first.values()) + 0.5, (first, final)


# ---------------------------------------------------------------------------------------------------------------------------------
# TwoDimPlanesModel options no shipped YAML sets (VERDICT r3 missing #4): through the generic kernels, against the reference (g22)
# ---------------------------------------------------------------------------------------------------------------------------------
G22_SID = "lego_DS8_PlRes9_5"


def _g22_model(hip, g, name, n_planes=3, state_prefix=None, **over):
    from test_oracle import G22_BOX, G22_KW
    from test_hip_parity import T
    pre = name + ".sd." if state_prefix is None else state_prefix
    sd = {k[len(pre):]: v for k, v in g.items() if k.startswith(pre)}
    rots = torch.nn.ParameterList([torch.nn.Parameter(torch.as_tensor(sd["coord_projector.rot_mats_NON_LEARNED.%d" % d])) for d in range(n_planes)])
    kw = dict(G22_KW)
    kw.update(over)
    m = hip.models.TwoDimPlanesModel(use_viewdirs=True, num_planes_or_rot_mats=rots, **kw)
    m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}, strict=True)
    m = m.to(DEV).eval()
    m.planes_ = torch.nn.ParameterDict({hip.models.get_plane_name(G22_SID, d): torch.nn.Parameter(T(g["%s.plane%d" % (name, d)])) for d in range(n_planes + 1)})
    m.box_coords = {G22_SID: torch.as_tensor(G22_BOX, dtype=torch.float64)}
    m.set_cur_scene_id(G22_SID)
    return m


def _g22_check_forward_and_gradients(hip, g, name, m, n_planes, before_forward=lambda: None):
    from test_hip_parity import N_, T
    params = m.decoder_parameters()
    plist = [m.planes_[hip.models.get_plane_name(G22_SID, d)] for d in range(n_planes + 1)]
    for t_ in list(params) + plist:
        t_.requires_grad_(True)
    before_forward()
    out = m(T(g[name + ".x"]))
    ref = g[name + ".out"]
    np.testing.assert_allclose(N_(out), ref, rtol=0, atol=2e-5 * max(1.0, float(np.abs(ref).max())), err_msg=name)
    (out * T(g[name + ".gout"])).sum().backward()
    gnat, refn = np.concatenate([N_(t_.grad).reshape(-1) for t_ in params]), g[name + ".gnat"]
    assert gnat.shape == refn.shape and np.linalg.norm(gnat - refn) / np.linalg.norm(refn) < 2e-5, name
    np.testing.assert_allclose(gnat, refn, rtol=0, atol=3e-5 * float(np.abs(refn).max()), err_msg=name + " decoder")
    for d in range(n_planes + 1):
        got, refp = N_(plist[d].grad), g[name + ".gplane%d" % d]
        assert got.shape == refp.shape and np.linalg.norm(got - refp) / np.linalg.norm(refp) < 2e-5, (name, d)
        np.testing.assert_allclose(got, refp, rtol=0, atol=3e-5 * float(np.abs(refp).max()), err_msg="%s plane %d" % (name, d))


def test_grid_sample_without_align_corners_vs_reference(hip):
    """TwoDimPlanesModel(align_corners=False) (models.py:303-309,320-326: -1 / +1 are the outer edges of the corner texels): forward and the
    gradients of planes and decoder against the reference's own (g22); not a geometry of the MFMA kernels -> the generic kernels"""
    from conftest import load_golden
    g = load_golden("g22_model_options.npz")
    m = _g22_model(hip, g, "align_false", align_corners=False, dec_channels=64)
    assert not m.is_native_geometry()
    _g22_check_forward_and_gradients(hip, g, "align_false", m, 3)
    # the same model with align_corners=True answers differently by far more than the tolerance
    m2 = _g22_model(hip, g, "align_false", align_corners=True, dec_channels=64)
    from test_hip_parity import N_, T
    with torch.no_grad():
        assert np.abs(N_(m2(T(g["align_false.x"]))) - g["align_false.out"]).max() > 1e-3


def test_five_position_planes_with_random_frames_vs_reference(hip):
    """num_planes_or_rot_mats = 5 (models.py:140,471-490): CoordProjector's random orthonormal frames (the reference's, from the fixture),
    'avg' over five planes for the density decoder, 'concat_pos' of five planes + the view plane for the colour decoder: forward and
    gradients of all six planes and of the decoder against the reference's (g22), and an eval_nerf render of a coarse / fine pair"""
    from conftest import load_golden
    from test_hip_parity import N_, T, make_options, psnr
    g = load_golden("g22_model_options.npz")
    m = _g22_model(hip, g, "planes5", n_planes=5, dec_channels=64)
    assert m.num_density_planes == 5 and not m.is_native_geometry()
    _g22_check_forward_and_gradients(hip, g, "planes5", m, 5)
    mc = _g22_model(hip, g, "planes5", n_planes=5, state_prefix="planes5.render.coarse.", dec_channels=64)
    mf = _g22_model(hip, g, "planes5", n_planes=5, state_prefix="planes5.render.fine.", dec_channels=64)
    mf.planes_ = mc.planes_
    H, W, focal = int(g["planes5.render.hwf"][0]), int(g["planes5.render.hwf"][1]), float(g["planes5.render.hwf"][2])
    pose = load_golden("g18_decoder_variants.npz")["pose"]
    ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, T(pose))
    opts, scfg = make_options(16, 16)
    rgb_c, _, _, rgb_f, *_ = hip.train_utils.eval_nerf(H, W, focal, mc, mf, ro, rd, opts, scene_id=G22_SID, scene_config=scfg)
    np.testing.assert_allclose(N_(rgb_c), g["planes5.render.rgb_coarse"], rtol=0, atol=3e-5)
    ef = np.abs(N_(rgb_f) - g["planes5.render.rgb_fine"]).max(-1)
    assert (ef <= 2e-4).mean() >= 0.95 and psnr(N_(rgb_f), g["planes5.render.rgb_fine"]) >= 70.0, ((ef <= 2e-4).mean(), ef.max())
    # more planes than the scene struct holds: refused loudly
    with pytest.raises((NotImplementedError, AssertionError)):
        hip.models.TwoDimPlanesModel(use_viewdirs=True, num_planes_or_rot_mats=16, proj_combination="avg",
                                     viewdir_proj_combination="concat_pos").generic_geometry()


def test_point_coords_noise_vs_reference(hip):
    """point_coords_noise (models.py:291-293): a training-mode forward adds N(0, (noise * 2 / (1 + plane resolution))^2) to the normalised
    sample positions, drawn by torch.normal from the CPU generator.  (a) one seeded model call: the jitter drawn here is the reference's,
    output and gradients are the reference's (g22); evaluation mode does not jitter and runs the MFMA kernels; (b) one training iteration
    through run_one_iter_of_nerf over three ray chunks and 44 network batches per chunk, with perturbed depths and density noise: every
    random tensor is drawn in the reference's order, so the rendered colours and the plane gradients are the reference's."""
    from conftest import load_golden
    from test_hip_parity import N_, T, Opt
    g = load_golden("g22_model_options.npz")
    pcn = float(g["noise.point_coords_noise"])
    m = _g22_model(hip, g, "noise", point_coords_noise=pcn)
    assert m.is_native_geometry() and m.jitter_std() == 0.0               # evaluation mode: no jitter, the shipped geometry
    m.train()
    assert not m.is_native_geometry() and abs(m.jitter_std() - float(g["noise.std"])) < 1e-12
    torch.manual_seed(2231)
    np.testing.assert_array_equal(N_(torch.normal(mean=0, std=m.jitter_std(), size=[g["noise.x"].shape[0], 3])), g["noise.jitter"])
    _g22_check_forward_and_gradients(hip, g, "noise", m, 3, before_forward=lambda: torch.manual_seed(2231))
    m.eval()
    with torch.no_grad():
        clean = N_(m(T(g["noise.x"])))
    assert np.abs(clean - g["noise.out"]).max() > 1e-3                    # (the jitter is no rounding matter)

    # (b) a training iteration
    nc, nf, std, chunk, H, W, focal = (float(v) for v in g["noise.train.params"])
    nc, nf, chunk, H, W = int(nc), int(nf), int(chunk), int(H), int(W)
    mc = _g22_model(hip, g, "noise", state_prefix="noise.train.coarse.", point_coords_noise=pcn)
    mf = _g22_model(hip, g, "noise", state_prefix="noise.train.fine.", point_coords_noise=pcn)
    mf.planes_ = mc.planes_
    mode = Opt(chunksize=chunk, perturb=True, num_coarse=nc, num_fine=nf, white_background=False, radiance_field_noise_std=std, lindisp=False)
    opts, scfg = Opt(nerf=Opt(use_viewdirs=True, train=mode, validation=mode)), Opt(near=2.0, far=6.0, no_ndc=True)
    mc.train(); mf.train()
    plist = [mc.planes_[hip.models.get_plane_name(G22_SID, d)] for d in range(4)]
    for t_ in plist:
        t_.requires_grad_(True)
    torch.manual_seed(2233)
    np.random.seed(2233)
    rc, _, _, rf, *_ = hip.train_utils.run_one_iter_of_nerf(H, W, focal, mc, mf, T(g["noise.train.rays"]), opts, G22_SID, mode="train", scene_config=scfg)
    np.testing.assert_allclose(N_(rc), g["noise.train.rgb_coarse"], rtol=0, atol=3e-5)
    ef = np.abs(N_(rf) - g["noise.train.rgb_fine"]).max(-1)
    assert (ef <= 2e-4).mean() >= 0.95, ((ef <= 2e-4).mean(), ef.max())
    target = T(g["noise.train.target"])
    loss = torch.nn.functional.mse_loss(rc, target) + torch.nn.functional.mse_loss(rf, target)
    assert abs(float(loss.detach()) - float(g["noise.train.loss"])) < 2e-4
    loss.backward()
    for d in range(4):
        got, ref = N_(plist[d].grad), g["noise.train.grad_plane%d" % d]
        assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 1e-2, (d, np.linalg.norm(got - ref) / np.linalg.norm(ref))    # (g11's bar)


def test_training_through_super_resolved_planes_on_a_generic_geometry(hip, oracle):
    """models.py:270-284 for a decoder geometry the MFMA kernels are not compiled for (dec_channels 64, 'sum' + 'concat_pos'): a training-mode
    model call super-resolves the region its points cover as part of the graph, and backward() reaches the EDSR weights and the LR planes.
    Checked without any backward oracle: the directional derivative of sum(out * G) along a random direction of the SR weights (and of one LR
    plane), by central differences of the FORWARD oracles (C planes_sr + the float64 decoder restatement), against <gradient, direction>."""
    from oracle.generic_decoder import decode
    from test_hip_parity import N_, T
    from test_oracle import G22_BOX
    rng = np.random.default_rng(71)
    R, Rv, hid, nb, C_ = 20, 6, 16, 2, 48
    kw = dict(skip_connect_every=3, proj_combination="sum", viewdir_proj_combination="concat_pos")
    torch.manual_seed(8)
    m = hip.models.TwoDimPlanesModel(use_viewdirs=True, dec_channels=64, **kw).to(DEV)
    sid = "lego_DS8_PlRes20_6"
    planes = [rng.standard_normal((1, C_, R, R), dtype=np.float32) * 0.5 for _ in range(3)] + [rng.standard_normal((1, C_, Rv, Rv), dtype=np.float32) * 0.5]
    m.planes_ = torch.nn.ParameterDict({hip.models.get_plane_name(sid, d): torch.nn.Parameter(T(planes[d])) for d in range(4)})
    m.box_coords = {sid: torch.as_tensor(G22_BOX, dtype=torch.float64)}
    m.set_cur_scene_id(sid)
    sr = hip.models.PlanesSR(hip.models.EDSR, 4, C_, C_, {"model": {"hidden_size": hid, "n_blocks": nb}}, "bilinear").to(DEV)
    with torch.no_grad():
        for p_ in sr.parameters():
            p_.mul_(10.0)
    m.assign_SR_model(sr, SR_viewdir=False)
    m.assign_LR_planes()
    assert not m.is_native_geometry()
    for p_ in list(m.decoder_parameters()) + list(m.planes_.values()):
        p_.requires_grad_(False)
    lr_names = [hip.models.get_plane_name(sid, d) for d in range(3)]
    for n_ in lr_names:
        sr.LR_planes[n_].requires_grad_(True)
    m.train(); sr.train()
    P = 300
    x = np.concatenate([rng.uniform(-1.5, 2.0, (P, 3)), rng.standard_normal((P, 3))], 1).astype(np.float32)     # a sub-box: the ROI is partial
    G = rng.standard_normal((P, 4)).astype(np.float32)
    out = m(T(x))
    (out * T(G)).sum().backward()
    got_w = np.concatenate([N_(w.grad).reshape(-1) for w in sr.inner_model.conv_weights()]).astype(np.float64)
    got_lr = N_(sr.LR_planes[lr_names[1]].grad).astype(np.float64)
    assert np.isfinite(got_w).all() and np.abs(got_w).max() > 0 and np.isfinite(got_lr).all() and np.abs(got_lr).max() > 0

    # ---- forward oracle of the same call as a function of (SR weight blob, LR plane 1)
    blob0 = np.concatenate([N_(w).reshape(-1) for w in sr.inner_model.conv_weights()])
    pad, over = int(sr.inner_model.required_padding), int(sr.HR_overpadding)
    sd_ = {k: N_(v) for k, v in m.state_dict().items() if "planes_" not in k and "SR_model" not in k}
    lo, rng_ = G22_BOX[0, :3].astype(np.float32), (G22_BOX[1, :3] - G22_BOX[0, :3]).astype(np.float32)
    n3 = (2 * (x[:, :3] - lo) / rng_ - 1).astype(np.float32)
    rois = []
    for d in range(3):
        grid = n3 @ N_(m.coord_projector.rot_mats_NON_LEARNED[d]).astype(np.float32)[:, 1:]
        rois.append(np.array([[grid[:, 1].min(), grid[:, 0].min()], [grid[:, 1].max(), grid[:, 0].max()]], np.float32))

    def loss(blob, lr1):
        hr = []
        for d in range(3):
            lr_d = lr1 if d == 1 else planes[d][0]
            h = oracle.planes_sr(lr_d, blob.astype(np.float32), hid, nb, 2, pad, over, roi=rois[d])
            assert np.isnan(h).any()                                  # the ROI path
            hr.append(np.nan_to_num(h)[None])
        o = decode(sd_, hr + [planes[3]], G22_BOX, x, dec_channels=64, **kw)
        return float((o * G.astype(np.float64)).sum())

    o0 = decode(sd_, [np.nan_to_num(oracle.planes_sr(planes[d][0], blob0, hid, nb, 2, pad, over, roi=rois[d]))[None] for d in range(3)] + [planes[3]],
                G22_BOX, x, dec_channels=64, **kw)
    np.testing.assert_allclose(N_(out), o0, rtol=0, atol=3e-5 * max(1.0, float(np.abs(o0).max())))
    dw = rng.standard_normal(blob0.shape) * np.abs(blob0).mean()
    eps = 2e-4                 # (the loss is piecewise linear in the weights -- ReLUs in the SR network and the decoder --: central differences across
                               #  kinks carry an O(eps) error; 2e-2 ... 1e-3 scatter by 3 %, 2e-4 is within 0.3 % of the gradient)
    fd_w = (loss(blob0 + eps * dw, planes[1][0]) - loss(blob0 - eps * dw, planes[1][0])) / (2 * eps)
    assert abs(fd_w - got_w @ dw) <= 2e-2 * abs(fd_w) + 1e-6, (fd_w, got_w @ dw)
    dl = rng.standard_normal(planes[1][0].shape).astype(np.float32) * 0.5
    fd_l = (loss(blob0, planes[1][0] + np.float32(eps) * dl) - loss(blob0, planes[1][0] - np.float32(eps) * dl)) / (2 * eps)
    assert abs(fd_l - (got_lr.reshape(-1) @ dl.reshape(-1).astype(np.float64))) <= 2e-2 * abs(fd_l) + 1e-6, (fd_l, got_lr.reshape(-1) @ dl.reshape(-1))


def test_planes_sr_without_align_corners_vs_reference(hip):
    """PlanesSR(align_corners=False) (models.py:858-859: the bilinear residual maps pixel centres): the full plane in evaluation mode, the ROI
    and the full plane in training mode with the gradients of the EDSR weights and of the LR plane, against the reference's (g22, the network
    of g09); a second PlanesSR with align_corners=True in the same process keeps answering g09's values (the flag is per call)"""
    from conftest import load_golden
    from test_hip_parity import N_, T, _rel, _sr_grad_blob, _sr_model
    g9, g = load_golden("g09_edsr.npz"), load_golden("g22_model_options.npz")
    sr, _ = _sr_model(hip, g9)
    sr.align_corners = False
    other, _ = _sr_model(hip, g9)
    sr.eval(); other.eval()
    sr.set_LR_plane(T(g9["lr"]), id="p", save_interpolated=False)
    other.set_LR_plane(T(g9["lr"]), id="p", save_interpolated=False)
    with torch.no_grad():
        full = N_(sr("p"))
        np.testing.assert_allclose(N_(other("p")), g9["sr_full"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(full, g["sr_align_false.full"], rtol=0, atol=1e-5)
    assert np.abs(full - g9["sr_full"]).max() > 1e-2                   # (the option is no rounding matter)
    sr.train()
    for tag, roi in (("roi", T(g["sr_align_false.roi"])), ("full", None)):
        lr = torch.nn.Parameter(T(g9["lr"]))
        sr.clear_SR_planes(all_planes=True)
        sr.set_LR_plane(lr, id="p", save_interpolated=False)
        sr.zero_grad(set_to_none=True)
        out = sr(("p", roi)) if roi is not None else sr("p")
        ref = g["sr_align_false.%s_out" % tag]
        valid = ~torch.isnan(out)
        assert np.array_equal(N_(valid), ~np.isnan(ref))
        np.testing.assert_allclose(np.nan_to_num(N_(out)), np.nan_to_num(ref), rtol=0, atol=1e-5)
        with torch.no_grad():
            other.clear_SR_planes()
            other("p")                                                # an align_corners=True call between this forward and its backward
        (torch.where(valid, out, torch.zeros_like(out)) * T(g["sr_align_false.%s_gout" % tag])).sum().backward()
        assert _rel(_sr_grad_blob(sr), g["sr_align_false.%s_gw" % tag]) < 2e-5, tag
        assert _rel(N_(lr.grad), g["sr_align_false.%s_glr" % tag]) < 2e-5, tag
        np.testing.assert_allclose(N_(lr.grad), g["sr_align_false.%s_glr" % tag], rtol=0, atol=2e-5 * np.abs(g["sr_align_false.%s_glr" % tag]).max())


def test_edsr_with_a_receptive_field_bound_vs_reference(hip):
    """EDSR(receptive_field_bound=8) (models.py:793-798): conv_input and the first block stay 3 x 3, the second block, conv_mid and the first
    up-scaling convolution fall back to 1 x 1, the second up-scaling convolution is 3 x 3 again (its increment is halved), conv_output 1 x 1.
    State dict shapes, required_padding / HR_overpadding, the network alone (output + gradients of weights and input), PlanesSR on the full
    plane and on a region with the gradients of the weights and the LR plane -- against the reference (g22)."""
    from conftest import load_golden
    from test_hip_parity import N_, T, _rel
    g = load_golden("g22_model_options.npz")
    Cc, hid, nblocks, sf, R, bound, pad, over = [int(v) for v in g["rf_bound.cfg"]]
    sr = hip.models.PlanesSR(hip.models.EDSR, sf, Cc, Cc, {"model": {"hidden_size": hid, "n_blocks": nblocks, "receptive_field_bound": bound}}, "bilinear")
    assert [int(w.shape[-1]) for w in sr.inner_model.conv_parameters()] == [int(k) for k in g["rf_bound.kernel_sizes"]]
    assert sr.inner_model.required_padding == pad and sr.HR_overpadding == over
    sr.load_state_dict({k[len("rf_bound.sd."):]: torch.as_tensor(v) for k, v in g.items() if k.startswith("rf_bound.sd.")}, strict=True)
    sr = sr.to(DEV)
    blob = lambda: np.concatenate([N_(w.grad).reshape(-1) for w in sr.inner_model.conv_parameters()])
    # the network alone
    sr.train()
    x = T(g["rf_bound.edsr_in"]).requires_grad_(True)
    out = sr.inner_model(x)
    assert tuple(out.shape) == g["rf_bound.edsr_out"].shape
    np.testing.assert_allclose(N_(out), g["rf_bound.edsr_out"], rtol=0, atol=1e-5)
    (out * T(g["rf_bound.edsr_gout"])).sum().backward()
    assert _rel(blob(), g["rf_bound.edsr_gw"]) < 2e-5 and _rel(N_(x.grad), g["rf_bound.edsr_gin"]) < 2e-5
    with torch.no_grad():
        np.testing.assert_allclose(N_(sr.inner_model(T(g["rf_bound.edsr_in"]))), g["rf_bound.edsr_out"], rtol=0, atol=1e-5)     # (the no-grad path)
    # PlanesSR: full plane, evaluation
    sr.eval()
    sr.set_LR_plane(T(g["rf_bound.lr"]), id="p", save_interpolated=False)
    with torch.no_grad():
        np.testing.assert_allclose(N_(sr("p")), g["rf_bound.full"], rtol=0, atol=1e-5)
    # PlanesSR: region of interest, training
    sr.train()
    lr = torch.nn.Parameter(T(g["rf_bound.lr"]))
    sr.clear_SR_planes(all_planes=True)
    sr.set_LR_plane(lr, id="p", save_interpolated=False)
    sr.zero_grad(set_to_none=True)
    out = sr(("p", T(g["rf_bound.roi"])))
    ref = g["rf_bound.roi_out"]
    valid = ~torch.isnan(out)
    assert np.array_equal(N_(valid), ~np.isnan(ref))
    np.testing.assert_allclose(np.nan_to_num(N_(out)), np.nan_to_num(ref), rtol=0, atol=1e-5)
    (torch.where(valid, out, torch.zeros_like(out)) * T(g["rf_bound.roi_gout"])).sum().backward()
    assert _rel(blob(), g["rf_bound.roi_gw"]) < 2e-5 and _rel(N_(lr.grad), g["rf_bound.roi_glr"]) < 2e-5


def test_one_decoder_for_both_passes_reuses_the_coarse_outputs(hip, oracle, monkeypatch):
    """models.fine.type == 'use_same' (train_nerf.py:353-355: model_fine IS model_coarse): the fine pass needs the decoder at
    sort(cat(z_coarse, z_samples)), a third of which the coarse pass has just evaluated with the same decoder on the same planes.
    nvsr_render_rays_shared_arith evaluates the importance samples only, merges the two lists and composites: the frame must agree with the
    path that recomputes everything (NVSR_NO_SHARED_DECODER=1) to compositing rounding, and with the oracle's render of the same model pair
    like any other frame; the coarse image is the same bits; white background, lindisp and density noise go through the same merge."""
    from bench import make_synthetic_scene, render_options
    from oracle.oracle import decoder_blob
    from test_hip_parity import N_
    mc, _, sid, pose = make_synthetic_scene(DEV, plane_res=64, view_res=16, seed=3)
    H = W = 256                                                         # 65 536 rays: a fused frame (>= nvsr_fused_min_rays)
    assert H * W >= hip.capi.fused_min_rays()
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, pose)
    tu = hip.train_utils
    for white, lindisp in ((False, False), (True, True)):
        opts, scfg = render_options(64, 128, white=white)
        opts.nerf.validation.lindisp = lindisp
        monkeypatch.delenv("NVSR_NO_SHARED_DECODER", raising=False)
        rgb_c, _, _, rgb_f, *_ = tu.eval_nerf(H, W, focal, mc, mc, ro, rd, opts, scene_id=sid, scene_config=scfg)
        monkeypatch.setenv("NVSR_NO_SHARED_DECODER", "1")
        ref_c, _, _, ref_f, *_ = tu.eval_nerf(H, W, focal, mc, mc, ro, rd, opts, scene_id=sid, scene_config=scfg)
        assert torch.equal(rgb_c, ref_c)
        d = (rgb_f - ref_f).abs()
        assert float(d.max()) <= 2e-5 and not torch.equal(rgb_f, torch.zeros_like(rgb_f)), float(d.max())
    # against the oracle (rows of the last frame: white background, lindisp)
    planes = [N_(mc.planes_[hip.models.get_plane_name(sid, d_)]) for d_ in range(4)]
    sc = oracle.scene(planes, mc.box_coords[sid].numpy())
    dec = oracle.decoder(decoder_blob({k: N_(v) for k, v in mc.state_dict().items()}))
    rows = slice(120 * W, 123 * W)
    rays = oracle.pack_rays(N_(ro).reshape(-1, 3)[rows], N_(rd).reshape(-1, 3)[rows], 2.0, 6.0)
    o = oracle.render_rays(sc, dec, dec, rays, 64, 128, lindisp=True, white_background=True)
    ef = np.abs(N_(rgb_f).reshape(-1, 3)[rows] - o["rgb_fine"]).max(-1)
    assert (ef <= 2e-4).mean() >= 0.98 and ef.max() <= 5e-3, ((ef <= 2e-4).mean(), ef.max())
    # density noise: explicit random inputs through run_one_iter_of_nerf (the merge gathers decoder outputs; the noise is added when compositing)
    monkeypatch.delenv("NVSR_NO_SHARED_DECODER", raising=False)
    opts, scfg = render_options(64, 128, noise=0.5)
    N = H * W
    g = torch.Generator(device=DEV).manual_seed(3)
    rnd = dict(noise_coarse=0.5 * torch.randn(N, 64, device=DEV, generator=g), noise_fine=0.5 * torch.randn(N, 192, device=DEV, generator=g))
    batch = torch.stack([ro.reshape(-1, 3), rd.reshape(-1, 3)], 0)
    a = tu.run_one_iter_of_nerf(H, W, focal, mc, mc, batch, opts, sid, mode="validation", scene_config=scfg, randoms=rnd)
    monkeypatch.setenv("NVSR_NO_SHARED_DECODER", "1")
    b = tu.run_one_iter_of_nerf(H, W, focal, mc, mc, batch, opts, sid, mode="validation", scene_config=scfg, randoms=rnd)
    assert torch.equal(a[0], b[0]) and float((a[3] - b[3]).abs().max()) <= 2e-5


def test_bicubic_planes_vs_reference(hip):
    """plane_interp='bicubic' (config/TrainModels.yml:72 lists it beside 'bilinear'): grid_sample(mode='bicubic', padding_mode='border') -- the
    coordinate is not clipped, each of the 4 x 4 taps is --, with and without align_corners: forward and the gradients of planes and decoder
    against the reference's (g22), through the generic kernels"""
    from conftest import load_golden
    from test_hip_parity import N_, T
    g = load_golden("g22_model_options.npz")
    for name, align in (("bicubic", True), ("bicubic_noalign", False)):
        m = _g22_model(hip, g, name, plane_interp="bicubic", align_corners=align, dec_channels=64)
        assert not m.is_native_geometry()
        _g22_check_forward_and_gradients(hip, g, name, m, 3)
    m2 = _g22_model(hip, g, "bicubic", plane_interp="bilinear", dec_channels=64)
    with torch.no_grad():
        assert np.abs(N_(m2(T(g["bicubic.x"]))) - g["bicubic.out"]).max() > 1e-3


def test_planes_sr_with_a_bicubic_residual_vs_reference(hip):
    """PlanesSR(plane_interp='bicubic') (models.py:858-859: the residual is F.interpolate(mode='bicubic')), with and without align_corners: the
    full plane in evaluation mode, a region in training mode with the gradients of the EDSR weights and of the LR plane (16 taps per HR texel),
    against the reference (g22, the network of g09)"""
    from conftest import load_golden
    from test_hip_parity import N_, T, _rel, _sr_grad_blob, _sr_model
    g9, g = load_golden("g09_edsr.npz"), load_golden("g22_model_options.npz")
    for tag, align in (("sr_bicubic", True), ("sr_bicubic_noalign", False)):
        sr, _ = _sr_model(hip, g9)
        sr.plane_interp, sr.align_corners = "bicubic", align
        sr.eval()
        sr.set_LR_plane(T(g9["lr"]), id="p", save_interpolated=False)
        with torch.no_grad():
            full = N_(sr("p"))
        np.testing.assert_allclose(full, g[tag + ".full"], rtol=0, atol=1e-5)
        assert np.abs(full - g9["sr_full"]).max() > 1e-2
        sr.train()
        lr = torch.nn.Parameter(T(g9["lr"]))
        sr.clear_SR_planes(all_planes=True)
        sr.set_LR_plane(lr, id="p", save_interpolated=False)
        sr.zero_grad(set_to_none=True)
        out = sr(("p", T(g[tag + ".roi"])))
        ref = g[tag + ".roi_out"]
        valid = ~torch.isnan(out)
        assert np.array_equal(N_(valid), ~np.isnan(ref))
        np.testing.assert_allclose(np.nan_to_num(N_(out)), np.nan_to_num(ref), rtol=0, atol=1e-5)
        (torch.where(valid, out, torch.zeros_like(out)) * T(g[tag + ".roi_gout"])).sum().backward()
        assert _rel(_sr_grad_blob(sr), g[tag + ".roi_gw"]) < 2e-5 and _rel(N_(lr.grad), g[tag + ".roi_glr"]) < 2e-5, tag


def test_sr_training_with_only_the_fine_model_super_resolving(hip, oracle):
    """The reference's default SR training (train_nerf.py:554-561, super_resolution.apply_2_coarse False -- the value of both shipped YAMLs): only
    model_fine gets the SR model, the coarse model samples the LR planes.
      sr_only  what: ['SR'] -- the coarse pass runs under torch.no_grad (optional_no_grad) and picks the importance samples, the fine loss reaches
               the EDSR weights through the super-resolved region;
      planes   the LR planes train along (SR network fed detached planes): the coarse loss reaches the LR planes, the fine loss the SR network --
               two sets of plane leaves in one iteration;
      joint    config/TrainModels.yml's what: ['LR_planes', 'decoder', 'SR'] -- both decoders, the SR network and the LR planes, which collect the
               coarse pass's gradient AND the fine pass's through the SR network.
    Oracle = its own SR forward, render backward on the HR planes (fine) / the LR planes (coarse) at the same importance depths, SR backward."""
    from conftest import load_golden
    from oracle.oracle import decoder_blob
    from test_hip_parity import N_, T, _decoder_grad_blob, _grad_models, _rel, _sr_grad_blob, make_options, sd
    g = load_golden("g08_render.npz")
    rng = np.random.default_rng(62)
    R, Rv, hid, nb = 24, 8, 16, 2
    planes = [rng.standard_normal((1, 48, R, R), dtype=np.float32) * 0.5 for _ in range(3)] + \
             [rng.standard_normal((1, 48, Rv, Rv), dtype=np.float32) * 0.5]
    sid = "lego_DS8_PlRes24_8"
    H = W = 12
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    N, nc, nf = 60, 24, 24
    opts, scfg = make_options(nc, nf)
    for mode in ("sr_only", "planes", "joint"):
        mc, mf = _grad_models(hip, g, planes, sid, what={"sr_only": (), "planes": ("planes",), "joint": ("planes", "decoder")}[mode])
        torch.manual_seed(6)
        sr = hip.models.PlanesSR(hip.models.EDSR, 4, 48, 48, {"model": {"hidden_size": hid, "n_blocks": nb}}, "bilinear").to(DEV)
        with torch.no_grad():
            for p_ in sr.parameters():
                p_.mul_(10.0)
        sr.train()
        mf.detach_LR_planes = mode != "joint"                       # (detached: the LR planes' gradient is the coarse pass's alone)
        mf.assign_SR_model(sr, SR_viewdir=False)
        mf.assign_LR_planes()
        assert not hasattr(mc, "SR_model")
        if mode == "sr_only":
            mc.optional_no_grad = torch.no_grad                     # train_nerf.py:560
        ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, T(g["pose"]))
        sel = torch.from_numpy(rng.permutation(H * W)[:N]).to(DEV)
        batch = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0)
        out = hip.train_utils.run_one_iter_of_nerf(H, W, focal, mc, mf, batch, opts, sid, mode="train", scene_config=scfg, randoms={})
        z_fine = N_(out[3].grad_fn.saved["z_f"])
        gc = T(rng.standard_normal((N, 3)).astype(np.float32) / N)
        gf = T(rng.standard_normal((N, 3)).astype(np.float32) / N)
        ((out[0] * gc).sum() * (0.0 if mode == "sr_only" else 1.0) + (out[3] * gf).sum()).backward()
        got = _sr_grad_blob(sr)
        # --- oracle
        blob = np.concatenate([N_(w).reshape(-1) for w in sr.inner_model.conv_weights()])
        pad, over = int(sr.inner_model.required_padding), int(sr.HR_overpadding)
        rays_np = oracle.pack_rays(N_(batch[0]), N_(batch[1]), 2.0, 6.0)
        box = np.asarray(g["box"], np.float64)
        ends = np.concatenate([rays_np[:, 0:3] + rays_np[:, 3:6] * rays_np[:, 6:7], rays_np[:, 0:3] + rays_np[:, 3:6] * rays_np[:, 7:8]], 0)
        n_ends = (2 * (ends - box[0, :3].astype(np.float32)) / (box[1, :3] - box[0, :3]).astype(np.float32) - 1).astype(np.float32)
        hr, rois = [], []
        for d in range(3):
            m_ = N_(mf.coord_projector.rot_mats_NON_LEARNED[d])[:, 1:]
            grid = n_ends @ m_
            roi = np.array([[grid[:, 1].min(), grid[:, 0].min()], [grid[:, 1].max(), grid[:, 0].max()]], np.float32)
            rois.append(roi)
            hr.append(oracle.planes_sr(planes[d][0], blob, hid, nb, 2, pad, over, roi=roi))
        hr_zero = [np.nan_to_num(h)[None] for h in hr] + [planes[3]]
        sc_hr, sc_lr = oracle.scene(hr_zero, g["box"]), oracle.scene(planes, g["box"])
        dec_c, dec_f = oracle.decoder(decoder_blob(sd(g, "coarse."))), oracle.decoder(decoder_blob(sd(g, "fine.")))
        o = oracle.render_rays(sc_lr, dec_c, dec_f, rays_np, nc, nf)
        np.testing.assert_allclose(N_(out[0]), o["rgb_coarse"], rtol=0, atol=3e-5)          # the coarse pass sampled the LR planes
        zero = np.zeros((N, 3), np.float32)
        gplanes = oracle.render_backward(sc_hr, [p.shape for p in hr_zero], dec_c, dec_f, rays_np, nc, nf, zero, N_(gf), z_fine=z_fine)
        ref = np.zeros_like(blob, dtype=np.float64)
        dlr = []
        for d in range(3):
            gw, dl = oracle.planes_sr_backward(planes[d][0], blob, hid, nb, 2, pad, over, gplanes[d], roi=rois[d], want_dlr=mode == "joint")
            ref += gw
            dlr.append(dl)
        assert _rel(got, ref) < 5e-3, "SR weight gradient (%s): relative L2 error %.2e" % (mode, _rel(got, ref))
        lr_params = [mc.planes_[hip.models.get_plane_name(sid, d)] for d in range(4)]
        if mode == "sr_only":
            assert all(p_.grad is None for p_ in lr_params) and not out[0].requires_grad
            continue
        glr = oracle.render_backward(sc_lr, [p.shape for p in planes], dec_c, dec_f, rays_np, nc, nf, N_(gc), zero, z_fine=z_fine)
        for d in range(3):
            want = glr[d] + (dlr[d].reshape(glr[d].shape) if mode == "joint" else 0.0)    # + the fine pass's share through the SR network
            assert _rel(N_(lr_params[d].grad)[0], want) < 5e-3, (mode, d, _rel(N_(lr_params[d].grad)[0], want))
        assert _rel(N_(lr_params[3].grad)[0], gplanes[3] + glr[3]) < 5e-3                    # the view plane: both passes
        if mode == "joint":
            gdc = oracle.render_backward_decoder(sc_lr, dec_c, dec_f, rays_np, nc, nf, N_(gc), zero, z_fine=z_fine)[0]
            gdf = oracle.render_backward_decoder(sc_hr, dec_c, dec_f, rays_np, nc, nf, zero, N_(gf), z_fine=z_fine)[1]
            assert _rel(_decoder_grad_blob(mc), gdc) < 1e-2 and _rel(_decoder_grad_blob(mf), gdf) < 1e-2, (
                _rel(_decoder_grad_blob(mc), gdc), _rel(_decoder_grad_blob(mf), gdf))


def test_default_sr_refinement_steps(hip):
    """training.TrainStep on the reference's default SR refinement (what: ['SR'], loss: 'fine', apply_2_coarse False: the SR model on the fine
    model only, the coarse pass under torch.no_grad -- train_nerf.py:554-561,883-889): the steps run, only the SR network moves, the fine
    loss falls, evaluate_view renders the scene with and without the SR model"""
    from conftest import load_golden
    from test_hip_parity import T, _grad_models, _gt_and_student, make_options
    g = load_golden("g11_grads.npz")
    sid = "lego_DS8_PlRes20_8"
    gt_planes, noisy = _gt_and_student(hip, g, sid, seed=82)
    H = W = 20
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    pose = T(load_golden("g08_render.npz")["pose"])
    opts, scfg = make_options(24, 24)
    mc, mf = _grad_models(hip, g, noisy, sid, what=())
    torch.manual_seed(8)
    sr = hip.models.PlanesSR(hip.models.EDSR, 4, 48, 48, {"model": {"hidden_size": 16, "n_blocks": 2}}, "bilinear").to(DEV)
    mf.assign_SR_model(sr, SR_viewdir=False)
    mf.assign_LR_planes()
    mc.optional_no_grad = torch.no_grad
    ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, pose)
    mf.skip_SR(True)
    with torch.no_grad():
        img = hip.train_utils.eval_nerf(H, W, focal, mc, mf, ro, rd, opts, scene_id=sid, scene_config=scfg)[3]   # target = the LR render
    mf.skip_SR(False)
    ev = hip.training.evaluate_view(mc, mf, opts, sid, scfg, img, pose, H, W, focal, SR_model=sr, sr_scene=True)
    sr_opt = torch.optim.Adam(sr.parameters(), lr=2e-3)
    step = hip.training.TrainStep(mc, mf, opts, {"SR"}, SR_optimizer=sr_opt, SR_model=sr, sr_loss="fine")
    np.random.seed(4)
    before = [p_.detach().clone() for p_ in sr.parameters()]
    for it in range(12):
        r = step(it, img, pose, H, W, focal, 1, sid, scfg, 200, sr_iter=True)
        assert r["coarse_loss"] is None and r["fine_loss"] is not None and np.isfinite(r["loss"])
    assert any(not torch.equal(a, b.detach()) for a, b in zip(before, sr.parameters()))
    assert all(p_.grad is None for m in (mc, mf) for p_ in list(m.decoder_parameters()) + list(m.planes_.values()))
    sr.clear_SR_planes()
    ev2 = hip.training.evaluate_view(mc, mf, opts, sid, scfg, img, pose, H, W, focal, SR_model=sr, sr_scene=True)
    assert ev2["loss"] < ev["loss"], (ev["loss"], ev2["loss"])


def test_image_consistency_iteration(hip):
    """An image-consistency iteration of config/RefineOnTestScene.yml (im_inconsistency_loss_w: 1; train_nerf.py:806-811,829-835,878-879): an LR
    target image, num_random_rays // ds^2 LR pixels drawn, each rendered as its ds x ds patch of HR rays through the super-resolved planes (SR
    model on the fine model only), the patches averaged back to LR pixels before the loss.  TrainStep's reported fine loss is the loss of that
    computation done by hand on the same draw; only the SR network receives a gradient and moves."""
    from conftest import load_golden
    from test_hip_parity import T, _grad_models, _gt_and_student, make_options
    tr = hip.training
    g = load_golden("g11_grads.npz")
    sid = "lego_DS8_PlRes20_8"
    _, noisy = _gt_and_student(hip, g, sid, seed=83)
    H = W = 10                                                      # the LR view; the iteration renders 40 x 40
    ds = 4
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    pose = T(load_golden("g08_render.npz")["pose"])
    opts, scfg = make_options(16, 16)
    mc, mf = _grad_models(hip, g, noisy, sid, what=())
    torch.manual_seed(9)
    sr = hip.models.PlanesSR(hip.models.EDSR, ds, 48, 48, {"model": {"hidden_size": 16, "n_blocks": 2}}, "bilinear").to(DEV)
    with torch.no_grad():
        for p_ in sr.parameters():
            p_.mul_(10.0)
    mf.assign_SR_model(sr, SR_viewdir=False)
    mf.assign_LR_planes()
    mc.optional_no_grad = torch.no_grad
    img = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(5)).to(DEV)
    n_rays = 320                                                    # -> 20 LR pixels x 16 HR rays
    sr.train()
    # by hand, on the draw the step will make
    np.random.seed(11)
    sel, target = tr.select_training_pixels(img, n_rays, ds)
    assert sel.shape == (320, 2) and target.shape == (20, 3) and int(sel.max()) < H * ds
    ro, rd = tr.get_ray_bundle_at(H * ds, W * ds, focal * ds, pose, sel, downsampling_offset=tr.downsampling_offset(ds // ds))
    with torch.no_grad():
        out = hip.train_utils.run_one_iter_of_nerf(H * ds, W * ds, focal * ds, mc, mf, (ro, rd), opts, sid, mode="train", scene_config=scfg)
    by_hand = float(torch.nn.functional.mse_loss(tr.avg_downsampling(out[3], ds), target))
    sr.clear_SR_planes()
    opt = torch.optim.SGD(sr.parameters(), lr=1e-3)
    step = tr.TrainStep(mc, mf, opts, {"SR"}, SR_optimizer=opt, SR_model=sr, sr_loss="fine", im_inconsistency_loss_w=1.0, ds_factor=ds)
    before = [p_.detach().clone() for p_ in sr.parameters()]
    np.random.seed(11)
    r = step(0, img, pose, H, W, focal, ds, sid, scfg, n_rays, sr_iter=True, im_consistency_iter=True)
    assert r["coarse_loss"] is None and abs(r["fine_loss"] - by_hand) <= 1e-5 * max(1.0, by_hand), (r["fine_loss"], by_hand)
    assert r["psnr"] is None                                        # (an image-consistency iteration logs no PSNR, train_nerf.py:893-899)
    assert any(not torch.equal(a, b.detach()) for a, b in zip(before, sr.parameters()))
    assert all(p_.grad is None for m in (mc, mf) for p_ in list(m.decoder_parameters()) + list(m.planes_.values()))
