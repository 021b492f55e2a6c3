"""GPU parity tests added in round 6 (-m gpu): the batched SR-training path directly against the ORACLE at wide channels, the
side-stream prologue on inputs produced anew every iteration, the gradient all-reduce overlapped with the SR backward."""
import numpy as np
import pytest
import torch

from test_hip_parity import DEV, N_, T, _rel

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------------------------------------------------------------------------
# PlanesSR.forward_many (ragged conv3x3_limb16_kernel launches, the one-pass-over-three-crops weight gradient) met the
# oracle only through its plane-by-plane sibling (hidden 16 directly).  Here: hidden 128 (the 16x16x32 kernels), three different regions.
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["f16x2", "bf16x3", "f32"])
def test_batched_sr_training_vs_oracle(hip, oracle, mode):
    """forward_many outputs vs oracle.planes_sr(..., roi=) within 3e-5 of the output range; EDSR weight gradients and LR-plane gradients vs
    oracle.planes_sr_backward summed over the three planes (models.py:884-926 under autograd; one weight-gradient pass per layer over all
    crops in the product).  Three regions of different sizes, one touching two borders of its plane (replicate padding of the crop).
    Gradient tolerance: relative L2 1e-4 like test_sr_gradients_vs_oracle_larger (a ReLU input within fp32 rounding of zero gates differently
    in the double-accumulating oracle: 1 flip in ~1e5 activations moves one activation's worth of gradient)."""
    R, hid, nb = 24, 128, 2
    rois = [[-0.9, -0.35, 0.1, 0.8], [-1.0, -1.0, 0.2, 0.3], [-0.2, -0.6, 0.95, 0.4]]
    torch.manual_seed(61)
    sr = hip.models.PlanesSR(hip.models.EDSR, 4, 48, 48, {"model": {"hidden_size": hid, "n_blocks": nb}}, "bilinear").to(DEV)
    with torch.no_grad():
        for p_ in sr.parameters():
            p_.mul_(3.0)
    sr.train()
    sr.inner_model.arithmetic = mode
    gen = torch.Generator(device=DEV).manual_seed(62)
    lrs = [torch.nn.Parameter(torch.randn(1, 48, R, R, device=DEV, generator=gen) * 0.5) for _ in range(3)]
    for k, t in enumerate(lrs):
        sr.set_LR_plane(t, id="p%d" % k, save_interpolated=False)
    outs = sr.forward_many([("p%d" % k, rois[k]) for k in range(3)])
    gouts = [torch.randn(o.shape, device=DEV, generator=gen) for o in outs]
    sum((torch.nan_to_num(o) * w).sum() for o, w in zip(outs, gouts)).backward()
    got_w = torch.cat([(w.grad if w.grad is not None else torch.zeros_like(w)).reshape(-1) for w in sr.inner_model.conv_parameters()])

    blob = np.concatenate([N_(w).reshape(-1) for w in sr.inner_model.conv_weights()])
    pad, over = int(sr.inner_model.required_padding), int(sr.HR_overpadding)
    ref_w = np.zeros(blob.shape, np.float64)
    shapes = set()
    for k in range(3):
        roi = np.asarray(rois[k], np.float32).reshape(2, 2)
        ref = oracle.planes_sr(N_(lrs[k])[0], blob, hid, nb, 2, pad, over, roi=roi)
        got = N_(outs[k])[0]
        inside = ~np.isnan(ref)
        assert inside.any() and not inside.all() and np.array_equal(inside, ~np.isnan(got)), k        # same region, and really a crop
        shapes.add(tuple(int(v) for v in (inside.any(0).sum(), inside.any(1).sum())))
        rng_ = float(ref[inside].max() - ref[inside].min())
        err = float(np.abs(got[inside] - ref[inside]).max())
        assert err <= 3e-5 * rng_, (mode, k, err, rng_)
        gw, glr = oracle.planes_sr_backward(N_(lrs[k])[0], blob, hid, nb, 2, pad, over, N_(gouts[k])[0], roi=roi)
        ref_w += gw
        assert _rel(N_(lrs[k].grad)[0], glr) < 1e-4, (mode, k, _rel(N_(lrs[k].grad)[0], glr))
    assert len(shapes) == 3                                                                            # three different crop sizes: ragged launches
    assert _rel(N_(got_w), ref_w) < 1e-4, (mode, _rel(N_(got_w), ref_w))
    # per layer as well: a layer whose gradient is wrong must not hide behind the norm of the others
    off = 0
    for w in sr.inner_model.conv_weights():
        n = w.numel()
        assert _rel(N_(got_w[off: off + n]), ref_w[off: off + n]) < 3e-4, (mode, tuple(w.shape))
        off += n


# ---------------------------------------------------------------------------------------------------------------------------------
# the side-stream prologue of an SR-training iteration on a target image / pose produced anew every iteration
# ---------------------------------------------------------------------------------------------------------------------------------
def test_prologue_ahead_on_targets_produced_every_iteration(hip):
    """TrainStep._draw_rays reads the target image and the pose on a side stream.  A target computed on the iteration's stream each iteration
    (here: behind ~50 ms of queued work, into a block the allocator recycles, version 0 every time) must be waited for: the iteration has to
    draw the pixels of THIS iteration's image.  Compared with the same iterations run with everything on the iteration's stream."""
    from conftest import load_golden
    from test_hip_parity import _grad_models, _gt_and_student, make_options
    g = load_golden("g11_grads.npz")
    sid = "lego_DS8_PlRes20_8"
    H = W = 20
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    pose0 = T(load_golden("g08_render.npz")["pose"])
    opts, scfg = make_options(24, 24)
    busy = torch.randn(4096, 4096, device=DEV)
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
        step = hip.training.TrainStep(mc, mf, opts, {"SR"}, SR_optimizer=torch.optim.SGD(sr.parameters(), lr=0.0), SR_model=sr, sr_loss="fine",
                                      pixel_sampler=hip.training.DevicePixelSampler(seed=9))
        step.prologue_ahead = ahead
        losses = []
        for it in range(4):
            torch.manual_seed(12 + it)
            x = busy
            for _ in range(30):                       # ~50 ms of work queued on the iteration's stream in front of the producer
                x = x @ busy * 1e-3
            img = torch.full((H, W, 3), 0.2 * it, device=DEV) + x[:H, :W * 3].reshape(H, W, 3).clamp(0, 0) \
                + torch.rand(H, W, 3, generator=torch.Generator().manual_seed(5 + it)).to(DEV) * 0.1
            pose = pose0 + x[:4, :4].clamp(0, 0)      # a pose "computed" behind the same queue
            m = step(it, img, pose, H, W, focal, 1, sid, scfg, 150, sr_iter=True)
            losses.append(m["loss"])
            del img, pose
        res[ahead] = losses
    assert all(abs(a - b) <= 1e-5 * max(1.0, abs(b)) for a, b in zip(res[True], res[False])), res
    assert len(set(round(v, 6) for v in res[False])) == 4          # (the four targets really differ: a stale image would show)


# ---------------------------------------------------------------------------------------------------------------------------------
# the EDSR gradient all-reduced bucket by bucket while the SR backward runs (nvsr_planes_sr_backward_batch_marks)
# ---------------------------------------------------------------------------------------------------------------------------------
def test_overlapped_sr_gradient_sync_through_rccl_on_one_rank():
    """distributed.OverlappedSRGradSync on backend "nccl" (= RCCL) with a process group of ONE rank (all a test box has): the batched SR backward
    records one event per bucket where that suffix of the weight-gradient blob is final, the collective stream waits for each event and RCCL
    all-reduces the bucket in place while the remaining layers are computed, the iteration's stream waits for the collectives.  Sums over one
    rank are the inputs: the gradients (weights and LR planes) must equal the unmarked backward's (same kernels; to the order of the float atomics of
    the partial-sum reductions), three ragged crops, several buckets."""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import os, sys, socket
        sys.path.insert(0, %r)
        import torch, torch.distributed as dist
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        import nvsr_amd
        D = nvsr_amd.distributed
        rois = [[-0.9, -0.35, 0.1, 0.8], [-1.0, -1.0, 0.2, 0.3], [-0.2, -0.6, 0.95, 0.4]]
        res = {}
        for overlapped in (False, True):
            torch.manual_seed(31)
            sr = nvsr_amd.models.PlanesSR(nvsr_amd.models.EDSR, 4, 48, 48, {"model": {"hidden_size": 128, "n_blocks": 3}}, "bilinear").to(dev)
            sr.train()
            gen = torch.Generator(device=dev).manual_seed(32)
            lrs = [torch.nn.Parameter(torch.randn(1, 48, 24, 24, device=dev, generator=gen) * 0.5) for _ in range(3)]
            for k, t in enumerate(lrs):
                sr.set_LR_plane(t, id="p%%d" %% k, save_interpolated=False)
            sync = None
            if overlapped:
                sync = D.OverlappedSRGradSync(sr, other_parameters=lrs, bucket_bytes=1 << 20, single_rank_ok=True)
            outs = sr.forward_many([("p%%d" %% k, rois[k]) for k in range(3)])
            ws = [torch.randn(o.shape, device=dev, generator=gen) for o in outs]
            sum((torch.nan_to_num(o) * w).sum() for o, w in zip(outs, ws)).backward()
            if overlapped:
                assert sync.reduced_in_backward and sync.stats["buckets"] >= 3, sync.stats
                sync()                                                         # the other parameters (here: the LR planes) through allreduce_gradients
                assert not sync.reduced_in_backward
            torch.cuda.synchronize()
            res[overlapped] = ([w.grad.clone() for w in sr.inner_model.conv_parameters()], [t.grad.clone() for t in lrs])
        for k, (a, b) in enumerate(zip(res[True][0] + res[True][1], res[False][0] + res[False][1])):
            rel = float((a - b).norm() / b.norm().clamp_min(1e-30))
            assert float(b.abs().max()) > 0 and rel <= 2e-6, (k, rel, float(a.abs().max()), float(b.abs().max()))     # (the order of float atomics)
        dist.barrier(); dist.destroy_process_group()
        print("RCCL_OVERLAPPED_SYNC_OK")
    """ % root)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL_OVERLAPPED_SYNC_OK" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])


# ---------------------------------------------------------------------------------------------------------------------------------
# planes super-resolved by a fallback frame stay cached; an exception inside a frame leaves the network's arithmetic alone
# ---------------------------------------------------------------------------------------------------------------------------------
def test_fallback_frames_keep_their_super_resolved_planes(hip):
    """An SR network beyond NVSR_ARITH_F16X2's range (a trunk weight of 300): the first frame raises bit 2, super-resolves and renders again in 'bf16x3'.
    Every LATER frame of the same parameters goes to 'bf16x3' directly and must find the planes that frame cached (round 5 dropped them and ran the SR
    stage again on every frame).  The network's configured arithmetic is untouched in between, also when a launch inside the frame raises."""
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
    with torch.no_grad():
        sr.inner_model.residual[0].conv1.weight[17, 3, 1, 1] = 300.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        first = ev()
    assert torch.isfinite(first[3]).all() and sr.inner_model.arithmetic is None and sr.__dict__.get("_planes_arith") == "bf16x3"
    cached = {k: v for k, v in sr.SR_planes.items()}
    assert len(cached) == 3
    second = ev()
    assert torch.equal(second[3], first[3]) and sr.inner_model.arithmetic is None
    assert all(sr.SR_planes[k] is v for k, v in cached.items()), "the fallback frame's planes were super-resolved again"
    sr.clear_SR_planes()                                        # (the caller changed a plane: the tag goes with the planes)
    assert "_planes_arith" not in sr.__dict__
    # a launch that raises inside a fallback frame: the arithmetic is put back, the tag is not set
    real = hip.train_utils.predict_and_render_radiance
    def boom(*a, **k):
        raise RuntimeError("boom")
    hip.train_utils.predict_and_render_radiance = boom
    try:
        with pytest.raises(RuntimeError, match="boom"):
            ev()
    finally:
        hip.train_utils.predict_and_render_radiance = real
    assert sr.inner_model.arithmetic is None and "_planes_arith" not in sr.__dict__
    third = ev()
    assert torch.equal(third[3], first[3])
