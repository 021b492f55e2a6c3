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
