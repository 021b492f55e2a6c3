"""GPU parity tests (-m gpu): the HIP path, called through the C ABI (ctypes) via the host-side mirror of the reference
interface, against (1) the golden vectors produced by the reference and (2) the C oracle on larger seeded inputs.

Stated tolerances (fp32 path, SURVEY.md 8d / DESIGN.md "Numerics"):
  ray generation ......................... bit-exact
  decoder output (raw rgb, sigma) ........ |err| <= 2e-5  (values are O(1); 10 chained fp32 GEMV layers, MFMA k-order vs sgemm)
  one render pass at identical depths .... |rgb|,|acc| <= 2e-5 ; rays whose LAST sample has |sigma| < 1e-4 are excluded
                                           (1e10 * sigma makes alpha a step function of sign(sigma) there)
  importance depths ...................... conditioned tolerance (conftest.sample_pdf_tolerance)
  end to end ............................. |rgb| <= 2e-4 and PSNR >= 80 dB vs the reference image
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, sample_pdf_tolerance
from oracle.oracle import decoder_blob

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a):
    return torch.as_tensor(np.ascontiguousarray(a), device=DEV)


def N_(t):
    return t.detach().cpu().numpy()


def assert_bits_equal(a, b):
    """bit-for-bit, i.e. also the sign of zero (atan2 in cart2az_el turns a -0.0 direction component into -pi instead of +pi)"""
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    assert a.shape == b.shape
    np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))


def sd(g, prefix):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


def psnr(a, b):
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return 200.0 if mse == 0 else -10.0 * np.log10(mse)


class Opt:
    """minimal stand-in for the CfgNode the reference passes around (attribute access)"""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def make_options(nc, nf, perturb=False, noise=0.0, white=False, lindisp=False):
    m = Opt(chunksize=131072, perturb=perturb, num_coarse=nc, num_fine=nf, white_background=white,
            radiance_field_noise_std=noise, lindisp=lindisp)
    return Opt(nerf=Opt(use_viewdirs=True, train=m, validation=m)), Opt(near=2.0, far=6.0, no_ndc=True)


def build_model(hip, state, planes, box, sid="lego_DS8_PlRes32_8"):
    m = hip.models.TwoDimPlanesModel(use_viewdirs=True, skip_connect_every=3, proj_combination="avg",
                                     viewdir_proj_combination="concat_pos", align_corners=True)
    missing = m.load_state_dict({k: torch.as_tensor(v) for k, v in state.items()}, strict=True)
    m = m.to(DEV)
    m.planes_ = torch.nn.ParameterDict({hip.models.get_plane_name(sid, d): torch.nn.Parameter(T(planes[d])) for d in range(4)})
    m.box_coords = {sid: torch.as_tensor(box, dtype=torch.float64)}
    m.set_cur_scene_id(sid)
    m.eval()
    return m, sid


# ---------------------------------------------------------------------------------------------------------------------
DEFAULT_ARITHMETIC = "f16x2"       # include/nvsr.h: NVSR_ARITH_DEFAULT


def test_library_loaded_and_version(hip):
    assert hip.capi.lib().nvsr_version() >= 100


def test_ray_bundle_bit_exact(hip):
    g = load_golden("g01_raybundle.npz")
    for i in range(int(g["n_cases"])):
        H, W, focal, pad, off = g["c%d_params" % i]
        ro, rd = hip.nerf_helpers.get_ray_bundle(int(H), int(W), float(focal), T(g["c%d_c2w" % i]), int(pad), float(off))
        assert tuple(ro.shape) == g["c%d_ro" % i].shape
        assert_bits_equal(N_(ro), g["c%d_ro" % i])
        assert_bits_equal(N_(rd), g["c%d_rd" % i])


def test_ray_bundle_full_size_vs_oracle(hip, oracle):
    c2w = load_golden("g08_render.npz")["pose"]
    focal = 0.5 * 800 / np.tan(0.5 * 0.6911112)
    ro, rd = hip.nerf_helpers.get_ray_bundle(800, 800, focal, T(c2w))
    ro_o, rd_o = oracle.get_ray_bundle(800, 800, focal, c2w)
    assert_bits_equal(N_(ro), ro_o)
    assert_bits_equal(N_(rd), rd_o)


def test_ndc_rays(hip):
    g = load_golden("g02_ndc.npz")
    H, W, focal, near = g["params"]
    o, d = hip.nerf_helpers.ndc_rays(int(H), int(W), float(focal), float(near), T(g["ro"]), T(g["rd"]))
    np.testing.assert_allclose(N_(o), g["ro_ndc"], rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(N_(d), g["rd_ndc"], rtol=2e-6, atol=1e-6)


def test_coarse_z(hip):
    g = load_golden("g03_coarse_z.npz")
    N = g["near"].size
    rays = np.zeros((N, 11), np.float32)
    rays[:, 6], rays[:, 7] = g["near"], g["far"]
    for i in range(int(g["n_cases"])):
        nc, lindisp, perturb = (int(v) for v in g["c%d_params" % i])
        z = torch.empty((N, nc), device=DEV)
        tr = T(g["c%d_t_rand" % i]) if perturb else None
        rays_d = T(rays)       # keep device buffers alive across the launch: capi.ptr() only captures the address
        hip.capi.call("nvsr_coarse_z", N, nc, hip.capi.ptr(rays_d), lindisp, hip.capi.ptr(tr), hip.capi.ptr(z), hip.capi.stream())
        np.testing.assert_allclose(N_(z), g["c%d_z" % i], rtol=0, atol=5e-7)


def test_plane_layout_round_trip(hip):
    torch.manual_seed(0)
    for shape in [(1, 48, 37, 53), (1, 48, 64, 64), (1, 6, 5, 130)]:
        p = torch.randn(shape, device=DEV)
        cl = hip.models.to_channel_last(p)
        assert torch.equal(cl, p[0].permute(1, 2, 0).contiguous())
        assert torch.equal(hip.models.from_channel_last(cl), p)


def test_decoder_golden(hip):
    g = load_golden("g04_decoder.npz")
    m, _ = build_model(hip, sd(g, "sd."), [g["plane%d" % d] for d in range(4)], g["box"], sid="lego_DS8_PlRes16_8")
    out = N_(m(T(g["x"])))
    assert out.shape == g["out"].shape
    np.testing.assert_allclose(out, g["out"], rtol=0, atol=2e-5)


def _random_scene(oracle, R, Rv, seed):
    rng = np.random.default_rng(seed)
    planes = [rng.standard_normal((1, 48, R, R), dtype=np.float32) * 0.5 for _ in range(3)]
    planes.append(rng.standard_normal((1, 48, Rv, Rv), dtype=np.float32) * 0.5)
    box = np.array([[-4.0, -4, -4, -np.pi, -np.pi / 2], [4, 4, 4, np.pi, np.pi / 2]])
    return planes, box


def test_decoder_large_vs_oracle(hip, oracle):
    """100k points (ragged: not a multiple of the 256-point tile), incl. points outside the box, planes 96x80 (non-square)"""
    g = load_golden("g04_decoder.npz")
    rng = np.random.default_rng(1)
    planes = [rng.standard_normal((1, 48, 96, 80), dtype=np.float32) * 0.5 for _ in range(3)] + \
             [rng.standard_normal((1, 48, 16, 24), dtype=np.float32) * 0.5]
    m, _ = build_model(hip, sd(g, "sd."), planes, g["box"])
    P = 100003
    x = np.concatenate([rng.uniform(-4.4, 4.4, (P, 3)), rng.standard_normal((P, 3))], -1).astype(np.float32)
    x[:, 3:] /= np.linalg.norm(x[:, 3:], axis=-1, keepdims=True)
    out = N_(m(T(x)))
    sc = oracle.scene(planes, g["box"])
    ref = oracle.triplane_decode(sc, oracle.decoder(decoder_blob(sd(g, "sd."))), x)
    np.testing.assert_allclose(out, ref, rtol=0, atol=2e-5)
    # empty and tiny inputs
    assert m(T(x[:0])).shape == (0, 4)
    np.testing.assert_allclose(N_(m(T(x[:1]))), ref[:1], rtol=0, atol=2e-5)
    # determinism: same launch twice is bit-identical
    assert torch.equal(m(T(x)), m(T(x)))


def test_composite_golden(hip):
    g = load_golden("g05_composite.npz")
    for i in range(int(g["n_cases"])):
        S, white, std = g["c%d_params" % i]
        noise = T(g["c%d_noise" % i]) if std > 0 else None
        rgb, disp, acc, w, depth = hip.volume_rendering_utils.volume_render_radiance_field(
            T(g["c%d_raw" % i]), T(g["c%d_z" % i]), T(g["c%d_rd" % i]), radiance_field_noise_std=float(std),
            white_background=bool(white), noise=noise)
        np.testing.assert_allclose(N_(w), g["c%d_weights" % i], rtol=0, atol=3e-6)
        np.testing.assert_allclose(N_(rgb), g["c%d_rgb" % i], rtol=0, atol=5e-6)
        np.testing.assert_allclose(N_(acc), g["c%d_acc" % i], rtol=0, atol=5e-6)
        np.testing.assert_allclose(N_(depth), g["c%d_depth" % i], rtol=3e-6, atol=1e-5)
        ref_disp = g["c%d_disp" % i]
        d = N_(disp)
        assert np.array_equal(np.isnan(d), np.isnan(ref_disp))
        mk = ~np.isnan(ref_disp)
        np.testing.assert_allclose(d[mk], ref_disp[mk], rtol=1e-5, atol=1e-6)


def test_sample_pdf_and_sort_golden(hip):
    g = load_golden("g06_sample_pdf.npz")
    for i in range(int(g["n_cases"])):
        nb, ns, det = (int(v) for v in g["c%d_params" % i])
        s = hip.nerf_helpers.sample_pdf_2(T(g["c%d_bins" % i]), T(g["c%d_weights" % i]), ns, det=bool(det),
                                          u=None if det else T(g["c%d_u" % i]))
        tol = sample_pdf_tolerance(g["c%d_bins" % i], g["c%d_weights" % i], g["c%d_u" % i])
        err = np.abs(N_(s).astype(np.float64) - g["c%d_samples" % i])
        assert (err <= tol).all(), "case %d: max err/tol %.2f" % (i, float((err / tol).max()))
    g = load_golden("g07_sort.npz")
    for i in range(int(g["n_cases"])):
        z = hip.nerf_helpers.sort_depths(T(np.concatenate([g["c%d_zc" % i], g["c%d_zs" % i]], -1)))
        np.testing.assert_array_equal(N_(z), g["c%d_sorted" % i])


def _excluded_last_sigma(raw_last_sigma):
    return np.abs(raw_last_sigma) < 1e-4


def test_render_staged_and_end_to_end_golden(hip, oracle):
    g = load_golden("g08_render.npz")
    planes = [g["plane%d" % d] for d in range(4)]
    mc, sid = build_model(hip, sd(g, "coarse."), planes, g["box"])
    mf, _ = build_model(hip, sd(g, "fine."), planes, g["box"])
    mf.planes_ = mc.planes_
    sc = oracle.scene(planes, g["box"])
    dec_f = oracle.decoder(decoder_blob(sd(g, "fine.")))
    H, W, focal = int(g["hwf"][0]), int(g["hwf"][1]), float(g["hwf"][2])
    ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, T(g["pose"]))
    rays_np = oracle.pack_rays(g["ro"], g["rd"], 2.0, 6.0)
    rays = hip.train_utils.pack_rays(ro, rd, 2.0, 6.0)
    np.testing.assert_allclose(N_(rays), rays_np, rtol=0, atol=1e-7)
    for i in range(int(g["n_eval"])):
        nc, nf, white, _, _ = (int(v) for v in g["e%d_params" % i])
        opts, scfg = make_options(nc, nf, white=bool(white))
        img_c, _, _, img_f, *_ = hip.train_utils.eval_nerf(H, W, focal, mc, mf, ro, rd, opts, scene_id=sid, scene_config=scfg)
        assert tuple(img_c.shape) == (H, W, 3)
        np.testing.assert_allclose(N_(img_c).reshape(-1, 3), g["e%d_rgb_coarse" % i], rtol=0, atol=2e-5)
        if nf == 0:
            assert img_f is None
            continue
        # fine pass at the reference's depths, through nvsr_render_pass
        zf = g["e%d_z_fine" % i]
        import ctypes as C
        N = zf.shape[0]
        rgb = torch.empty((N, 3), device=DEV)
        disp, acc = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
        scn, keep = mf.native_scene()
        capi = hip.capi
        zf_d = T(zf)
        capi.call("nvsr_render_pass", C.byref(scn), capi.ptr(mf.packed_decoder()), N, nc + nf, capi.ptr(rays), capi.ptr(zf_d), None,
                  white, capi.ptr(rgb), capi.ptr(disp), capi.ptr(acc), None, None, capi.stream())
        o = oracle.render_given_z(sc, dec_f, rays_np, zf, white_background=bool(white), want_raw=True)
        ok = ~_excluded_last_sigma(o["raw"][:, -1, 3])
        assert ok.mean() > 0.95
        np.testing.assert_allclose(N_(rgb)[ok], g["e%d_rgb_fine" % i][ok], rtol=0, atol=2e-5)
        np.testing.assert_allclose(N_(acc)[ok], g["e%d_acc_fine" % i][ok], rtol=0, atol=2e-5)
        np.testing.assert_allclose(N_(disp)[ok], g["e%d_disp_fine" % i][ok], rtol=1e-4, atol=1e-5)
        # end to end
        f = N_(img_f).reshape(-1, 3)
        err = np.abs(f - g["e%d_rgb_fine" % i]).max(-1)
        assert np.mean(err <= 2e-4) >= 0.98 and err.max() <= 2e-3, "frac %.4f max %.2e" % (np.mean(err <= 2e-4), err.max())
        assert psnr(f, g["e%d_rgb_fine" % i]) >= 80.0


def test_render_train_mode_golden(hip):
    g = load_golden("g08_render.npz")
    planes = [g["plane%d" % d] for d in range(4)]
    mc, sid = build_model(hip, sd(g, "coarse."), planes, g["box"])
    mf, _ = build_model(hip, sd(g, "fine."), planes, g["box"])
    sel = g["t_sel"]
    nc, nf, std = int(g["t_params"][0]), int(g["t_params"][1]), float(g["t_params"][4])
    opts, scfg = make_options(nc, nf, perturb=True, noise=std)
    batch = torch.stack([T(g["ro"].reshape(-1, 3)[sel]), T(g["rd"].reshape(-1, 3)[sel])], 0)
    rnd = dict(t_rand=T(g["t_t_rand"]), u=T(g["t_u"]), noise_coarse=T(g["t_noise_coarse"]), noise_fine=T(g["t_noise_fine"]))
    out = hip.train_utils.run_one_iter_of_nerf(16, 16, float(g["hwf"][2]), mc, mf, batch, opts, sid, mode="train",
                                               scene_config=scfg, randoms=rnd)
    assert len(out) == 9 and out[6] is None
    np.testing.assert_allclose(N_(out[0]), g["t_rgb_coarse"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(N_(out[3]), g["t_rgb_fine"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(N_(out[5]), g["t_acc_fine"], rtol=0, atol=1e-3)
    # the same call without explicit randoms draws them on the CPU generator in the reference's order
    torch.manual_seed(88)
    out2 = hip.train_utils.run_one_iter_of_nerf(16, 16, float(g["hwf"][2]), mc, mf, batch, opts, sid, mode="train", scene_config=scfg)
    for a, b in zip(out[:6], out2[:6]):
        assert torch.equal(a, b)


def test_importance_resample_vs_oracle(hip, oracle):
    rng = np.random.default_rng(5)
    N, Nc, Nf = 3001, 64, 128
    z = np.sort(rng.uniform(2, 6, (N, Nc)).astype(np.float32), -1)
    w = (rng.uniform(0, 1, (N, Nc)) ** 6).astype(np.float32)
    w[::7] = 0.0
    w[::7, 20] = 0.9     # opaque rays with empty bins sit on the 1e-5 threshold
    for u in (None, rng.uniform(0, 1, (N, Nf)).astype(np.float32)):
        zf = torch.empty((N, Nc + Nf), device=DEV)
        capi = hip.capi
        z_d, w_d, u_d = T(z), T(w), (None if u is None else T(u))   # kept alive: capi.ptr() only captures the address
        capi.call("nvsr_importance_resample", N, Nc, Nf, capi.ptr(z_d), capi.ptr(w_d), capi.ptr(u_d), capi.ptr(zf), capi.stream())
        zf = N_(zf)
        assert (np.diff(zf, axis=-1) >= 0).all()
        zm = 0.5 * (z[:, 1:] + z[:, :-1])
        uu = np.broadcast_to(np.linspace(0, 1, Nf, dtype=np.float32), (N, Nf)) if u is None else u
        ref = oracle.sort_rows(np.concatenate([z, oracle.sample_pdf(zm, w[:, 1:-1], uu)], -1))
        tol = sample_pdf_tolerance(zm, w[:, 1:-1], uu).max(-1, keepdims=True)
        assert (np.abs(zf.astype(np.float64) - ref) <= tol).all()
        # every coarse depth survives the merge bit-exactly
        for r in (0, 7, N - 1):
            assert np.isin(z[r], zf[r]).all()


def test_importance_resample_merge_paths_bit_identical(hip):
    """The fused kernel ranks the elements of a sorted run by binary search (merge_sort_wave); whichever path a ray takes -- both runs
    sorted, only the coarse depths sorted, neither (a caller's unsorted depths), ties between and inside the runs, odd sizes -- the
    result is bit for bit torch.sort(cat(z, sample_pdf(...))) as the separate kernels (general rank sort) produce it."""
    capi = hip.capi
    rng = np.random.default_rng(11)
    for (N, Nc, Nf) in ((1537, 64, 128), (130, 33, 77), (64, 3, 1), (257, 256, 256)):
        z = np.sort(rng.uniform(2, 6, (N, Nc)).astype(np.float32), -1)
        z[1::5, Nc // 2:] = z[1::5, Nc // 2 - 1:Nc // 2]              # runs of equal coarse depths
        w = (rng.uniform(0, 1, (N, Nc)) ** 4).astype(np.float32)
        w[::3] = 0.0                                                  # flat pdf: samples land exactly on bin edges / coarse mids
        z_unsorted = rng.permuted(z, axis=-1)
        u_rand = rng.uniform(0, 1, (N, Nf)).astype(np.float32)
        u_ties = np.repeat(u_rand[:, : (Nf + 1) // 2], 2, axis=-1)[:, :Nf].copy()   # duplicated samples, unsorted
        for zz in (z, z_unsorted):
            for u in (None, u_rand, u_ties):
                z_d, w_d, u_d = T(zz), T(w), (None if u is None else T(u))
                zf = torch.empty((N, Nc + Nf), device=DEV)
                capi.call("nvsr_importance_resample", N, Nc, Nf, capi.ptr(z_d), capi.ptr(w_d), capi.ptr(u_d), capi.ptr(zf), capi.stream())
                zm = (0.5 * (z_d[:, 1:] + z_d[:, :-1])).contiguous()
                wi = w_d[:, 1:-1].contiguous()
                smp = torch.empty((N, Nf), device=DEV)
                capi.call("nvsr_sample_pdf", N, Nc - 1, Nf, capi.ptr(zm), capi.ptr(wi), capi.ptr(u_d), capi.ptr(smp), capi.stream())
                cat = torch.cat([z_d, smp], -1).contiguous()
                ref = torch.empty_like(cat)
                capi.call("nvsr_sort_rows", N, Nc + Nf, capi.ptr(cat), capi.ptr(ref), capi.stream())
                assert torch.equal(zf, ref)
                assert torch.equal(ref, torch.sort(cat, -1).values)


def test_full_size_frame_properties_and_oracle_subset(hip, oracle):
    """BASELINE config 2 at full size: 800x800 rays, 64+128 samples, planes 800^2 (+32^2 view plane).
    Size-independent properties on the whole frame + the oracle, stage by stage, on a seeded subset of its rays."""
    import ctypes as C
    g = load_golden("g08_render.npz")
    torch.manual_seed(0)
    R, Rv = 800, 32
    # band-limited planes (random 100^2 grids, bilinearly upsampled): feature planes of a trained scene are smooth at texel
    # scale; white noise at 800^2 would make the radiance field a chaotic function of the sample depth
    up = lambda r, src: torch.nn.functional.interpolate(torch.randn(1, 48, src, src, device=DEV) * 0.7, size=(r, r), mode="bilinear",
                                                        align_corners=True)
    planes = [up(R, 100) for _ in range(3)] + [up(Rv, 8)]
    planes_np = [N_(p) for p in planes]
    mc, sid = build_model(hip, sd(g, "coarse."), planes_np, g["box"], sid="lego_DS1_PlRes800_32")
    mf, _ = build_model(hip, sd(g, "fine."), planes_np, g["box"], sid="lego_DS1_PlRes800_32")
    mf.planes_ = mc.planes_
    H = W = 800
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, T(g["pose"]))
    opts, scfg = make_options(64, 128)
    batch = torch.stack([ro.reshape(-1, 3), rd.reshape(-1, 3)], 0)
    rc, dc, ac, rf, df, af, *_ = hip.train_utils.run_one_iter_of_nerf(H, W, focal, mc, mf, batch, opts, sid, mode="validation",
                                                                      scene_config=scfg)
    torch.cuda.synchronize()
    for a in (ac, af):
        assert float(a.min()) >= 0.0 and float(a.max()) <= 1.0 + 1e-5
    for c in (rc, rf):
        assert torch.isfinite(c).all() and float(c.min()) >= 0.0 and float(c.max()) <= 1.0 + 1e-5
    assert 0.05 < float(af.mean()) < 0.99          # the synthetic scene is neither empty nor saturated
    # ray-order independence: a shuffled subset rendered on its own gives bit-identical pixels (no cross-ray state).
    # 70 001 rays stay on the fused path (>= NVSR_FUSED_MIN_RAYS); a small batch takes the un-fused path (wave-scan compositing,
    # different summation order) and agrees to the stated tolerance instead.
    perm = torch.randperm(H * W, device=DEV)
    big = perm[:70001]
    sub = hip.train_utils.run_one_iter_of_nerf(H, W, focal, mc, mf, batch[:, big], opts, sid, mode="validation", scene_config=scfg)
    assert torch.equal(sub[3], rf[big]) and torch.equal(sub[0], rc[big])
    idx = perm[:4099]
    small = hip.train_utils.run_one_iter_of_nerf(H, W, focal, mc, mf, batch[:, idx], opts, sid, mode="validation", scene_config=scfg)
    np.testing.assert_allclose(N_(small[0]), N_(rc[idx]), rtol=0, atol=2e-5)
    e_small = (small[3] - rf[idx]).abs().max(-1)[0]
    assert float((e_small <= 2e-4).float().mean()) >= 0.98

    # ---- oracle on a seeded subset, stage by stage --------------------------------------------------------------------
    ids = N_(idx[:1500])
    sc = oracle.scene(planes_np, g["box"])
    dec_c, dec_f = oracle.decoder(decoder_blob(sd(g, "coarse."))), oracle.decoder(decoder_blob(sd(g, "fine.")))
    rays_np = oracle.pack_rays(N_(ro).reshape(-1, 3)[ids], N_(rd).reshape(-1, 3)[ids], 2.0, 6.0)
    o = oracle.render_rays(sc, dec_c, dec_f, rays_np, 64, 128, want_aux=True)
    np.testing.assert_allclose(N_(rc)[ids], o["rgb_coarse"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(N_(ac)[ids], o["acc_coarse"], rtol=0, atol=2e-5)
    capi = hip.capi
    n = len(ids)
    rays_d = T(rays_np)
    z_c = torch.empty((n, 64), device=DEV)
    w_c = torch.empty((n, 64), device=DEV)
    z_f = torch.empty((n, 192), device=DEV)
    o3, o1a, o1b = torch.empty((n, 3), device=DEV), torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    scn, keep = mc.native_scene()
    st = capi.stream()
    capi.call("nvsr_coarse_z", n, 64, capi.ptr(rays_d), 0, None, capi.ptr(z_c), st)
    capi.call("nvsr_render_pass", C.byref(scn), capi.ptr(mc.packed_decoder()), n, 64, capi.ptr(rays_d), capi.ptr(z_c), None, 0,
              capi.ptr(o3), capi.ptr(o1a), capi.ptr(o1b), capi.ptr(w_c), None, st)
    capi.call("nvsr_importance_resample", n, 64, 128, capi.ptr(z_c), capi.ptr(w_c), None, capi.ptr(z_f), st)
    np.testing.assert_allclose(N_(w_c), o["weights_coarse"], rtol=0, atol=2e-5)
    zc_np = N_(z_c)
    tol = sample_pdf_tolerance(0.5 * (zc_np[:, 1:] + zc_np[:, :-1]), o["weights_coarse"][:, 1:-1],
                               np.broadcast_to(np.linspace(0, 1, 128, dtype=np.float32), (n, 128)), w_noise=2e-6).max(-1, keepdims=True)
    assert (np.abs(N_(z_f).astype(np.float64) - o["z_fine"]) <= tol).all()
    # fine pass at the oracle's depths
    zf_d = T(o["z_fine"])
    capi.call("nvsr_render_pass", C.byref(scn), capi.ptr(mf.packed_decoder()), n, 192, capi.ptr(rays_d), capi.ptr(zf_d), None, 0,
              capi.ptr(o3), capi.ptr(o1a), capi.ptr(o1b), None, None, st)
    fo = oracle.render_given_z(sc, dec_f, rays_np, o["z_fine"], want_raw=True)
    ok = ~_excluded_last_sigma(fo["raw"][:, -1, 3])
    assert ok.mean() > 0.95
    np.testing.assert_allclose(N_(o3)[ok], fo["rgb"][ok], rtol=0, atol=2e-5)
    np.testing.assert_allclose(N_(o1b)[ok], fo["acc"][ok], rtol=0, atol=2e-5)
    # end to end: depths regenerated by each side
    err = np.abs(N_(rf)[ids] - o["rgb_fine"]).max(-1)
    assert np.mean(err <= 2e-4) >= 0.98 and err.max() <= 5e-3, "fine rgb: %.4f of rays within 2e-4 (max %.2e)" % (np.mean(err <= 2e-4), err.max())
    assert psnr(N_(rf)[ids], o["rgb_fine"]) >= 75.0


# ---------------------------------------------------------------------------------------------------------------------
# feature-plane super-resolution
def _sr_model(hip, g):
    Cc, hid, nblocks, sf, R, pad, over = [int(v) for v in g["cfg"]]
    sr = hip.models.PlanesSR(hip.models.EDSR, sf, Cc, Cc, {"model": {"hidden_size": hid, "n_blocks": nblocks}}, "bilinear")
    sr.load_state_dict({k[3:]: torch.as_tensor(v) for k, v in g.items() if k.startswith("sd.")}, strict=True)
    sr = sr.to(DEV)
    assert sr.inner_model.required_padding == pad and sr.HR_overpadding == over
    return sr, (Cc, hid, nblocks, sf, R, pad, over)


def test_edsr_and_planes_sr_golden(hip):
    g = load_golden("g09_edsr.npz")
    sr, (Cc, hid, nblocks, sf, R, pad, over) = _sr_model(hip, g)
    sr.eval()
    # single residual block through the conv entry point with its fused epilogues
    capi = hip.capi
    blk_in = T(g["block_in"][0])
    h, w = blk_in.shape[-2:]
    w1, w2 = T(g["sd.inner_model.residual.0.conv1.weight"]), T(g["sd.inner_model.residual.0.conv2.weight"])
    pk = [torch.empty(capi.lib().nvsr_conv3x3_packed_floats(hid, hid), device=DEV) for _ in range(2)]
    for wt, p in zip((w1, w2), pk):
        capi.call("nvsr_pack_conv3x3", capi.ptr(wt), hid, hid, capi.ptr(p), capi.stream())
    t1 = torch.empty((hid, h - 2, w - 2), device=DEV)
    t2 = torch.empty((hid, h - 4, w - 4), device=DEV)
    capi.call("nvsr_conv3x3", capi.ptr(blk_in), hid, h, w, capi.ptr(pk[0]), hid, 1, None, capi.ptr(t1), capi.stream())
    capi.call("nvsr_conv3x3", capi.ptr(t1), hid, h - 2, w - 2, capi.ptr(pk[1]), hid, 2, capi.ptr(blk_in), capi.ptr(t2), capi.stream())
    np.testing.assert_allclose(N_(t2), g["block_out"][0], rtol=0, atol=3e-6)
    # whole network, then PlanesSR full plane and ROI
    out = sr.inner_model(T(g["edsr_in"]))
    assert tuple(out.shape) == g["edsr_out"].shape
    np.testing.assert_allclose(N_(out), g["edsr_out"], rtol=0, atol=1e-5)
    sr.set_LR_plane(T(g["lr"]), id="p", save_interpolated=False)
    full = sr("p")
    assert tuple(full.shape) == (1, Cc, R * sf, R * sf)
    np.testing.assert_allclose(N_(full), g["sr_full"], rtol=0, atol=1e-5)
    assert sr("p") is full                      # cached like the reference's SR_planes
    sr.clear_SR_planes()
    sr.train()
    roi = N_(sr(("p", T(g["roi"]))))
    ref = g["sr_roi"]
    assert np.array_equal(np.isnan(roi), np.isnan(ref)) and np.isnan(ref).any()
    m = ~np.isnan(ref)
    np.testing.assert_allclose(roi[m], ref[m], rtol=0, atol=1e-5)


def test_conv3x3_shapes_vs_oracle(hip, oracle):
    """channel counts that exercise both workgroup shapes and the padding of Cin/Cout (48->256, 256->256, 256->48, 16->64 shuffle,
    256->1024 shuffle), in both arithmetic modes (the wide layers run on the bf16-limb kernel by default: near-f32 products (bound in include/nvsr.h), the same
    tolerance; the others always on the f32 kernel)"""
    rng = np.random.default_rng(3)
    capi = hip.capi
    assert capi.get_conv_arithmetic() == "f16x2"            # include/nvsr.h NVSR_CONV_ARITH_DEFAULT
    try:
        for Cin, Cout, H, W, epi in [(48, 256, 21, 45, 0), (256, 256, 14, 40, 1), (256, 48, 37, 35, 0), (16, 64, 9, 70, 3), (5, 7, 3, 3, 0),
                                     (256, 1024, 9, 37, 3), (256, 256, 5, 131, 0)]:
            x = rng.standard_normal((Cin, H, W), dtype=np.float32)
            w = (rng.standard_normal((Cout, Cin, 3, 3), dtype=np.float32) / np.sqrt(9 * Cin)).astype(np.float32)
            xd, wd = T(x), T(w)
            pk = torch.empty(capi.lib().nvsr_conv3x3_packed_floats(Cin, Cout), device=DEV)
            capi.call("nvsr_pack_conv3x3", capi.ptr(wd), Cin, Cout, capi.ptr(pk), capi.stream())
            ref = oracle.conv3x3(x, w, relu=(epi == 1))
            if epi == 3:
                ref = ref.reshape(Cout // 4, 2, 2, H - 2, W - 2).transpose(0, 3, 1, 4, 2).reshape(Cout // 4, 2 * (H - 2), 2 * (W - 2))
            for mode in ("f16x2", "bf16x3", "f32"):
                capi.set_conv_arithmetic(mode)
                out = torch.full(ref.shape, -7.0, device=DEV)
                capi.call("nvsr_conv3x3", capi.ptr(xd), Cin, H, W, capi.ptr(pk), Cout, epi, None, capi.ptr(out), capi.stream())
                # unit-variance data: fp32 accumulation over K = 9*Cin terms, |out| up to ~4  ->  ~sqrt(K)*2^-24*|out|
                np.testing.assert_allclose(N_(out), ref, rtol=0, atol=3e-5, err_msg=str((Cin, Cout, H, W, epi, mode)))
        # data gradient of a wide layer (the same kernels with a virtual zero border and the flipped, transposed weights)
        Cin, Cout, H, W = 256, 256, 11, 38
        w = (rng.standard_normal((Cout, Cin, 3, 3), dtype=np.float32) / np.sqrt(9 * Cin)).astype(np.float32)
        dy = rng.standard_normal((Cout, H - 2, W - 2), dtype=np.float32)
        pk = torch.empty(capi.lib().nvsr_conv3x3_packed_floats(Cout, Cin), device=DEV)
        capi.call("nvsr_pack_conv3x3_dgrad", capi.ptr(T(w)), Cin, Cout, capi.ptr(pk), capi.stream())
        wt = np.ascontiguousarray(w.transpose(1, 0, 2, 3)[:, :, ::-1, ::-1])
        ref = oracle.conv3x3(np.pad(dy, ((0, 0), (2, 2), (2, 2))), wt)
        for mode in ("f16x2", "bf16x3", "f32"):               # (f16x2: the data gradient runs 2 f16 limbs with the dy tensor's own scale)
            capi.set_conv_arithmetic(mode)
            dx = torch.full((Cin, H, W), -7.0, device=DEV)
            capi.call("nvsr_conv3x3_dgrad", capi.ptr(T(dy)), Cin, H, W, capi.ptr(pk), Cout, capi.ptr(dx), capi.stream())
            np.testing.assert_allclose(N_(dx), ref, rtol=0, atol=3e-5, err_msg="dgrad " + mode)
    finally:
        capi.set_conv_arithmetic("f16x2")


def test_planes_sr_batch_is_bit_identical_to_one_by_one(hip):
    """the 3 position planes of a scene through the SR net in one batched pass == three separate passes, bit for bit"""
    for hidden, blocks in ((64, 3), (256, 1)):          # narrow (<= 64 channels) and wide (256) workgroup shapes of the conv kernels
        torch.manual_seed(12)
        sr = hip.models.PlanesSR(hip.models.EDSR, 4, 48, 48, {"model": {"hidden_size": hidden, "n_blocks": blocks}}, "bilinear").to(DEV)
        with torch.no_grad():
            for p_ in sr.parameters():
                p_.mul_(10.0)
        sr.eval()
        names = ["a", "b", "c"]
        for n in names:
            sr.set_LR_plane(torch.randn(1, 48, 37, 29, device=DEV) * 0.5, id=n, save_interpolated=False)
        with torch.no_grad():
            single = [sr(n).clone() for n in names]
            sr.clear_SR_planes()
            sr.super_resolve_many(names)
            assert sorted(sr.SR_planes) == names
            for n, ref in zip(names, single):
                assert sr(n) is sr.SR_planes[n] and torch.equal(sr(n), ref)
            # EDSR entry point, batch of 2 against two single calls
            capi = hip.capi
            x = torch.randn(2, 48, 30, 41, device=DEV)
            cin, cout, hid, nb, n_up = sr.inner_model.geometry
            one = torch.stack([sr.inner_model(x[b:b + 1])[0] for b in range(2)])
            out = torch.empty_like(one)
            ws = torch.empty(2 * capi.lib().nvsr_edsr_workspace_floats(hid, nb, n_up, 30, 41), device=DEV)
            capi.call("nvsr_edsr_forward_batch", capi.ptr(x), 2, 48, 30, 41, capi.ptr(sr.inner_model.packed_weights()), cout, hid, nb, n_up,
                      capi.ptr(out), capi.ptr(ws), capi.stream())
            assert torch.equal(out, one)


def test_planes_sr_row_bands_equal_the_full_plane(hip):
    """the band a rank computes in the sharded SR stage (distributed.super_resolve_planes_sharded: PlanesSR's ROI path with the
    rows of distributed.band_roi) holds exactly the values of the full-plane pass -- the halo comes from real rows, not padding"""
    torch.manual_seed(13)
    sr = hip.models.PlanesSR(hip.models.EDSR, 4, 48, 48, {"model": {"hidden_size": 32, "n_blocks": 3}}, "bilinear").to(DEV)
    with torch.no_grad():
        for p_ in sr.parameters():
            p_.mul_(10.0)
    sr.eval()
    R0, R1 = 45, 38
    sr.set_LR_plane(torch.randn(1, 48, R0, R1, device=DEV) * 0.5, id="p", save_interpolated=False)
    D = hip.distributed
    with torch.no_grad():
        full = sr("p")
        for world in (2, 3):
            for rank in range(world):
                lo, hi = D.shard_bounds(R0, rank, world)
                band = sr(("p", D.band_roi(lo, hi, R0).to(DEV)))
                assert torch.equal(band[..., lo * 4: hi * 4, :], full[..., lo * 4: hi * 4, :])
                assert not torch.isnan(band[..., lo * 4: hi * 4, :]).any()
    assert D.super_resolve_planes_sharded(sr, ["p"])[0] is full          # world size 1: the plain cached path


def test_conv3x3_backward_vs_oracle(hip, oracle):
    """data and weight gradients of the valid 3x3 conv through the C ABI; sizes that exercise partial tiles in every dimension
    (channels not multiples of 64, width not a multiple of 32, fewer rows than row slabs; the last shape gives the weight-gradient kernel
    408 equal pieces of 8 row steps over 16 tiles, i.e. pieces that cross column chunks and tiles)"""
    rng = np.random.default_rng(33)
    capi = hip.capi
    for Cin, Cout, H, W in [(48, 256, 21, 45), (256, 256, 14, 40), (256, 48, 37, 35), (70, 130, 9, 70), (5, 7, 3, 3), (64, 1024, 10, 12), (256, 256, 70, 70)]:
        x = rng.standard_normal((Cin, H, W), dtype=np.float32)
        w = (rng.standard_normal((Cout, Cin, 3, 3), dtype=np.float32) / np.sqrt(9 * Cin)).astype(np.float32)
        dy = rng.standard_normal((Cout, H - 2, W - 2), dtype=np.float32)
        dx_ref, dw_ref = oracle.conv3x3_backward(x, w, dy)
        xd, wd, dyd = T(x), T(w), T(dy)
        pk = torch.empty(capi.lib().nvsr_conv3x3_packed_floats(Cout, Cin), device=DEV)
        capi.call("nvsr_pack_conv3x3_dgrad", capi.ptr(wd), Cin, Cout, capi.ptr(pk), capi.stream())
        dx = torch.empty((Cin, H, W), device=DEV)
        capi.call("nvsr_conv3x3_dgrad", capi.ptr(dyd), Cin, H, W, capi.ptr(pk), Cout, capi.ptr(dx), capi.stream())
        # K = 9*Cout unit-variance products of magnitude 1/sqrt(9 Cin)
        np.testing.assert_allclose(N_(dx), dx_ref, rtol=0, atol=3e-5 * max(1.0, np.sqrt(Cout / Cin)), err_msg="dgrad " + str((Cin, Cout, H, W)))
        ws = torch.empty(capi.lib().nvsr_conv3x3_wgrad_workspace_floats(Cin, H, W, Cout), device=DEV)
        dw = torch.full((Cout, Cin, 3, 3), 1.0, device=DEV)           # accumulates on top of what is there, scaled
        capi.call("nvsr_conv3x3_wgrad", capi.ptr(dyd), capi.ptr(xd), Cin, H, W, Cout, 0.5, capi.ptr(dw), capi.ptr(ws), capi.stream())
        got = (N_(dw) - 1.0) / 0.5
        K = (H - 2) * (W - 2)
        np.testing.assert_allclose(got, dw_ref, rtol=0, atol=2e-5 * np.sqrt(K) + 1e-5, err_msg="wgrad " + str((Cin, Cout, H, W)))
        assert np.linalg.norm(got - dw_ref) / np.linalg.norm(dw_ref) < 2e-6
        dw2 = torch.full((Cout, Cin, 3, 3), 1.0, device=DEV)          # deterministic: fixed-order reduction, no atomics
        capi.call("nvsr_conv3x3_wgrad", capi.ptr(dyd), capi.ptr(xd), Cin, H, W, Cout, 0.5, capi.ptr(dw2), capi.ptr(ws), capi.stream())
        assert torch.equal(dw, dw2)


def _sr_grad_blob(sr):
    return np.concatenate([N_(w.grad).reshape(-1) for w in sr.inner_model.conv_weights()])


def _rel(a, b):
    return np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b)


def test_sr_gradients_golden(hip):
    """loss.backward() through EDSR / PlanesSR (ROI = training path, and full plane) vs torch.autograd through the reference (g14)"""
    g9, g = load_golden("g09_edsr.npz"), load_golden("g14_sr_grads.npz")
    sr, (Cc, hid, nblocks, sf, R, pad, over) = _sr_model(hip, g9)
    sr.train()
    x = T(g["edsr_in"]).requires_grad_(True)
    out = sr.inner_model(x)
    (out * T(g["edsr_gout"])).sum().backward()
    assert _rel(_sr_grad_blob(sr), g["edsr_gw"]) < 2e-5
    assert _rel(N_(x.grad), g["edsr_gin"]) < 2e-5
    np.testing.assert_allclose(_sr_grad_blob(sr), g["edsr_gw"], rtol=0, atol=2e-5 * np.abs(g["edsr_gw"]).max())
    for tag, roi in (("roi", T(g["roi"])), ("full", None)):
        lr = torch.nn.Parameter(T(g9["lr"]))
        sr.clear_SR_planes(all_planes=True)
        sr.set_LR_plane(lr, id="p", save_interpolated=False)
        sr.zero_grad(set_to_none=True)
        out = sr(("p", roi)) if roi is not None else sr("p")
        assert out.requires_grad and "p" not in sr.SR_planes          # the training result is never cached
        valid = ~torch.isnan(out)
        assert abs(float(valid.float().mean()) - float(g["sr_%s_valid_frac" % tag])) < 1e-6
        (torch.where(valid, out, torch.zeros_like(out)) * T(g["sr_%s_gout" % tag])).sum().backward()
        assert _rel(_sr_grad_blob(sr), g["sr_%s_gw" % tag]) < 2e-5, tag
        assert _rel(N_(lr.grad), g["sr_%s_glr" % tag]) < 2e-5, tag
        np.testing.assert_allclose(N_(lr.grad), g["sr_%s_glr" % tag], rtol=0, atol=2e-5 * np.abs(g["sr_%s_glr" % tag]).max())
    # LR plane detached (models.py:272): weights only
    sr.clear_SR_planes(all_planes=True)
    sr.set_LR_plane(T(g9["lr"]), id="p", save_interpolated=False)
    sr.zero_grad(set_to_none=True)
    out = sr("p")
    (out * T(g["sr_full_gout"])).sum().backward()
    assert _rel(_sr_grad_blob(sr), g["sr_full_gw"]) < 2e-5
    # evaluation never builds a graph and is cached
    sr.eval()
    with torch.no_grad():
        full = sr("p")
    assert not full.requires_grad and sr("p") is full
    np.testing.assert_allclose(N_(full), g9["sr_full"], rtol=0, atol=1e-5)


def test_sr_gradients_vs_oracle_larger(hip, oracle):
    """48 -> 48 channels, hidden 64, 3 blocks, x4, 34 x 46 input (hidden width 64 = one partial weight-gradient tile in ci and co)"""
    rng = np.random.default_rng(51)
    torch.manual_seed(51)
    Cc, hid, nb, n_up, H, W = 48, 64, 3, 2, 34, 46
    sr = hip.models.PlanesSR(hip.models.EDSR, 4, Cc, Cc, {"model": {"hidden_size": hid, "n_blocks": nb}}, "bilinear").to(DEV)
    with torch.no_grad():
        for p_ in sr.parameters():
            p_.mul_(10.0)
    sr.train()
    x = rng.standard_normal((1, Cc, H, W), dtype=np.float32)
    xd = T(x).requires_grad_(True)
    out = sr.inner_model(xd)
    gout = rng.standard_normal(tuple(out.shape), dtype=np.float32)
    (out * T(gout)).sum().backward()
    blob = np.concatenate([N_(w).reshape(-1) for w in sr.inner_model.conv_weights()])
    np.testing.assert_allclose(N_(out)[0], oracle.edsr_forward(x[0], blob, Cc, hid, nb, n_up), rtol=0, atol=2e-5)
    gw, gx = oracle.edsr_backward(x[0], blob, Cc, hid, nb, n_up, gout[0])
    # a ReLU input within fp32 rounding of zero can gate differently in the fp32 kernels and the double-accumulating oracle; each such
    # flip moves one activation's worth of gradient (1 flip in ~10^5 activations at this size)
    assert _rel(_sr_grad_blob(sr), gw) < 1e-4 and _rel(N_(xd.grad)[0], gx) < 1e-4


def test_edsr_forward_backward_vs_torch_reference(hip):
    """EDSR forward + gradients (all conv weights, input) against the same network written with torch.nn.functional on the GPU
    (conv2d / relu / pixel_shuffle, fp32 autograd): an independent reference next to the C oracle"""
    import torch.nn.functional as F
    torch.manual_seed(21)
    Cc, hid, nb = 48, 64, 4
    sr = hip.models.PlanesSR(hip.models.EDSR, 4, Cc, Cc, {"model": {"hidden_size": hid, "n_blocks": nb}}, "bilinear").to(DEV)
    with torch.no_grad():
        for p_ in sr.parameters():
            p_.mul_(10.0)
    sr.train()
    net = sr.inner_model
    x = torch.randn(1, Cc, 44, 61, device=DEV)
    xa = x.clone().requires_grad_(True)
    out = net(xa)
    Gm = torch.randn_like(out)
    (out * Gm).sum().backward()
    got_w = [w.grad.clone() for w in net.conv_weights()]
    # reference
    ws = [w.detach().clone().requires_grad_(True) for w in net.conv_weights()]
    xb = x.clone().requires_grad_(True)
    k = 0
    h = F.conv2d(xb, ws[k]); k += 1
    for b in range(nb):
        t = F.conv2d(F.relu(F.conv2d(h, ws[k])), ws[k + 1]); k += 2
        h = t * 0.1 + h[..., 2:-2, 2:-2]
    h = F.conv2d(h, ws[k]); k += 1
    for u in range(2):
        h = F.pixel_shuffle(F.conv2d(h, ws[k]), 2); k += 1
    ref = F.conv2d(h, ws[k])
    assert k == len(ws) - 1 and tuple(ref.shape) == tuple(out.shape)
    np.testing.assert_allclose(N_(out), N_(ref), rtol=0, atol=3e-5)
    (ref * Gm).sum().backward()
    for i, (a, b) in enumerate(zip(got_w, ws)):
        assert _rel(N_(a), N_(b.grad)) < 1e-4, "conv weight %d" % i
    assert _rel(N_(xa.grad), N_(xb.grad)) < 1e-4


def test_train_step_through_super_resolved_planes(hip, oracle):
    """SR refinement step (what: ['SR']): rays -> ROI -> PlanesSR(ROI) x3 -> render -> loss.backward() fills the EDSR weights' .grad;
    oracle = its own SR forward, render backward wrt the HR planes, SR backward, chained on the host"""
    g = load_golden("g08_render.npz")
    rng = np.random.default_rng(61)
    R, Rv, hid, nb = 24, 8, 16, 2
    planes = [rng.standard_normal((1, 48, R, R), dtype=np.float32) * 0.5 for _ in range(3)] + \
             [rng.standard_normal((1, 48, Rv, Rv), dtype=np.float32) * 0.5]
    sid = "lego_DS8_PlRes24_8"
    mc, mf = _grad_models(hip, g, planes, sid, what=())
    torch.manual_seed(6)
    sr = hip.models.PlanesSR(hip.models.EDSR, 4, 48, 48, {"model": {"hidden_size": hid, "n_blocks": nb}}, "bilinear").to(DEV)
    with torch.no_grad():
        for p_ in sr.parameters():
            p_.mul_(10.0)
    sr.train()
    for m in (mc, mf):
        m.assign_SR_model(sr, SR_viewdir=False)
    mf.assign_LR_planes()
    H = W = 12
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, T(g["pose"]))
    sel = torch.from_numpy(rng.permutation(H * W)[:60]).to(DEV)     # a partial batch: its ROI does not cover the planes
    batch = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0)
    N, nc, nf = 60, 24, 24
    opts, scfg = make_options(nc, nf)
    out = hip.train_utils.run_one_iter_of_nerf(H, W, focal, mc, mf, batch, opts, sid, mode="train", scene_config=scfg, randoms={})
    z_fine = N_(out[3].grad_fn.saved["z_f"])
    gc = T(rng.standard_normal((N, 3)).astype(np.float32) / N)
    gf = T(rng.standard_normal((N, 3)).astype(np.float32) / N)
    ((out[0] * gc).sum() + (out[3] * gf).sum()).backward()
    got = _sr_grad_blob(sr)
    # --- oracle chain
    blob = np.concatenate([N_(w).reshape(-1) for w in sr.inner_model.conv_weights()])
    pad, over = int(sr.inner_model.required_padding), int(sr.HR_overpadding)
    rays_np = oracle.pack_rays(N_(batch[0]), N_(batch[1]), 2.0, 6.0)
    box = np.asarray(g["box"], np.float64)
    ends = np.concatenate([rays_np[:, 0:3] + rays_np[:, 3:6] * rays_np[:, 6:7], rays_np[:, 0:3] + rays_np[:, 3:6] * rays_np[:, 7:8]], 0)
    n_ends = (2 * (ends - box[0, :3].astype(np.float32)) / (box[1, :3] - box[0, :3]).astype(np.float32) - 1).astype(np.float32)
    hr, rois = [], []
    for d in range(3):
        m = N_(mf.coord_projector.rot_mats_NON_LEARNED[d])[:, 1:]
        grid = n_ends @ m
        roi = np.array([[grid[:, 1].min(), grid[:, 0].min()], [grid[:, 1].max(), grid[:, 0].max()]], np.float32)
        rois.append(roi)
        hr.append(oracle.planes_sr(planes[d][0], blob, hid, nb, 2, pad, over, roi=roi))
    assert all(np.isnan(h).any() for h in hr)                       # the ROI path really ran
    hr_zero = [np.nan_to_num(h)[None] for h in hr] + [planes[3]]
    sc = oracle.scene(hr_zero, g["box"])
    dec_c, dec_f = oracle.decoder(decoder_blob(sd(g, "coarse."))), oracle.decoder(decoder_blob(sd(g, "fine.")))
    o = oracle.render_rays(sc, dec_c, dec_f, rays_np, nc, nf)
    np.testing.assert_allclose(N_(out[0]), o["rgb_coarse"], rtol=0, atol=3e-5)
    gplanes = oracle.render_backward(sc, [p.shape for p in hr_zero], dec_c, dec_f, rays_np, nc, nf, N_(gc), N_(gf), z_fine=z_fine)
    ref = np.zeros_like(blob, dtype=np.float64)
    for d in range(3):
        gw, _ = oracle.planes_sr_backward(planes[d][0], blob, hid, nb, 2, pad, over, gplanes[d], roi=rois[d], want_dlr=False)
        ref += gw
    assert _rel(got, ref) < 5e-3, "SR weight gradient through the renderer: relative L2 error %.2e" % _rel(got, ref)
    for m in (mc, mf):                                             # nothing else received a gradient
        assert all(p_.grad is None for p_ in m.decoder_parameters()) and all(p_.grad is None for p_ in m.planes_.values())


def test_render_through_super_resolved_planes(hip, oracle):
    """BASELINE config 3 in miniature: the fine model samples planes produced by PlanesSR(EDSR) (x4), the coarse model the LR
    planes (models.py:270-284, 289-310); oracle = its own planes_sr + render."""
    g = load_golden("g08_render.npz")
    rng = np.random.default_rng(11)
    R, Rv, hid, nb = 24, 8, 16, 2
    planes = [rng.standard_normal((1, 48, R, R), dtype=np.float32) * 0.5 for _ in range(3)] + \
             [rng.standard_normal((1, 48, Rv, Rv), dtype=np.float32) * 0.5]
    sid = "lego_DS8_PlRes24_8"
    mc, _ = build_model(hip, sd(g, "coarse."), planes, g["box"], sid=sid)
    mf, _ = build_model(hip, sd(g, "fine."), planes, g["box"], sid=sid)
    mf.planes_ = mc.planes_
    torch.manual_seed(5)
    sr = hip.models.PlanesSR(hip.models.EDSR, 4, 48, 48, {"model": {"hidden_size": hid, "n_blocks": nb}}, "bilinear").to(DEV)
    with torch.no_grad():
        for p_ in sr.parameters():
            p_.mul_(10.0)
    sr.eval()
    mf.assign_SR_model(sr, SR_viewdir=False)
    mf.assign_LR_planes()
    assert sorted(sr.LR_planes) == sorted(hip.models.get_plane_name(sid, d) for d in range(3))   # the view plane is never SR'd
    H = W = 12
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, T(g["pose"]))
    opts, scfg = make_options(32, 32)
    _, _, _, img_f, *_ = hip.train_utils.eval_nerf(H, W, focal, mc, mf, ro, rd, opts, scene_id=sid, scene_config=scfg)
    # oracle
    from oracle.oracle import Oracle
    blob, _ = Oracle.edsr_blob({k: N_(v) for k, v in sr.state_dict().items()}, n_up=2)
    pad, over = sr.inner_model.required_padding, sr.HR_overpadding
    hr = [oracle.planes_sr(planes[d][0], blob, hid, nb, 2, pad, over)[None] for d in range(3)] + [planes[3]]
    for d in range(3):
        np.testing.assert_allclose(N_(sr(hip.models.get_plane_name(sid, d))), hr[d], rtol=0, atol=2e-5)
    rays = oracle.pack_rays(N_(ro), N_(rd), 2.0, 6.0)
    o_c = oracle.render_rays(oracle.scene(planes, g["box"]), oracle.decoder(decoder_blob(sd(g, "coarse."))),
                             oracle.decoder(decoder_blob(sd(g, "fine."))), rays, 32, 32, want_aux=True)
    fo = oracle.render_given_z(oracle.scene(hr, g["box"]), oracle.decoder(decoder_blob(sd(g, "fine."))), rays, o_c["z_fine"])
    err = np.abs(N_(img_f).reshape(-1, 3) - fo["rgb"]).max(-1)
    assert np.mean(err <= 2e-4) >= 0.97 and psnr(N_(img_f).reshape(-1, 3), fo["rgb"]) >= 70.0, (np.mean(err <= 2e-4), err.max())
    # skip_SR(True) falls back to the LR planes (train_nerf.py:701-705)
    mf.skip_SR(True)
    _, _, _, img_lr, *_ = hip.train_utils.eval_nerf(H, W, focal, mc, mf, ro, rd, opts, scene_id=sid, scene_config=scfg)
    np.testing.assert_allclose(N_(img_lr).reshape(-1, 3), o_c["rgb_fine"], rtol=0, atol=1e-3)
    assert not torch.equal(img_lr, img_f)


def test_positional_encoding_and_nerf_mlp_golden(hip):
    g = load_golden("g10_posenc.npz")
    pe = hip.nerf_helpers.positional_encoding
    np.testing.assert_allclose(N_(pe(T(g["x"]), 6, True)), g["pe_L6"], rtol=0, atol=4e-6)
    np.testing.assert_allclose(N_(pe(T(g["x"]), 4, True)), g["pe_L4"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(N_(pe(T(g["x"]), 4, False)), g["pe_L4_noinput"], rtol=0, atol=2e-6)
    m = hip.models.FlexibleNeRFModel(num_layers=4, hidden_size=128, skip_connect_every=3, num_encoding_fn_xyz=6, num_encoding_fn_dir=4)
    m.load_state_dict({k[3:]: torch.as_tensor(v) for k, v in g.items() if k.startswith("sd.")}, strict=True)
    m = m.to(DEV)
    x = torch.cat([pe(T(g["nerf_pts"]), 6), pe(T(g["nerf_dirs"]), 4)], -1)
    np.testing.assert_allclose(N_(m(x)), g["nerf_out"], rtol=0, atol=1e-5)


# ---------------------------------------------------------------------------------------------------------------------
# training: gradients with respect to the feature planes
def _grad_models(hip, g, planes, sid, what=("planes",)):
    mc, _ = build_model(hip, sd(g, "coarse."), planes, g["box"], sid=sid)
    mf, _ = build_model(hip, sd(g, "fine."), planes, g["box"], sid=sid)
    mf.planes_ = mc.planes_
    for m in (mc, mf):
        for n, p in m.named_parameters():
            # ("planes",): decoder frozen (Feature_Planes_Only.yml); ("decoder",) / both: nerf.train.what of train_nerf.py:75-77
            is_plane = "planes_" in n
            p.requires_grad_(("planes" in what) if is_plane else ("decoder" in what and "rot_mats" not in n))
        m.train()
    return mc, mf


def _decoder_grad_blob(model):
    return np.concatenate([N_(p.grad).reshape(-1) for p in model.decoder_parameters()])


def test_plane_gradients_golden(hip):
    """loss.backward() through run_one_iter_of_nerf vs torch.autograd through the reference (g11)"""
    g = load_golden("g11_grads.npz")
    planes = [g["plane%d" % d] for d in range(4)]
    sid = "lego_DS8_PlRes16_8"
    mc, mf = _grad_models(hip, g, planes, sid)
    rays = T(g["rays"])
    target = T(g["target"])
    for ci in range(int(g["n_cases"])):
        nc, nf, perturb, std = g["c%d_params" % ci]
        opts, scfg = make_options(int(nc), int(nf), perturb=bool(perturb), noise=float(std))
        rnd = {k: T(g["c%d_%s" % (ci, k)]) for k in ("t_rand", "u", "noise_coarse", "noise_fine") if "c%d_%s" % (ci, k) in g}
        for p_ in mc.planes_.values():
            p_.grad = None
        out = hip.train_utils.run_one_iter_of_nerf(16, 16, float(g["hwf"][2]), mc, mf, rays, opts, sid, mode="train", scene_config=scfg,
                                                   randoms=rnd)
        np.testing.assert_allclose(N_(out[0]), g["c%d_rgb_coarse" % ci], rtol=0, atol=2e-5)
        loss = torch.nn.functional.mse_loss(out[0], target) + torch.nn.functional.mse_loss(out[3], target)
        assert abs(float(loss.detach()) - float(g["c%d_loss" % ci])) < 2e-4
        loss.backward()
        for d in range(4):
            got = N_(mc.planes_[hip.models.get_plane_name(sid, d)].grad)
            ref = g["c%d_grad_plane%d" % (ci, d)]
            assert got.shape == ref.shape
            # the fine depths are regenerated by each side: a few importance samples move (sample_pdf conditioning) and shift
            # gradient between neighbouring texels, so the end-to-end comparison is in aggregate
            rel = np.linalg.norm(got - ref) / np.linalg.norm(ref)
            assert rel < 1e-2, "case %d plane %d: relative L2 error %.2e" % (ci, d, rel)
            assert np.abs(got - ref).max() <= 3e-2 * np.abs(ref).max()


def test_decoder_gradients_golden(hip):
    """loss.backward() fills the decoder parameters' .grad of both models like torch.autograd through the reference (g13); planes
    and decoder trained together, then the decoder alone (planes frozen)"""
    g, gd = load_golden("g11_grads.npz"), load_golden("g13_decoder_grads.npz")
    planes = [g["plane%d" % d] for d in range(4)]
    sid = "lego_DS8_PlRes16_8"
    rays, target = T(g["rays"]), T(g["target"])
    for what in (("planes", "decoder"), ("decoder",)):
        mc, mf = _grad_models(hip, g, planes, sid, what=what)
        for ci in range(int(g["n_cases"])):
            nc, nf, perturb, std = g["c%d_params" % ci]
            opts, scfg = make_options(int(nc), int(nf), perturb=bool(perturb), noise=float(std))
            rnd = {k: T(g["c%d_%s" % (ci, k)]) for k in ("t_rand", "u", "noise_coarse", "noise_fine") if "c%d_%s" % (ci, k) in g}
            for m in (mc, mf):
                m.zero_grad(set_to_none=True)
            out = hip.train_utils.run_one_iter_of_nerf(16, 16, float(g["hwf"][2]), mc, mf, rays, opts, sid, mode="train", scene_config=scfg,
                                                       randoms=rnd)
            loss = torch.nn.functional.mse_loss(out[0], target) + torch.nn.functional.mse_loss(out[3], target)
            loss.backward()
            for tag, m in (("coarse", mc), ("fine", mf)):
                got, ref = _decoder_grad_blob(m), gd["c%d_%s_grad" % (ci, tag)]
                assert got.shape == ref.shape
                rel = np.linalg.norm(got - ref) / np.linalg.norm(ref)
                # coarse: identical depths on both sides -> fp32 summation order only; fine: depths regenerated (sample_pdf conditioning)
                assert rel < (1e-4 if tag == "coarse" else 1e-2), "case %d %s decoder %s: relative L2 error %.2e" % (ci, tag, what, rel)
                assert np.abs(got - ref).max() <= (1e-4 if tag == "coarse" else 3e-2) * np.abs(ref).max()
            p0 = mc.planes_[hip.models.get_plane_name(sid, 0)]
            if "planes" in what:
                ref0 = g["c%d_grad_plane0" % ci]
                assert np.linalg.norm(N_(p0.grad) - ref0) / np.linalg.norm(ref0) < 1e-2
            else:
                assert p0.grad is None


def test_decoder_gradients_vs_oracle_larger(hip, oracle):
    """700 rays (not a multiple of the 256-ray tile: padding slots of the record must contribute exact zeros), 40+56 samples,
    non-square planes; at the SAME fine depths the HIP weight gradients match the analytic double-precision oracle"""
    g = load_golden("g11_grads.npz")
    rng = np.random.default_rng(41)
    planes = [rng.standard_normal((1, 48, 36, 44), dtype=np.float32) * 0.5 for _ in range(3)] + \
             [rng.standard_normal((1, 48, 10, 14), dtype=np.float32) * 0.5]
    sid = "lego_DS8_PlRes36_10"
    mc, mf = _grad_models(hip, g, planes, sid, what=("decoder",))
    N, nc, nf = 700, 40, 56
    H = W = 40
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, T(load_golden("g08_render.npz")["pose"]))
    sel = torch.from_numpy(rng.permutation(H * W)[:N]).to(DEV)
    batch = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0)
    opts, scfg = make_options(nc, nf)
    old, old_max = hip.train_utils.RECORD_RAYS, hip.train_utils.RECORD_FORWARD_MAX_POINTS
    try:
        results = []
        # (1) the forward records, gate-driven backward; (2) recomputing backward, one launch; (3) recomputing, 256 + 256 + 188 rays
        for rec_rays, fwd_max in ((old, old_max), (old, 0), (256, 0)):
            hip.train_utils.RECORD_RAYS, hip.train_utils.RECORD_FORWARD_MAX_POINTS = rec_rays, fwd_max
            for m in (mc, mf):
                m.zero_grad(set_to_none=True)
            out = hip.train_utils.run_one_iter_of_nerf(H, W, focal, mc, mf, batch, opts, sid, mode="train", scene_config=scfg, randoms={})
            z_fine = N_(out[3].grad_fn.saved["z_f"])
            if not results:
                gc = T(rng.standard_normal((N, 3)).astype(np.float32) / N)
                gf = T(rng.standard_normal((N, 3)).astype(np.float32) / N)
            ((out[0] * gc).sum() + (out[3] * gf).sum()).backward()
            results.append((_decoder_grad_blob(mc), _decoder_grad_blob(mf)))
            assert (out[3].grad_fn.saved["rec_f"] is not None) == (fwd_max > 0)
    finally:
        hip.train_utils.RECORD_RAYS, hip.train_utils.RECORD_FORWARD_MAX_POINTS = old, old_max
    sc = oracle.scene(planes, g["box"])
    rays_np = oracle.pack_rays(N_(batch[0]), N_(batch[1]), 2.0, 6.0)
    dec_c, dec_f = oracle.decoder(decoder_blob(sd(g, "coarse."))), oracle.decoder(decoder_blob(sd(g, "fine.")))
    ref = oracle.render_backward_decoder(sc, dec_c, dec_f, rays_np, nc, nf, N_(gc), N_(gf), z_fine=z_fine)
    for got in results:
        for tag, a, b in zip(("coarse", "fine"), got, ref):
            rel = np.linalg.norm(a - b) / np.linalg.norm(b)
            # ReLU masks of pre-activations within fp32 noise of zero flip between the fp32 kernel and the double oracle
            assert rel < 2e-3, "%s decoder: relative L2 error %.2e" % (tag, rel)
            assert np.abs(a - b).max() <= 5e-3 * np.abs(b).max()
    # the two recomputing paths differ by summation order only; since round 4 the recording forward runs the arithmetic of the pass (2 f16 limbs
    # by default, like the forward without a record: same raw, same fine depths), so the recording path differs from them by the backward
    # kernels' summation order and limb arithmetic alone
    for a, b in zip(results[1], results[2]):
        assert np.linalg.norm(a - b) / np.linalg.norm(b) < 1e-5
    for a, b in zip(results[0], results[1]):
        assert np.linalg.norm(a - b) / np.linalg.norm(b) < 1e-5


def test_plane_gradients_gate_path_equals_recompute_path(hip):
    """with the decoder frozen the backward is driven by the ReLU gates the forward published; with the decoder trained it
    recomputes the forward: the plane gradients of the two kernels agree to float-atomic summation order"""
    g = load_golden("g11_grads.npz")
    rng = np.random.default_rng(91)
    planes = [rng.standard_normal((1, 48, 40, 56), dtype=np.float32) * 0.5 for _ in range(3)] + \
             [rng.standard_normal((1, 48, 12, 12), dtype=np.float32) * 0.5]
    sid = "lego_DS8_PlRes40_12"
    N, nc, nf = 700, 40, 56                                  # 700 rays: partial 256-ray and 128-ray tiles
    H = W = 40
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    res = []
    for what in (("planes",), ("planes", "decoder")):
        mc, mf = _grad_models(hip, g, planes, sid, what=what)
        ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, T(load_golden("g08_render.npz")["pose"]))
        sel = torch.from_numpy(np.random.default_rng(92).permutation(H * W)[:N]).to(DEV)
        batch = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0)
        opts, scfg = make_options(nc, nf, white=True)
        old_max = hip.train_utils.RECORD_FORWARD_MAX_POINTS
        hip.train_utils.RECORD_FORWARD_MAX_POINTS = 0        # decoder gradients through the RECOMPUTING kernel in this test
        try:
            out = hip.train_utils.run_one_iter_of_nerf(H, W, focal, mc, mf, batch, opts, sid, mode="train", scene_config=scfg, randoms={})
        finally:
            hip.train_utils.RECORD_FORWARD_MAX_POINTS = old_max
        saved = out[3].grad_fn.saved
        assert (saved["gates_f"] is not None) == (what == ("planes",))
        gc = T(np.random.default_rng(93).standard_normal((N, 3)).astype(np.float32) / N)
        gf = T(np.random.default_rng(94).standard_normal((N, 3)).astype(np.float32) / N)
        ((out[0] * gc).sum() + (out[3] * gf).sum()).backward()
        res.append([N_(mc.planes_[hip.models.get_plane_name(sid, d)].grad) for d in range(4)])
    for d in range(4):
        rel = np.linalg.norm(res[0][d] - res[1][d]) / np.linalg.norm(res[1][d])
        assert rel < (1e-5 if d < 3 else 2e-4), "plane %d: gate path vs recompute path %.2e" % (d, rel)


def test_decoder_weight_grad_contraction(hip):
    """nvsr_decoder_weight_grad alone: a synthetic record (random G / X / H / g4) against float64 matmuls, through the C ABI; an even
    and an odd number of rows (rows are consumed in pairs; allocation is rounded up to 8 rows that must be ignored)"""
    capi = hip.capi
    # both kernels: exact-f32 MFMA (rows in pairs) and the default 3-limb bf16 MFMA (16 rows per step, [128 x 128] blocks); row counts that
    # are / are not multiples of 16, one that spans several slabs per block; the allocation padding holds NaNs
    for mode, N, S in (("f32", 300, 7), ("f32", 301, 7), ("bf16x3", 300, 7), ("bf16x3", 301, 7), ("bf16x3", 2, 8), ("bf16x3", 1101, 33),
                       ("f32", 1101, 33)):
        P = N * S
        n = capi.lib().nvsr_decoder_record_floats(N, S)
        Pp = (P + 7) // 8 * 8 + 32                  # (round 6: + the 32 dump rows the staged record stores send a partial tile's padding points to)
        assert n == Pp * 2308
        g_ = torch.Generator(device="cpu").manual_seed(5 + N)
        rec = torch.randn(n, generator=g_, dtype=torch.float32)
        r = rec.numpy()
        o = 0

        def take(cols, k=1):
            nonlocal o
            a = r[o:o + k * cols * Pp].reshape(k, Pp, cols)
            a[:, P:] = np.nan                                           # rows >= P are allocation padding: never read
            o += k * cols * Pp
            return a[:, :P].astype(np.float64)

        Xd, Hd, Gd, Xr, Hr, Gr, g4 = take(64)[0], take(128, 4), take(128, 4), take(192)[0], take(128, 4), take(128, 4), take(4)[0]
        rec_d = rec.to(DEV)
        grad = torch.zeros(capi.DECODER_NATURAL_FLOATS, device=DEV)
        capi.set_decoder_arithmetic(mode)
        try:
            capi.call("nvsr_decoder_weight_grad", N, S, capi.ptr(rec_d), capi.ptr(grad), capi.stream())
            capi.call("nvsr_decoder_weight_grad", N, S, capi.ptr(rec_d), capi.ptr(grad), capi.stream())    # accumulates: twice the gradient
        finally:
            capi.set_decoder_arithmetic(DEFAULT_ARITHMETIC)
        got = N_(grad) / 2

        parts = []
        for X, H, G, width, head in ((Xd, Hd, Gd, 48, g4[:, 3:4]), (Xr, Hr, Gr, 192, g4[:, :3])):
            parts += [(G[0].T @ X[:, :width]).ravel(), G[0].sum(0)]
            for l in range(1, 4):
                parts += [(G[l].T @ H[l - 1]).ravel(), G[l].sum(0)]
            parts += [(head.T @ H[3]).ravel(), head.sum(0)]
        ref = np.concatenate(parts)
        assert ref.size == got.size
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-4 * np.sqrt(P))     # sums of P N(0,1) products in fp32
        assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 1e-5, mode


def test_composite_backward_vs_autograd_formula(hip):
    """nvsr_composite_backward against torch.autograd of the same formula evaluated in float64 on the host"""
    rng = np.random.default_rng(21)
    N, S = 37, 96
    raw = rng.standard_normal((N, S, 4)).astype(np.float32) * 2
    raw[..., 3] = raw[..., 3] * 3 - 1
    raw[3, 10, 3] = 60.0                      # an opaque sample: everything behind it has T ~ 1e-10
    z = np.sort(rng.uniform(2, 6, (N, S)).astype(np.float32), -1)
    rd = rng.standard_normal((N, 3)).astype(np.float32)
    noise = (rng.standard_normal((N, S)) * 0.2).astype(np.float32)
    g_rgb = rng.standard_normal((N, 3)).astype(np.float32)
    g_acc = rng.standard_normal(N).astype(np.float32)
    for white in (0, 1):
        r64 = torch.tensor(raw, dtype=torch.float64, requires_grad=True)
        z64, rd64, n64 = (torch.tensor(a, dtype=torch.float64) for a in (z, rd, noise))
        dists = torch.cat([z64[:, 1:] - z64[:, :-1], torch.full((N, 1), 1e10, dtype=torch.float64)], -1) * rd64.norm(dim=-1, keepdim=True)
        alpha = 1.0 - torch.exp(-torch.relu(r64[..., 3] + n64) * dists)
        T_ = torch.cumprod(torch.cat([torch.ones(N, 1, dtype=torch.float64), (1.0 - alpha + 1e-10)[:, :-1]], -1), -1)
        w = alpha * T_
        rgb = (w[..., None] * torch.sigmoid(r64[..., :3])).sum(1)
        acc = w.sum(-1)
        if white:
            rgb = rgb + (1.0 - acc[:, None])
        (rgb * torch.tensor(g_rgb, dtype=torch.float64)).sum().add((acc * torch.tensor(g_acc, dtype=torch.float64)).sum()).backward()
        ref = r64.grad.numpy()
        capi = hip.capi
        raw_d, z_d, rd_d, n_d, gr_d, ga_d = T(raw), T(z), T(rd), T(noise), T(g_rgb), T(g_acc)
        out = torch.empty((N, S, 4), device=DEV)
        capi.call("nvsr_composite_backward", N, S, capi.ptr(raw_d), capi.ptr(z_d), capi.ptr(rd_d), capi.ptr(n_d), white, capi.ptr(gr_d),
                  capi.ptr(ga_d), capi.ptr(out), capi.stream())
        got = N_(out)
        scale = np.abs(ref).max()
        np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-6 * scale)


def test_model_call_is_differentiable_vs_torch_reference(hip):
    """TwoDimPlanesModel.forward(points) in training mode carries gradients for the planes and the decoder; reference = the same op
    in plain PyTorch fp32 on the GPU (grid_sample + linear layers, autograd)"""
    import torch.nn.functional as F
    g = load_golden("g11_grads.npz")
    rng = np.random.default_rng(77)
    planes = [rng.standard_normal((1, 48, 20, 28), dtype=np.float32) * 0.5 for _ in range(3)] + \
             [rng.standard_normal((1, 48, 9, 11), dtype=np.float32) * 0.5]
    sid = "lego_DS8_PlRes20_9"
    mc, _ = _grad_models(hip, g, planes, sid, what=("planes", "decoder"))
    P = 1000                                    # ragged against the 128- and 256-point tiles
    x = np.concatenate([rng.uniform(-4.2, 4.2, (P, 3)), rng.standard_normal((P, 3))], -1).astype(np.float32)
    x[:, 3:] /= np.linalg.norm(x[:, 3:], axis=-1, keepdims=True)
    G = T(rng.standard_normal((P, 4)).astype(np.float32))
    out = mc(T(x))
    assert out.requires_grad
    (out * G).sum().backward()
    got_planes = [N_(mc.planes_[hip.models.get_plane_name(sid, d)].grad) for d in range(4)]
    got_dec = _decoder_grad_blob(mc)
    # ---- torch reference
    box = torch.as_tensor(g["box"], dtype=torch.float64)
    pl = [T(p).requires_grad_(True) for p in planes]
    sdc = {k: T(v).requires_grad_(True) for k, v in sd(g, "coarse.").items() if "rot_mats" not in k}
    xt = T(x)
    az = torch.atan2(xt[:, 4], xt[:, 3]); el = torch.atan2(xt[:, 5], torch.sqrt(xt[:, 3] ** 2 + xt[:, 4] ** 2))
    x5 = torch.cat([xt[:, :3], az[:, None], el[:, None]], -1)
    n5 = 2 * (x5 - box[0].float().to(DEV)) / (box[1] - box[0]).float().to(DEV) - 1
    rot = [N_(mc.coord_projector.rot_mats_NON_LEARNED[d])[:, 1:] for d in range(3)]
    feats = []
    for d in range(3):
        grid = (n5[:, :3] @ T(rot[d])).reshape(1, P, 1, 2)
        feats.append(F.grid_sample(pl[d], grid, mode="bilinear", align_corners=True, padding_mode="border")[0, :, :, 0].t())
    fv = F.grid_sample(pl[3], n5[:, 3:].reshape(1, P, 1, 2), mode="bilinear", align_corners=True, padding_mode="border")[0, :, :, 0].t()
    hden = torch.stack(feats, 0).mean(0)
    for l in range(4):
        hden = torch.relu(F.linear(hden, sdc["density_dec.0.%d.weight" % l], sdc["density_dec.0.%d.bias" % l]))
    sigma = F.linear(hden, sdc["fc_alpha.0.weight"], sdc["fc_alpha.0.bias"])
    hrgb = torch.cat(feats + [fv], -1)
    for l in range(4):
        hrgb = torch.relu(F.linear(hrgb, sdc["rgb_dec.0.%d.weight" % l], sdc["rgb_dec.0.%d.bias" % l]))
    rgb = F.linear(hrgb, sdc["fc_rgb.0.weight"], sdc["fc_rgb.0.bias"])
    ref = torch.cat([rgb, sigma], -1)
    np.testing.assert_allclose(N_(out), N_(ref), rtol=0, atol=3e-5)
    (ref * G).sum().backward()
    for d in range(4):
        r = N_(pl[d].grad)
        assert np.linalg.norm(got_planes[d] - r) / np.linalg.norm(r) < 2e-4, "plane %d" % d
    keys = ["density_dec.0.%d" % i for i in range(4)] + ["fc_alpha.0"] + ["rgb_dec.0.%d" % i for i in range(4)] + ["fc_rgb.0"]
    ref_dec = np.concatenate([N_(sdc[k + "." + leaf].grad).reshape(-1) for k in keys for leaf in ("weight", "bias")])
    assert np.linalg.norm(got_dec - ref_dec) / np.linalg.norm(ref_dec) < 2e-4
    # evaluation mode / no_grad: the plain inference kernel, no graph
    mc.eval()
    assert not mc(T(x)).requires_grad


def _torch_decoder(hip, model, planes_t, state, box, x):
    """TwoDimPlanesModel.forward in plain PyTorch (grid_sample + linear layers) for [P,6] points"""
    import torch.nn.functional as F
    P = x.shape[0]
    az = torch.atan2(x[:, 4], x[:, 3]); el = torch.atan2(x[:, 5], torch.sqrt(x[:, 3] ** 2 + x[:, 4] ** 2))
    x5 = torch.cat([x[:, :3], az[:, None], el[:, None]], -1)
    n5 = 2 * (x5 - box[0].float().to(DEV)) / (box[1] - box[0]).float().to(DEV) - 1
    feats = []
    for d in range(3):
        rot = model.coord_projector.rot_mats_NON_LEARNED[d].detach()[:, 1:]
        grid = (n5[:, :3] @ rot).reshape(1, P, 1, 2)
        feats.append(F.grid_sample(planes_t[d], grid, mode="bilinear", align_corners=True, padding_mode="border")[0, :, :, 0].t())
    fv = F.grid_sample(planes_t[3], n5[:, 3:].reshape(1, P, 1, 2), mode="bilinear", align_corners=True, padding_mode="border")[0, :, :, 0].t()
    hden = torch.stack(feats, 0).mean(0)
    for l in range(4):
        hden = torch.relu(F.linear(hden, state["density_dec.0.%d.weight" % l], state["density_dec.0.%d.bias" % l]))
    sigma = F.linear(hden, state["fc_alpha.0.weight"], state["fc_alpha.0.bias"])
    hrgb = torch.cat(feats + [fv], -1)
    for l in range(4):
        hrgb = torch.relu(F.linear(hrgb, state["rgb_dec.0.%d.weight" % l], state["rgb_dec.0.%d.bias" % l]))
    return torch.cat([F.linear(hrgb, state["fc_rgb.0.weight"], state["fc_rgb.0.bias"]), sigma], -1)


def test_render_step_vs_torch_reference(hip):
    """3 000 rays, 48 + 96 samples: the whole predict_and_render_radiance chain restated with torch ops on the GPU (linspace depths,
    grid_sample decoder, cumprod compositing, cumsum / searchsorted inverse-CDF sampling, sort) against run_one_iter_of_nerf; forward
    and the gradient of a random linear functional of both images with respect to the planes"""
    g = load_golden("g11_grads.npz")
    rng = np.random.default_rng(55)
    planes = [rng.standard_normal((1, 48, 48, 40), dtype=np.float32) * 0.5 for _ in range(3)] + \
             [rng.standard_normal((1, 48, 12, 10), dtype=np.float32) * 0.5]
    sid = "lego_DS8_PlRes48_12"
    mc, mf = _grad_models(hip, g, planes, sid, what=("planes",))
    N, nc, nf = 3000, 48, 96
    H = W = 64
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, T(load_golden("g08_render.npz")["pose"]))
    sel = torch.from_numpy(rng.permutation(H * W)[:N]).to(DEV)
    ro, rd = ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]
    opts, scfg = make_options(nc, nf)
    out = hip.train_utils.run_one_iter_of_nerf(H, W, focal, mc, mf, torch.stack([ro, rd], 0), opts, sid, mode="train", scene_config=scfg, randoms={})
    gc, gf = T(rng.standard_normal((N, 3)).astype(np.float32) / N), T(rng.standard_normal((N, 3)).astype(np.float32) / N)
    ((out[0] * gc).sum() + (out[3] * gf).sum()).backward()
    got = [N_(mc.planes_[hip.models.get_plane_name(sid, d)].grad) for d in range(4)]
    # ---- reference chain
    box = torch.as_tensor(g["box"], dtype=torch.float64)
    pl = [T(p).requires_grad_(True) for p in planes]
    st_c = {k: T(v) for k, v in sd(g, "coarse.").items()}
    st_f = {k: T(v) for k, v in sd(g, "fine.").items()}
    vd = rd / rd.norm(dim=-1, keepdim=True)

    def render(model, st, z):
        pts = ro[:, None, :] + rd[:, None, :] * z[..., None]
        x = torch.cat([pts, vd[:, None, :].expand_as(pts)], -1).reshape(-1, 6)
        raw = _torch_decoder(hip, model, pl, st, box, x).reshape(N, -1, 4)
        dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full((N, 1), 1e10, device=DEV)], -1) * rd.norm(dim=-1, keepdim=True)
        alpha = 1.0 - torch.exp(-torch.relu(raw[..., 3]) * dists)
        Tr = torch.cumprod(torch.cat([torch.ones(N, 1, device=DEV), (1.0 - alpha + 1e-10)[:, :-1]], -1), -1)
        w = alpha * Tr
        return (w[..., None] * torch.sigmoid(raw[..., :3])).sum(1), w

    t = torch.linspace(0.0, 1.0, nc, device=DEV)
    z_c = (2.0 * (1.0 - t) + 6.0 * t).expand(N, nc)
    rgb_c, w_c = render(mc, st_c, z_c)
    with torch.no_grad():
        z_mid = 0.5 * (z_c[:, 1:] + z_c[:, :-1])
        wts = w_c[:, 1:-1] + 1e-5
        cdf = torch.cat([torch.zeros(N, 1, device=DEV), torch.cumsum(wts / wts.sum(-1, keepdim=True), -1)], -1)
        u = torch.linspace(0.0, 1.0, nf, device=DEV).expand(N, nf).contiguous()
        inds = torch.searchsorted(cdf, u, right=True)
        below, above = torch.clamp(inds - 1, min=0), torch.clamp(inds, max=cdf.shape[-1] - 1)
        c0, c1 = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
        b0, b1 = torch.gather(z_mid, 1, below), torch.gather(z_mid, 1, above)
        den = c1 - c0
        den = torch.where(den < 1e-5, torch.ones_like(den), den)
        z_f = torch.sort(torch.cat([z_c, b0 + (u - c0) / den * (b1 - b0)], -1), -1)[0]
    rgb_f, _ = render(mf, st_f, z_f)
    np.testing.assert_allclose(N_(out[0]), N_(rgb_c), rtol=0, atol=3e-5)
    err = (out[3] - rgb_f).abs().max(-1)[0]
    assert float((err <= 2e-4).float().mean()) >= 0.98 and psnr(N_(out[3]), N_(rgb_f)) >= 70.0
    ((rgb_c * gc).sum() + (rgb_f * gf).sum()).backward()
    for d in range(4):
        r = N_(pl[d].grad)
        assert np.linalg.norm(got[d] - r) / np.linalg.norm(r) < 1e-2, "plane %d" % d


def test_coarse_only_training_and_empty_batches(hip, oracle):
    """BASELINE config 1 shape (num_fine = 0): the train step has one pass only; empty ray batches are legal everywhere"""
    g = load_golden("g11_grads.npz")
    planes = [g["plane%d" % d] for d in range(4)]
    sid = "lego_DS8_PlRes16_8"
    mc, mf = _grad_models(hip, g, planes, sid, what=("planes", "decoder"))
    rays = T(g["rays"])
    N = rays.shape[1]
    opts, scfg = make_options(24, 0)
    out = hip.train_utils.run_one_iter_of_nerf(16, 16, float(g["hwf"][2]), mc, mf, rays, opts, sid, mode="train", scene_config=scfg, randoms={})
    assert out[3] is None and out[0].requires_grad
    gc = T(np.random.default_rng(3).standard_normal((N, 3)).astype(np.float32) / N)
    (out[0] * gc).sum().backward()
    sc = oracle.scene(planes, g["box"])
    dc, df = oracle.decoder(decoder_blob(sd(g, "coarse."))), oracle.decoder(decoder_blob(sd(g, "fine.")))
    rays_np = oracle.pack_rays(g["rays"][0], g["rays"][1], 2.0, 6.0)
    ref_p = oracle.render_backward(sc, [p.shape for p in planes], dc, df, rays_np, 24, 0, N_(gc), None)
    ref_d, _ = oracle.render_backward_decoder(sc, dc, df, rays_np, 24, 0, N_(gc), None)
    for d in range(4):
        got = N_(mc.planes_[hip.models.get_plane_name(sid, d)].grad)[0]
        assert np.linalg.norm(got - ref_p[d]) / np.linalg.norm(ref_p[d]) < (2e-3 if d < 3 else 5e-3)
    assert np.linalg.norm(_decoder_grad_blob(mc) - ref_d) / np.linalg.norm(ref_d) < 2e-3
    assert all(p_.grad is None for p_ in mf.decoder_parameters())          # the fine model never ran
    # empty batches
    empty = rays[:, :0]
    for mode in ("train", "validation"):
        o = hip.train_utils.run_one_iter_of_nerf(16, 16, 10.0, mc, mf, empty, opts, sid, mode=mode, scene_config=scfg, randoms={})
        assert o[0].shape == (0, 3) and o[2].shape == (0,)
    opts2, _ = make_options(8, 8)
    o = hip.train_utils.run_one_iter_of_nerf(16, 16, 10.0, mc, mf, empty, opts2, sid, mode="validation", scene_config=scfg)
    assert o[3].shape == (0, 3)


def test_ray_block_loop_matches_a_single_launch(hip):
    """run_one_iter_of_nerf splits batches beyond MAX_RAYS_PER_LAUNCH into ray blocks (the reference's ray-chunk loop, train_utils.py:
    228-247); rays are independent, so the blocked result is bit-identical -- also with train-mode random inputs, which are sliced per block"""
    g = load_golden("g08_render.npz")
    planes = [g["plane%d" % d] for d in range(4)]
    mc, sid = build_model(hip, sd(g, "coarse."), planes, g["box"], sid="lego_DS8_PlRes32_8")
    mf, _ = build_model(hip, sd(g, "fine."), planes, g["box"], sid=sid)
    mf.planes_ = mc.planes_
    H = W = 50
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, T(g["pose"]))
    batch = torch.stack([ro.reshape(-1, 3), rd.reshape(-1, 3)], 0)
    N = H * W
    opts, scfg = make_options(16, 24, perturb=True, noise=0.1)
    gen = torch.Generator().manual_seed(5)
    rnd = dict(t_rand=torch.rand(N, 16, generator=gen), u=torch.rand(N, 24, generator=gen), noise_coarse=0.1 * torch.randn(N, 16, generator=gen),
               noise_fine=0.1 * torch.randn(N, 40, generator=gen))
    old = hip.train_utils.MAX_RAYS_PER_LAUNCH
    try:
        with torch.no_grad():
            one = hip.train_utils.run_one_iter_of_nerf(H, W, focal, mc, mf, batch, opts, sid, mode="train", scene_config=scfg, randoms=rnd)
            hip.train_utils.MAX_RAYS_PER_LAUNCH = 777           # 2500 rays -> blocks of 777, 777, 777, 169
            blk = hip.train_utils.run_one_iter_of_nerf(H, W, focal, mc, mf, batch, opts, sid, mode="train", scene_config=scfg, randoms=rnd)
    finally:
        hip.train_utils.MAX_RAYS_PER_LAUNCH = old
    for a, b in zip(one[:6], blk[:6]):
        assert a.shape == b.shape
        assert_bits_equal(N_(a), N_(b))


def test_render_pass_generations_are_bit_identical(hip):
    """second-generation fused kernel (two tiles per wave, render2.hip) == first generation, bit for bit, on every output: 20 011 rays
    (partial last workgroup and a wave whose second tile is empty), 37 samples, density noise, white background, weights / depth / raw"""
    import ctypes as C
    import os
    g = load_golden("g08_render.npz")
    capi = hip.capi
    rng = np.random.default_rng(17)
    planes = [rng.standard_normal((1, 48, 64, 48), dtype=np.float32) * 0.5 for _ in range(3)] + \
             [rng.standard_normal((1, 48, 8, 12), dtype=np.float32) * 0.5]
    m, _ = build_model(hip, sd(g, "fine."), planes, g["box"])
    for N in (20011, 16384 + 33):
        S = 37
        H, W = 150, 160
        focal = 0.5 * W / np.tan(0.5 * 0.6911112)
        ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, T(g["pose"]))
        rays = hip.train_utils.pack_rays(ro, rd, 2.0, 6.0)[:N].contiguous()
        z = T(np.sort(rng.uniform(2, 6, (N, S)).astype(np.float32), -1))
        noise = T((rng.standard_normal((N, S)) * 0.3).astype(np.float32))
        sc, keep = m.native_scene()
        outs = []
        capi.set_decoder_arithmetic("f32")       # the two f32-MFMA generations; the bf16-limb kernel has its own test below
        for gen in ("1", None):
            if gen:
                os.environ["NVSR_RENDER_V1"] = gen
            else:
                os.environ.pop("NVSR_RENDER_V1", None)
            o = dict(rgb=torch.full((N, 3), -7.0, device=DEV), disp=torch.full((N,), -7.0, device=DEV), acc=torch.full((N,), -7.0, device=DEV),
                     w=torch.full((N, S), -7.0, device=DEV), depth=torch.full((N,), -7.0, device=DEV), raw=torch.full((N, S, 4), -7.0, device=DEV))
            capi.call("nvsr_render_pass_ex", C.byref(sc), capi.ptr(m.packed_decoder()), N, S, capi.ptr(rays), capi.ptr(z), capi.ptr(noise), 1,
                      capi.ptr(o["rgb"]), capi.ptr(o["disp"]), capi.ptr(o["acc"]), capi.ptr(o["w"]), capi.ptr(o["depth"]), capi.ptr(o["raw"]),
                      capi.stream())
            torch.cuda.synchronize()
            outs.append(o)
        os.environ.pop("NVSR_RENDER_V1", None)
        capi.set_decoder_arithmetic(DEFAULT_ARITHMETIC)
        for k in outs[0]:
            assert_bits_equal(N_(outs[0][k]), N_(outs[1][k]))
            assert not (N_(outs[1][k]) == -7.0).all()


def _limbs_of(words, limbs):
    """packed fragment words [.., 4] uint32 -> the limb values (as float64) of both halves: bf16 (3 limbs) or f16 of W * 2^8 (2 limbs)"""
    if limbs == 2:
        return ((words & 0xffff).astype(np.uint16).view(np.float16).astype(np.float64) / 256.0,
                (words >> 16).astype(np.uint16).view(np.float16).astype(np.float64) / 256.0)
    lo = (words & 0xffff).astype(np.uint32) << 16
    hi = (words & 0xffff0000).astype(np.uint32)
    return lo.view(np.float32).astype(np.float64), hi.view(np.float32).astype(np.float64)


def test_limb_fragments_reproduce_the_weights(hip):
    """packed blob, limb regions (include/nvsr.h, csrc/limb_core.h): the 3 bf16 limbs of every weight sum to the f32 weight EXACTLY;
    the 2 f16 limbs (of W * 2^8, rounded to nearest) to within 2^-23 |w| (one f32 ulp); fragment order = [K-block][out block][limb][lane][word] with the k-order of the C/D register layout"""
    g = load_golden("g08_render.npz")
    rng = np.random.default_rng(5)
    planes = [rng.standard_normal((1, 48, 8, 8), dtype=np.float32) for _ in range(4)]
    m, _ = build_model(hip, sd(g, "fine."), planes, g["box"])
    nat = N_(m.natural_blob())
    packed = N_(m.packed_decoder()).view(np.uint32)
    assert packed.size == hip.capi.DECODER_PACKED_FLOATS
    F32 = 130576
    KB = 63
    for limbs, base in ((3, F32), (2, F32 + KB * 3072)):
        fr = packed[base:base + KB * 4 * limbs * 256].reshape(KB, 4, limbs, 64, 4)          # [rec][ob][limb][lane][word]
        lo, hi = _limbs_of(fr, limbs)
        w = np.stack([lo, hi], -1).sum(2)                                                   # [rec][ob][lane][word][half]: limb sum
        lane = np.arange(64)
        mrow, h = lane & 31, lane >> 5
        for rec in (0, 2, 3, 11, 12, 19, 35, 36, 38, 39, 62):
            for ob in range(4):
                i = 32 * ob + mrow                                                          # output channel per lane
                e = 2 * np.arange(4)[None, :, None] + np.arange(2)[None, None, :]           # element index [1][word][half]
                if rec < 12:
                    plane = 3 if rec < 3 else rec // 3 - 1
                    src = 55937 + i[:, None, None] * 192 + 48 * plane + 24 * h[:, None, None] + 8 * (rec % 3) + e
                elif 36 <= rec < 39:
                    src = 0 + i[:, None, None] * 48 + 24 * h[:, None, None] + 8 * (rec - 36) + e
                else:
                    rgb = rec < 36
                    r = rec - (12 if rgb else 39)
                    layer, kb = r // 8, r % 8
                    k = 32 * (kb >> 1) + 16 * (kb & 1) + 8 * (e >> 2) + 4 * h[:, None, None] + (e & 3)
                    src = (80641 if rgb else 6272) + layer * 16512 + i[:, None, None] * 128 + k
                ref = nat[src].astype(np.float64)
                got = w[rec, ob]
                if limbs == 3:
                    assert np.array_equal(got, ref), (rec, ob)
                else:
                    assert np.all(np.abs(got - ref) <= np.abs(ref) * 2.0 ** -23 + 2.0 ** -33), (rec, ob)     # (+ the subnormal low limb below |w| = 2^-10)


def test_render_pass_limb_arithmetic(hip, oracle):
    """nvsr_render_pass with the decoder GEMMs on the bf16 matrix pipe (render3.hip): 3 limbs per operand = near-f32 products (bound in include/nvsr.h)
    (tolerances of the f32 kernels), 2 limbs = 16-bit operands (stated looser tolerance); against the f32-MFMA kernel and the oracle.
    20 011 rays (partial last workgroup, a wave whose second tile is empty), 37 samples, density noise, white background."""
    import ctypes as C
    g = load_golden("g08_render.npz")
    capi = hip.capi
    rng = np.random.default_rng(23)
    planes = [rng.standard_normal((1, 48, 64, 48), dtype=np.float32) * 0.5 for _ in range(3)] + \
             [rng.standard_normal((1, 48, 8, 12), dtype=np.float32) * 0.5]
    m, _ = build_model(hip, sd(g, "fine."), planes, g["box"])
    N, S = 20011, 37
    H, W = 150, 160
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, T(g["pose"]))
    rays = hip.train_utils.pack_rays(ro, rd, 2.0, 6.0)[:N].contiguous()
    z = T(np.sort(rng.uniform(2, 6, (N, S)).astype(np.float32), -1))
    noise = T((rng.standard_normal((N, S)) * 0.3).astype(np.float32))
    sc, keep = m.native_scene()
    res = {}
    try:
        for mode in ("f32", "bf16x3", "f16x2"):
            capi.set_decoder_arithmetic(mode)
            assert capi.get_decoder_arithmetic() == mode
            o = dict(rgb=torch.full((N, 3), -7.0, device=DEV), disp=torch.full((N,), -7.0, device=DEV), acc=torch.full((N,), -7.0, device=DEV),
                     w=torch.full((N, S), -7.0, device=DEV), depth=torch.full((N,), -7.0, device=DEV), raw=torch.full((N, S, 4), -7.0, device=DEV))
            capi.call("nvsr_render_pass_ex", C.byref(sc), capi.ptr(m.packed_decoder()), N, S, capi.ptr(rays), capi.ptr(z), capi.ptr(noise), 1,
                      capi.ptr(o["rgb"]), capi.ptr(o["disp"]), capi.ptr(o["acc"]), capi.ptr(o["w"]), capi.ptr(o["depth"]), capi.ptr(o["raw"]),
                      capi.stream())
            torch.cuda.synchronize()
            res[mode] = {k: N_(v).astype(np.float64) for k, v in o.items()}
    finally:
        capi.set_decoder_arithmetic(DEFAULT_ARITHMETIC)
    for k, v in res["bf16x3"].items():
        assert not (v == -7.0).any(), k
    scale = np.abs(res["f32"]["raw"]).max()
    # stated tolerances: both limb arithmetics hold the f32 kernels' own (2e-5 on pixels, 1e-5 of the range on decoder outputs)
    err_raw = {}
    for mode, t_raw, t_pix in (("bf16x3", 1e-5, 2e-5), ("f16x2", 1e-5, 2e-5)):
        err_raw[mode] = np.abs(res[mode]["raw"] - res["f32"]["raw"]).max() / scale
        print("%s: max |raw - raw_f32| / range = %.3g, rms = %.3g" % (mode, err_raw[mode], np.sqrt(np.mean((res[mode]["raw"] - res["f32"]["raw"]) ** 2)) / scale))
        assert np.abs(res[mode]["raw"] - res["f32"]["raw"]).max() <= t_raw * scale, mode
        for k in ("rgb", "acc", "w"):
            assert np.abs(res[mode][k] - res["f32"][k]).max() <= t_pix, (mode, k)
    # the oracle on a subset of the rays
    ids = rng.choice(N, 600, replace=False)
    osc = oracle.scene(planes, g["box"])
    dec = oracle.decoder(decoder_blob(sd(g, "fine.")))
    fo = oracle.render_given_z(osc, dec, N_(rays)[ids], N_(z)[ids], noise=N_(noise)[ids], white_background=True, want_raw=True)
    ok = ~_excluded_last_sigma(fo["raw"][:, -1, 3] + N_(noise)[ids][:, -1])
    for mode, tol in (("f32", 2e-5), ("bf16x3", 2e-5), ("f16x2", 2e-5)):
        print("%s: max |rgb - oracle| = %.3g" % (mode, np.abs(res[mode]["rgb"][ids][ok] - fo["rgb"][ok]).max()))
        np.testing.assert_allclose(res[mode]["rgb"][ids][ok], fo["rgb"][ok], rtol=0, atol=tol, err_msg=mode)
        np.testing.assert_allclose(res[mode]["acc"][ids][ok], fo["acc"][ok], rtol=0, atol=tol, err_msg=mode)


def test_render_pass_limb_kernel_edge_shapes_and_repeatability(hip):
    """bf16-limb render pass at the edges of its loop structure (1, 2 and 5 samples; a ray count that leaves a half-empty wave), without the
    optional outputs, twice: bit-identical launch to launch (the weight ring and the counted vector-memory waits have no race) and within
    the stated tolerance of the f32-MFMA kernel; rays whose last sigma is within the arithmetic's noise of zero excluded as everywhere"""
    import ctypes as C
    g = load_golden("g08_render.npz")
    capi = hip.capi
    rng = np.random.default_rng(31)
    planes = [rng.standard_normal((1, 48, 40, 56), dtype=np.float32) * 0.5 for _ in range(3)] + \
             [rng.standard_normal((1, 48, 8, 12), dtype=np.float32) * 0.5]
    m, _ = build_model(hip, sd(g, "fine."), planes, g["box"])
    H, W = 150, 160
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, T(g["pose"]))
    rays_all = hip.train_utils.pack_rays(ro, rd, 2.0, 6.0)
    sc, keep = m.native_scene()
    try:
        for N, S in ((16384 + 1, 1), (20000, 2), (23999, 5)):
            rays = rays_all[:N].contiguous()
            z = T(np.sort(rng.uniform(2, 6, (N, S)).astype(np.float32), -1))
            outs = {}
            for mode in ("f32", "bf16x3", "bf16x3", "f16x2", "f16x2"):
                capi.set_decoder_arithmetic(mode)
                o = dict(rgb=torch.full((N, 3), -7.0, device=DEV), disp=torch.full((N,), -7.0, device=DEV), acc=torch.full((N,), -7.0, device=DEV),
                         raw=torch.full((N, S, 4), -7.0, device=DEV))
                capi.call("nvsr_render_pass_ex", C.byref(sc), capi.ptr(m.packed_decoder()), N, S, capi.ptr(rays), capi.ptr(z), None, 0,
                          capi.ptr(o["rgb"]), capi.ptr(o["disp"]), capi.ptr(o["acc"]), None, None, capi.ptr(o["raw"]) if mode == "f32" else None,
                          capi.stream())
                torch.cuda.synchronize()
                if mode in outs:
                    for k in ("rgb", "disp", "acc"):
                        assert_bits_equal(N_(o[k]), N_(outs[mode][k]))
                outs[mode] = o
            ok = np.abs(N_(outs["f32"]["raw"])[:, -1, 3]) > 1e-4
            assert ok.mean() > 0.9
            for k in ("rgb", "acc"):
                for mode in ("bf16x3", "f16x2"):
                    assert not (N_(outs[mode][k]) == -7.0).any()
                    np.testing.assert_allclose(N_(outs[mode][k])[ok], N_(outs["f32"][k])[ok], rtol=0, atol=1e-4, err_msg=str((mode, N, S, k)))
    finally:
        capi.set_decoder_arithmetic(DEFAULT_ARITHMETIC)


def test_volume_render_radiance_field_is_differentiable(hip):
    """the mirror of volume_render_radiance_field carries a gradient for the radiance field (rgb_map, acc_map), equal to float64
    autograd of the same formula"""
    rng = np.random.default_rng(23)
    N, S = 19, 40
    raw = rng.standard_normal((N, S, 4)).astype(np.float32)
    z = np.sort(rng.uniform(2, 6, (N, S)).astype(np.float32), -1)
    rd = rng.standard_normal((N, 3)).astype(np.float32)
    g_rgb = rng.standard_normal((N, 3)).astype(np.float32)
    g_acc = rng.standard_normal(N).astype(np.float32)
    r = T(raw).requires_grad_(True)
    rgb, disp, acc, w, depth = hip.volume_rendering_utils.volume_render_radiance_field(r, T(z), T(rd), white_background=True)
    assert rgb.requires_grad and acc.requires_grad and not w.requires_grad
    ((rgb * T(g_rgb)).sum() + (acc * T(g_acc)).sum()).backward()
    r64 = torch.tensor(raw, dtype=torch.float64, requires_grad=True)
    z64, rd64 = torch.tensor(z, dtype=torch.float64), torch.tensor(rd, dtype=torch.float64)
    dists = torch.cat([z64[:, 1:] - z64[:, :-1], torch.full((N, 1), 1e10, dtype=torch.float64)], -1) * rd64.norm(dim=-1, keepdim=True)
    alpha = 1.0 - torch.exp(-torch.relu(r64[..., 3]) * dists)
    T_ = torch.cumprod(torch.cat([torch.ones(N, 1, dtype=torch.float64), (1.0 - alpha + 1e-10)[:, :-1]], -1), -1)
    wt = alpha * T_
    rgb64 = (wt[..., None] * torch.sigmoid(r64[..., :3])).sum(1) + (1.0 - wt.sum(-1))[:, None]
    ((rgb64 * torch.tensor(g_rgb, dtype=torch.float64)).sum() + (wt.sum(-1) * torch.tensor(g_acc, dtype=torch.float64)).sum()).backward()
    np.testing.assert_allclose(N_(rgb), rgb64.detach().numpy(), rtol=0, atol=2e-6)
    ref = r64.grad.numpy()
    np.testing.assert_allclose(N_(r.grad), ref, rtol=2e-4, atol=2e-6 * np.abs(ref).max())


def test_plane_gradients_vs_oracle_larger(hip, oracle):
    """600 rays, planes 40x56 (non-square) + view 12x12, 48+80 samples, white background: HIP backward vs the analytic oracle,
    first at the SAME fine depths (isolates the backward kernels), then end to end (depths regenerated by each side)."""
    g = load_golden("g11_grads.npz")
    rng = np.random.default_rng(31)
    planes = [rng.standard_normal((1, 48, 40, 56), dtype=np.float32) * 0.5 for _ in range(3)] + \
             [rng.standard_normal((1, 48, 12, 12), dtype=np.float32) * 0.5]
    sid = "lego_DS8_PlRes40_12"
    mc, mf = _grad_models(hip, g, planes, sid)
    N, nc, nf = 600, 48, 80
    H = W = 40
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, T(load_golden("g08_render.npz")["pose"]))
    sel = torch.from_numpy(rng.permutation(H * W)[:N]).to(DEV)
    batch = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0)
    opts, scfg = make_options(nc, nf, white=True)
    out = hip.train_utils.run_one_iter_of_nerf(H, W, focal, mc, mf, batch, opts, sid, mode="train", scene_config=scfg, randoms={})
    z_fine = N_(out[3].grad_fn.saved["z_f"]) if hasattr(out[3].grad_fn, "saved") else None
    gc = T(rng.standard_normal((N, 3)).astype(np.float32) / N)
    gf = T(rng.standard_normal((N, 3)).astype(np.float32) / N)
    ((out[0] * gc).sum() + (out[3] * gf).sum()).backward()
    sc = oracle.scene(planes, g["box"])
    rays_np = oracle.pack_rays(N_(batch[0]), N_(batch[1]), 2.0, 6.0)
    dec_c, dec_f = oracle.decoder(decoder_blob(sd(g, "coarse."))), oracle.decoder(decoder_blob(sd(g, "fine.")))
    grads = [N_(mc.planes_[hip.models.get_plane_name(sid, d)].grad)[0] for d in range(4)]
    assert z_fine is not None
    same_z = oracle.render_backward(sc, [p.shape for p in planes], dec_c, dec_f, rays_np, nc, nf, N_(gc), N_(gf), white_background=True,
                                    z_fine=z_fine)
    for d in range(4):
        rel = np.linalg.norm(grads[d] - same_z[d]) / np.linalg.norm(same_z[d])
        # the view-direction plane sums every sample of a ray into the same 4 texels with mixed signs: its net gradient is ~1000x
        # smaller than the terms added, so fp32 summation order shows (the reference's own fp32 autograd has the same noise)
        # position planes: ReLU pre-activations / last-sample sigmas within fp32 noise of zero flip their mask between the fp32
        # kernel and the double oracle for a handful of points; each flip moves a finite amount of gradient
        assert rel < (2e-3 if d < 3 else 5e-3), "same depths, plane %d: relative L2 error %.2e" % (d, rel)
    e2e = oracle.render_backward(sc, [p.shape for p in planes], dec_c, dec_f, rays_np, nc, nf, N_(gc), N_(gf), white_background=True)
    for d in range(4):
        rel = np.linalg.norm(grads[d] - e2e[d]) / np.linalg.norm(e2e[d])
        assert rel < 1e-2, "end to end, plane %d: relative L2 error %.2e" % (d, rel)
    # a second backward accumulates into .grad like any autograd leaf
    g0 = mc.planes_[hip.models.get_plane_name(sid, 0)].grad.clone()
    out = hip.train_utils.run_one_iter_of_nerf(H, W, focal, mc, mf, batch, opts, sid, mode="train", scene_config=scfg, randoms={})
    ((out[0] * gc).sum() + (out[3] * gf).sum()).backward()
    assert torch.allclose(mc.planes_[hip.models.get_plane_name(sid, 0)].grad, 2 * g0, rtol=1e-3, atol=1e-7)


def test_ndc_render_golden(hip):
    """BASELINE config 5 in miniature: scene_config.no_ndc = False routes the rays through nvsr_ndc_rays (train_utils.py:215-218)"""
    g = load_golden("g12_ndc_render.npz")
    planes = [g["plane%d" % d] for d in range(4)]
    sid = "fern_DS8_PlRes24_8"
    mc, _ = build_model(hip, sd(g, "coarse."), planes, g["box"], sid=sid)
    mf, _ = build_model(hip, sd(g, "fine."), planes, g["box"], sid=sid)
    mf.planes_ = mc.planes_
    H, W, focal = int(g["hwf"][0]), int(g["hwf"][1]), float(g["hwf"][2])
    ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, T(g["pose"]))
    assert_bits_equal(N_(rd), g["rd"])        # incl. the sign of the zero y-components of the middle image row
    opts, _ = make_options(64, 128)
    scfg = Opt(near=0, far=1, no_ndc=False)
    img_c, _, _, img_f, *_ = hip.train_utils.eval_nerf(H, W, focal, mc, mf, ro, rd, opts, scene_id=sid, scene_config=scfg)
    np.testing.assert_allclose(N_(img_c).reshape(-1, 3), g["rgb_coarse"], rtol=0, atol=3e-5)
    err = np.abs(N_(img_f).reshape(-1, 3) - g["rgb_fine"]).max(-1)
    assert np.mean(err <= 2e-4) >= 0.97 and psnr(N_(img_f).reshape(-1, 3), g["rgb_fine"]) >= 70.0, (np.mean(err <= 2e-4), err.max())


# ---------------------------------------------------------------------------------------------------------------------
# train() / evaluate() glue
# ---------------------------------------------------------------------------------------------------------------------
def test_ray_bundle_at_selected_pixels_bit_exact(hip):
    """rays of selected pixels only == the same rows of the full bundle, bit for bit (train_nerf.py:814,842-844)"""
    g = load_golden("g08_render.npz")
    rng = np.random.default_rng(71)
    for H, W, off in [(37, 53, 0.0), (100, 80, 0.4375)]:
        focal = 0.5 * W / np.tan(0.5 * 0.6911112)
        ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, T(g["pose"]), downsampling_offset=off)
        sel = torch.from_numpy(np.stack([rng.integers(0, H, 500), rng.integers(0, W, 500)], -1)).to(DEV)
        ro_s, rd_s = hip.training.get_ray_bundle_at(H, W, focal, T(g["pose"]), sel, downsampling_offset=off)
        assert_bits_equal(N_(rd_s), N_(rd[sel[:, 0], sel[:, 1]]))
        assert_bits_equal(N_(ro_s), N_(ro[sel[:, 0], sel[:, 1]]))
    e = hip.training.get_ray_bundle_at(8, 8, 10.0, T(g["pose"]), torch.zeros((0, 2), dtype=torch.int64, device=DEV))
    assert e[0].shape == (0, 3)


def _gt_and_student(hip, g, sid, R=20, Rv=8, seed=81):
    rng = np.random.default_rng(seed)
    gt_planes = [rng.standard_normal((1, 48, R, R), dtype=np.float32) * 0.5 for _ in range(3)] + \
                [rng.standard_normal((1, 48, Rv, Rv), dtype=np.float32) * 0.5]
    noisy = [p + rng.standard_normal(p.shape, dtype=np.float32) * 0.3 for p in gt_planes]
    return gt_planes, noisy


def test_train_step_glue_reduces_loss(hip):
    """TrainStep (train_nerf.py:790-923): planes + decoder trained against images rendered from ground-truth planes; the loss falls,
    virtual batches gate the decoder optimizer, the planes optimizer steps every iteration"""
    g = load_golden("g11_grads.npz")
    sid = "lego_DS8_PlRes20_8"
    gt_planes, noisy = _gt_and_student(hip, g, sid)
    H = W = 24
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    pose = T(load_golden("g08_render.npz")["pose"])
    opts, scfg = make_options(24, 24, perturb=True, noise=0.0)
    gt_c, gt_f = _grad_models(hip, g, gt_planes, sid, what=())
    ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, pose)
    with torch.no_grad():
        img = hip.train_utils.eval_nerf(H, W, focal, gt_c, gt_f, ro, rd, opts, scene_id=sid, scene_config=scfg)[3]
    mc, mf = _grad_models(hip, g, noisy, sid, what=("planes", "decoder"))
    dec_params = list({id(p_): p_ for m in (mc, mf) for p_ in m.decoder_parameters()}.values())
    opt = torch.optim.Adam(dec_params, lr=1e-4)
    popt = torch.optim.Adam(list(mc.planes_.values()), lr=2e-2)
    step = hip.training.TrainStep(mc, mf, opts, {"LR_planes", "decoder"}, optimizer=opt, planes_optimizer=popt, virtual_batch_size=2)
    np.random.seed(3)
    torch.manual_seed(3)
    w0 = mc.fc_alpha["0"].weight.detach().clone()
    p0 = mc.planes_[hip.models.get_plane_name(sid, 0)].detach().clone()
    r = step(0, img, pose, H, W, focal, 1, sid, scfg, 256)
    assert torch.equal(mc.fc_alpha["0"].weight, w0)                      # first half of a virtual batch: no decoder step yet
    assert not torch.equal(mc.planes_[hip.models.get_plane_name(sid, 0)], p0)   # the planes step every iteration (:904)
    losses = [r["loss"]]
    for it in range(1, 40):
        losses.append(step(it, img, pose, H, W, focal, 1, sid, scfg, 256)["loss"])
    assert not torch.equal(mc.fc_alpha["0"].weight, w0)
    assert np.mean(losses[-8:]) < 0.5 * np.mean(losses[:4]), losses
    assert r["psnr"] is not None and r["coarse_loss"] is not None and r["fine_loss"] is not None
    # the metrics are read lazily from pinned memory (training.StepMetrics): same keys and python floats as the reference's .item() calls
    assert set(r) == {"loss", "psnr", "coarse_loss", "fine_loss"} and all(isinstance(v, float) for v in dict(r).values())
    assert abs(r["loss"] - (r["coarse_loss"] + r["fine_loss"])) <= 1e-6 * r["loss"]
    assert r["psnr"] == hip.nerf_helpers.mse2psnr(r["loss"])


def test_sr_train_step_and_evaluate_view(hip):
    """what = ['SR']: only the SR optimizer moves, only the fine loss is taken with loss: 'fine' (:885-889); evaluate_view renders the
    SR scene twice (with and without the SR model) and reports the PSNR gain (:690-713)"""
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
    for m in (mc, mf):
        m.assign_SR_model(sr, SR_viewdir=False)
    mf.assign_LR_planes()
    ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, pose)
    for m in (mc, mf):
        m.skip_SR(True)
    with torch.no_grad():
        img = hip.train_utils.eval_nerf(H, W, focal, mc, mf, ro, rd, opts, scene_id=sid, scene_config=scfg)[3]   # target = the LR render
    for m in (mc, mf):
        m.skip_SR(False)
    ev = hip.training.evaluate_view(mc, mf, opts, sid, scfg, img, pose, H, W, focal, SR_model=sr, sr_scene=True)
    assert ev["fine_loss"] == 0.0 and ev["rgb_SR"] is not None and ev["psnr"] < 50.0     # the no-SR render IS the target
    assert abs(ev["SR_psnr_gain"] - (ev["psnr"] - 50.0)) < 1e-9                          # mse2psnr(0) = 50 dB by convention (:265-269)
    sr_opt = torch.optim.Adam(sr.parameters(), lr=2e-3)
    step = hip.training.TrainStep(mc, mf, opts, {"SR"}, SR_optimizer=sr_opt, SR_model=sr, sr_loss="fine")
    np.random.seed(4)
    first = None
    for it in range(12):
        r = step(it, img, pose, H, W, focal, 1, sid, scfg, 200, sr_iter=True)
        assert r["coarse_loss"] is None and r["fine_loss"] is not None
        first = first if first is not None else r["loss"]
    sr.clear_SR_planes()
    ev2 = hip.training.evaluate_view(mc, mf, opts, sid, scfg, img, pose, H, W, focal, SR_model=sr, sr_scene=True)
    assert ev2["loss"] < ev["loss"], (ev["loss"], ev2["loss"])           # the SR net learned to reproduce the target better
    assert all(p_.grad is None for p_ in mc.decoder_parameters())


@pytest.mark.parametrize("workload", ["render", "train", "sr"])
def test_bench_two_rank_rehearsal(workload):
    """bench.py's N > 1 path (barriers, max-over-ranks timing, whole-job value, the gradient all-reduce of the train workload) launched
    exactly as the driver launches it, with two ranks sharing this box's one GPU over gloo (NVSR_BENCH_REHEARSAL=1)."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, NVSR_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", workload,
           "--res", "160", "--plane-res", "128", "--no-cpu-baseline", "--no-modes"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, "\n".join(l for l in p.stderr.splitlines() if "[rank" in l)[-4000:] or p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                       # rank 0 prints ONE line
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["scaling"] == "weak" and r["value"] > 0
    per_gpu = r["config"]["planes_per_step_per_gpu" if workload == "sr" else "rays_per_step_per_gpu"]
    assert abs(r["value"] - 2 * per_gpu * 2 / (r["ms_per_step"] * 2e-3)) <= 1e-6 * r["value"]   # whole-job aggregate over both ranks
    assert "roofline" in r and "cpu_baseline" not in r


def test_loaded_scenes_feed_the_render_path(hip, tmp_path):
    """the data formats on the input side: a Blender-layout and an LLFF-layout scene loaded from disk drive
    get_ray_bundle / eval_nerf directly -- per-image [H, W, focal] and the camera-to-world matrices as the loaders return them, NDC rays
    for the forward-facing scene; and a train step draws its pixels from a loaded image"""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_host import _write_toy_scenes
    from bench import make_synthetic_scene, render_options
    g = load_golden("g15_loaders.npz")
    bdir, ldir = _write_toy_scenes(g, str(tmp_path))
    mc, mf, sid, _ = make_synthetic_scene(DEV, plane_res=32, view_res=8, seed=5)
    imgs, poses, render_poses, (H, W, focal, ds), i_split = hip.load_blender.load_blender_data(bdir, downsampling_factor=2, splits2use=["train", "val"])
    opts, scfg = render_options(16, 16)
    k = int(i_split[0][0])
    pose = poses[k].to(DEV)
    ro, rd = hip.nerf_helpers.get_ray_bundle(H[k], W[k], focal[k], pose, downsampling_offset=hip.training.downsampling_offset(ds[k]))
    out = hip.train_utils.eval_nerf(H[k], W[k], focal[k], mc, mf, ro, rd, opts, scene_id=sid, scene_config=scfg)
    assert out[3].shape == imgs[k].shape and torch.isfinite(out[3]).all()
    out = hip.train_utils.eval_nerf(H[k], W[k], focal[k], mc, mf, *hip.nerf_helpers.get_ray_bundle(H[k], W[k], focal[k], render_poses[7].float().to(DEV)),
                                    opts, scene_id=sid, scene_config=scfg)
    assert torch.isfinite(out[3]).all()
    # one optimisation step on pixels of the loaded image
    for m in (mc, mf):
        for n_, p_ in m.named_parameters():
            p_.requires_grad_("planes_" in n_)
    popt = torch.optim.Adam(list(mc.planes_.values()), lr=1e-2)
    topts, _ = render_options(16, 16, perturb=True, noise=0.0)
    step = hip.training.TrainStep(mc, mf, topts, {"LR_planes"}, planes_optimizer=popt)
    np.random.seed(0)
    r = step(0, imgs[k].to(DEV), pose, H[k], W[k], focal[k], ds[k], sid, scfg, 16)
    assert np.isfinite(r["loss"]) and r["psnr"] is not None
    # LLFF: [H, W, focal] ride in the fifth column of every pose; forward-facing scenes render through NDC rays
    images, lposes, bds, lrender, i_test, _ = hip.load_llff.load_llff_data(ldir, factor=2, base_factor=1, max_factor=4)
    h, w, f = (float(v) for v in lposes[i_test, :3, -1])
    c2w = lposes[i_test, :3, :4].to(DEV)
    ro, rd = hip.nerf_helpers.get_ray_bundle(int(h), int(w), f, c2w)
    ndc_cfg = dict(no_ndc=False, near=0.0, far=1.0)
    out = hip.train_utils.eval_nerf(int(h), int(w), f, mc, mf, ro, rd, opts, scene_id=sid, scene_config=ndc_cfg)
    assert out[3].shape == images[i_test].shape and torch.isfinite(out[0]).all()


def test_training_kernels_limb_vs_f32(hip):
    """The training forward (raw, ReLU gates, layer-input record) and the gate-driven backward (plane gradients, view rows, gradient record)
    in the default 3-limb arithmetic against the exact-f32 kernels on the same inputs, through the C ABI: ragged ray counts (a partial last
    workgroup, a single ray), 1 / 3 / 37 samples.  Stated tolerances: raw and every record row within 1e-5 of the output range; gates equal
    except where a pre-activation lies within that noise of zero (< 0.1 % of the bits); plane gradients 2e-5 relative L2 (float atomics
    reorder sums); the forward is bit-identical launch to launch (its weight ring has no race)."""
    import ctypes as C
    g = load_golden("g08_render.npz")
    capi = hip.capi
    rng = np.random.default_rng(41)
    planes = [rng.standard_normal((1, 48, 40, 56), dtype=np.float32) * 0.5 for _ in range(3)] + \
             [rng.standard_normal((1, 48, 8, 12), dtype=np.float32) * 0.5]
    m, _ = build_model(hip, sd(g, "fine."), planes, g["box"])
    H, W = 40, 50
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, T(g["pose"]))
    all_rays = hip.train_utils.pack_rays(ro, rd, 2.0, 6.0)
    sc, keep = m.native_scene()
    lib = capi.lib()
    for N, S in ((1, 1), (131, 3), (1000, 37)):
        rays = all_rays[:N].contiguous()
        z = T(np.sort(rng.uniform(2, 6, (N, S)).astype(np.float32), -1))
        g_raw = T((rng.standard_normal((N, S, 4)) * 1e-2).astype(np.float32))
        nrec = lib.nvsr_decoder_record_floats(N, S)
        res = {}
        try:
            for mode in ("f32", "bf16x3", "bf16x3"):
                capi.set_decoder_arithmetic(mode)
                raw = torch.full((N, S, 4), -7.0, device=DEV)
                gates = torch.zeros(N * S * 32, dtype=torch.int32, device=DEV)
                rec = torch.full((nrec,), float("nan"), device=DEV)
                capi.call("nvsr_decode_rays_ex", C.byref(sc), capi.ptr(m.packed_decoder()), N, S, capi.ptr(rays), capi.ptr(z), capi.ptr(raw),
                          capi.ptr(gates), capi.ptr(rec), capi.stream())
                fwd_rec = rec.clone()
                gpl = [torch.zeros_like(k) for k in keep]
                gptrs = (C.c_void_p * 4)(*[t.data_ptr() for t in gpl])
                vws = torch.zeros(lib.nvsr_view_grad_workspace_floats(N, S), device=DEV)
                capi.call("nvsr_render_pass_backward_gates", C.byref(sc), capi.ptr(m.packed_decoder()), capi.ptr(m.packed_decoder_bwd()), N, S,
                          capi.ptr(rays), capi.ptr(z), capi.ptr(g_raw), capi.ptr(gates), gptrs, capi.ptr(vws), capi.ptr(rec), capi.stream())
                torch.cuda.synchronize()
                cur = dict(raw=N_(raw), gates=N_(gates), fwd_rec=N_(fwd_rec), rec=N_(rec), gpl=[N_(t).astype(np.float64) for t in gpl])
                if mode in res:         # second launch in the same arithmetic
                    assert np.array_equal(cur["raw"], res[mode]["raw"]) and np.array_equal(cur["gates"], res[mode]["gates"])
                    assert np.array_equal(cur["fwd_rec"], res[mode]["fwd_rec"], equal_nan=True)
                res[mode] = cur
        finally:
            capi.set_decoder_arithmetic(DEFAULT_ARITHMETIC)
        a, b = res["f32"], res["bf16x3"]
        assert not (b["raw"] == -7.0).any()
        scale = max(1.0, np.abs(a["raw"]).max())
        assert np.abs(a["raw"].astype(np.float64) - b["raw"]).max() <= 1e-5 * scale, (N, S)
        flips = np.unpackbits((a["gates"] ^ b["gates"]).view(np.uint8)).sum()
        assert flips <= 1e-3 * N * S * 8 * 128 + 2, (N, S, flips)
        # record: rows < P of every array; the allocation padding behind them is never written (still NaN in both) -- except the last 32 rows of
        # every array: round 6's dump rows, where the limb kernels' staged stores send the padding points of a partial tile
        P, Pp = N * S, nrec // 2308
        assert Pp == (N * S + 7) // 8 * 8 + 32
        o = 0
        for cols, k in ((64, 1), (128, 4), (128, 4), (192, 1), (128, 4), (128, 4), (4, 1)):
            xa, xb = (res[mm]["rec"][o:o + k * cols * Pp].reshape(k, Pp, cols)[:, :Pp - 32] for mm in ("f32", "bf16x3"))
            assert np.array_equal(np.isnan(xa), np.isnan(xb)), (N, S, cols)
            for name in ("fwd_rec", "rec"):
                x, y = (res[mm][name][o:o + k * cols * Pp].reshape(k, Pp, cols)[:, :P].astype(np.float64) for mm in ("f32", "bf16x3"))
                if np.isnan(x).all():
                    continue                     # the gradient half before the backward has run
                if flips == 0:
                    rng_ = max(1e-3, np.abs(x).max())
                    assert np.abs(x - y).max() <= 1e-5 * rng_ + 1e-9, (N, S, name, cols)
                else:                            # a flipped gate zeroes / releases whole gradient rows: compare in the L2 sense
                    assert np.linalg.norm(x - y) <= 2e-3 * np.linalg.norm(x) + 1e-9, (N, S, name, cols)
            o += k * cols * Pp
        for d in range(4):
            na = np.linalg.norm(a["gpl"][d])
            assert np.linalg.norm(a["gpl"][d] - b["gpl"][d]) <= (2e-5 if flips == 0 else 2e-3) * na + 1e-12, (N, S, d)


def test_composite_mip_branch_golden_and_gradient(hip):
    """volume_render_radiance_field(mip_nerf=True) -- the Mip-NeRF baseline's intervals (volume_rendering_utils.py:19-26,41-42) -- against the
    reference's outputs and its autograd wrt the radiance field (fixture g16); 130 intervals = three 64-lane chunks per ray"""
    g = load_golden("g16_composite_mip.npz")
    for tag in ("a", "b"):
        raw = T(g[tag + "_raw"]).requires_grad_(True)
        out = hip.volume_rendering_utils.volume_render_radiance_field(raw, T(g[tag + "_z"]), T(g[tag + "_rd"]), white_background=bool(g[tag + "_white"]),
                                                                      mip_nerf=True, noise=T(g[tag + "_noise"]))
        rgb, disp, acc, w, depth = out
        np.testing.assert_allclose(N_(rgb), g[tag + "_rgb"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(N_(acc), g[tag + "_acc"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(N_(w), g[tag + "_weights"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(N_(depth), g[tag + "_depth"], rtol=0, atol=1e-5)
        np.testing.assert_allclose(N_(disp), g[tag + "_disp"], rtol=1e-5, atol=0, equal_nan=True)
        ((rgb * T(g[tag + "_g_rgb"])).sum() + (acc * T(g[tag + "_g_acc"])).sum()).backward()
        ref = g[tag + "_g_raw"]
        assert np.abs(N_(raw.grad) - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
        with torch.no_grad():        # the no-graph path takes the same kernel
            out2 = hip.volume_rendering_utils.volume_render_radiance_field(raw.detach(), T(g[tag + "_z"]), T(g[tag + "_rd"]),
                                                                           white_background=bool(g[tag + "_white"]), mip_nerf=True, noise=T(g[tag + "_noise"]))
        assert all(torch.equal(a, b) or (torch.isnan(a) == torch.isnan(b)).all() for a, b in zip(out, out2))


def test_channels_last_planes_are_used_in_place(hip):
    """A plane parameter in torch.channels_last memory format IS the kernels' [H][W][C] layout: rendering and a train step give bit-identical
    images / the same gradients as with the reference's NCHW parameters, the gradient comes back channels_last, and no copy is made"""
    from bench import make_synthetic_scene, render_options
    outs = {}
    for cl in (False, True):
        mc, mf, sid, pose = make_synthetic_scene(DEV, plane_res=48, view_res=8, seed=9, channels_last=cl)
        H = W = 20
        focal = 0.5 * W / np.tan(0.5 * 0.6911112)
        ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, pose)
        opts, scfg = render_options(16, 16)
        with torch.no_grad():
            img = hip.train_utils.eval_nerf(H, W, focal, mc, mf, ro, rd, opts, scene_id=sid, scene_config=scfg)[3]
        p0 = mc.planes_[hip.models.get_plane_name(sid, 0)]
        assert hip.models.is_native_layout(p0) == cl
        if cl:
            assert hip.models.to_channel_last(p0.detach()).data_ptr() == p0.data_ptr()        # a view, not a copy
        for m in (mc, mf):
            for n_, p_ in m.named_parameters():
                p_.requires_grad_("planes_" in n_)
            m.train()
        batch = torch.stack([ro.reshape(-1, 3)[:64], rd.reshape(-1, 3)[:64]], 0)
        out = hip.train_utils.run_one_iter_of_nerf(H, W, focal, mc, mf, batch, opts, sid, mode="train", scene_config=scfg, randoms={})
        (out[0].sum() + out[3].sum()).backward()
        assert p0.grad.shape == p0.shape and hip.models.is_native_layout(p0.grad) == cl
        outs[cl] = (img, [mc.planes_[hip.models.get_plane_name(sid, d)].grad.contiguous() for d in range(4)])
    assert torch.equal(outs[False][0], outs[True][0])
    for a, b in zip(outs[False][1], outs[True][1]):
        assert torch.allclose(a, b, rtol=0, atol=2e-5 * float(a.abs().max()) + 1e-12)     # float atomics reorder sums
