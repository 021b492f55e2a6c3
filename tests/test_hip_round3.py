"""GPU parity tests added in round 3 (-m gpu): the texel-deduplicating plane scatter of the training backward against the oracle, the
accumulate-into form of the backward operator, cumprod_exclusive as a differentiable operator, the exported fused-path threshold, and
(further down) the multi-GPU rehearsals with the HIP kernels and the small parity holes VERDICT r2 listed."""
import ctypes as C

import numpy as np
import pytest
import torch

from test_hip_parity import DEV, N_, T, build_model

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------------------------------------------------------------------------
# cumprod_exclusive: registered operator with autograd (nerf_helpers.py:409-430 is differentiable torch code in the reference)
# ---------------------------------------------------------------------------------------------------------------------------------
def test_cumprod_exclusive_is_differentiable(hip):
    """Forward = the golden-checked kernel; backward (nvsr_cumprod_exclusive_backward) against float64 torch.cumprod + roll autograd -- the
    reference helper verbatim in double precision -- on rows with zeros, ones, negative entries and a length-1 row."""
    rng = np.random.default_rng(5)
    for shape in [(7, 64), (3, 5, 33), (4, 1), (2, 192)]:
        x = rng.uniform(0.2, 1.1, size=shape).astype(np.float32)
        x.reshape(-1)[::7] *= -1.0
        if x.shape[-1] > 4:
            x[..., 3] = 0.0                      # a zero in the product: torch.cumprod's backward special-cases it, the recurrence here does not care
            x[0, ..., 1] = 1.0
        g = rng.standard_normal(shape).astype(np.float32)
        xt = T(x).requires_grad_(True)
        out = hip.nerf_helpers.cumprod_exclusive(xt)
        assert out.requires_grad, "cumprod_exclusive dropped the autograd graph"
        (out * T(g)).sum().backward()
        xd = torch.tensor(x, dtype=torch.float64, requires_grad=True)
        ref = torch.roll(torch.cumprod(xd, -1), 1, -1).clone()
        ref[..., 0] = 1.0
        (ref * torch.tensor(g, dtype=torch.float64)).sum().backward()
        np.testing.assert_allclose(N_(out.detach()), ref.detach().numpy(), rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(N_(xt.grad), xd.grad.numpy(), rtol=1e-5, atol=1e-6)
    torch.library.opcheck(torch.ops.nvsr.cumprod_exclusive, (T(x),), test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
    with pytest.raises(hip.capi.NvsrError):
        hip.nerf_helpers.cumprod_exclusive(torch.ones(2, 3))      # CPU tensor: no fallback


def test_fused_threshold_comes_from_the_library(hip):
    """train_utils' pass-by-pass path and the patch order use nvsr_fused_min_rays() of the LOADED library (a Python copy of
    NVSR_FUSED_MIN_RAYS could silently disagree with a retuned library)."""
    n = hip.capi.fused_min_rays()
    assert n == hip.capi.lib().nvsr_fused_min_rays() and n > 0
    assert hip.train_utils.PATCH_ORDER_MIN_RAYS == n
    # the workspace size switches at exactly that ray count
    lib = hip.capi.lib()
    assert lib.nvsr_render_workspace_floats(n, 64, 128) == n * (2 * 64 + 192)
    assert lib.nvsr_render_workspace_floats(n - 1, 64, 128) > (n - 1) * (2 * 64 + 192)


# ---------------------------------------------------------------------------------------------------------------------------------
# plane scatter of the limb backward: every texel of a wave tile written once (csrc/bwd_core.h scatter_plane_cached)
# ---------------------------------------------------------------------------------------------------------------------------------
def _backward_inputs(hip, N, S, plane_res, seed, z_kind):
    from bench import make_synthetic_scene

    mc, mf, sid, pose = make_synthetic_scene(DEV, plane_res=plane_res, view_res=8, seed=seed)
    g = torch.Generator(device="cpu").manual_seed(seed)
    H = W = 64
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    sel = torch.randint(0, H, (N, 2), generator=g).to(DEV)
    ro, rd = hip.training.get_ray_bundle_at(H, W, focal, pose, sel)
    rays = hip.train_utils.pack_rays(ro, rd, 2.0, 6.0)
    if z_kind == "dense":            # many samples per texel: long runs inside one cell
        z = torch.sort(torch.rand(N, S, generator=g) * 0.3 + 3.0, -1)[0]
    elif z_kind == "sparse":         # steps of several texels: no reuse at all
        z = torch.sort(torch.rand(N, S, generator=g) * 4.0 + 2.0, -1)[0]
    else:                            # unsorted depths: the path is NOT monotone, a slot's texel may come back after it was flushed
        z = torch.rand(N, S, generator=g) * 4.0 + 2.0
    return mf, rays, z.to(DEV).contiguous()


@pytest.mark.parametrize("z_kind,S,plane_res", [("dense", 96, 16), ("sparse", 40, 200), ("unsorted", 70, 48), ("dense", 33, 200)])
def test_deduplicated_scatter_matches_per_point_scatter(hip, z_kind, S, plane_res):
    """The limb backward sums, per wave tile (32 consecutive samples of a ray), everything that lands on one texel before it touches the
    plane (four slots keyed by texel parity).  Whatever the depths -- long runs in a cell, no reuse, even a non-monotone path where a
    flushed texel returns -- the gradient planes must equal those of the exact-f32 gate backward (one atomic set per point,
    render_bwd.hip) up to summation order and the limb arithmetic of the transposed layers: 1e-5 of the plane's largest gradient (measured
    <= 4e-6 with ~100 contributions per texel; one lost or doubled contribution would be >= 1e-3)."""
    nv = torch.ops.nvsr
    N = 257
    mf, rays, z = _backward_inputs(hip, N, S, plane_res, 11, z_kind)
    planes, consts = mf.scene_args()
    g_raw = torch.randn(N, S, 4, device=DEV) * 1e-2
    need = [True, True, True, True]
    out = {}
    # ONE set of ReLU gates (the f32 forward's) for both backward kernels: a gate whose pre-activation is rounding noise around zero may
    # differ between the two forwards, and this test is about the scatter
    _, gates, _ = nv.decode_rays(planes, consts, mf.packed_decoder(), rays, z, True, False, 0)
    for name, arith in (("f32", 0), ("bf16x3", 3)):
        out[name] = nv.decode_rays_backward(planes, consts, mf.packed_decoder(), mf.packed_decoder_bwd(), rays, z, g_raw, gates, None, need, arith)
    for d in range(4):
        a, b = out["f32"][d], out["bf16x3"][d]
        scale = float(a.abs().max())
        assert scale > 0
        assert float((a - b).abs().max()) <= 1e-5 * scale + 1e-12, (d, float((a - b).abs().max()), scale)


def test_backward_accumulates_into_existing_gradient_planes(hip):
    """torch.ops.nvsr.decode_rays_backward_: the fine pass scatters into the coarse pass's gradient planes.  Two passes
    accumulated in place == the sum of two functional calls, up to the order of float atomics; mis-shaped buffers are refused."""
    nv = torch.ops.nvsr
    N, S = 300, 48
    mf, rays, z = _backward_inputs(hip, N, S, 64, 3, "dense")
    z2 = (z + 0.7).contiguous()
    planes, consts = mf.scene_args()
    need = [True, True, True, True]
    g1 = torch.randn(N, S, 4, device=DEV) * 1e-2
    g2 = torch.randn(N, S, 4, device=DEV) * 1e-2
    packed, packed_bwd = mf.packed_decoder(), mf.packed_decoder_bwd()
    _, gates1, _ = nv.decode_rays(planes, consts, packed, rays, z, True, False, 3)
    _, gates2, _ = nv.decode_rays(planes, consts, packed, rays, z2, True, False, 3)
    a = nv.decode_rays_backward(planes, consts, packed, packed_bwd, rays, z, g1, gates1, None, need, 3)
    b = nv.decode_rays_backward(planes, consts, packed, packed_bwd, rays, z2, g2, gates2, None, need, 3)
    acc = nv.decode_rays_backward(planes, consts, packed, packed_bwd, rays, z, g1, gates1, None, need, 3)
    nv.decode_rays_backward_(planes, consts, packed, packed_bwd, rays, z2, g2, gates2, None, need, 3, acc)
    for d in range(4):
        ref = a[d] + b[d]
        assert float((acc[d] - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    with pytest.raises(ValueError):
        nv.decode_rays_backward_(planes, consts, packed, packed_bwd, rays, z2, g2, gates2, None, need, 3, [t[..., :1].contiguous() for t in acc])


# ---------------------------------------------------------------------------------------------------------------------------------
# the stated multi-GPU partitions of SURVEY.md 8e with the HIP kernels, rehearsed with two ranks on this box's one GPU (gloo)
# ---------------------------------------------------------------------------------------------------------------------------------
def _bench_rehearsal(args, timeout=900):
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NVSR_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--no-cpu-baseline", "--no-modes"] + args      # (bench.py spawns its ranks)
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0]), p.stderr


@pytest.mark.parametrize("what", ["planes", "planes+decoder"])
def test_training_partition_two_ranks_equal_one_rank(what):
    """SURVEY 8e "Training partition": 4096 rays -> 4096 / N per GPU, same scene and pixels on all ranks, all-reduced gradients.
    `bench.py --workload train --rays-global 512 --gpus 2` (two ranks on cuda:0, gloo): inside the run every rank takes a step on its half
    of a ray set, all-reduces (distributed.allreduce_gradients) and compares with the gradients of the one-rank step on the whole set
    (the same HIP kernels): planes within 1e-5 relative L2 (order of the float atomics), decoders within 1e-4
    (bench.train_partition_check prints TRAIN_GRADIENTS_MATCH per rank and exits non-zero otherwise).  Reference hook: train_nerf.py:839-860."""
    r, err = _bench_rehearsal(["--workload", "train", "--rays-global", "512", "--train-what", what, "--steps", "2", "--warmup", "1"])
    for rank in (0, 1):
        assert "TRAIN_GRADIENTS_MATCH rank %d of 2" % rank in err, err[-3000:]
    assert r["scaling"] == "strong" and r["n_gpus"] == 2 and r["config"]["rays_per_step_per_gpu"] == 256
    assert abs(r["value"] - 512 / (r["ms_per_step"] * 1e-3)) <= 1e-6 * r["value"]


def test_band_sharded_sr_stage_is_bit_identical_to_one_rank():
    """SURVEY 8e "SR stage": `bench.py --workload sr --partition bands --gpus 2` drives the REAL PlanesSR(EDSR 256 x 32) through
    distributed.super_resolve_planes_sharded -- each rank one horizontal band of every 200^2 plane with its 68-pixel LR halo, one
    all_gather per plane -- and asserts inside the run that the assembled 800^2 planes torch.equal the single-rank planes
    (SR_BANDS_IDENTICAL per rank).  Reference: models.py:884-926."""
    r, err = _bench_rehearsal(["--workload", "sr", "--partition", "bands", "--steps", "1", "--warmup", "0"])
    for rank in (0, 1):
        assert "SR_BANDS_IDENTICAL rank %d of 2" % rank in err, err[-3000:]
    assert r["scaling"] == "strong" and r["config"]["partition"] == "bands"


# ---------------------------------------------------------------------------------------------------------------------------------
# small parity holes of VERDICT r2 #5
# ---------------------------------------------------------------------------------------------------------------------------------
def _sr_model_g20(hip, g):
    Cc, hid, nblocks, sf, R, pad, over = [int(v) for v in g["cfg"]]
    sr = hip.models.PlanesSR(hip.models.EDSR, sf, Cc, Cc, {"model": {"hidden_size": hid, "n_blocks": nblocks}, "input_normalization": True}, "bilinear")
    sr.load_state_dict({k[3:]: torch.as_tensor(v) for k, v in g.items() if k.startswith("sd.")}, strict=True)
    sr = sr.to(DEV)
    assert sr.inner_model.required_padding == pad and sr.HR_overpadding == over
    return sr, (Cc, hid, nblocks, sf, R)


def test_planes_sr_input_normalization_golden(hip):
    """PlanesSR's input normalisation (models.py:855-857,899-901 -> sr_prepare_kernel): g09 never sets planes_mean/std_NON_LEARNED, g20 does.
    Full plane (eval), ROI (training mode), the batched full-plane pass and the differentiable training forward against the reference."""
    from conftest import load_golden
    g = load_golden("g20_sr_options.npz")
    sr, (Cc, hid, nblocks, sf, R) = _sr_model_g20(hip, g)
    sr.eval()
    sr.set_LR_plane(T(g["lr"]), id="p", save_interpolated=False)
    sr.set_LR_plane(T(g["lr"]), id="q", save_interpolated=False)
    full = sr("p")
    np.testing.assert_allclose(N_(full), g["sr_full"], rtol=0, atol=1e-5)
    sr.clear_SR_planes()
    sr.super_resolve_many(["p", "q"])                       # batched pass takes the same mean / std
    assert torch.equal(sr.SR_planes["p"], full) and torch.equal(sr.SR_planes["q"], full)
    sr.clear_SR_planes()
    sr.train()
    ref = g["sr_roi"]
    m = ~np.isnan(ref)
    roi = N_(sr(("p", T(g["roi"]))))
    assert np.array_equal(np.isnan(roi), np.isnan(ref)) and np.isnan(ref).any()
    np.testing.assert_allclose(roi[m], ref[m], rtol=0, atol=1e-5)
    for p_ in sr.inner_model.parameters():                  # differentiable forward: same values
        p_.requires_grad_(True)
    roi_t = sr(("p", T(g["roi"])))
    assert roi_t.requires_grad
    np.testing.assert_allclose(N_(roi_t.detach())[m], ref[m], rtol=0, atol=1e-5)


def test_planes_sr_training_noise_matches_seeded_reference(hip):
    """sr_input_noise / sr_output_noise (models.py:896-897,920-921): training-mode Gaussian noise on the network input (std = level x
    LR.std()) and on the super-resolved region (std = level x difference.std()), drawn with torch.normal from the CPU generator.  Seeded
    like the fixture generator, the build draws the same numbers in the same order: outputs equal the reference's within 2e-5 (noise
    amplitudes are ~0.1, so a different draw, a missing noise or noise on the wrong tensor is off by 1e-1)."""
    from conftest import load_golden
    g = load_golden("g20_sr_options.npz")
    sr, _ = _sr_model_g20(hip, g)
    sr.set_LR_plane(T(g["lr"]), id="p", save_interpolated=False)
    sr.train()
    clean = g["sr_roi"]
    m = ~np.isnan(clean)
    for tag, (ni, no), seed in zip(("in", "out", "both"), g["noise_levels"], g["noise_seeds"]):
        sr.input_noise, sr.output_noise = float(ni), float(no)
        torch.manual_seed(int(seed))
        got = N_(sr(("p", T(g["roi"]))))
        ref = g["sr_noise_" + tag]
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        assert np.abs(ref[m] - clean[m]).max() > 2e-2, tag         # the fixture's noise is visible
        np.testing.assert_allclose(got[m], ref[m], rtol=0, atol=2e-5, err_msg=tag)
    # differentiable path: gradients flow through the noisy forward (noise is an additive constant)
    sr.input_noise, sr.output_noise = 0.2, 0.15
    for p_ in sr.inner_model.parameters():
        p_.requires_grad_(True)
    torch.manual_seed(103)
    out = sr(("p", T(g["roi"])))
    np.testing.assert_allclose(N_(out.detach())[m], g["sr_noise_both"][m], rtol=0, atol=2e-5)
    out[~torch.isnan(out.detach())].sum().backward()
    assert all(p_.grad is not None and torch.isfinite(p_.grad).all() for p_ in sr.inner_model.parameters())
    sr.eval()
    sr.input_noise = sr.output_noise = 0


def test_run_network_mirror_matches_golden_raw_field(hip):
    """The stand-alone run_network mirror (train_utils.py:15-64: flatten the points, expand the batch's view directions to every sample,
    call the model, reshape to [N,S,4]) against the reference's raw radiance field on g08's fine model (fixture g21: 40 rays x 32 depths,
    chunked in the reference) -- the fused passes never call the mirror, so it had no numeric test (VERDICT r2 weak #1a).  Tolerance = the
    decoder's: 2e-5."""
    from conftest import load_golden
    g8, g = load_golden("g08_render.npz"), load_golden("g21_run_network.npz")
    mf, sid = build_model(hip, {k[5:]: v for k, v in g8.items() if k.startswith("fine.")}, [g8["plane%d" % d] for d in range(4)], g8["box"])
    ident = hip.train_utils.identity_encoding
    raw = hip.train_utils.run_network(mf, T(g["pts"]), T(g["ray_batch"]), int(g["chunksize"]), ident, ident, sid)
    assert tuple(raw.shape) == g["raw"].shape == (40, 32, 4)
    np.testing.assert_allclose(N_(raw), g["raw"], rtol=0, atol=2e-5)
    # and it agrees with what the fused coarse pass composites from: the same points through the decode operator
    flat = torch.cat([T(g["pts"]).reshape(-1, 3), T(g["ray_batch"])[:, None, -3:].expand(40, 32, 3).reshape(-1, 3)], -1)
    assert torch.equal(mf(flat).reshape(40, 32, 4), raw)


# ---------------------------------------------------------------------------------------------------------------------------------
# resampler: the deterministic-u fast path (csrc/aux.hip resample_fast_wave)
# ---------------------------------------------------------------------------------------------------------------------------------
def test_resampler_fast_path_is_bit_identical_to_the_separate_kernels(hip):
    """Inference frames (deterministic u, Nc <= 64, Nf <= 128) take resample_fast_wave: branch-free 6-step searches, sample ranks from the
    bin index + two verified compares, coarse-depth ranks from a histogram + wave scan.  Whatever the ray -- spiked pdfs that put every
    sample into one bin, flat pdfs (samples exactly on bin edges), runs of equal depths, zero-width rays (near == far), lindisp depths
    computed in the kernel -- and on the rays that must FALL BACK to the general path (NaN or negative weights, unsorted depths), the
    result is bit for bit sort(cat(z, sample_pdf(z_mid, w[1:-1]))) as the separate kernels produce it (train_utils.py:144-155)."""
    capi = hip.capi
    rng = np.random.default_rng(33)
    for (N, Nc, Nf) in ((2051, 64, 128), (300, 64, 64), (257, 17, 128), (129, 3, 5), (64, 64, 1)):
        z = np.sort(rng.uniform(2, 6, (N, Nc)).astype(np.float32), -1)
        z[1::7, Nc // 3:] = z[1::7, Nc // 3 - 1: Nc // 3]               # long runs of equal depths
        z[2::11] = 3.25                                                  # zero-width rays
        w = (rng.uniform(0, 1, (N, Nc)) ** 6).astype(np.float32)
        w[::3] = 0.0                                                     # flat pdf
        w[4::9] = 0.0
        w[4::9, min(Nc - 2, 1 + Nc // 2)] = 50.0                         # one spike: (almost) every sample in one bin
        w[5::13, Nc // 2] = np.nan                                       # fallback: NaN
        w[6::17, 1:] = -0.5                                              # fallback: negative pdf
        zz = z.copy()
        zz[7::19] = rng.permuted(z[7::19], axis=-1)                      # fallback: unsorted depths
        z_d, w_d = T(zz), T(w)
        zf = torch.empty((N, Nc + Nf), device=DEV)
        capi.call("nvsr_importance_resample", N, Nc, Nf, capi.ptr(z_d), capi.ptr(w_d), None, capi.ptr(zf), capi.stream())
        zm = (0.5 * (z_d[:, 1:] + z_d[:, :-1])).contiguous()
        smp = torch.empty((N, Nf), device=DEV)
        capi.call("nvsr_sample_pdf", N, Nc - 1, Nf, capi.ptr(zm), capi.ptr(w_d[:, 1:-1].contiguous()), None, capi.ptr(smp), capi.stream())
        cat = torch.cat([z_d, smp], -1).contiguous()
        ref = torch.empty_like(cat)
        capi.call("nvsr_sort_rows", N, Nc + Nf, capi.ptr(cat), capi.ptr(ref), capi.stream())
        # (rows poisoned by a NaN weight: NaNs sort last like torch.sort's, in the fused kernel and in nvsr_sort_rows alike)
        tref = torch.sort(cat.cpu(), -1).values.to(DEV)     # (the documented order -- NaNs last -- as the CPU implementation gives it)
        assert torch.equal(torch.isnan(zf), torch.isnan(tref)) and torch.equal(torch.isnan(ref), torch.isnan(tref))
        assert torch.equal(zf.nan_to_num(nan=-1.0), tref.nan_to_num(nan=-1.0)), (N, Nc, Nf)
        assert torch.equal(ref.nan_to_num(nan=-1.0), tref.nan_to_num(nan=-1.0))
    # depths computed in the kernel from the packed rays' near / far (nvsr_importance_resample_rays), linear and lindisp
    N, Nc, Nf = 1000, 64, 128
    rays = torch.zeros((N, 11), device=DEV)
    rays[:, 6] = T(rng.uniform(0.5, 2.5, N).astype(np.float32))
    rays[:, 7] = rays[:, 6] + T(rng.uniform(0.0, 5.0, N).astype(np.float32))
    w_d = T((rng.uniform(0, 1, (N, Nc)) ** 3).astype(np.float32))
    for lindisp in (0, 1):
        z_d = torch.empty((N, Nc), device=DEV)
        capi.call("nvsr_coarse_z", N, Nc, capi.ptr(rays), lindisp, None, capi.ptr(z_d), capi.stream())
        a = torch.empty((N, Nc + Nf), device=DEV)
        b = torch.empty_like(a)
        capi.call("nvsr_importance_resample_rays", N, Nc, Nf, capi.ptr(rays), lindisp, capi.ptr(w_d), None, capi.ptr(a), capi.stream())
        capi.call("nvsr_importance_resample", N, Nc, Nf, capi.ptr(z_d), capi.ptr(w_d), None, capi.ptr(b), capi.stream())
        zm = (0.5 * (z_d[:, 1:] + z_d[:, :-1])).contiguous()
        smp = torch.empty((N, Nf), device=DEV)
        capi.call("nvsr_sample_pdf", N, Nc - 1, Nf, capi.ptr(zm), capi.ptr(w_d[:, 1:-1].contiguous()), None, capi.ptr(smp), capi.stream())
        ref = torch.sort(torch.cat([z_d, smp], -1), -1).values
        assert torch.equal(a, ref) and torch.equal(b, ref), lindisp


# ---------------------------------------------------------------------------------------------------------------------------------
# the limb arithmetics as a primitive (csrc/limb_core.h; include/nvsr.h NVSR_ARITH_*): error against float64 on chosen operands
# ---------------------------------------------------------------------------------------------------------------------------------
def _gemm_probe(hip, mode, W, X):
    Y, Wd, Xd = torch.empty((32, 32), device=DEV), T(np.ascontiguousarray(W)), T(np.ascontiguousarray(X))
    hip.capi.call("nvsr_limb_gemm_probe", hip.capi.ARITHMETIC[mode], W.shape[1], hip.capi.ptr(Wd), hip.capi.ptr(Xd), hip.capi.ptr(Y),
                  hip.capi.stream())
    return N_(Y).astype(np.float64)


def test_limb_gemm_error_bounds(hip):
    """Y = W X through the products of every decoder arithmetic -- the operands split exactly as the render kernels and the weight packer
    split them -- against float64, relative to sum_k |W_k||X_k| per output.  K = 192 (the widest decoder layer, models.py:409-413).
    Operand sets: random; adversarial mantissas (all ones / low 16 bits set -- the worst case of truncation limbs) with all products of
    a row positive; magnitudes spread over 2^-12 .. 2^8 (activations) to exercise F16X2's static scales and its subnormal low limbs.
    Asserted per arithmetic (include/nvsr.h):
        f32     <= (K / 2) 2^-24                                   one rounding per accumulating MFMA
        bf16x3  <= 2^-21 + 2^-30 + (6 K / 16) 2^-24                 dropped limb products + accumulation
        f16x2   <= 2^-21 + (3 K / 16) 2^-24 + floor                 representation 2 x 2^-23 + dropped Wl xl 2^-22 + accumulation; floor =
                                                                     the absolute error of subnormal low limbs (2^-29 per activation,
                                                                     2^-33 per weight) relative to the row's sum |W||X|
    and: F16X2 is not worse than BF16X3 on the random and the adversarial sets (max and rms), which is what makes it the default."""
    rng = np.random.default_rng(11)
    K = 192

    def adversarial(shape, pattern, sign=None):
        e = rng.integers(-3, 3, size=shape)
        mant = {"ones": 0x7FFFFF, "low16": 0x00FFFF}[pattern]
        bits = ((127 + e).astype(np.uint32) << 23) | np.uint32(mant)
        v = bits.view(np.float32)
        s = np.where(rng.random(shape) < 0.5, -1.0, 1.0).astype(np.float32) if sign is None else sign
        return (v * s).astype(np.float32)

    sets = {}
    sets["random"] = (rng.standard_normal((32, K)).astype(np.float32) * 0.1, np.maximum(rng.standard_normal((K, 32)), 0).astype(np.float32))
    for pat in ("ones", "low16"):
        X = adversarial((K, 32), pat)
        W = adversarial((32, K), pat, sign=np.sign(X[:, 0])[None, :].repeat(32, 0)) / np.float32(64)        # column 0: every product positive
        sets["adversarial " + pat] = (W, X)
    mag = np.exp2(rng.uniform(-12, 8, size=(K, 32))).astype(np.float32)
    sets["wide range"] = (rng.standard_normal((32, K)).astype(np.float32) * np.exp2(rng.uniform(-9, 2, size=(32, K))).astype(np.float32),
                          (mag * np.where(rng.random((K, 32)) < 0.5, -1, 1)).astype(np.float32))
    bound = {"f32": lambda floor: (K // 2) * 2.0 ** -24, "bf16x3": lambda floor: 2.0 ** -21 + 2.0 ** -30 + (6 * K // 16) * 2.0 ** -24,
             "f16x2": lambda floor: 2.0 ** -21 + (3 * K // 16) * 2.0 ** -24 + floor}
    report = {}
    for name, (W, X) in sets.items():
        ref = W.astype(np.float64) @ X.astype(np.float64)
        magn = np.abs(W.astype(np.float64)) @ np.abs(X.astype(np.float64))
        floor = (2.0 ** -29 * np.abs(W.astype(np.float64)).sum(1)[:, None] + 2.0 ** -33 * np.abs(X.astype(np.float64)).sum(0)[None, :]) / magn
        for mode in ("f32", "bf16x3", "f16x2"):
            rel = (_gemm_probe(hip, mode, W, X) - ref) / magn
            report[(name, mode)] = (np.abs(rel).max(), np.sqrt((rel ** 2).mean()), rel.mean())
            assert np.all(np.abs(rel) <= bound[mode](floor if mode == "f16x2" else 0.0)), (name, mode, np.abs(rel).max())
    for (name, mode), (mx, rms, mean) in report.items():
        print("%-18s %-7s max %.3g  rms %.3g  mean %+.3g   (of sum |w||x|)" % (name, mode, mx, rms, mean))
    for name in ("random", "adversarial ones", "adversarial low16"):
        assert report[(name, "f16x2")][0] <= report[(name, "bf16x3")][0] * 1.05, name
        assert report[(name, "f16x2")][1] <= report[(name, "bf16x3")][1] * 1.05, name
    # an activation beyond the static range (>= 4094) must come out non-finite, never as a wrong number
    W, X = sets["random"][0].copy(), sets["random"][1].copy()
    X[5, 7] = 5000.0
    W[:, 5] = np.where(W[:, 5] == 0, 0.1, W[:, 5])
    bad = _gemm_probe(hip, "f16x2", W, X).astype(np.float32)
    assert not np.isfinite(bad[:, 7]).any()
    # ... as the matrix pipe's default NaN, 0xFFC00000 (negative): csrc/render3.hip's ReLU relies on it (source negation + integer max)
    assert (bad[:, 7].view(np.uint32) == 0xFFC00000).all(), [hex(v) for v in bad[:, 7].view(np.uint32)[:4]]
    assert np.isfinite(_gemm_probe(hip, "bf16x3", W, X)).all()


def test_f16_limb_range_overflow_is_loud(hip):
    """NVSR_ARITH_F16X2 carries static scales (include/nvsr.h): a feature / activation >= 4094 or a weight >= 255 does not fit its f16 limbs.
    Such a render must come out NaN -- never a finite wrong number (max(NaN, 0) = 0 inside the decoder would otherwise hide the overflow:
    csrc/render3.hip reads the conversion's sticky IEEE overflow flag per sample; the weight packer poisons the blob) -- while BF16X3 renders
    the same scene with finite pixels, and an in-range scene is untouched by the checks (finite, equal to a second launch)."""
    from bench import make_synthetic_scene
    capi = hip.capi
    mc, mf, sid, pose = make_synthetic_scene(DEV, plane_res=64, view_res=16, seed=3)
    H = W = 136                                               # 18 496 rays: the fused kernels
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, pose)
    rays = hip.train_utils.pack_rays(ro, rd, 2.0, 6.0)
    N, S = rays.shape[0], 24
    z = torch.sort(torch.rand(N, S, device=DEV) * 4 + 2, -1).values.contiguous()

    def render(mode):
        sc, keep = mf.native_scene()
        o = [torch.empty((N, 3), device=DEV), torch.empty(N, device=DEV), torch.empty(N, device=DEV)]
        capi.call("nvsr_render_pass_arith", C.byref(sc), capi.ptr(mf.packed_decoder()), N, S, capi.ptr(rays), capi.ptr(z), None, 1,
                  *[capi.ptr(b) for b in o], None, None, None, capi.ARITHMETIC[mode], capi.stream())
        torch.cuda.synchronize()
        return o[0]

    a, b = render("f16x2"), render("f16x2")
    assert torch.isfinite(a).all() and torch.equal(a, b)
    assert (a - render("bf16x3")).abs().max() < 2e-5
    # (1) features beyond the range: planes x 20000
    name = hip.models.get_plane_name(sid, 0)
    with torch.no_grad():
        saved = mf.planes_[name].detach().clone()
        mf.planes_[name].mul_(20000.0)
    try:
        assert torch.isfinite(render("bf16x3")).all()
        bad = render("f16x2")
        assert torch.isnan(bad).all(), "an out-of-range feature produced finite pixels"
    finally:
        with torch.no_grad():
            mf.planes_[name].copy_(saved)
    assert torch.equal(render("f16x2"), a)
    # (2) a weight beyond the range
    with torch.no_grad():
        w = mf.rgb_dec["0"][2].weight
        keep_w = float(w[5, 7])
        w[5, 7] = 300.0
    try:
        assert torch.isfinite(render("bf16x3")).all()
        assert torch.isnan(render("f16x2")).all(), "an out-of-range weight produced finite pixels"
    finally:
        with torch.no_grad():
            w[5, 7] = keep_w
    assert torch.equal(render("f16x2"), a)


# ---------------------------------------------------------------------------------------------------------------------------------
# disp_map / depth_map are differentiable (volume_rendering_utils.py:42-46; VERDICT r2 missing #5)
# ---------------------------------------------------------------------------------------------------------------------------------
def _vrrf_float64(raw, z, rd, white, mip):
    """the reference's formula in float64 torch (autograd gives the gradients to compare with)"""
    dists = z[..., 1:] - z[..., :-1]
    if not mip:
        dists = torch.cat((dists, torch.full_like(z[..., :1], 1e10)), -1)
    dists = dists * rd[..., None, :].norm(p=2, dim=-1)
    rgb = torch.sigmoid(raw[..., :3])
    alpha = 1.0 - torch.exp(-torch.relu(raw[..., 3]) * dists)
    t = torch.cumprod(1.0 - alpha + 1e-10, -1)
    t = torch.cat((torch.ones_like(t[..., :1]), t[..., :-1]), -1)
    w = alpha * t
    rgb_map = (w[..., None] * rgb).sum(-2)
    zz = 0.5 * (z[:, :-1] + z[:, 1:]) if mip else z
    depth = (w * zz).sum(-1)
    acc = w.sum(-1)
    disp = 1.0 / torch.max(1e-10 * torch.ones_like(depth), depth / acc)
    if white:
        rgb_map = rgb_map + (1.0 - acc[..., None])
    return rgb_map, disp, acc, w, depth


def test_disparity_and_depth_gradients(hip):
    """Gradients of a loss on (rgb_map, disp_map, acc_map, depth_map) with respect to the radiance field: the registered autograd of
    torch.ops.nvsr.composite (disp folded into depth / acc, nvsr_composite_backward_depth) against float64 autograd of the reference's
    formula, both branches (mip_nerf False / True), white background on / off; and through the fused training path
    (run_one_iter_of_nerf), whose disp outputs used to be non-differentiable."""
    rng = np.random.default_rng(9)
    N, S = 257, 37
    for mip in (False, True):
        for white in (False, True):
            raw = rng.standard_normal((N, S, 4)).astype(np.float32)
            raw[..., 3] = raw[..., 3] * 2.0 + 0.5
            z = np.sort(rng.uniform(2, 6, (N, S + (1 if mip else 0))).astype(np.float32), -1)
            rd = rng.standard_normal((N, 3)).astype(np.float32)
            g = [rng.standard_normal(s).astype(np.float32) for s in ((N, 3), (N,), (N,), (N,))]
            rt = T(raw).requires_grad_(True)
            out = hip.volume_rendering_utils.volume_render_radiance_field(rt, T(z), T(rd), white_background=white, mip_nerf=mip)
            assert out[1].requires_grad and out[4].requires_grad and not out[3].requires_grad
            (out[0] * T(g[0])).sum().add((out[1] * T(g[1])).sum()).add((out[2] * T(g[2])).sum()).add((out[4] * T(g[3])).sum()).backward()
            rdd = torch.tensor(raw, dtype=torch.float64, requires_grad=True)
            ref = _vrrf_float64(rdd, torch.tensor(z, dtype=torch.float64), torch.tensor(rd, dtype=torch.float64), white, mip)
            gd = [torch.tensor(a, dtype=torch.float64) for a in g]
            ((ref[0] * gd[0]).sum() + (ref[1] * gd[1]).sum() + (ref[2] * gd[2]).sum() + (ref[4] * gd[3]).sum()).backward()
            np.testing.assert_allclose(N_(out[1].detach()), ref[1].detach().numpy(), rtol=2e-5, atol=1e-6)
            got, want = N_(rt.grad).astype(np.float64), rdd.grad.numpy()
            # rays behind an opaque sample divide by (1 - alpha + 1e-10): compare in the norm, and element-wise where the reference is well conditioned
            assert np.linalg.norm(got - want) <= 2e-4 * np.linalg.norm(want), (mip, white, np.linalg.norm(got - want) / np.linalg.norm(want))
            # disp alone: a gradient that only the new path produces
            rt2 = T(raw).requires_grad_(True)
            o2 = hip.volume_rendering_utils.volume_render_radiance_field(rt2, T(z), T(rd), white_background=white, mip_nerf=mip)
            (o2[1] * T(g[1])).sum().backward()
            rd2 = torch.tensor(raw, dtype=torch.float64, requires_grad=True)
            r2 = _vrrf_float64(rd2, torch.tensor(z, dtype=torch.float64), torch.tensor(rd, dtype=torch.float64), white, mip)
            (r2[1] * torch.tensor(g[1], dtype=torch.float64)).sum().backward()
            g2, w2 = N_(rt2.grad).astype(np.float64), rd2.grad.numpy()
            assert np.abs(w2).max() > 0 and np.linalg.norm(g2 - w2) <= 2e-4 * np.linalg.norm(w2), (mip, white)
    # the fused training path: d (sum disp_fine) / d planes is finite and non-zero, and equals the chain through the standalone operators
    from bench import make_synthetic_scene, render_options
    mc, mf, sid, pose = make_synthetic_scene(DEV, plane_res=48, view_res=16, seed=5)
    H = W = 24
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    ro, rd_ = hip.nerf_helpers.get_ray_bundle(H, W, focal, pose)
    batch = torch.stack([ro.reshape(-1, 3)[:300], rd_.reshape(-1, 3)[:300]], 0)
    opts, scfg = render_options(16, 24)
    for m in (mc, mf):
        m.train()
    out = hip.train_utils.run_one_iter_of_nerf(H, W, focal, mc, mf, batch, opts, sid, mode="train", scene_config=scfg, randoms={})
    assert out[4].requires_grad, "disp_fine has no gradient path"
    (out[4] * 0.01).sum().backward()
    gp = [p.grad for p in mf.planes_.values() if p.grad is not None]
    assert gp and all(torch.isfinite(g_).all() for g_ in gp) and sum(float(g_.abs().sum()) for g_ in gp) > 0


def test_evaluation_falls_back_to_bf16x3_outside_the_f16_range(hip):
    """eval_nerf / run_one_iter_of_nerf(mode='validation') of a model whose planes or weights do not fit NVSR_ARITH_F16X2's static scales
    renders in the 3-bf16-limb arithmetic (models.render_arithmetic; one cached reduction per tensor version) instead of NaN pixels; an explicit
    model.arithmetic = 'f16x2' ... still renders through the check (the choice is about range, not preference), in range nothing changes."""
    import warnings
    from bench import make_synthetic_scene, render_options
    mc, mf, sid, pose = make_synthetic_scene(DEV, plane_res=64, view_res=16, seed=4)
    H = W = 136
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, pose)
    opts, scfg = render_options(16, 24)
    code = hip.capi.ARITHMETIC
    planes, _ = mf.scene_args()
    assert mf.render_arithmetic(planes, training=False) == code["f16x2"]
    base = hip.train_utils.eval_nerf(H, W, focal, mc, mf, ro, rd, opts, scene_id=sid, scene_config=scfg)
    assert torch.isfinite(base[3]).all()
    name = hip.models.get_plane_name(sid, 1)
    with torch.no_grad():
        mf.planes_[name].mul_(20000.0)          # (mc and mf share the planes)
    planes, _ = mf.scene_args()
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter("always")
        assert mf.render_arithmetic(planes, training=False) == code["bf16x3"]
        assert mf.render_arithmetic(planes, training=True) == code["f16x2"]       # training: no host read per step; NaN is the signal there
        out = hip.train_utils.eval_nerf(H, W, focal, mc, mf, ro, rd, opts, scene_id=sid, scene_config=scfg)
    assert any("bf16x3" in str(w.message) for w in wl)
    assert torch.isfinite(out[0]).all() and torch.isfinite(out[3]).all()


def test_training_steps_do_not_retain_their_graphs(hip):
    """A train step with decoder gradients keeps a multi-GB forward record alive in its autograd context; nothing may hold that context past
    the step.  (Round 3 regression: the differentiable-disparity change stored OUTPUT tensors in the context -- a reference cycle ctx -> saved
    -> output -> grad_fn -> ctx that only the cyclic garbage collector breaks: 29 steps of the bench filled 288 GB and the step went from 5 to
    200 ms.)  With the collector disabled the allocated memory after step 6 must equal that after step 2."""
    import gc
    from bench import make_synthetic_scene, render_options
    mc, mf, sid, pose = make_synthetic_scene(DEV, plane_res=48, view_res=16, seed=6)
    for m in (mc, mf):
        for n, p in m.named_parameters():
            p.requires_grad_("rot_mats" not in n)
        m.train()
    H = W = 32
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    ro, rd_ = hip.nerf_helpers.get_ray_bundle(H, W, focal, pose)
    batch = torch.stack([ro.reshape(-1, 3)[:512], rd_.reshape(-1, 3)[:512]], 0)
    opts, scfg = render_options(16, 16)
    gc.collect()
    gc.disable()
    try:
        mem = []
        for it in range(6):
            for m in (mc, mf):
                m.zero_grad(set_to_none=True)
            out = hip.train_utils.run_one_iter_of_nerf(H, W, focal, mc, mf, batch, opts, sid, mode="train", scene_config=scfg, randoms={})
            assert out[3].grad_fn.saved["rec_f"] is not None            # the step does carry a record
            (out[0].sum() + out[3].sum()).backward()
            del out
            torch.cuda.synchronize()
            mem.append(torch.cuda.memory_allocated())
    finally:
        gc.enable()
    assert mem[5] == mem[1], mem


def test_f16_conv_range_overflow_is_loud(hip):
    """The f16-limb SR convolution (csrc/sr.hip conv3x3_limb16_kernel<., ., 2>) carries the same static scales as the render pass: an input
    value >= 4094 or a weight >= 255 must give NaN outputs where it contributes (its ReLU epilogue is a compare + select, not v_max_f32) and
    leave every other output untouched; the 3-bf16-limb kernel computes the same layer with finite outputs."""
    rng = np.random.default_rng(2)
    capi = hip.capi
    Cin, Cout, H, W = 64, 128, 12, 40
    x = rng.standard_normal((Cin, H, W), dtype=np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3), dtype=np.float32) / np.sqrt(9 * Cin)).astype(np.float32)

    def conv(xa, wa, mode, epi):
        pk = torch.empty(capi.lib().nvsr_conv3x3_packed_floats(Cin, Cout), device=DEV)
        wd, xd = T(wa), T(xa)
        capi.call("nvsr_pack_conv3x3", capi.ptr(wd), Cin, Cout, capi.ptr(pk), capi.stream())
        out = torch.full((Cout, H - 2, W - 2), -7.0, device=DEV)
        capi.call("nvsr_conv3x3_arith", capi.ptr(xd), Cin, H, W, capi.ptr(pk), Cout, epi, None, capi.ptr(out), capi.ARITHMETIC[mode], 0, capi.stream())
        torch.cuda.synchronize()
        return N_(out)

    for epi in (0, 1):                                   # no epilogue / ReLU
        base = conv(x, w, "f16x2", epi)
        assert np.isfinite(base).all()
        xb = x.copy()
        xb[5, 6, 20] = 6000.0                            # reaches outputs (.., 4..6, 18..20)
        bad = conv(xb, w, "f16x2", epi)
        hit = np.zeros_like(bad, bool)
        hit[:, 4:7, 18:21] = True
        assert np.isnan(bad[hit]).all() and np.array_equal(bad[~hit], base[~hit]), epi
        assert np.isfinite(conv(xb, w, "bf16x3", epi)).all()
        wb = w.copy()
        wb[17, 3, 1, 1] = 300.0                          # output channel 17
        bad = conv(x, wb, "f16x2", epi)
        assert np.isnan(bad[17]).all() and np.array_equal(np.delete(bad, 17, 0), np.delete(base, 17, 0)), epi
        assert np.isfinite(conv(x, wb, "bf16x3", epi)).all()


def test_f16_backward_matches_the_f32_backward(hip):
    """The gate-driven backward without a weight-gradient record on 2 f16 limbs (csrc/render_bwd_limb.hip, LF = 2: unscaled transposed weights,
    every wave tile's gradients scaled by its own power of two) against the exact-f32 backward ON THE SAME GATES, with dL/draw magnitudes spread
    over 8 decades from ray to ray: relative L2 error of the plane gradients <= 2e-5 and not worse than the 3-bf16-limb backward's (both sit at
    the float-atomics' ordering noise, ~1e-6)."""
    from bench import make_synthetic_scene
    capi = hip.capi
    lib = capi.lib()
    rng = np.random.default_rng(3)
    for pr, N, S in ((40, 1500, 65), (200, 3000, 32), (17, 700, 128)):
        mc, mf, sid, pose = make_synthetic_scene(DEV, plane_res=pr, view_res=8, seed=int(rng.integers(1 << 20)), channels_last=True)
        H = W = 80
        focal = 0.5 * W / np.tan(0.5 * 0.6911112)
        ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, pose)
        rays = hip.train_utils.pack_rays(ro, rd, 2.0, 6.0)[torch.from_numpy(rng.integers(0, H * W, N)).to(DEV)].contiguous()
        z = torch.sort(torch.rand(N, S, device=DEV) * 4 + 2, -1).values.contiguous()
        mag = T(np.exp(rng.uniform(np.log(1e-6), np.log(1e2), (N, 1, 1))).astype(np.float32))
        g_raw = (torch.randn(N, S, 4, device=DEV) * mag).contiguous()
        sc, keep = mf.native_scene()
        raw = torch.empty((N, S, 4), device=DEV)
        gates = torch.zeros(N * S * 32, dtype=torch.int32, device=DEV)
        capi.call("nvsr_decode_rays_arith", C.byref(sc), capi.ptr(mf.packed_decoder()), N, S, capi.ptr(rays), capi.ptr(z), capi.ptr(raw),
                  capi.ptr(gates), None, capi.ARITHMETIC["f32"], capi.stream())
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
        rel = {m: float((res[m] - res["f32"]).norm() / res["f32"].norm()) for m in ("bf16x3", "f16x2")}
        print("planes %d N %d S %d: relative L2 vs the f32 backward %s" % (pr, N, S, rel))
        assert rel["f16x2"] <= 2e-5 and rel["f16x2"] <= 3.0 * rel["bf16x3"] + 1e-6, rel


def test_f16_conv_data_gradient_scales_with_the_gradient(hip, oracle):
    """The data gradient of a wide SR convolution on 2 f16 limbs (csrc/sr.hip: the dy tensor scaled by the power of two that puts its largest
    magnitude into [2^12, 2^13), absmax_kernel) against the oracle, for gradients of magnitude 1, 1e-7 and 1e+5: the error relative to the
    largest |dx| stays at the 3-bf16-limb kernel's level whatever the magnitude (a static scale would flush a 1e-7 gradient to zero)."""
    rng = np.random.default_rng(8)
    capi = hip.capi
    Cin, Cout, H, W = 256, 256, 13, 40
    w = (rng.standard_normal((Cout, Cin, 3, 3), dtype=np.float32) / np.sqrt(9 * Cin)).astype(np.float32)
    pk = torch.empty(capi.lib().nvsr_conv3x3_packed_floats(Cout, Cin), device=DEV)
    wd = T(w)
    capi.call("nvsr_pack_conv3x3_dgrad", capi.ptr(wd), Cin, Cout, capi.ptr(pk), capi.stream())
    wt = np.ascontiguousarray(w.transpose(1, 0, 2, 3)[:, :, ::-1, ::-1])
    dy0 = rng.standard_normal((Cout, H - 2, W - 2), dtype=np.float32)
    for mag in (1.0, 1e-7, 1e5):
        dy = (dy0 * np.float32(mag)).astype(np.float32)
        ref = oracle.conv3x3(np.pad(dy, ((0, 0), (2, 2), (2, 2))), wt)
        dyd = T(dy)
        err = {}
        for mode in ("bf16x3", "f16x2"):
            for rows in ((0,) if mode == "bf16x3" else (0, 18, 19, 20, 22)):
                dx = torch.full((Cin, H, W), -7.0, device=DEV)
                capi.call("nvsr_conv3x3_dgrad_arith", capi.ptr(dyd), Cin, H, W, capi.ptr(pk), Cout, capi.ptr(dx), capi.ARITHMETIC[mode], rows, capi.stream())
                e = float(np.abs(N_(dx).astype(np.float64) - ref).max() / np.abs(ref).max())
                err[(mode, rows)] = e
                assert e <= 1e-5, (mag, mode, rows, e)
        assert max(v for k, v in err.items() if k[0] == "f16x2") <= 1.5 * err[("bf16x3", 0)] + 2e-7, (mag, err)
        # ... and the weight gradient dW = sum_pixels dy (x) X (csrc/sr_bwd.hip conv3x3_wgrad_limb_kernel<2>: X with the static scale, dy with its tensor's)
        if mag == 1.0:
            xin = rng.standard_normal((Cin, H, W), dtype=np.float32)
            xd = T(xin)
        _, dw_ref = oracle.conv3x3_backward(xin, w, dy)
        ws = torch.empty(capi.lib().nvsr_conv3x3_wgrad_workspace_floats(Cin, H, W, Cout), device=DEV)
        rel = {}
        for mode in ("bf16x3", "f16x2"):
            dw = torch.zeros((Cout, Cin, 3, 3), device=DEV)
            capi.call("nvsr_conv3x3_wgrad_arith", capi.ptr(dyd), capi.ptr(xd), Cin, H, W, Cout, 1.0, capi.ptr(dw), capi.ptr(ws), capi.ARITHMETIC[mode], capi.stream())
            rel[mode] = float(np.linalg.norm(N_(dw).astype(np.float64) - dw_ref) / np.linalg.norm(dw_ref))
        assert rel["f16x2"] < 2e-6 and rel["f16x2"] <= 1.5 * rel["bf16x3"] + 2e-7, (mag, rel)


def test_device_pixel_sampler_is_the_specified_permutation(hip):
    """nvsr_sample_pixels (include/nvsr.h) against its numpy restatement: integer work, bit-exact -- (row, col) pairs of entries
    [first, first + n) of the keyed permutation, for the training size (4096 of 800 x 800), a window in the middle of it, a whole small
    non-square image (every pixel exactly once) and a one-pixel image; the gathered targets are image[row, col, :]."""
    from oracle.oracle import sample_pixels
    capi = hip.capi
    g = torch.Generator(device=DEV).manual_seed(3)
    for H, W, key, first, n in ((800, 800, 0x1234567890ABCDEF, 0, 4096), (800, 800, 7, 300000, 4096), (5, 7, 99, 0, 35), (1, 1, 5, 0, 1),
                                (37, 3, 2 ** 64 - 1, 11, 50)):
        img = torch.rand(H, W, 4, device=DEV, generator=g)
        rc = torch.full((n, 2), -1, dtype=torch.int32, device=DEV)
        tgt = torch.empty((n, 4), device=DEV)
        capi.call("nvsr_sample_pixels", H * W, H, W, key, first, n, capi.ptr(img), 4, capi.ptr(rc), capi.ptr(tgt), capi.stream())
        want = sample_pixels(H * W, H, key, first, n)
        got = rc.cpu().numpy()
        assert np.array_equal(got, want), (H, W, key)
        assert torch.equal(tgt, img[rc[:, 0].long(), rc[:, 1].long()])
        assert len({(int(r), int(c)) for r, c in got}) == n and got[:, 0].max() < H and got[:, 1].max() < W and got.min() >= 0
    # argument checks of the boundary
    lib = capi.lib()
    assert lib.nvsr_sample_pixels(10, 5, 2, 0, 8, 4, None, 0, rc.data_ptr(), None, None) != 0        # first + n > total
    assert lib.nvsr_sample_pixels(11, 5, 2, 0, 0, 4, None, 0, rc.data_ptr(), None, None) != 0        # total != H * W
    assert lib.nvsr_sample_pixels(10, 5, 2, 0, 0, 4, None, 3, rc.data_ptr(), tgt.data_ptr(), None) != 0   # targets without an image


def test_device_pixel_sampler_in_a_train_step(hip):
    """training.DevicePixelSampler as TrainStep's pixel_sampler: a new draw per call, the same draws for the same seed, rank shares of one
    global draw are disjoint and tile it; image-consistency iterations expand every drawn LR pixel to its ds x ds patch like
    select_training_pixels; uniformity over many calls (chi-square over 50 bands of 16 columns, 49 degrees of freedom: < 100 is p > 2e-5)."""
    tr = hip.training
    img = torch.rand(96, 64, 3, device=DEV)
    a, b = tr.DevicePixelSampler(seed=5), tr.DevicePixelSampler(seed=5)
    s1, t1 = a(img, 512)
    s2, _ = a(img, 512)
    r1, _ = b(img, 512)
    assert s1.dtype == torch.int32 and s1.shape == (512, 2) and torch.equal(s1, r1) and not torch.equal(s1, s2)
    assert torch.equal(t1, img[s1[:, 0].long(), s1[:, 1].long()])
    assert len(set(map(tuple, s1.tolist()))) == 512
    whole, _ = tr.DevicePixelSampler(seed=9, n_draw=512, lo=0)(img, 512)
    parts = [tr.DevicePixelSampler(seed=9, n_draw=512, lo=lo)(img, 128)[0] for lo in (0, 128, 256, 384)]
    assert torch.equal(torch.cat(parts, 0), whole)
    sel, tgt = tr.DevicePixelSampler(seed=1)(img[:48, :32], 64, consistency_ds=2)
    assert sel.shape == (64, 2) and tgt.shape == (16, 3)
    blocks = sel.reshape(16, 2, 2, 2)
    assert bool((blocks[:, :, :, 0] == blocks[:, :1, :1, 0] + torch.arange(2, device=DEV).reshape(1, 2, 1)).all())
    assert bool((blocks[:, :, :, 1] == blocks[:, :1, :1, 1] + torch.arange(2, device=DEV).reshape(1, 1, 2)).all())
    assert torch.equal(tgt, img[:48, :32][(blocks[:, 0, 0, 0] // 2).long(), (blocks[:, 0, 0, 1] // 2).long()])
    big = torch.zeros(800, 800, 3, device=DEV)
    smp = tr.DevicePixelSampler(seed=2)
    cnt = torch.zeros(50, device=DEV)
    for _ in range(100):
        s, _ = smp(big, 4096)
        cnt += torch.bincount(s[:, 1].long() // 16, minlength=50)
    e = float(cnt.sum()) / 50
    assert float(((cnt - e) ** 2 / e).sum()) < 100.0
    with pytest.raises(RuntimeError):
        smp(big.cpu(), 16)


def test_mse_loss_pair_matches_two_mse_losses(hip):
    """training.mse_loss_pair = (F.mse_loss(a, t), F.mse_loss(b, t)) in one launch, values and gradients (float32 sums in another order:
    relative 1e-6); a loss that is not used gets no gradient; shapes the kernel does not take fall back to torch."""
    tr = hip.training
    g = torch.Generator(device=DEV).manual_seed(8)
    for n in (1, 7, 4096, 70001):
        a = torch.rand(n, 3, device=DEV, generator=g, requires_grad=True)
        b = torch.rand(n, 3, device=DEV, generator=g, requires_grad=True)
        t = torch.rand(n, 3, device=DEV, generator=g)
        la, lb = tr.mse_loss_pair(a, b, t)
        ra, rb = torch.nn.functional.mse_loss(a.detach().double(), t.double()), torch.nn.functional.mse_loss(b.detach().double(), t.double())
        assert abs(float(la.detach()) - float(ra)) <= 2e-6 * float(ra) and abs(float(lb.detach()) - float(rb)) <= 2e-6 * float(rb)
        (0.25 * la + 3.0 * lb).backward()
        assert torch.allclose(a.grad, 0.25 * 2 * (a.detach() - t) / (3 * n), rtol=1e-6, atol=1e-12)
        assert torch.allclose(b.grad, 3.0 * 2 * (b.detach() - t) / (3 * n), rtol=1e-6, atol=1e-12)
    a = torch.rand(64, 3, device=DEV, requires_grad=True)
    b = torch.rand(64, 3, device=DEV, requires_grad=True)
    la, lb = tr.mse_loss_pair(a, b, torch.rand(64, 3, device=DEV))
    la.backward()
    assert b.grad is None and a.grad is not None
    la, lb = tr.mse_loss_pair(a.cpu(), b.cpu(), torch.rand(64, 3))                 # not a CUDA batch: torch's own mse_loss
    assert la.device.type == "cpu" and la.requires_grad
