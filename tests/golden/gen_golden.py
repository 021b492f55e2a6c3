#!/usr/bin/env python3
"""Golden-vector generator (BUILD CONTAINER ONLY).

Imports the upstream reference from /root/reference (read-only, pure PyTorch) behind the
shims documented in SURVEY.md Appendix A, runs its hot-path functions on small seeded
inputs on the CPU, and writes inputs + expected outputs as .npz fixtures next to this
script.  The fixtures are data only; nothing of the reference travels with the repo.

Re-run:  python tests/golden/gen_golden.py          (needs /root/reference)

Fixture index:
  g01_raybundle.npz   get_ray_bundle                    nerf_helpers.py:507-549
  g02_ndc.npz         ndc_rays                          nerf_helpers.py:578-605
  g03_coarse_z.npz    stratified coarse depths          train_utils.py:95-111
  g04_decoder.npz     TwoDimPlanesModel.forward         models.py:381-421 (+ intermediates)
  g05_composite.npz   volume_render_radiance_field      volume_rendering_utils.py:6-51
  g06_sample_pdf.npz  sample_pdf_2                      nerf_helpers.py:668-702
  g07_sort.npz        sort(cat(z, z_samples))           train_utils.py:155
  g08_render.npz      eval_nerf / run_one_iter_of_nerf  train_utils.py:185-331 (16x16 rays)
  g09_edsr.npz        EDSR + PlanesSR (mini net)        models.py:769-926
  g10_posenc.npz      positional_encoding, FlexibleNeRFModel   nerf_helpers.py:552-575, models.py:14-108
  g11_grads.npz       autograd of one train step wrt the planes (train_nerf.py:860-903)
  g12_ndc_render.npz  eval_nerf of a forward-facing (LLFF-style) view through NDC rays (train_utils.py:215-218)
  g14_sr_grads.npz    autograd through EDSR / PlanesSR (full plane and ROI): weights, network input and LR plane ('SR' in what)
  g13_decoder_grads.npz autograd of one train step wrt the decoder parameters of both models (what: ['decoder'], train_nerf.py:75-77)
  g16_composite_mip.npz volume_render_radiance_field(mip_nerf=True) + its autograd wrt the radiance field (volume_rendering_utils.py:19-26,41-42)
  g18_decoder_variants.npz  TwoDimPlanesModel.forward for other decoder geometries (widths, channel counts, combinations, skip layers)
  g17_store(.npz + g17_store/)  plane file, decoder checkpoint and SR checkpoint WRITTEN by the reference (PlanesOptimizer.save_params,
                      safe_saving; models.py:640-670, nerf_helpers.py:19-48, train_nerf.py:996-1008) + what it renders from them
  g20_sr_options.npz  PlanesSR with input_normalization (models.py:855-857,899-901) and the training noises sr_input_noise / sr_output_noise
                      (models.py:896-897,920-921; drawn from the seeded CPU generator)
  g22_model_options.npz TwoDimPlanesModel options no shipped YAML sets: align_corners=False, five position planes with random frames
                      (models.py:471-490), point_coords_noise in training mode (models.py:291-293; forward, gradients, one training iteration)
  g21_run_network.npz run_network (train_utils.py:15-64) on g08's fine model: points + view directions -> raw [N,S,4], chunked
  g15_loaders.npz     load_blender_data / load_llff_data on two tiny synthetic scenes (load_blender.py:232-332, load_llff.py:70-360).
                      imageio and cv2 are absent here: the harness reads the PNGs with PIL and gives cv2.resize(INTER_AREA) its
                      definition for integer factors (block mean), so the resampling itself is pinned by definition only; the JSON /
                      poses_bounds parsing, splits, intrinsics, pose algebra, render paths and hold-out choice are the reference's.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def import_reference():
    for m in ["cv2", "torchvision", "imageio", "magic", "deepdiff"]:
        sys.modules[m] = types.ModuleType(m)
    sys.modules["magic"].from_file = lambda *a, **k: None
    sys.modules["deepdiff"].DeepDiff = dict
    import scipy.signal
    import scipy.signal.windows

    scipy.signal.gaussian = scipy.signal.windows.gaussian
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    # the reference targets torch 1.12, whose torch.load unpickles arbitrary objects (its plane files hold an nn.ParameterDict);
    # torch >= 2.6 defaults to weights_only=True
    _load = torch.load
    torch.load = lambda *a, **k: _load(*a, **{"weights_only": False, **k})
    sys.path.insert(0, REF)
    import nerf_helpers, volume_rendering_utils, models, train_utils  # noqa
    from cfgnode import CfgNode

    return nerf_helpers, volume_rendering_utils, models, train_utils, CfgNode


nh, vru, models, tu, CfgNode = import_reference()

BOX = [[-4.0, -4.0, -4.0, -np.pi, -np.pi / 2], [4.0, 4.0, 4.0, np.pi, np.pi / 2]]
POSE = np.array(
    [
        [-0.9999, 0.0042, -0.0133, -0.0538],
        [-0.0140, -0.2997, 0.9539, 3.8455],
        [0.0, 0.9540, 0.2997, 1.2081],
        [0.0, 0.0, 0.0, 1.0],
    ],
    dtype=np.float32,
)


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print("wrote %-22s %8.1f KB  (%d arrays)" % (name, os.path.getsize(path) / 1024, len(arrs)))


def build_models(R, Rv, plane_std, seed, ds=8):
    """Coarse + fine TwoDimPlanesModel wired the way PlanesOptimizer would (models.py:545-550,601-604)."""
    torch.manual_seed(seed)
    np.random.seed(seed)
    sid = models.get_scene_id("lego", ds, (R, Rv))
    sc = models.SceneCoupler([sid], planes_res="LR", num_pos_planes=3, training_scenes=[sid])
    kw = dict(
        use_viewdirs=True,
        skip_connect_every=3,
        proj_combination="avg",
        viewdir_proj_combination="concat_pos",
        align_corners=True,
        scene_coupler=sc,
    )
    mc = models.TwoDimPlanesModel(**kw)
    mc.optional_no_grad = nh.null_with
    mf = models.TwoDimPlanesModel(num_planes_or_rot_mats=mc.rot_mats(), **kw)
    mf.optional_no_grad = nh.null_with
    planes = nn.ParameterDict(
        [
            (models.get_plane_name(sid, d), models.create_plane(R if d < 3 else Rv, 48, plane_std))
            for d in range(4)
        ]
    )
    box = torch.tensor(BOX, dtype=torch.float64)
    for m in (mc, mf):
        m.planes_ = planes
        m.plane_rank = None
        m.generated_planes = {}
        m.downsampled_planes = {}
        m.coverages = {}
        m.box_coords = {sid: box}
        m.set_cur_scene_id(sid)
    # Calibrate fc_alpha so that sigma straddles zero and acc_map is spread ("degenerate scenes").
    with torch.no_grad():
        g = torch.Generator().manual_seed(seed + 1)
        pts = torch.rand(4096, 3, generator=g) * 6 - 3
        d = torch.randn(4096, 3, generator=g)
        d = d / d.norm(dim=-1, keepdim=True)
        x = torch.cat([pts, d], -1)
        for m in (mc, mf):
            m.eval()
            raw = m(x)[:, 3]
            scale = 1.0 / float(raw.std())
            m.fc_alpha["0"].weight.mul_(scale)
            m.fc_alpha["0"].bias.mul_(scale)
            raw = m(x)[:, 3]
            m.fc_alpha["0"].bias.add_(-float(raw.mean()) - 0.5)
    return sid, mc, mf, planes, box


def state_arrays(prefix, model):
    return {prefix + k: npy(v) for k, v in model.state_dict().items() if "planes_" not in k}


def make_cfg(v_train, v_val, ndc=False, near=2.0, far=6.0):
    return CfgNode(
        {
            "nerf": {"use_viewdirs": True, "train": v_train, "validation": v_val},
            "dataset": {"synt": {"near": near, "far": far, "no_ndc": not ndc}},
        }
    )


def mode_cfg(nc, nf, perturb=False, noise=0.0, white=False, lindisp=False, chunk=131072):
    return dict(
        chunksize=chunk,
        perturb=perturb,
        num_coarse=nc,
        num_fine=nf,
        white_background=white,
        radiance_field_noise_std=noise,
        lindisp=lindisp,
    )


# --------------------------------------------------------------------------------------
def g01_raybundle():
    torch.manual_seed(1)
    out = {}
    cases = []
    for i, (H, W, ds, pad) in enumerate([(5, 7, 1, 0), (5, 7, 2, 0), (6, 4, 8, 0), (4, 6, 2, 2)]):
        q, _ = torch.linalg.qr(torch.randn(3, 3))
        c2w = torch.eye(4)
        c2w[:3, :3] = q
        c2w[:3, 3] = torch.randn(3) * 2
        focal = 0.5 * W / np.tan(0.5 * 0.6911112)
        off = (ds - 1) / (2 * ds)
        ro, rd = nh.get_ray_bundle(H, W, focal, c2w, padding_size=pad, downsampling_offset=off)
        out.update(
            {
                "c%d_c2w" % i: npy(c2w),
                "c%d_params" % i: np.array([H, W, focal, pad, off], dtype=np.float64),
                "c%d_ro" % i: npy(ro.contiguous()),
                "c%d_rd" % i: npy(rd),
            }
        )
        cases.append(i)
    # Blender-style pose used by the render fixtures
    H = W = 8
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    ro, rd = nh.get_ray_bundle(H, W, focal, torch.from_numpy(POSE))
    i = len(cases)
    out.update(
        {
            "c%d_c2w" % i: POSE,
            "c%d_params" % i: np.array([H, W, focal, 0, 0.0], dtype=np.float64),
            "c%d_ro" % i: npy(ro.contiguous()),
            "c%d_rd" % i: npy(rd),
        }
    )
    out["n_cases"] = np.array(i + 1)
    save("g01_raybundle.npz", **out)


def g02_ndc():
    torch.manual_seed(2)
    H, W, focal = 378, 504, 407.5
    ro = torch.randn(37, 3) * 0.3
    rd = torch.randn(37, 3)
    rd[:, 2] = -rd[:, 2].abs() - 0.2
    o, d = nh.ndc_rays(H, W, focal, 1.0, ro, rd)
    save(
        "g02_ndc.npz",
        params=np.array([H, W, focal, 1.0], dtype=np.float64),
        ro=npy(ro),
        rd=npy(rd),
        ro_ndc=npy(o),
        rd_ndc=npy(d),
    )


def g03_coarse_z():
    """Drives predict_and_render_radiance's own depth code (train_utils.py:95-109): run_network is
    swapped for a recorder that captures the z_vals argument and returns a zero radiance field."""
    out = {}
    N = 9
    near = torch.full((N, 1), 2.0)
    far = torch.full((N, 1), 6.0)
    far[3:] = 5.5
    near[5:] = 0.5
    torch.manual_seed(3)
    ro = torch.randn(N, 3)
    rd = torch.randn(N, 3)
    rays = torch.cat([ro, rd, near, far, rd / rd.norm(dim=-1, keepdim=True)], -1)
    captured = {}

    def recorder(network_fn, pts, ray_batch, chunksize, embed_fn, embeddirs_fn, scene_id, **kw):
        captured["z"] = kw["z_vals"].clone()
        captured["pts"] = pts.clone()
        return torch.zeros(list(pts.shape[:-1]) + [4])

    class Stub:
        optional_no_grad = nh.null_with

    real = tu.run_network
    tu.run_network = recorder
    try:
        for ci, (nc, lindisp, perturb) in enumerate([(64, False, False), (32, True, False), (64, False, True), (16, True, True)]):
            cfg = make_cfg(mode_cfg(nc, 0, perturb=perturb, lindisp=lindisp), mode_cfg(nc, 0))
            torch.manual_seed(30 + ci)
            tu.predict_and_render_radiance(rays, Stub(), Stub(), cfg, scene_id="x", mode="train")
            torch.manual_seed(30 + ci)
            t_rand = torch.rand(N, nc)
            out.update(
                {
                    "c%d_params" % ci: np.array([nc, int(lindisp), int(perturb)]),
                    "c%d_t_rand" % ci: npy(t_rand),
                    "c%d_z" % ci: npy(captured["z"].contiguous()),
                    "c%d_pts" % ci: npy(captured["pts"].contiguous()),
                }
            )
    finally:
        tu.run_network = real
    out["ro"] = npy(ro)
    out["rd"] = npy(rd)
    out["near"] = npy(near[:, 0])
    out["far"] = npy(far[:, 0])
    out["n_cases"] = np.array(4)
    save("g03_coarse_z.npz", **out)


def g04_decoder():
    R, Rv = 16, 8
    sid, mc, mf, planes, box = build_models(R, Rv, 0.5, seed=4)
    m = mc
    torch.manual_seed(40)
    P = 257
    pts = torch.rand(P, 3) * 8.4 - 4.2  # some points outside the box -> border clamp
    pts[0] = torch.tensor([-4.0, -4.0, -4.0])
    pts[1] = torch.tensor([4.0, 4.0, 4.0])
    pts[2] = torch.tensor([0.0, 0.0, 0.0])
    d = torch.randn(P, 3)
    d = d / d.norm(dim=-1, keepdim=True)
    d[3] = torch.tensor([0.0, 0.0, 1.0])
    d[4] = torch.tensor([-1.0, 0.0, 0.0])
    x = torch.cat([pts, d], -1)
    with torch.no_grad():
        out = m(x)
        x5 = torch.cat([x[..., :3], nh.cart2az_el(x[..., 3:])], -1)
        n5 = m.normalize_coords(x5)
        pos = m.project_xyz(n5[..., :3])
        view = m.project_viewdir(n5[..., 3:])
        dens_in = m.combine_pos_planes(pos)
        rgb_in = m.combine_all_planes(pos_planes=1 * pos, viewdir_planes=view)
    arrs = dict(
        x=npy(x),
        out=npy(out),
        norm_coords=npy(n5),
        feat0=npy(pos[0]),
        feat1=npy(pos[1]),
        feat2=npy(pos[2]),
        feat_view=npy(view),
        density_in=npy(dens_in),
        rgb_in=npy(rgb_in),
        box=npy(box),
    )
    for dnum in range(4):
        arrs["plane%d" % dnum] = npy(planes[models.get_plane_name(sid, dnum)])
    arrs.update(state_arrays("sd.", m))
    save("g04_decoder.npz", **arrs)


def g05_composite():
    out = {}
    torch.manual_seed(5)
    ci = 0
    for S in (8, 64, 192):
        for white in (False, True):
            for noise_std in (0.0, 0.2):
                N = 11
                raw = torch.randn(N, S, 4) * 2.0
                raw[..., 3] = raw[..., 3] * 3.0 - 2.0
                raw[0, :, 3] = -1.0  # sigma all-zero ray -> acc = 0, disp = nan
                raw[1, :, 3] = 50.0  # opaque at first sample
                z = torch.sort(torch.rand(N, S) * 4 + 2, dim=-1)[0]
                rd = torch.randn(N, 3)
                torch.manual_seed(500 + ci)
                rgb, disp, acc, w, depth = vru.volume_render_radiance_field(
                    raw, z, rd, radiance_field_noise_std=noise_std, white_background=white
                )
                torch.manual_seed(500 + ci)
                noise = torch.randn(N, S) * noise_std if noise_std > 0 else torch.zeros(N, S)
                out.update(
                    {
                        "c%d_params" % ci: np.array([S, int(white), noise_std]),
                        "c%d_raw" % ci: npy(raw),
                        "c%d_z" % ci: npy(z),
                        "c%d_rd" % ci: npy(rd),
                        "c%d_noise" % ci: npy(noise),
                        "c%d_rgb" % ci: npy(rgb),
                        "c%d_disp" % ci: npy(disp),
                        "c%d_acc" % ci: npy(acc),
                        "c%d_weights" % ci: npy(w),
                        "c%d_depth" % ci: npy(depth),
                    }
                )
                ci += 1
    out["n_cases"] = np.array(ci)
    # cumprod_exclusive on its own (nerf_helpers.py:409-430)
    t = torch.rand(5, 17)
    out["cumprod_in"] = npy(t)
    out["cumprod_out"] = npy(nh.cumprod_exclusive(t))
    save("g05_composite.npz", **out)


def g06_sample_pdf():
    out = {}
    torch.manual_seed(6)
    ci = 0
    for nb, ns, det, kind in [
        (63, 128, True, "rand"),
        (63, 64, True, "spike"),
        (63, 128, False, "rand"),
        (63, 128, True, "flat"),
        (31, 17, False, "zero"),
        (7, 5, True, "rand"),
        (63, 128, False, "spike"),
    ]:
        N = 13
        bins = torch.sort(torch.rand(N, nb) * 4 + 2, dim=-1)[0]
        if kind == "rand":
            w = torch.rand(N, nb - 1)
        elif kind == "flat":
            w = torch.ones(N, nb - 1) * 0.3
        elif kind == "zero":
            w = torch.zeros(N, nb - 1)
        else:
            w = torch.zeros(N, nb - 1)
            w[torch.arange(N), torch.randint(0, nb - 1, (N,))] = 0.9
        torch.manual_seed(600 + ci)
        s = nh.sample_pdf_2(bins, w, ns, det=det)
        torch.manual_seed(600 + ci)
        u = torch.linspace(0.0, 1.0, ns).expand(N, ns) if det else torch.rand(N, ns)
        out.update(
            {
                "c%d_params" % ci: np.array([nb, ns, int(det)]),
                "c%d_bins" % ci: npy(bins),
                "c%d_weights" % ci: npy(w),
                "c%d_u" % ci: npy(u.contiguous()),
                "c%d_samples" % ci: npy(s),
            }
        )
        ci += 1
    out["n_cases"] = np.array(ci)
    save("g06_sample_pdf.npz", **out)


def g07_sort():
    torch.manual_seed(7)
    out = {}
    for ci, (nc, nf) in enumerate([(64, 128), (64, 64), (5, 3)]):
        N = 9
        zc = torch.sort(torch.rand(N, nc) * 4 + 2, -1)[0]
        zs = torch.rand(N, nf) * 4 + 2  # unsorted (train-mode importance samples)
        zs[0, : min(nf, nc)] = zc[0, : min(nf, nc)]  # ties
        z, _ = torch.sort(torch.cat((zc, zs), dim=-1), dim=-1)
        out.update({"c%d_zc" % ci: npy(zc), "c%d_zs" % ci: npy(zs), "c%d_sorted" % ci: npy(z)})
    out["n_cases"] = np.array(3)
    save("g07_sort.npz", **out)


def g08_render():
    R, Rv = 32, 8
    sid, mc, mf, planes, box = build_models(R, Rv, 0.5, seed=8)
    H = W = 16
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    pose = torch.from_numpy(POSE)
    ro, rd = nh.get_ray_bundle(H, W, focal, pose)
    arrs = dict(box=npy(box), pose=POSE, hwf=np.array([H, W, focal], dtype=np.float64), ro=npy(ro.contiguous()), rd=npy(rd))
    for dnum in range(4):
        arrs["plane%d" % dnum] = npy(planes[models.get_plane_name(sid, dnum)])
    arrs.update(state_arrays("coarse.", mc))
    arrs.update(state_arrays("fine.", mf))
    mc.eval()
    mf.eval()
    ci = 0
    for nc, nf, white in [(32, 0, False), (64, 64, False), (64, 128, False), (64, 128, True)]:
        v = mode_cfg(nc, nf, white=white)
        cfg = make_cfg(v, v)
        zrec = []
        real_rn = tu.run_network

        def rec_rn(*a, **k):            # record the depths each pass is evaluated at (train_utils.py:122,167)
            zrec.append(k["z_vals"].clone())
            return real_rn(*a, **k)

        tu.run_network = rec_rn
        try:
            with torch.no_grad():
                rc, dc, ac, rf, df, af, *_ = tu.run_one_iter_of_nerf(
                    H, W, focal, mc, mf, torch.stack([ro.reshape(-1, 3), rd.reshape(-1, 3)], 0), cfg,
                    scene_id=sid, mode="validation", scene_config=cfg.dataset["synt"],
                )
        finally:
            tu.run_network = real_rn
        if nf > 0:
            arrs["e%d_z_fine" % ci] = npy(zrec[1])
        with torch.no_grad():
            img_c, _, _, img_f, *_ = tu.eval_nerf(H, W, focal, mc, mf, ro, rd, cfg, scene_id=sid, scene_config=cfg.dataset["synt"])
        assert torch.equal(img_c.reshape(-1, 3), rc)
        arrs["e%d_params" % ci] = np.array([nc, nf, int(white), 0, 0.0])
        arrs["e%d_rgb_coarse" % ci] = npy(rc)
        arrs["e%d_disp_coarse" % ci] = npy(dc)
        arrs["e%d_acc_coarse" % ci] = npy(ac)
        if nf > 0:
            arrs["e%d_rgb_fine" % ci] = npy(rf)
            arrs["e%d_disp_fine" % ci] = npy(df)
            arrs["e%d_acc_fine" % ci] = npy(af)
        print("   eval %s: acc_coarse mean %.3f  min %.3f max %.3f%s" % (
            (nc, nf, white), float(ac.mean()), float(ac.min()), float(ac.max()),
            "" if nf == 0 else "  acc_fine mean %.3f" % float(af.mean())))
        ci += 1
    # train mode: perturb + density noise + random u; RNG draws captured by re-seeding (CPU generator,
    # order: t_rand -> coarse noise -> u -> fine noise; train_utils.py:108, volume_rendering_utils.py:32,
    # nerf_helpers.py:683)
    nc, nf, std = 64, 64, 0.2
    vt = mode_cfg(nc, nf, perturb=True, noise=std)
    cfg = make_cfg(vt, mode_cfg(nc, nf))
    sel = torch.arange(0, H * W, 5)
    rays = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0)
    N = rays.shape[1]
    mc.train()
    mf.train()
    torch.manual_seed(88)
    zrec = []
    real_rn = tu.run_network

    def rec_rn(*a, **k):
        zrec.append(k["z_vals"].clone())
        return real_rn(*a, **k)

    tu.run_network = rec_rn
    try:
        with torch.no_grad():
            rc, dc, ac, rf, df, af, *_ = tu.run_one_iter_of_nerf(
                H, W, focal, mc, mf, rays, cfg, scene_id=sid, mode="train", scene_config=cfg.dataset["synt"])
    finally:
        tu.run_network = real_rn
    arrs["t_z_coarse"] = npy(zrec[0])
    arrs["t_z_fine"] = npy(zrec[1])
    torch.manual_seed(88)
    t_rand = torch.rand(N, nc)
    noise_c = torch.randn(N, nc) * std
    u = torch.rand(N, nf)
    noise_f = torch.randn(N, nc + nf) * std
    arrs.update(
        t_params=np.array([nc, nf, 0, 1, std]),
        t_sel=npy(sel),
        t_t_rand=npy(t_rand),
        t_noise_coarse=npy(noise_c),
        t_u=npy(u),
        t_noise_fine=npy(noise_f),
        t_rgb_coarse=npy(rc), t_disp_coarse=npy(dc), t_acc_coarse=npy(ac),
        t_rgb_fine=npy(rf), t_disp_fine=npy(df), t_acc_fine=npy(af),
    )
    arrs["n_eval"] = np.array(ci)
    save("g08_render.npz", **arrs)


def g09_edsr():
    torch.manual_seed(9)
    C, hidden, nblocks, sf, R = 6, 16, 2, 4, 20
    sr = models.PlanesSR(models.EDSR, sf, C, C, CfgNode({"model": {"hidden_size": hidden, "n_blocks": nblocks}}), "bilinear")
    sr.align_corners = True
    sr.eval()
    # default init is N(0, sqrt(2/(9*Cout))/10) (models.py:843-848): scale up so the net output is not ~0
    with torch.no_grad():
        for p in sr.parameters():
            p.mul_(10.0)
    lr = torch.randn(1, C, R, R) * 0.5
    sr.set_LR_plane(lr, id="p", save_interpolated=False)
    arrs = dict(
        cfg=np.array([C, hidden, nblocks, sf, R, sr.inner_model.required_padding, sr.HR_overpadding]),
        lr=npy(lr),
    )
    arrs.update({"sd." + k: npy(v) for k, v in sr.state_dict().items()})
    with torch.no_grad():
        x = torch.randn(1, C, 30, 26)
        arrs["edsr_in"] = npy(x)
        arrs["edsr_out"] = npy(sr.inner_model(x))
        blk = sr.inner_model.residual[0]
        h = torch.randn(1, hidden, 12, 9)
        arrs["block_in"] = npy(h)
        arrs["block_out"] = npy(blk(h.clone()))
        arrs["upsampled_lr"] = npy(sr.interpolate_LR("p"))
        hr = sr("p")
        arrs["sr_full"] = npy(hr)
        sr.clear_SR_planes()
        sr.train()  # ROI path is only taken in training mode (models.py:277); noise knobs are 0
        roi = torch.tensor([[-0.35, -0.6], [0.2, 0.15]])  # rows=(min,max), cols=(y,x) as built at models.py:278-279
        arrs["roi"] = npy(roi)
        arrs["sr_roi"] = npy(sr(("p", roi)))
        sr.eval()
    save("g09_edsr.npz", **arrs)


def g10_posenc():
    torch.manual_seed(10)
    x = torch.randn(19, 3) * 2
    arrs = dict(x=npy(x))
    arrs["pe_L6"] = npy(nh.positional_encoding(x, 6, True))
    arrs["pe_L4"] = npy(nh.positional_encoding(x, 4, True))
    arrs["pe_L4_noinput"] = npy(nh.positional_encoding(x, 4, False))
    m = models.FlexibleNeRFModel(num_layers=4, hidden_size=128, skip_connect_every=3, num_encoding_fn_xyz=6, num_encoding_fn_dir=4)
    m.eval()
    P = 33
    pts = torch.randn(P, 3)
    d = torch.randn(P, 3)
    d = d / d.norm(dim=-1, keepdim=True)
    inp = torch.cat([nh.positional_encoding(pts, 6), nh.positional_encoding(d, 4)], -1)
    with torch.no_grad():
        arrs["nerf_out"] = npy(m(inp))
    arrs["nerf_pts"] = npy(pts)
    arrs["nerf_dirs"] = npy(d)
    arrs.update({"sd." + k: npy(v) for k, v in m.state_dict().items()})
    save("g10_posenc.npz", **arrs)


def g11_grads():
    """d(mse_coarse + mse_fine)/d planes through run_one_iter_of_nerf (train_nerf.py:860,884-891,903), decoder frozen like
    Feature_Planes_Only.yml; case 0: train mode (perturb + density noise + random u), case 1: deterministic sampling."""
    R, Rv = 16, 8
    sid, mc, mf, planes, box = build_models(R, Rv, 0.5, seed=11)
    H = W = 16
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    ro, rd = nh.get_ray_bundle(H, W, focal, torch.from_numpy(POSE))
    arrs = dict(box=npy(box), hwf=np.array([H, W, focal], dtype=np.float64))
    for dnum in range(4):
        arrs["plane%d" % dnum] = npy(planes[models.get_plane_name(sid, dnum)])
    arrs.update(state_arrays("coarse.", mc))
    arrs.update(state_arrays("fine.", mf))
    torch.manual_seed(110)
    sel = torch.randperm(H * W)[:64]
    rays = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0)
    target = torch.rand(64, 3)
    arrs.update(rays=npy(rays), target=npy(target))
    N = 64
    for ci, (nc, nf, perturb, std) in enumerate([(32, 32, True, 0.2), (24, 40, False, 0.0)]):
        vt = mode_cfg(nc, nf, perturb=perturb, noise=std)
        cfg = make_cfg(vt, vt)
        mc.train(); mf.train()
        for p_ in planes.values():
            p_.grad = None
        torch.manual_seed(111 + ci)
        rc, dc, ac, rf, df, af, *_ = tu.run_one_iter_of_nerf(H, W, focal, mc, mf, rays, cfg, scene_id=sid, mode="train",
                                                             scene_config=cfg.dataset["synt"])
        loss = torch.nn.functional.mse_loss(rc, target) + torch.nn.functional.mse_loss(rf, target)
        loss.backward()
        torch.manual_seed(111 + ci)
        if perturb:
            arrs["c%d_t_rand" % ci] = npy(torch.rand(N, nc))
        if std > 0:
            arrs["c%d_noise_coarse" % ci] = npy(torch.randn(N, nc) * std)
        if perturb:
            arrs["c%d_u" % ci] = npy(torch.rand(N, nf))
        if std > 0:
            arrs["c%d_noise_fine" % ci] = npy(torch.randn(N, nc + nf) * std)
        arrs["c%d_params" % ci] = np.array([nc, nf, int(perturb), std])
        arrs["c%d_rgb_coarse" % ci] = npy(rc)
        arrs["c%d_rgb_fine" % ci] = npy(rf)
        arrs["c%d_loss" % ci] = np.array(float(loss))
        for dnum in range(4):
            g_ = planes[models.get_plane_name(sid, dnum)].grad
            arrs["c%d_grad_plane%d" % (ci, dnum)] = npy(g_)
        print("   grads case %d: loss %.5f, |grad| per plane %s, nonzero texel fraction %.2f" % (
            ci, float(loss), ["%.2e" % float(planes[models.get_plane_name(sid, d)].grad.abs().mean()) for d in range(4)],
            float((planes[models.get_plane_name(sid, 0)].grad.abs().sum(1) > 0).float().mean())))
    arrs["n_cases"] = np.array(2)
    save("g11_grads.npz", **arrs)


def g12_ndc_render():
    """BASELINE config 5 in miniature: no_ndc=False, near=0, far=1 (config dataset.llff), 64+128 samples."""
    R, Rv = 24, 8
    torch.manual_seed(12)
    np.random.seed(12)
    sid = models.get_scene_id("fern", 8, (R, Rv))
    sc = models.SceneCoupler([sid], planes_res="LR", num_pos_planes=3, training_scenes=[sid])
    kw = dict(use_viewdirs=True, skip_connect_every=3, proj_combination="avg", viewdir_proj_combination="concat_pos",
              align_corners=True, scene_coupler=sc)
    mc = models.TwoDimPlanesModel(**kw); mc.optional_no_grad = nh.null_with
    mf = models.TwoDimPlanesModel(num_planes_or_rot_mats=mc.rot_mats(), **kw); mf.optional_no_grad = nh.null_with
    planes = nn.ParameterDict([(models.get_plane_name(sid, d), models.create_plane(R if d < 3 else Rv, 48, 0.5)) for d in range(4)])
    # NDC box: x,y in [-1.2,1.2], z in [-1,1]; view directions of a forward-facing rig
    box = torch.tensor([[-1.2, -1.2, -1.05, -np.pi, -np.pi / 2], [1.2, 1.2, 1.05, np.pi, np.pi / 2]], dtype=torch.float64)
    for m in (mc, mf):
        m.planes_ = planes; m.plane_rank = None; m.generated_planes = {}; m.downsampled_planes = {}; m.coverages = {}
        m.box_coords = {sid: box}; m.set_cur_scene_id(sid); m.eval()
    with torch.no_grad():
        g = torch.Generator().manual_seed(13)
        pts = torch.rand(4096, 3, generator=g) * 2 - 1
        d = torch.randn(4096, 3, generator=g); d = d / d.norm(dim=-1, keepdim=True)
        x = torch.cat([pts, d], -1)
        for m in (mc, mf):
            raw = m(x)[:, 3]
            scale = 8.0 / float(raw.std())         # NDC depths span [0,1]: 4x shorter rays than the synthetic scenes
            m.fc_alpha["0"].weight.mul_(scale); m.fc_alpha["0"].bias.mul_(scale)
            raw = m(x)[:, 3]
            m.fc_alpha["0"].bias.add_(-float(raw.mean()) - 3.0)
    H, W, focal = 12, 16, 14.0
    pose = torch.eye(4)
    pose[:3, 3] = torch.tensor([0.05, -0.03, 0.1])
    c, s_ = np.cos(0.1), np.sin(0.1)
    pose[:3, :3] = torch.tensor([[c, 0, s_], [0, 1, 0], [-s_, 0, c]], dtype=torch.float32)
    ro, rd = nh.get_ray_bundle(H, W, focal, pose)
    v = mode_cfg(64, 128)
    cfg = CfgNode({"nerf": {"use_viewdirs": True, "train": v, "validation": v},
                   "dataset": {"llff": {"near": 0, "far": 1, "no_ndc": False}}})
    zrec = []
    real_rn = tu.run_network

    def rec_rn(*a, **k):
        zrec.append(k["z_vals"].clone())
        return real_rn(*a, **k)

    tu.run_network = rec_rn
    try:
        with torch.no_grad():
            rc, dc, ac, rf, df, af, *_ = tu.run_one_iter_of_nerf(H, W, focal, mc, mf, torch.stack([ro.reshape(-1, 3), rd.reshape(-1, 3)], 0),
                                                                 cfg, scene_id=sid, mode="validation", scene_config=cfg.dataset["llff"])
    finally:
        tu.run_network = real_rn
    print("   ndc eval: acc_coarse mean %.3f [%.3f, %.3f]  acc_fine mean %.3f" % (float(ac.mean()), float(ac.min()), float(ac.max()), float(af.mean())))
    arrs = dict(box=npy(box), pose=npy(pose), hwf=np.array([H, W, focal], dtype=np.float64), ro=npy(ro.contiguous()), rd=npy(rd),
                rgb_coarse=npy(rc), acc_coarse=npy(ac), disp_coarse=npy(dc), rgb_fine=npy(rf), acc_fine=npy(af), disp_fine=npy(df),
                z_fine=npy(zrec[1]))
    for dnum in range(4):
        arrs["plane%d" % dnum] = npy(planes[models.get_plane_name(sid, dnum)])
    arrs.update(state_arrays("coarse.", mc))
    arrs.update(state_arrays("fine.", mf))
    save("g12_ndc_render.npz", **arrs)


DEC_KEYS = (["density_dec.0.%d" % i for i in range(4)] + ["fc_alpha.0"] + ["rgb_dec.0.%d" % i for i in range(4)] + ["fc_rgb.0"])


def g13_decoder_grads():
    """d(mse_coarse + mse_fine)/d(decoder weights and biases) of model_coarse and model_fine for the g11 train step (same
    planes, rays, targets and random draws: g11_grads.npz holds the inputs).  Gradients are stored flattened in state-dict
    order (weight, bias per layer: density_dec 0..3, fc_alpha, rgb_dec 0..3, fc_rgb), the layout of the 'natural' blob."""
    R, Rv = 16, 8
    sid, mc, mf, planes, box = build_models(R, Rv, 0.5, seed=11)
    H = W = 16
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    ro, rd = nh.get_ray_bundle(H, W, focal, torch.from_numpy(POSE))
    torch.manual_seed(110)
    sel = torch.randperm(H * W)[:64]
    rays = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0)
    target = torch.rand(64, 3)
    arrs = {}
    for ci, (nc, nf, perturb, std) in enumerate([(32, 32, True, 0.2), (24, 40, False, 0.0)]):
        vt = mode_cfg(nc, nf, perturb=perturb, noise=std)
        cfg = make_cfg(vt, vt)
        mc.train(); mf.train()
        for m in (mc, mf):
            m.zero_grad(set_to_none=True)
        for p_ in planes.values():
            p_.grad = None
        torch.manual_seed(111 + ci)
        rc, dc, ac, rf, df, af, *_ = tu.run_one_iter_of_nerf(H, W, focal, mc, mf, rays, cfg, scene_id=sid, mode="train",
                                                             scene_config=cfg.dataset["synt"])
        loss = torch.nn.functional.mse_loss(rc, target) + torch.nn.functional.mse_loss(rf, target)
        loss.backward()
        arrs["c%d_loss" % ci] = np.array(float(loss))
        for tag, m in (("coarse", mc), ("fine", mf)):
            sd = dict(m.named_parameters())
            flat = []
            for k in DEC_KEYS:
                for leaf in ("weight", "bias"):
                    g_ = sd[k + "." + leaf].grad
                    assert g_ is not None, k
                    flat.append(npy(g_).reshape(-1))
            arrs["c%d_%s_grad" % (ci, tag)] = np.concatenate(flat).astype(np.float32)
        print("   decoder grads case %d: loss %.5f  |g| coarse %.3e fine %.3e" % (
            ci, float(loss), np.abs(arrs["c%d_coarse_grad" % ci]).mean(), np.abs(arrs["c%d_fine_grad" % ci]).mean()))
    arrs["n_cases"] = np.array(2)
    save("g13_decoder_grads.npz", **arrs)


def g14_sr_grads():
    """torch.autograd through the SR network of g09 (same seed -> same weights / LR plane): (1) EDSR alone, gradient of
    sum(out * G) wrt every conv weight and the input; (2) PlanesSR.forward on the ROI (training path) and on the full plane,
    gradient of sum(nan_to_zero(out) * G) wrt the weights and the (non-detached) LR plane."""
    torch.manual_seed(9)
    C, hidden, nblocks, sf, R = 6, 16, 2, 4, 20
    sr = models.PlanesSR(models.EDSR, sf, C, C, CfgNode({"model": {"hidden_size": hidden, "n_blocks": nblocks}}), "bilinear")
    sr.align_corners = True
    with torch.no_grad():
        for p in sr.parameters():
            p.mul_(10.0)
    lr = torch.randn(1, C, R, R) * 0.5
    g9 = np.load(os.path.join(HERE, "g09_edsr.npz"))
    assert np.array_equal(g9["lr"], npy(lr)), "g14 must rebuild the g09 network"
    keys = [k for k, _ in sr.inner_model.named_parameters()]
    arrs = dict(cfg=np.array([C, hidden, nblocks, sf, R, sr.inner_model.required_padding, sr.HR_overpadding]))

    def blob_grad():
        return np.concatenate([npy(p.grad).reshape(-1) for _, p in sr.inner_model.named_parameters()]).astype(np.float32)

    # (1) EDSR alone
    torch.manual_seed(140)
    x = torch.randn(1, C, 30, 26, requires_grad=True)
    sr.train()
    out = sr.inner_model(x)
    G = torch.randn_like(out)
    sr.zero_grad(set_to_none=True)
    (out * G).sum().backward()
    arrs.update(edsr_in=npy(x), edsr_gout=npy(G), edsr_gw=blob_grad(), edsr_gin=npy(x.grad))
    # (2) PlanesSR: ROI (training) and full plane
    roi = torch.tensor([[-0.35, -0.6], [0.2, 0.15]])
    arrs["roi"] = npy(roi)
    for tag, arg in (("roi", ("p", roi)), ("full", "p")):
        lrp = nn.Parameter(lr.clone())
        sr.clear_SR_planes(all_planes=True)
        sr.set_LR_plane(lrp, id="p", save_interpolated=False)
        sr.zero_grad(set_to_none=True)
        out = sr(arg)
        Gp = torch.randn_like(out)
        valid = ~torch.isnan(out)
        (torch.where(valid, out, torch.zeros_like(out)) * Gp).sum().backward()
        arrs.update({"sr_%s_gout" % tag: npy(Gp), "sr_%s_gw" % tag: blob_grad(), "sr_%s_glr" % tag: npy(lrp.grad),
                     "sr_%s_valid_frac" % tag: np.array(float(valid.float().mean()))})
        sr.clear_SR_planes(all_planes=True)
    arrs["param_order"] = np.array(keys)
    save("g14_sr_grads.npz", **arrs)


def g15_loaders():
    import json
    import shutil
    import tempfile

    from PIL import Image

    def pil_imread(path, **kw):
        with Image.open(path) as im:
            return np.asarray(im)

    def area_resize(img, dsize, interpolation=None):
        w, h = dsize
        fy, fx = img.shape[0] // h, img.shape[1] // w
        assert fy * h == img.shape[0] and fx * w == img.shape[1]
        return img.reshape(h, fy, w, fx, -1).mean(axis=(1, 3)).astype(img.dtype).reshape((h, w) + img.shape[2:])

    sys.modules["imageio"].imread = pil_imread
    sys.modules["cv2"].resize = area_resize
    sys.modules["cv2"].INTER_AREA = 3
    import load_blender as ref_blender
    import load_llff as ref_llff

    rng = np.random.default_rng(15)
    arrs = {}
    root = tempfile.mkdtemp(prefix="g15_")
    try:
        # ---- Blender layout: 3 train / 4 val / 2 test RGBA frames of 8 x 12 pixels ----
        bdir = os.path.join(root, "toyscene")
        counts = {"train": 3, "val": 4, "test": 2}
        for split, n in counts.items():
            os.makedirs(os.path.join(bdir, split))
            frames = []
            for k in range(n):
                rgba = rng.integers(0, 256, (8, 12, 4), dtype=np.uint8)
                rgba[..., 3] = np.where(rng.random((8, 12)) < 0.3, 0, rgba[..., 3])      # some fully transparent pixels
                Image.fromarray(rgba, "RGBA").save(os.path.join(bdir, split, "r_%d.png" % k))
                pose = ref_blender.pose_spherical(float(rng.uniform(-180, 180)), float(rng.uniform(-60, 0)), 4.0)
                frames.append({"file_path": "./%s/r_%d" % (split, k), "transform_matrix": pose.tolist()})
                arrs["blender_%s_%d" % (split, k)] = rgba
                arrs["blender_%s_%d_pose" % (split, k)] = np.asarray(pose)
            with open(os.path.join(bdir, "transforms_%s.json" % split), "w") as fp:
                json.dump({"camera_angle_x": 0.6911112, "frames": frames}, fp)
        arrs["blender_counts"] = np.array([counts[s] for s in ("train", "val", "test")])
        for tag, kw in (("a", dict(downsampling_factor=2, val_downsampling_factor=1, testskip=2, splits2use=["train", "val"])),
                        ("b", dict(downsampling_factor=4, splits2use=["train", "val", "test"]))):
            imgs, poses, render_poses, (H, W, focal, ds), i_split = ref_blender.load_blender_data(bdir, **kw)
            for k, im in enumerate(imgs):
                arrs["blender_%s_img%d" % (tag, k)] = npy(im)
            arrs.update({"blender_%s_poses" % tag: npy(poses), "blender_%s_render_poses" % tag: npy(render_poses),
                         "blender_%s_H" % tag: np.array(H), "blender_%s_W" % tag: np.array(W), "blender_%s_focal" % tag: np.array(focal),
                         "blender_%s_ds" % tag: np.array(ds)})
            for k, idx in enumerate(i_split):
                arrs["blender_%s_split%d" % (tag, k)] = np.asarray(idx)
        # ---- LLFF layout: 6 views of 14 x 18 RGB pixels (odd multiples: cropped to 12 x 16 for max_factor 4) ----
        ldir = os.path.join(root, "toyfern")
        os.makedirs(os.path.join(ldir, "images"))
        n = 6
        pb = np.zeros((n, 17))
        for k in range(n):
            rgb = rng.integers(0, 256, (14, 18, 3), dtype=np.uint8)
            Image.fromarray(rgb, "RGB").save(os.path.join(ldir, "images", "view_%02d.png" % k))
            arrs["llff_img_%d" % k] = rgb
            q, _ = np.linalg.qr(rng.standard_normal((3, 3)) * 0.15 + np.eye(3))
            m = np.concatenate([q, rng.uniform(-0.5, 0.5, (3, 1)), np.array([[14.0], [18.0], [21.0]])], 1)
            pb[k, :15] = m.reshape(-1)
            pb[k, 15:] = [rng.uniform(1.0, 1.5), rng.uniform(8.0, 12.0)]
        np.save(os.path.join(ldir, "poses_bounds.npy"), pb)
        arrs["llff_poses_bounds"] = pb
        for tag, kw in (("fwd", dict(factor=2, base_factor=1, max_factor=4)),
                        ("sph", dict(factor=4, base_factor=1, max_factor=4, spherify=True)),
                        ("flat", dict(factor=2, base_factor=1, max_factor=2, path_zflat=True, bd_factor=None))):
            if tag == "flat":        # the reference turns N_views into a float there, which numpy >= 1.18 rejects in linspace
                _orig = ref_llff.render_path_spiral
                ref_llff.render_path_spiral = lambda *a, **k: _orig(*a, **{**k, "N": int(k["N"])})
            images, poses, bds, render_poses, i_test, (bf, marg) = ref_llff.load_llff_data(ldir, **kw)
            arrs.update({"llff_%s_images" % tag: npy(images), "llff_%s_poses" % tag: npy(poses), "llff_%s_bds" % tag: np.asarray(bds),
                         "llff_%s_render_poses" % tag: np.asarray(render_poses), "llff_%s_i_test" % tag: np.array(int(i_test)),
                         "llff_%s_base_factor" % tag: np.array(bf),
                         "llff_%s_margins" % tag: np.array([-1, -1]) if marg is None else np.asarray(marg)})
    finally:
        shutil.rmtree(root)
    save("g15_loaders.npz", **arrs)


def g16_composite_mip():
    torch.manual_seed(16)
    arrs = {}
    for tag, (N, S, white) in {"a": (37, 24, False), "b": (5, 130, True)}.items():
        raw = torch.randn(N, S, 4) * 2.0
        raw[..., 3] = raw[..., 3] * 3.0 - 1.0
        raw[1, :, 3] = -5.0                                            # an empty ray: acc = 0, disp = NaN
        z = torch.sort(torch.rand(N, S + 1) * 4.0 + 2.0, -1).values    # S + 1 interval edges
        rd = torch.randn(N, 3)
        noise = torch.randn(N, S) * 0.3
        g_rgb, g_acc = torch.randn(N, 3), torch.randn(N)
        r = (raw.clone()).requires_grad_(True)
        # the reference draws its own noise; add it to the density channel instead (same arithmetic: relu(raw + noise))
        rin = torch.cat([r[..., :3], (r[..., 3] + noise)[..., None]], -1)
        rgb, disp, acc, w, depth = vru.volume_render_radiance_field(rin, z, rd, radiance_field_noise_std=0.0, white_background=white, mip_nerf=True)
        ((rgb * g_rgb).sum() + (acc * g_acc).sum()).backward()
        arrs.update({tag + "_raw": npy(raw), tag + "_z": npy(z), tag + "_rd": npy(rd), tag + "_noise": npy(noise), tag + "_white": np.array(white),
                     tag + "_rgb": npy(rgb), tag + "_disp": npy(disp), tag + "_acc": npy(acc), tag + "_weights": npy(w), tag + "_depth": npy(depth),
                     tag + "_g_rgb": npy(g_rgb), tag + "_g_acc": npy(g_acc), tag + "_g_raw": npy(r.grad)})
    save("g16_composite_mip.npz", **arrs)


def g17_store():
    """Files WRITTEN BY THE REFERENCE for the plane store / checkpoint formats:
      g17_store/planes/coarse_<sid>.par      PlanesOptimizer.__init__(init_params) -> draw_scenes -> one Adam step -> save_params
                                              (models.py:499-581,640-670,683-726) through safe_saving (nerf_helpers.py:19-48)
      g17_store/checkpoint00007.ckpt         {model_coarse_state_dict, model_fine_state_dict, optimizer} with the key filtering of
                                              train_nerf.py:996-1008, through safe_saving
      g17_store/SR_checkpoint00007.ckpt      {SR_model, SR_optimizer} (train_nerf.py:996-999)
    and, in g17_store.npz, what the reference renders / super-resolves from those files after reading them back itself
    (PlanesOptimizer.load_scene, load_state_dict(strict=False), safe_loading)."""
    import shutil
    from collections import OrderedDict

    out_dir = os.path.join(HERE, "g17_store")
    shutil.rmtree(out_dir, ignore_errors=True)
    planes_dir = os.path.join(out_dir, "planes") + "/"
    os.makedirs(planes_dir)
    R, Rv = 12, 6
    sid, mc, mf, _, box = build_models(R, Rv, 0.5, seed=17)
    # -- the plane store, written by PlanesOptimizer itself ---------------------------------------------------------------
    torch.manual_seed(170)
    opt = models.PlanesOptimizer(optimizer_type="Adam", scene_id_plane_resolution={sid: (R, Rv)}, options=CfgNode({"steps_per_buffer": -1}),
                                 save_location=[planes_dir], lr=5e-3, model_coarse=mc, model_fine=mf, use_coarse_planes=True, init_params=True,
                                 optimize=True, training_scenes=[sid], coords_normalization={sid: box}, available_scenes=[sid],
                                 STD_factor=2.0, run_time_signature=0)
    opt.draw_scenes()                                     # reads the initial file, builds Adam
    for m in (mc, mf):
        m.set_cur_scene_id(sid)
    opt.cur_id = sid
    g = torch.Generator().manual_seed(171)
    for p_ in mc.planes_.values():                        # one real Adam step so that the saved opt_states are populated
        p_.grad = torch.randn(p_.shape, generator=g) * 1e-2
    opt.step()
    opt.save_params()                                     # safe_saving({'params','opt_states','coords_normalization'})
    # -- decoder checkpoint with train_nerf.py's key filtering -------------------------------------------------------------
    optimizer = torch.optim.Adam([p_ for k, p_ in mc.named_parameters() if "planes_" not in k and "NON_LEARNED" not in k], lr=1e-4)
    tokens_2_exclude = ["planes_.", "SR_model"]
    ck = {"model_coarse_state_dict": mc.state_dict(), "model_fine_state_dict": mf.state_dict()}
    ck["model_fine_state_dict"] = OrderedDict([(k, v) for k, v in ck["model_fine_state_dict"].items() if all(t not in k for t in tokens_2_exclude + ["rot_mats"])])
    ck["model_coarse_state_dict"] = OrderedDict([(k, v) for k, v in ck["model_coarse_state_dict"].items() if all(t not in k for t in tokens_2_exclude)])
    ck["optimizer"] = optimizer.state_dict()
    nh.safe_saving(os.path.join(out_dir, "checkpoint00007.ckpt"), content=ck, suffix="ckpt", best=False, run_time_signature=0)
    # -- SR checkpoint -----------------------------------------------------------------------------------------------------
    torch.manual_seed(172)
    C, hidden, nblocks, sf = 48, 16, 2, 4
    sr = models.PlanesSR(models.EDSR, sf, C, C, CfgNode({"model": {"hidden_size": hidden, "n_blocks": nblocks}}), "bilinear")
    with torch.no_grad():
        for p_ in sr.parameters():
            p_.mul_(10.0)
    sr_opt = torch.optim.Adam(sr.parameters(), lr=1e-4)
    nh.safe_saving(os.path.join(out_dir, "SR_checkpoint00007.ckpt"), content={"SR_model": sr.state_dict(), "SR_optimizer": sr_opt.state_dict()},
                   suffix="ckpt", best=False, run_time_signature=0)
    # -- read everything back WITH THE REFERENCE into fresh models and record what it computes ---------------------------------
    sid2, mc2, mf2, _, _ = build_models(R, Rv, 0.5, seed=999)          # different weights / planes: everything must come from the files
    assert sid2 == sid
    opt2 = models.PlanesOptimizer(optimizer_type="Adam", scene_id_plane_resolution={sid: (R, Rv)}, options=CfgNode({"steps_per_buffer": -1}),
                                  save_location=[planes_dir], lr=5e-3, model_coarse=mc2, model_fine=mf2, use_coarse_planes=True,
                                  init_params=False, optimize=False, training_scenes=[sid], available_scenes=[sid], run_time_signature=0)
    opt2.load_scene(sid)
    ckpt_path = tu.find_latest_checkpoint(out_dir, sr=False)
    sr_path = tu.find_latest_checkpoint(out_dir, sr=True)
    assert ckpt_path.endswith("checkpoint00007.ckpt") and sr_path.endswith("SR_checkpoint00007.ckpt")
    loaded = nh.safe_loading(ckpt_path, suffix="ckpt")
    mc2.load_state_dict(loaded["model_coarse_state_dict"], strict=False)
    mf2.load_state_dict(mf2.rot_mat_backward_support(loaded["model_fine_state_dict"]), strict=False)
    sr2 = models.PlanesSR(models.EDSR, sf, C, C, CfgNode({"model": {"hidden_size": hidden, "n_blocks": nblocks}}), "bilinear")
    sr2.load_state_dict(nh.safe_loading(sr_path, suffix="ckpt")["SR_model"])
    for m in (mc2, mf2):
        m.eval()
        m.set_cur_scene_id(sid)
    H = W = 12
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    ro, rd = nh.get_ray_bundle(H, W, focal, torch.from_numpy(POSE))
    cfg = make_cfg(mode_cfg(32, 32), mode_cfg(32, 32))
    with torch.no_grad():
        rgb_c, _, _, rgb_f, *_ = tu.eval_nerf(H, W, focal, mc2, mf2, ro, rd, cfg, scene_id=sid, scene_config=cfg.dataset["synt"])
        sr2.align_corners = True
        sr2.eval()
        name0 = models.get_plane_name(sid, 0)
        sr2.set_LR_plane(mc2.planes_[name0].detach(), id=name0, save_interpolated=False)
        sr_plane = sr2(name0)
    par = nh.safe_loading(os.path.join(planes_dir, "coarse_%s.par" % sid), suffix="par")
    save("g17_store.npz", sid=np.array(sid), cfg=np.array([H, W, 32, 32, R, Rv, C, hidden, nblocks, sf]), focal=np.float64(focal), pose=POSE,
         rgb_coarse=npy(rgb_c), rgb_fine=npy(rgb_f), sr_plane0=npy(sr_plane), box=np.asarray(par["coords_normalization"], np.float64),
         plane0=npy(par["params"][name0]), adam_step=np.float64(float(par["opt_states"][0]["step"])),
         adam_exp_avg0=npy(par["opt_states"][0]["exp_avg"]), fc_alpha_w=npy(mc2.fc_alpha["0"].weight))
    for f in sorted(os.listdir(out_dir)) + sorted(os.listdir(planes_dir)):
        fp = os.path.join(out_dir, f) if os.path.exists(os.path.join(out_dir, f)) else os.path.join(planes_dir, f)
        if os.path.isfile(fp):
            print("wrote g17_store/%-28s %8.1f KB" % (os.path.relpath(fp, out_dir), os.path.getsize(fp) / 1024))


# geometry variants of TwoDimPlanesModel that the shipped YAMLs list as alternatives (config/TrainModels.yml:78,82,92): constructor kwargs
G18_VARIANTS = {
    "wide256": dict(dec_channels=256, proj_combination="avg", viewdir_proj_combination="concat_pos", num_plane_channels=48),
    "sum_sum": dict(dec_channels=128, proj_combination="sum", viewdir_proj_combination=None, num_plane_channels=48),
    "avg_mult": dict(dec_channels=64, proj_combination="avg", viewdir_proj_combination="mult", num_plane_channels=48),
    "concat24": dict(dec_channels=128, proj_combination="concat", viewdir_proj_combination="concat", num_plane_channels=24),
    "skip2": dict(dec_channels=128, proj_combination="avg", viewdir_proj_combination="concat_pos", num_plane_channels=48, skip_connect_every=2),
    "deep5_c24": dict(dec_channels=96, proj_combination="sum", viewdir_proj_combination="concat_pos", num_plane_channels=24, dec_density_layers=5,
                      dec_rgb_layers=3, skip_connect_every=2),
}


def g18_decoder_variants():
    """TwoDimPlanesModel.forward (models.py:381-421) for decoder geometries other than the shipped one: other widths, plane channel counts,
    proj_combination sum / concat, viewdir_proj_combination sum / mult / concat, skip layers, unequal layer counts.  Per variant: the
    points, planes, box, state dict and the reference's output; for one variant also an eval_nerf render (8 x 8 rays, 16 + 16 samples).
    Also writes g19_decoder_variant_grads.npz: the reference's autograd gradients of planes and decoder parameters for five of the variants."""
    arrs, grads = {}, {}
    R, Rv, P = 10, 6, 203
    for vi, (name, kw) in enumerate(G18_VARIANTS.items()):
        torch.manual_seed(180 + vi)
        np.random.seed(180 + vi)
        sid = models.get_scene_id("lego", 8, (R, Rv))
        sc = models.SceneCoupler([sid], planes_res="LR", num_pos_planes=3, training_scenes=[sid])
        kwargs = dict(use_viewdirs=True, skip_connect_every=3, align_corners=True, scene_coupler=sc)
        kwargs.update(kw)
        m = models.TwoDimPlanesModel(**kwargs)
        m.optional_no_grad = nh.null_with
        Cc = kwargs["num_plane_channels"]
        planes = nn.ParameterDict([(models.get_plane_name(sid, d), models.create_plane(R if d < 3 else Rv, Cc, 0.5)) for d in range(4)])
        box = torch.tensor(BOX, dtype=torch.float64)
        m.planes_, m.plane_rank, m.generated_planes, m.downsampled_planes, m.coverages = planes, None, {}, {}, {}
        m.box_coords = {sid: box}
        m.set_cur_scene_id(sid)
        m.eval()
        pts = torch.rand(P, 3) * 8.4 - 4.2
        d = torch.randn(P, 3)
        x = torch.cat([pts, d / d.norm(dim=-1, keepdim=True)], -1)
        mf = None
        if name == "concat24":             # this variant is also rendered: a fine model, densities spread so that compositing is exercised
            mf = models.TwoDimPlanesModel(num_planes_or_rot_mats=m.rot_mats(), **{k: v for k, v in kwargs.items()})
            mf.optional_no_grad = nh.null_with
            mf.planes_, mf.plane_rank, mf.generated_planes, mf.downsampled_planes, mf.coverages = planes, None, {}, {}, {}
            mf.box_coords = {sid: box}
            mf.set_cur_scene_id(sid)
            mf.eval()
            with torch.no_grad():
                for mm in (m, mf):
                    raw = mm(x)[:, 3]
                    s_ = 1.0 / float(raw.std())
                    mm.fc_alpha["0"].weight.mul_(s_)
                    mm.fc_alpha["0"].bias.mul_(s_)
                    mm.fc_alpha["0"].bias.add_(-float(mm(x)[:, 3].mean()) - 0.5)
        with torch.no_grad():
            out = m(x)
        pre = "%s." % name
        arrs[pre + "x"], arrs[pre + "out"] = npy(x), npy(out)
        for dnum in range(4):
            arrs[pre + "plane%d" % dnum] = npy(planes[models.get_plane_name(sid, dnum)])
        arrs.update(state_arrays(pre + "sd.", m))
        if name != "wide256":
            # g19: torch.autograd through the same forward -- gradients of the four planes and of the decoder parameters (state-dict order:
            # density_dec.0.{l}.{weight,bias}, fc_alpha.0, rgb_dec.0.{l}, fc_rgb.0) for a fixed cotangent (its own generator: g18 is unchanged)
            gout = torch.randn(P, 4, generator=torch.Generator().manual_seed(1900 + vi))
            sdp = dict(m.named_parameters())
            keys = ["density_dec.0.%d" % i for i in range(kwargs.get("dec_density_layers", 4))] + ["fc_alpha.0"] + \
                   ["rgb_dec.0.%d" % i for i in range(kwargs.get("dec_rgb_layers", 4))] + ["fc_rgb.0"]
            params = [sdp[k + sfx] for k in keys for sfx in (".weight", ".bias")]
            plist = [planes[models.get_plane_name(sid, dnum)] for dnum in range(4)]
            for t_ in params + plist:
                t_.requires_grad_(True)
                t_.grad = None
            (m(x) * gout).sum().backward()
            grads[pre + "gout"] = npy(gout)
            grads[pre + "gnat"] = np.concatenate([npy(t_.grad).reshape(-1) for t_ in params])
            for dnum in range(4):
                grads[pre + "gplane%d" % dnum] = npy(plist[dnum].grad)
        if name == "concat24":
            arrs.update(state_arrays(pre + "render.coarse.", m))
            arrs.update(state_arrays(pre + "render.fine.", mf))
            H = W = 8
            focal = 0.5 * W / np.tan(0.5 * 0.6911112)
            ro, rd = nh.get_ray_bundle(H, W, focal, torch.from_numpy(POSE))
            cfg = make_cfg(mode_cfg(16, 16), mode_cfg(16, 16))
            with torch.no_grad():
                rgb_c, _, _, rgb_f, *_ = tu.eval_nerf(H, W, focal, m, mf, ro, rd, cfg, scene_id=sid, scene_config=cfg.dataset["synt"])
            arrs[pre + "render.rgb_coarse"], arrs[pre + "render.rgb_fine"] = npy(rgb_c), npy(rgb_f)
            arrs[pre + "render.hwf"] = np.array([H, W, focal])
    arrs["box"] = np.array(BOX, np.float64)
    arrs["pose"] = POSE
    save("g18_decoder_variants.npz", **arrs)
    save("g19_decoder_variant_grads.npz", **grads)


def g20_sr_options():
    """PlanesSR with the options g09 leaves at their defaults: `input_normalization` (models.py:855-857,899-901: the network
    sees (LR - mean) / std per channel, the bilinear residual the raw plane) -- full plane in eval mode and an ROI in training mode -- and the
    training noises `sr_input_noise` / `sr_output_noise` (models.py:896-897,920-921), drawn by torch.normal from the CPU generator seeded
    here: the build draws the same numbers in the same order."""
    torch.manual_seed(20)
    C, hidden, nblocks, sf, R = 6, 16, 2, 4, 20
    cfg = CfgNode({"model": {"hidden_size": hidden, "n_blocks": nblocks}, "input_normalization": True})
    sr = models.PlanesSR(models.EDSR, sf, C, C, cfg, "bilinear")
    sr.align_corners = True
    sr.eval()
    with torch.no_grad():
        for p_ in sr.inner_model.parameters():
            p_.mul_(10.0)
    lr = torch.randn(1, C, R, R) * 0.5 + torch.linspace(-1.0, 1.0, C).reshape(1, C, 1, 1)
    mean, std = lr.mean(dim=(0, 2, 3)), lr.std(dim=(0, 2, 3)) * 1.7
    sr.normalization_params({"mean": mean.clone(), "std": std.clone()})
    sr.set_LR_plane(lr, id="p", save_interpolated=False)
    arrs = dict(cfg=np.array([C, hidden, nblocks, sf, R, sr.inner_model.required_padding, sr.HR_overpadding]), lr=npy(lr), mean=npy(mean),
                std=npy(std))
    arrs.update({"sd." + k: npy(v) for k, v in sr.state_dict().items()})
    roi = torch.tensor([[-0.5, -0.2], [0.3, 0.65]])
    arrs["roi"] = npy(roi)
    with torch.no_grad():
        arrs["sr_full"] = npy(sr("p"))
        sr.clear_SR_planes()
        sr.train()
        arrs["sr_roi"] = npy(sr(("p", roi)))
        # training noises: input only, output only, both; each call after torch.manual_seed(seed)
        for tag, (ni, no), seed in (("in", (0.3, 0.0), 101), ("out", (0.0, 0.25), 102), ("both", (0.2, 0.15), 103)):
            sr.input_noise, sr.output_noise = ni, no
            torch.manual_seed(seed)
            arrs["sr_noise_" + tag] = npy(sr(("p", roi)))
        arrs["noise_levels"] = np.array([[0.3, 0.0], [0.0, 0.25], [0.2, 0.15]])
        arrs["noise_seeds"] = np.array([101, 102, 103])
        sr.input_noise = sr.output_noise = 0
        sr.eval()
    save("g20_sr_options.npz", **arrs)


def g21_run_network():
    """run_network (train_utils.py:15-64) alone: g08's fine model (same seed -> same weights and planes; checked against g08's arrays) on 40
    rays x 32 coarse depths, chunked (chunksize 100 of 1280 points, so the loop and the final cat run), identity encodings for points
    and directions like the planes model -> raw radiance field [40, 32, 4]."""
    R, Rv = 32, 8
    sid, mc, mf, planes, box = build_models(R, Rv, 0.5, seed=8)
    g8 = np.load(os.path.join(HERE, "g08_render.npz"))
    assert np.array_equal(g8["plane0"], npy(planes[models.get_plane_name(sid, 0)])), "g21 must rebuild the g08 scene"
    assert np.array_equal(g8["fine.fc_rgb.0.weight"], npy(mf.fc_rgb["0"].weight))
    H = W = 16
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    ro, rd = nh.get_ray_bundle(H, W, focal, torch.from_numpy(POSE))
    sel = torch.arange(3, 3 + 40 * 5, 5)
    ro, rd = ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]
    vd = rd / rd.norm(p=2, dim=-1, keepdim=True)
    near, far = 2.0 * torch.ones_like(ro[..., :1]), 6.0 * torch.ones_like(ro[..., :1])
    ray_batch = torch.cat((ro, rd, near, far, vd), -1)
    z = near * (1.0 - torch.linspace(0.0, 1.0, 32)) + far * torch.linspace(0.0, 1.0, 32)
    pts = ro[..., None, :] + rd[..., None, :] * z[..., :, None]
    mf.eval()
    with torch.no_grad():
        raw = tu.run_network(mf, pts, ray_batch, 100, tu.identity_encoding, tu.identity_encoding, sid)
    save("g21_run_network.npz", pts=npy(pts), ray_batch=npy(ray_batch), raw=npy(raw), chunksize=np.array(100))


def _g22_models(R, Rv, seed, n_planes=3, **over):
    """coarse + fine TwoDimPlanesModel like build_models, with constructor options overridden and n_planes position planes"""
    torch.manual_seed(seed)
    np.random.seed(seed)
    sid = models.get_scene_id("lego", 8, (R, Rv))
    sc = models.SceneCoupler([sid], planes_res="LR", num_pos_planes=n_planes, training_scenes=[sid])
    kw = dict(use_viewdirs=True, skip_connect_every=3, proj_combination="avg", viewdir_proj_combination="concat_pos", align_corners=True,
              scene_coupler=sc)
    kw.update(over)
    mc = models.TwoDimPlanesModel(num_planes_or_rot_mats=n_planes, **kw)
    mc.optional_no_grad = nh.null_with
    mf = models.TwoDimPlanesModel(num_planes_or_rot_mats=mc.rot_mats(), **kw)
    mf.optional_no_grad = nh.null_with
    planes = nn.ParameterDict([(models.get_plane_name(sid, d), models.create_plane(R if d < n_planes else Rv, 48, 0.5)) for d in range(n_planes + 1)])
    box = torch.tensor(BOX, dtype=torch.float64)
    for m in (mc, mf):
        m.planes_, m.plane_rank, m.generated_planes, m.downsampled_planes, m.coverages = planes, None, {}, {}, {}
        m.box_coords = {sid: box}
        m.set_cur_scene_id(sid)
    return sid, mc, mf, planes, box


def g22_model_options():
    """The options of TwoDimPlanesModel no shipped YAML sets (VERDICT r3 missing #4), each through the reference itself:
      align_false  grid_sample(align_corners=False) (models.py:303-309,320-326): forward + autograd gradients of planes and decoder
      planes5      five position planes with CoordProjector's random orthonormal frames (models.py:471-490; numpy generator seeded here):
                   the frames, forward, gradients, and an eval_nerf render (chunk size divided by 5/3, train_utils.py:229-230)
      noise        point_coords_noise (models.py:291-293) in training mode: the jitter drawn inside one model call (seeded), forward +
                   gradients; and run_one_iter_of_nerf(mode='train') with perturb + density noise over several ray chunks and network
                   batches -- every random tensor in the order the reference draws it
      sr_align_false  PlanesSR(align_corners=False): the bilinear residual F.interpolate(..., align_corners=False) (models.py:858-859)
      bicubic[_noalign], sr_bicubic[_noalign]  plane_interp='bicubic': grid_sample(mode='bicubic') in the planes model, F.interpolate(mode='bicubic')
                   as PlanesSR's residual
      rf_bound     EDSR(receptive_field_bound=8): a mix of 3 x 3 and 1 x 1 convolutions (models.py:793-798), alone and inside PlanesSR"""
    arrs = {}
    P = 157

    def points(seed):
        g = torch.Generator().manual_seed(seed)
        pts = torch.rand(P, 3, generator=g) * 8.4 - 4.2              # some outside the box: border clamping
        d = torch.randn(P, 3, generator=g)
        return torch.cat([pts, d / d.norm(dim=-1, keepdim=True)], -1)

    def fwd_and_grads(pre, m, planes, sid, x, n_planes, before_each_forward=lambda: None):
        sdp = dict(m.named_parameters())
        keys = ["density_dec.0.%d" % i for i in range(4)] + ["fc_alpha.0"] + ["rgb_dec.0.%d" % i for i in range(4)] + ["fc_rgb.0"]
        params = [sdp[k + sfx] for k in keys for sfx in (".weight", ".bias")]
        plist = [planes[models.get_plane_name(sid, d)] for d in range(n_planes + 1)]
        for t_ in params + plist:
            t_.requires_grad_(True)
            t_.grad = None
        gout = torch.randn(P, 4, generator=torch.Generator().manual_seed(2200))
        before_each_forward()
        out = m(x)
        (out * gout).sum().backward()
        arrs[pre + "x"], arrs[pre + "out"], arrs[pre + "gout"] = npy(x), npy(out), npy(gout)
        arrs[pre + "gnat"] = np.concatenate([npy(t_.grad).reshape(-1) for t_ in params])
        for d in range(n_planes + 1):
            arrs[pre + "plane%d" % d] = npy(plist[d])
            arrs[pre + "gplane%d" % d] = npy(plist[d].grad)
        arrs.update(state_arrays(pre + "sd.", m))

    # ---- align_corners=False ---------------------------------------------------------------------------------------------
    sid, mc, mf, planes, box = _g22_models(9, 5, 221, align_corners=False, dec_channels=64)
    mc.eval()
    fwd_and_grads("align_false.", mc, planes, sid, points(2210), 3)

    # ---- plane_interp='bicubic' (config/TrainModels.yml:72 lists it beside 'bilinear'), with and without align_corners ---------------------
    for tag, align in (("bicubic", True), ("bicubic_noalign", False)):
        sid, mc, mf, planes, box = _g22_models(9, 5, 224 + int(align), plane_interp="bicubic", align_corners=align, dec_channels=64)
        mc.eval()
        fwd_and_grads(tag + ".", mc, planes, sid, points(2240 + int(align)), 3)

    # ---- five position planes ------------------------------------------------------------------------------------------------
    sid, mc, mf, planes, box = _g22_models(9, 5, 222, n_planes=5, dec_channels=64)
    mc.eval(); mf.eval()
    for d in range(5):
        arrs["planes5.rot%d" % d] = npy(mc.rot_mats()[d])
    arrs["planes5.np_seed"] = np.array(222)
    with torch.no_grad():
        x = points(2221)
        for mm in (mc, mf):                        # densities spread so that compositing is exercised (as in g18)
            s_ = 1.0 / float(mm(x)[:, 3].std())
            mm.fc_alpha["0"].weight.mul_(s_)
            mm.fc_alpha["0"].bias.mul_(s_)
            mm.fc_alpha["0"].bias.add_(-float(mm(x)[:, 3].mean()) - 0.5)
    fwd_and_grads("planes5.", mc, planes, sid, points(2220), 5)
    arrs.update(state_arrays("planes5.render.coarse.", mc))
    arrs.update(state_arrays("planes5.render.fine.", mf))
    H = W = 8
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    ro, rd = nh.get_ray_bundle(H, W, focal, torch.from_numpy(POSE))
    cfg = make_cfg(mode_cfg(16, 16), mode_cfg(16, 16))
    with torch.no_grad():
        rgb_c, _, _, rgb_f, *_ = tu.eval_nerf(H, W, focal, mc, mf, ro, rd, cfg, scene_id=sid, scene_config=cfg.dataset["synt"])
    arrs["planes5.render.rgb_coarse"], arrs["planes5.render.rgb_fine"] = npy(rgb_c), npy(rgb_f)
    arrs["planes5.render.hwf"] = np.array([H, W, focal])

    # ---- point_coords_noise ----------------------------------------------------------------------------------------------------
    PCN = 0.75
    sid, mc, mf, planes, box = _g22_models(9, 5, 223, point_coords_noise=PCN)
    res = 9                                        # '(?<=PlRes)(\d)+(?=_)' of the scene id
    arrs["noise.point_coords_noise"], arrs["noise.std"] = np.array(PCN), np.array(PCN * 2 / (1 + res))
    mc.train()
    x = points(2230)
    fwd_and_grads("noise.", mc, planes, sid, x, 3, before_each_forward=lambda: (torch.manual_seed(2231), np.random.seed(2231)))
    torch.manual_seed(2231)
    arrs["noise.jitter"] = npy(torch.normal(mean=0, std=PCN * 2 / (1 + res), size=[P, 3]))
    # one training iteration: 40 rays in ray chunks of 16 (chunksize 16 * ... see below), network batches of `chunksize` points
    H = W = 8
    ro, rd = nh.get_ray_bundle(H, W, focal, torch.from_numpy(POSE))
    torch.manual_seed(2232)
    sel = torch.randperm(H * W)[:40]
    rays = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0)
    target = torch.rand(40, 3)
    nc, nf, std, chunk = 12, 20, 0.3, 16
    vt = mode_cfg(nc, nf, perturb=True, noise=std, chunk=chunk)
    cfg = make_cfg(vt, vt)
    mc.train(); mf.train()
    plist = [planes[models.get_plane_name(sid, d)] for d in range(4)]
    for t_ in list(mc.parameters()) + list(mf.parameters()) + plist:
        t_.grad = None
    torch.manual_seed(2233)
    np.random.seed(2233)
    rc, dc, ac, rf, df, af, *_ = tu.run_one_iter_of_nerf(H, W, focal, mc, mf, rays, cfg, scene_id=sid, mode="train", scene_config=cfg.dataset["synt"])
    loss = torch.nn.functional.mse_loss(rc, target) + torch.nn.functional.mse_loss(rf, target)
    loss.backward()
    arrs.update(state_arrays("noise.train.coarse.", mc))
    arrs.update(state_arrays("noise.train.fine.", mf))
    arrs["noise.train.rays"], arrs["noise.train.target"] = npy(rays), npy(target)
    arrs["noise.train.params"] = np.array([nc, nf, std, chunk, H, W, focal])
    arrs["noise.train.rgb_coarse"], arrs["noise.train.rgb_fine"], arrs["noise.train.loss"] = npy(rc), npy(rf), np.array(float(loss))
    for d in range(4):
        arrs["noise.train.grad_plane%d" % d] = npy(plist[d].grad)
    # ---- PlanesSR with align_corners=False (models.py:858-859: assign_SR_model copies the planes model's flag) ------------------------
    # the network of g09 / g14 (same seed -> same weights and LR plane): full plane in evaluation mode, then the ROI and the full plane in
    # training mode with the gradients of the weights and of the (non-detached) LR plane.  Only the bilinear residual differs from g14.
    torch.manual_seed(9)
    C, hidden, nblocks, sf, R = 6, 16, 2, 4, 20
    sr = models.PlanesSR(models.EDSR, sf, C, C, CfgNode({"model": {"hidden_size": hidden, "n_blocks": nblocks}}), "bilinear")
    sr.align_corners = False
    with torch.no_grad():
        for p in sr.parameters():
            p.mul_(10.0)
    lr = torch.randn(1, C, R, R) * 0.5
    g9 = np.load(os.path.join(HERE, "g09_edsr.npz"))
    assert np.array_equal(g9["lr"], npy(lr)), "the align_corners=False case must rebuild the g09 network"
    sr.eval()
    sr.set_LR_plane(lr, id="p", save_interpolated=False)
    with torch.no_grad():
        arrs["sr_align_false.full"] = npy(sr("p")).copy()
    sr.train()
    roi = torch.tensor([[-0.35, -0.6], [0.2, 0.15]])
    arrs["sr_align_false.roi"] = npy(roi)
    for tag, arg in (("roi", ("p", roi)), ("full", "p")):
        lrp = nn.Parameter(lr.clone())
        sr.clear_SR_planes(all_planes=True)
        sr.set_LR_plane(lrp, id="p", save_interpolated=False)
        sr.zero_grad(set_to_none=True)
        out = sr(arg)
        Gp = torch.randn(out.shape, generator=torch.Generator().manual_seed(2290))
        valid = ~torch.isnan(out)
        (torch.where(valid, out, torch.zeros_like(out)) * Gp).sum().backward()
        gw = np.concatenate([npy(p.grad).reshape(-1) for _, p in sr.inner_model.named_parameters()]).astype(np.float32)
        arrs.update({"sr_align_false.%s_out" % tag: npy(out).copy(), "sr_align_false.%s_gout" % tag: npy(Gp), "sr_align_false.%s_gw" % tag: gw,
                     "sr_align_false.%s_glr" % tag: npy(lrp.grad).copy()})
        sr.clear_SR_planes(all_planes=True)
    # ---- PlanesSR with plane_interp='bicubic' (the residual is F.interpolate(mode='bicubic'), models.py:858-859), both align_corners ---------
    for tag, align in (("sr_bicubic", True), ("sr_bicubic_noalign", False)):
        torch.manual_seed(9)
        C, hidden, nblocks, sf, R = 6, 16, 2, 4, 20
        sr = models.PlanesSR(models.EDSR, sf, C, C, CfgNode({"model": {"hidden_size": hidden, "n_blocks": nblocks}}), "bicubic")
        sr.align_corners = align
        with torch.no_grad():
            for p in sr.parameters():
                p.mul_(10.0)
        lr = torch.randn(1, C, R, R) * 0.5
        assert np.array_equal(np.load(os.path.join(HERE, "g09_edsr.npz"))["lr"], npy(lr)), "the bicubic cases must rebuild the g09 network"
        sr.eval()
        sr.set_LR_plane(lr, id="p", save_interpolated=False)
        with torch.no_grad():
            arrs[tag + ".full"] = npy(sr("p")).copy()
        sr.train()
        roi = torch.tensor([[-0.35, -0.6], [0.2, 0.15]])
        arrs[tag + ".roi"] = npy(roi)
        lrp = nn.Parameter(lr.clone())
        sr.clear_SR_planes(all_planes=True)
        sr.set_LR_plane(lrp, id="p", save_interpolated=False)
        sr.zero_grad(set_to_none=True)
        out = sr(("p", roi))
        Gp = torch.randn(out.shape, generator=torch.Generator().manual_seed(2293))
        valid = ~torch.isnan(out)
        (torch.where(valid, out, torch.zeros_like(out)) * Gp).sum().backward()
        arrs.update({tag + ".roi_out": npy(out).copy(), tag + ".roi_gout": npy(Gp), tag + ".roi_glr": npy(lrp.grad).copy(),
                     tag + ".roi_gw": np.concatenate([npy(p.grad).reshape(-1) for _, p in sr.inner_model.named_parameters()]).astype(np.float32)})

    # ---- EDSR with receptive_field_bound (models.py:789-822: layers beyond the bound fall back to 1 x 1 convolutions) --------------------
    # bound 8 with 2 blocks and x4: conv_input 3x3, block 0 3x3, block 1 1x1, conv_mid 1x1, first up-scaling conv 1x1, second 3x3 (its
    # receptive-field increment is halved), conv_output 1x1
    torch.manual_seed(229)
    C, hidden, nblocks, sf, R = 6, 16, 2, 4, 20
    sr = models.PlanesSR(models.EDSR, sf, C, C, CfgNode({"model": {"hidden_size": hidden, "n_blocks": nblocks, "receptive_field_bound": 8}}), "bilinear")
    sr.align_corners = True
    with torch.no_grad():
        for p in sr.parameters():
            p.mul_(10.0)
    ks = [int(m.kernel_size[0]) for m in sr.inner_model.modules() if isinstance(m, nn.Conv2d)]
    assert ks == [3, 3, 3, 1, 1, 1, 1, 3, 1], ks
    arrs["rf_bound.cfg"] = np.array([C, hidden, nblocks, sf, R, 8, sr.inner_model.required_padding, sr.HR_overpadding])
    arrs["rf_bound.kernel_sizes"] = np.array(ks)
    arrs.update({"rf_bound.sd." + k: npy(v).copy() for k, v in sr.state_dict().items()})
    lr = torch.randn(1, C, R, R) * 0.5
    arrs["rf_bound.lr"] = npy(lr)
    sr.train()
    x = torch.randn(1, C, 30, 26, requires_grad=True)
    out = sr.inner_model(x)
    G = torch.randn(out.shape, generator=torch.Generator().manual_seed(2291))
    sr.zero_grad(set_to_none=True)
    (out * G).sum().backward()
    blob_grad = lambda: np.concatenate([npy(p.grad).reshape(-1) for _, p in sr.inner_model.named_parameters()]).astype(np.float32)
    arrs.update({"rf_bound.edsr_in": npy(x).copy(), "rf_bound.edsr_out": npy(out).copy(), "rf_bound.edsr_gout": npy(G), "rf_bound.edsr_gw": blob_grad(),
                 "rf_bound.edsr_gin": npy(x.grad).copy()})
    sr.eval()
    sr.set_LR_plane(lr, id="p", save_interpolated=False)
    with torch.no_grad():
        arrs["rf_bound.full"] = npy(sr("p")).copy()
    sr.train()
    roi = torch.tensor([[-0.35, -0.6], [0.2, 0.15]])
    arrs["rf_bound.roi"] = npy(roi)
    lrp = nn.Parameter(lr.clone())
    sr.clear_SR_planes(all_planes=True)
    sr.set_LR_plane(lrp, id="p", save_interpolated=False)
    sr.zero_grad(set_to_none=True)
    out = sr(("p", roi))
    Gp = torch.randn(out.shape, generator=torch.Generator().manual_seed(2292))
    valid = ~torch.isnan(out)
    (torch.where(valid, out, torch.zeros_like(out)) * Gp).sum().backward()
    arrs.update({"rf_bound.roi_out": npy(out).copy(), "rf_bound.roi_gout": npy(Gp), "rf_bound.roi_gw": blob_grad(), "rf_bound.roi_glr": npy(lrp.grad).copy()})
    save("g22_model_options.npz", **arrs)


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["g01", "g02", "g03", "g04", "g05", "g06", "g07", "g08", "g09", "g10", "g11", "g12", "g13", "g14", "g15", "g16", "g17", "g18", "g20", "g21", "g22"]
    for name, fn in list(globals().items()):
        if callable(fn) and name[:3] in which and name.startswith("g"):
            fn()
