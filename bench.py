#!/usr/bin/env python3
"""Benchmark of the rendering hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json configs[1]): one "step" renders one 800x800 view (640 000 rays) of a synthetic Lego-like scene with
64 coarse + 128 fine samples per ray (256 decoder evaluations per ray) through the tri-plane decoder, planes 3 x 800^2 x 48 +
32^2 x 48, everything resident in HBM before the timed region.  Weak scaling: every rank renders its own view of the same
(replicated) scene; there is no data-path collective.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_EVAL = 259072          # 129 536 MAC per decoded point
GATHER_BYTES_PER_EVAL = 3072    # 16 texels x 48 ch x 4 B
PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32
CAMERA_ANGLE_X = 0.6911112


class Opt:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def render_options(nc, nf, perturb=False, noise=0.0, white=False):
    m = Opt(chunksize=131072, perturb=perturb, num_coarse=nc, num_fine=nf, white_background=white,
            radiance_field_noise_std=noise, lindisp=False)
    return Opt(nerf=Opt(use_viewdirs=True, train=m, validation=m)), Opt(near=2.0, far=6.0, no_ndc=True)


def pose_spherical(theta, phi, radius):
    """Blender-style camera-to-world (the reference's load_blender.pose_spherical, load_blender.py:34-39)"""
    t = np.eye(4); t[2, 3] = radius
    p = np.deg2rad(phi)
    rp = np.array([[1, 0, 0, 0], [0, np.cos(p), -np.sin(p), 0], [0, np.sin(p), np.cos(p), 0], [0, 0, 0, 1.0]])
    th = np.deg2rad(theta)
    rt = np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1.0]])
    c2w = rt @ rp @ t
    c2w = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1.0]]) @ c2w
    return c2w.astype(np.float32)


def make_synthetic_scene(dev, plane_res=800, view_res=32, seed=0, theta=30.0):
    """Random-init decoder pair + random planes (no dataset / checkpoint is reachable), calibrated so that the density is
    neither empty nor saturated (SURVEY.md 7 'degenerate synthetic scenes')."""
    import nvsr_amd
    M = nvsr_amd.models
    torch.manual_seed(seed)
    sid = M.get_scene_id("lego", 1, (plane_res, view_res))
    kw = dict(use_viewdirs=True, skip_connect_every=3, proj_combination="avg", viewdir_proj_combination="concat_pos", align_corners=True)
    mc = M.TwoDimPlanesModel(**kw)
    mf = M.TwoDimPlanesModel(num_planes_or_rot_mats=mc.rot_mats(), **kw)
    # band-limited random planes (random res/8 grids, bilinearly up-sampled): trained feature planes are smooth at texel scale,
    # white noise at 800^2 would make the radiance field a chaotic function of depth.  Values do not change the work done.
    def smooth_plane(res):
        src = max(res // 8, 4)
        low = 0.7 * torch.randn(1, 48, src, src, device=dev)
        return torch.nn.Parameter(torch.nn.functional.interpolate(low, size=(res, res), mode="bilinear", align_corners=True).contiguous())
    mc, mf = mc.to(dev), mf.to(dev)
    planes = torch.nn.ParameterDict({M.get_plane_name(sid, d): smooth_plane(plane_res if d < 3 else view_res) for d in range(4)})
    box = torch.tensor([[-4.0, -4, -4, -np.pi, -np.pi / 2], [4, 4, 4, np.pi, np.pi / 2]], dtype=torch.float64)
    g = torch.Generator().manual_seed(seed + 1)
    pts = torch.rand(8192, 3, generator=g) * 6 - 3
    d = torch.randn(8192, 3, generator=g)
    x = torch.cat([pts, d / d.norm(dim=-1, keepdim=True)], -1).to(dev)
    for m in (mc, mf):
        m.planes_, m.box_coords = planes, {sid: box}
        m.set_cur_scene_id(sid)
        m.eval()
        with torch.no_grad():
            raw = m(x)[:, 3]
            scale = 1.0 / float(raw.std())
            m.fc_alpha["0"].weight.mul_(scale)
            m.fc_alpha["0"].bias.mul_(scale)
            raw = m(x)[:, 3]
            m.fc_alpha["0"].bias.add_(-float(raw.mean()) - 0.5)
    pose = torch.from_numpy(pose_spherical(theta, -30.0, 4.0)).to(dev)
    return mc, mf, sid, pose


def time_fine_pass_kernel(nvsr_amd, mf, rays, z_fine, reps=3):
    """Average duration of the dominant kernel (fused fine render pass, S = Nc+Nf) measured with HIP events on the launch stream."""
    import ctypes as C
    capi = nvsr_amd.capi
    N, S = z_fine.shape
    dev = rays.device
    rgb = torch.empty((N, 3), device=dev); disp = torch.empty(N, device=dev); acc = torch.empty(N, device=dev)
    sc, keep = mf.native_scene()
    packed = mf.packed_decoder()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        capi.call("nvsr_render_pass", C.byref(sc), capi.ptr(packed), N, S, capi.ptr(rays), capi.ptr(z_fine), None, 0, capi.ptr(rgb),
                  capi.ptr(disp), capi.ptr(acc), None, None, capi.stream())
        b.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in ev])) * 1e-3


def cpu_baseline(nvsr_amd, mc, mf, sid, rays, rgb_fine_gpu, budget_s=15.0):
    """The oracle (plain-C port of the reference algorithm, fp32, OpenMP over all host cores) on a bounded sample of the same
    rays; also the PSNR of the GPU pixels against the (double-accumulating) checker on that sample."""
    from oracle.oracle import Oracle, decoder_blob
    fast, chk = Oracle(f32=True), Oracle(f32=False)
    planes = [mc.planes_[nvsr_amd.models.get_plane_name(sid, d)].detach().cpu().numpy() for d in range(4)]
    box = mc.box_coords[sid].numpy()
    sdc = {k: v.detach().cpu().numpy() for k, v in mc.state_dict().items()}
    sdf = {k: v.detach().cpu().numpy() for k, v in mf.state_dict().items()}
    N = rays.shape[0]
    rng = np.random.default_rng(0)
    ids = np.sort(rng.choice(N, size=min(N, 262144), replace=False))
    rays_np = rays[torch.from_numpy(ids).to(rays.device)].cpu().numpy()

    def run(o, n):
        sc = o.scene(planes, box)
        t0 = time.perf_counter()
        out = o.render_rays(sc, o.decoder(decoder_blob(sdc)), o.decoder(decoder_blob(sdf)), rays_np[:n], 64, 128)
        return time.perf_counter() - t0, out

    run(fast, 256)                                  # warm-up (OpenMP team, page faults)
    t_probe, _ = run(fast, 2048)
    n = int(min(len(ids), max(2048, 2048 * budget_s / max(t_probe, 1e-3))))
    t, _ = run(fast, n)
    n_chk = min(n, 2048)
    _, ref = run(chk, n_chk)
    gpu = rgb_fine_gpu[torch.from_numpy(ids[:n_chk]).to(rgb_fine_gpu.device)].cpu().numpy()
    mse = float(np.mean((gpu.astype(np.float64) - ref["rgb_fine"]) ** 2))
    psnr = 200.0 if mse == 0 else -10.0 * np.log10(mse)
    cores = os.cpu_count() or 1
    return {"value": n / t, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": "%d rays of the same 800x800 / 64+128 / planes 800^2 frame, %.1f s, C oracle fp32 -Ofast OpenMP (%d threads)" % (n, t, cores),
            "evals_per_s": n * 256 / t}, psnr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--res", type=int, default=800, help="image side (default 800 = BASELINE config)")
    ap.add_argument("--plane-res", type=int, default=800)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import nvsr_amd
    nvsr_amd.capi.lib()  # fail loudly if the HIP library is not built

    H = W = args.res
    focal = 0.5 * W / np.tan(0.5 * CAMERA_ANGLE_X)
    # same (replicated) scene on every rank, a different view per rank
    mc, mf, sid, pose = make_synthetic_scene(dev, args.plane_res, 32, seed=0, theta=30.0 + 45.0 * rank)
    opts, scfg = render_options(64, 128)
    ro, rd = nvsr_amd.nerf_helpers.get_ray_bundle(H, W, focal, pose)

    def step():
        r, d = nvsr_amd.nerf_helpers.get_ray_bundle(H, W, focal, pose)   # ray generation is part of the path
        return nvsr_amd.train_utils.eval_nerf(H, W, focal, mc, mf, r, d, opts, scene_id=sid, scene_config=scfg)

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    rays_per_step = H * W
    value = world * rays_per_step * args.steps / elapsed
    result = {
        "metric": "rendered rays/sec (64+128 samples) at 800x800 Lego-like view",
        "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "Blender-'lego'-like %dx%d view, 64 coarse + 128 fine samples, tri-plane decoder (3x%d^2x48 + 32^2x48 planes, "
                               "4+4x128 MLP), 1 view per GPU per step" % (H, W, args.plane_res),
                   "rays_per_step_per_gpu": rays_per_step, "decoder_evals_per_ray": 256, "parallelism": "rays sharded by view, no collective"},
        "decoder_evals_per_s_per_gpu": value * 256 / world,
    }
    if rank == 0:
        # dominant kernel: fused fine render pass (192 of the 256 evaluations per ray)
        rays = nvsr_amd.train_utils.pack_rays(ro, rd, 2.0, 6.0)
        N = rays.shape[0]
        import ctypes as C
        capi = nvsr_amd.capi
        ws = torch.empty(capi.lib().nvsr_render_workspace_floats(N, 64, 128), device=dev)
        bufs = [torch.empty((N, 3), device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev),
                torch.empty((N, 3), device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev)]
        sc, keep = mc.native_scene()
        capi.call("nvsr_render_rays", C.byref(sc), capi.ptr(mc.packed_decoder()), capi.ptr(mf.packed_decoder()), N, 64, 128, capi.ptr(rays),
                  0, 0, None, None, None, None, *[capi.ptr(b) for b in bufs], capi.ptr(ws), capi.stream())
        z_fine = ws[2 * N * 64:].view(N, 192)
        dt = time_fine_pass_kernel(nvsr_amd, mf, rays, z_fine)
        flops = FLOP_PER_EVAL * N * 192
        achieved = flops / dt / 1e12
        # HBM-side traffic of that launch: PMC counters cannot be read from inside this process; the value comes from the
        # committed rocprofv3 --pmc passes of this same command (profiles/pmc_latest.json), corrected as the guide prescribes
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc) and H == 800 and args.plane_res == 800:
            traffic = json.load(open(pmc)).get("traffic_bytes")
        result["roofline"] = {"kernel": "render_pass_kernel (fine pass, S=192)", "bound": "mfma", "achieved": achieved,
                              "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                              "kernel_ms": dt * 1e3, "algorithmic_flop_per_launch": flops,
                              "algorithmic_gather_bytes_per_launch": GATHER_BYTES_PER_EVAL * N * 192}
        if world == 1 and not args.no_cpu_baseline:
            cb, psnr = cpu_baseline(nvsr_amd, mc, mf, sid, rays, bufs[3])
            result["cpu_baseline"] = cb
            result["psnr_vs_oracle_db"] = psnr
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
