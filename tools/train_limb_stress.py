"""randomised check of the limb training kernels (decode_limb.hip, render_bwd_limb.hip, decoder_wgrad limb) against the exact-f32 kernels:
24 random (rays, samples, plane shapes) configurations, forward raw / gates / record, backward plane gradients / record, weight gradients.
usage: train_limb_stress.py [seed]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
import nvsr_amd
from bench import make_synthetic_scene
capi = nvsr_amd.capi
lib = capi.lib()
dev = torch.device("cuda", 0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = dict(raw=0.0, rec=0.0, gpl=0.0, wg=0.0, flips=0.0)
for it in range(24):
    pr = int(rng.choice([8, 17, 40, 64, 200]))
    mc, mf, sid, pose = make_synthetic_scene(dev, plane_res=pr, view_res=int(rng.choice([4, 8, 32])), seed=int(rng.integers(1 << 20)),
                                             channels_last=bool(rng.integers(2)))
    N, S = int(rng.integers(1, 5000)), int(rng.choice([1, 2, 31, 32, 33, 64, 65, 96, 128, 150]))
    H = W = 80
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    ro, rd = nvsr_amd.nerf_helpers.get_ray_bundle(H, W, focal, pose)
    rays = nvsr_amd.train_utils.pack_rays(ro, rd, 2.0, 6.0)[torch.from_numpy(rng.integers(0, H * W, N)).to(dev)].contiguous()
    z = torch.sort(torch.rand(N, S, device=dev) * 4 + 2, -1).values.contiguous()
    g_raw = (torch.randn(N, S, 4, device=dev) * 1e-2).contiguous()
    sc, keep = mf.native_scene()
    nrec = lib.nvsr_decoder_record_floats(N, S)
    res = {}
    for mode in ("f32", "bf16x3"):
        capi.set_decoder_arithmetic(mode)
        raw = torch.full((N, S, 4), -7.0, device=dev)
        gates = torch.zeros(N * S * 32, dtype=torch.int32, device=dev)
        rec = torch.full((nrec,), float("nan"), device=dev)
        capi.call("nvsr_decode_rays_ex", C.byref(sc), capi.ptr(mf.packed_decoder()), N, S, capi.ptr(rays), capi.ptr(z), capi.ptr(raw), capi.ptr(gates),
                  capi.ptr(rec), capi.stream())
        gpl = [torch.zeros_like(k) for k in keep]
        gptrs = (C.c_void_p * 4)(*[t.data_ptr() for t in gpl])
        vws = torch.zeros(lib.nvsr_view_grad_workspace_floats(N, S), device=dev)
        capi.call("nvsr_render_pass_backward_gates", C.byref(sc), capi.ptr(mf.packed_decoder()), capi.ptr(mf.packed_decoder_bwd()), N, S, capi.ptr(rays),
                  capi.ptr(z), capi.ptr(g_raw), capi.ptr(gates), gptrs, capi.ptr(vws), capi.ptr(rec), capi.stream())
        wg = torch.zeros(capi.DECODER_NATURAL_FLOATS, device=dev)
        capi.call("nvsr_decoder_weight_grad", N, S, capi.ptr(rec), capi.ptr(wg), capi.stream())
        torch.cuda.synchronize()
        res[mode] = dict(raw=raw.double().cpu().numpy(), gates=gates.cpu().numpy(), rec=rec.double().cpu().numpy(),
                         gpl=[t.double().cpu().numpy() for t in gpl], wg=wg.double().cpu().numpy())
    capi.set_decoder_arithmetic("bf16x3")
    a, b = res["f32"], res["bf16x3"]
    assert not (b["raw"] == -7.0).any()
    assert np.array_equal(np.isnan(a["rec"]), np.isnan(b["rec"])), "record rows written differ"
    flips = int(np.unpackbits((a["gates"] ^ b["gates"]).view(np.uint8)).sum())
    e = dict(raw=np.abs(a["raw"] - b["raw"]).max() / max(1.0, np.abs(a["raw"]).max()),
             rec=np.linalg.norm(np.nan_to_num(a["rec"] - b["rec"])) / max(1e-30, np.linalg.norm(np.nan_to_num(a["rec"]))),
             gpl=max(np.linalg.norm(x - y) / max(1e-30, np.linalg.norm(x)) for x, y in zip(a["gpl"], b["gpl"])),
             wg=np.linalg.norm(a["wg"] - b["wg"]) / max(1e-30, np.linalg.norm(a["wg"])), flips=flips / (N * S * 1024.0))
    ok = e["raw"] <= 1e-5 and e["flips"] <= 1e-3 + 2.0 / (N * S * 1024) and all(e[k] <= (2e-5 if flips == 0 else 3e-3) for k in ("rec", "gpl", "wg"))
    print("%2d  N=%4d S=%3d planes %3d^2 %s: raw %.1e rec %.1e planes %.1e weights %.1e gate flips %d  %s"
          % (it, N, S, pr, "cl  " if nvsr_amd.models.is_native_layout(list(mf.planes_.values())[0]) else "nchw", e["raw"], e["rec"], e["gpl"], e["wg"], flips,
             "ok" if ok else "FAIL"), flush=True)
    assert ok
    for k in worst:
        worst[k] = max(worst[k], e[k])
print("all 24 configurations within tolerance; worst:", {k: float("%.2e" % v) for k, v in worst.items()})
