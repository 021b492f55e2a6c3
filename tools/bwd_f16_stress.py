"""The gate-driven backward (planes only, no record) in every arithmetic on the SAME gates (those of an exact-f32 forward): relative L2 error of
the plane gradients against the exact-f32 backward.  dL/draw magnitudes span 8 decades from ray to ray (the f16-limb kernel scales every wave
tile by its own power of two).  usage: bwd_f16_stress.py [seed]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
import nvsr_amd
from bench import make_synthetic_scene
capi = nvsr_amd.capi
lib = capi.lib()
dev = torch.device("cuda", 0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = {"bf16x3": 0.0, "f16x2": 0.0}
for it in range(12):
    pr = int(rng.choice([17, 40, 64, 200]))
    mc, mf, sid, pose = make_synthetic_scene(dev, plane_res=pr, view_res=int(rng.choice([8, 32])), seed=int(rng.integers(1 << 20)), channels_last=True)
    N, S = int(rng.integers(50, 5000)), int(rng.choice([31, 32, 64, 65, 128]))
    H = W = 80
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    ro, rd = nvsr_amd.nerf_helpers.get_ray_bundle(H, W, focal, pose)
    rays = nvsr_amd.train_utils.pack_rays(ro, rd, 2.0, 6.0)[torch.from_numpy(rng.integers(0, H * W, N)).to(dev)].contiguous()
    z = torch.sort(torch.rand(N, S, device=dev) * 4 + 2, -1).values.contiguous()
    mag = torch.from_numpy(np.exp(rng.uniform(np.log(1e-6), np.log(1e2), (N, 1, 1))).astype(np.float32)).to(dev)
    g_raw = (torch.randn(N, S, 4, device=dev) * mag).contiguous()
    sc, keep = mf.native_scene()
    raw = torch.empty((N, S, 4), device=dev)
    gates = torch.zeros(N * S * 32, dtype=torch.int32, device=dev)
    capi.call("nvsr_decode_rays_arith", C.byref(sc), capi.ptr(mf.packed_decoder()), N, S, capi.ptr(rays), capi.ptr(z), capi.ptr(raw), capi.ptr(gates), None,
              capi.ARITHMETIC["f32"], capi.stream())
    res = {}
    for mode in ("f32", "bf16x3", "f16x2"):
        gpl = [torch.zeros_like(k) for k in keep]
        gptrs = (C.c_void_p * 4)(*[t.data_ptr() for t in gpl])
        vws = torch.zeros(lib.nvsr_view_grad_workspace_floats(N, S), device=dev)
        capi.call("nvsr_render_pass_backward_gates_arith", C.byref(sc), capi.ptr(mf.packed_decoder()), capi.ptr(mf.packed_decoder_bwd()), N, S, capi.ptr(rays),
                  capi.ptr(z), capi.ptr(g_raw), capi.ptr(gates), gptrs, capi.ptr(vws), None, capi.ARITHMETIC[mode], capi.stream())
        torch.cuda.synchronize()
        res[mode] = torch.cat([g.reshape(-1).double() for g in gpl])
    line = "N %4d S %3d planes %3d:" % (N, S, pr)
    for mode in ("bf16x3", "f16x2"):
        assert torch.isfinite(res[mode]).all(), mode
        rel = float((res[mode] - res["f32"]).norm() / res["f32"].norm())
        worst[mode] = max(worst[mode], rel)
        line += "  %s rel L2 %.2e" % (mode, rel)
    print(line)
print("worst relative L2 error against the exact-f32 backward:", worst, "(float atomics: the f32 kernel against itself differs by ~1e-7)")
assert worst["f16x2"] < 2e-5
