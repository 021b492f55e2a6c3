"""Randomised comparison of the bf16-limb render pass with the f32-MFMA pass: ray counts with partial workgroups, 1..192 samples, with and
without density noise / white background / optional outputs, repeated launches (a race on the weight ring or on the counted vector-memory
waits would show as large, launch-dependent differences)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvsr_amd as hip
from bench import make_synthetic_scene

capi = hip.capi
dev = "cuda:0"
mc, mf, sid, pose = make_synthetic_scene(dev, plane_res=200, view_res=32, seed=5)
H, W = 220, 230
focal = 0.5 * W / np.tan(0.5 * 0.6911112)
ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, pose)
rays_all = hip.train_utils.pack_rays(ro, rd, 2.0, 6.0).contiguous()
sc, keep = mf.native_scene()
packed = mf.packed_decoder()
rng = np.random.default_rng(1)
worst = {"bf16x3": 0.0, "f16x2": 0.0}
for trial in range(24):
    N = int(rng.integers(16384, 50000))
    S = int(rng.choice([1, 2, 3, 5, 17, 64, 129, 192]))
    use_noise, white, want = bool(rng.integers(2)), int(rng.integers(2)), bool(rng.integers(2))
    rays = rays_all[rng.permutation(rays_all.shape[0])[:N]].contiguous()
    z = torch.as_tensor(np.sort(rng.uniform(2, 6, (N, S)).astype(np.float32), -1), device=dev)
    noise = torch.as_tensor((rng.standard_normal((N, S)) * 0.3).astype(np.float32), device=dev) if use_noise else None
    res = {}
    for mode in ("f32", "bf16x3", "f16x2", "bf16x3"):
        capi.set_decoder_arithmetic(mode)
        o = dict(rgb=torch.full((N, 3), -7.0, device=dev), disp=torch.full((N,), -7.0, device=dev), acc=torch.full((N,), -7.0, device=dev),
                 w=torch.full((N, S), -7.0, device=dev), depth=torch.full((N,), -7.0, device=dev), raw=torch.full((N, S, 4), -7.0, device=dev))
        capi.call("nvsr_render_pass_ex", C.byref(sc), capi.ptr(packed), N, S, capi.ptr(rays), capi.ptr(z), capi.ptr(noise), white,
                  capi.ptr(o["rgb"]), capi.ptr(o["disp"]), capi.ptr(o["acc"]), capi.ptr(o["w"]) if (want or mode == "f32") else None,
                  capi.ptr(o["depth"]) if (want or mode == "f32") else None, capi.ptr(o["raw"]) if (want or mode == "f32") else None, capi.stream())
        torch.cuda.synchronize()
        if mode in res:                                   # second launch of the same mode: bit-identical
            for k in o:
                assert torch.equal(o[k].view(torch.int32), res[mode][k].view(torch.int32)), (trial, mode, k)     # bits: disp may be NaN
        res[mode] = o
    # the last interval is 1e10 long: alpha of the last sample is a step function of the sign of its sigma -- rays whose last sigma is within
    # the arithmetic's noise of zero are excluded (as in tests/test_hip_parity.py)
    sig_last = res["f32"]["raw"][:, -1, 3] + (noise[:, -1] if use_noise else 0.0)
    for mode in ("bf16x3", "f16x2"):
        ok = sig_last.abs() > (1e-4 if mode == "bf16x3" else 5e-3)
        for k in ("rgb", "acc") + (("raw", "w") if want else ()):
            d = float((res[mode][k] - res["f32"][k])[ok].abs().max())
            scale = float(res["f32"][k].abs().max()) if k == "raw" else 1.0
            worst[mode] = max(worst[mode], d / max(scale, 1e-30))
            lim = ((2e-5 if k == "raw" else 1e-4) if mode == "bf16x3" else (1e-3 if k == "raw" else 1e-2))   # composited values amplify raw errors by sigma * dist
            assert d <= lim * max(scale, 1.0), (trial, N, S, mode, k, d)
    print("trial %2d N %5d S %3d noise %d white %d outputs %d ok" % (trial, N, S, use_noise, white, want))
print("worst relative difference:", worst)
