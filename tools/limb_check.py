"""Compare the bf16-limb fused render pass (render3.hip) with the f32-MFMA pass (render2.hip) on the outputs of one launch."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvsr_amd as hip
from bench import make_synthetic_scene

capi = hip.capi
dev = "cuda:0"
mc, mf, sid, pose = make_synthetic_scene(dev, plane_res=200, view_res=32, seed=3)
N, S = 20011, 37
H, W = 150, 160
focal = 0.5 * W / np.tan(0.5 * 0.6911112)
ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, pose)
rays = hip.train_utils.pack_rays(ro, rd, 2.0, 6.0)[:N].contiguous()
rng = np.random.default_rng(17)
z = torch.as_tensor(np.sort(rng.uniform(2, 6, (N, S)).astype(np.float32), -1), device=dev)
sc, keep = mf.native_scene()
packed = mf.packed_decoder()
outs = {}
for mode in ("f32", "bf16x3", "f16x2"):
    capi.set_decoder_arithmetic(mode)
    o = dict(rgb=torch.full((N, 3), -7.0, device=dev), disp=torch.full((N,), -7.0, device=dev), acc=torch.full((N,), -7.0, device=dev),
             w=torch.full((N, S), -7.0, device=dev), depth=torch.full((N,), -7.0, device=dev), raw=torch.full((N, S, 4), -7.0, device=dev))
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        capi.call("nvsr_render_pass_ex", C.byref(sc), capi.ptr(packed), N, S, capi.ptr(rays), capi.ptr(z), None, 1,
                  capi.ptr(o["rgb"]), capi.ptr(o["disp"]), capi.ptr(o["acc"]), capi.ptr(o["w"]), capi.ptr(o["depth"]), capi.ptr(o["raw"]), capi.stream())
        torch.cuda.synchronize(); dt = time.time() - t0
    outs[mode] = {k: v.cpu().numpy().astype(np.float64) for k, v in o.items()}
    print(mode, "%.2f ms" % (dt * 1e3), "raw range", outs[mode]["raw"].min(), outs[mode]["raw"].max())
ref = outs["f32"]
for mode in ("bf16x3", "f16x2"):
    for k in ("raw", "rgb", "acc", "w", "depth"):
        d = np.abs(outs[mode][k] - ref[k])
        print("%-7s %-5s max|d| %.3e  mean|d| %.3e   (max|ref| %.3e)" % (mode, k, d.max(), d.mean(), np.abs(ref[k]).max()))
