"""times the training forward kernels alone, one-tile (decode_limb.hip) against tile-pair (decode_pair.hip), f16x2 with gate words:
   4096 rays x {64, 128} sorted random depths, planes 200^2.     python tools/fwd_pair_time.py"""
import sys, os, ctypes as C; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nvsr_amd
from bench import make_synthetic_scene
dev = torch.device("cuda", 0); capi = nvsr_amd.capi; lib = capi.lib()
mc, mf, sid, pose = make_synthetic_scene(dev, 200, 32, seed=0)
H = W = 800; focal = 0.5 * W / np.tan(0.5 * 0.6911112)
g = torch.Generator(device=dev).manual_seed(1)
sel = torch.randint(0, H, (4096, 2), device=dev, generator=g)
ro, rd = nvsr_amd.training.get_ray_bundle_at(H, W, focal, pose, sel)
rays = nvsr_amd.train_utils.pack_rays(ro, rd, 2.0, 6.0)
sc, keep = mf.native_scene()
packed = mf.packed_decoder()
for S in (64, 128):
    N = 4096
    z = torch.sort(torch.rand(N, S, device=dev, generator=g) * 4 + 2, -1)[0].contiguous()
    raw = torch.empty(N, S, 4, device=dev); gates = torch.empty(N, S, 32, dtype=torch.int32, device=dev)
    for name in ("limb", "pair", "limb", "pair"):
        ts = []
        for i in range(8):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            if name == "limb":
                st = lib.nvsr_decode_rays_limb_launch(C.c_int(2), C.byref(sc), capi.ptr(packed), C.c_int64(N), C.c_int(S), capi.ptr(rays), capi.ptr(z), capi.ptr(raw), capi.ptr(gates), None, capi.stream())
            else:
                st = lib.nvsr_decode_rays_pair_launch(C.byref(sc), capi.ptr(packed), C.c_int64(N), C.c_int(S), capi.ptr(rays), capi.ptr(z), capi.ptr(raw), capi.ptr(gates), capi.stream())
            b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
            assert st == 0
        t = min(ts[2:])
        print("S=%d %s: %.3f ms = %.1f TFLOP/s of f32 work (min of 6; median %.3f)" % (S, name, t, 259072.0 * N * S / t / 1e9, float(np.median(ts[2:]))), flush=True)
