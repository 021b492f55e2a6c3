"""pair forward kernel alone, S = 64 and 128 (NVSR_HIP_LIB selects a variant): prints min ms"""
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
out = []
for S in (64, 128):
    N = 4096
    z = torch.sort(torch.rand(N, S, device=dev, generator=g) * 4 + 2, -1)[0].contiguous()
    raw = torch.empty(N, S, 4, device=dev); gates = torch.empty(N, S, 32, dtype=torch.int32, device=dev)
    ts = []
    for i in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        st = lib.nvsr_decode_rays_pair_launch(C.byref(sc), capi.ptr(packed), C.c_int64(N), C.c_int(S), capi.ptr(rays), capi.ptr(z), capi.ptr(raw), capi.ptr(gates), capi.stream())
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    out.append("S=%d %.3f" % (S, min(ts[2:])))
print("  ".join(out))
