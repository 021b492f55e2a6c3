"""gate-driven backward kernel alone (f16x2, planes only, all four planes + view rows), 4096 rays x {64, 128} sorted random depths, planes 200^2.
   NVSR_HIP_LIB selects a variant.  Prints min ms of 8 launches.  BWD_ZERO_WEIGHTS=1: the decoder matrices are zeroed after the gates were made (same
   instruction stream and traffic, operands that toggle fewer bits: does the kernel run on a power-managed clock?)."""
import sys, os, ctypes as C; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nvsr_amd
from bench import make_synthetic_scene
dev = torch.device("cuda", 0); capi = nvsr_amd.capi; lib = capi.lib()
mc, mf, sid, pose = make_synthetic_scene(dev, 200, 32, seed=0, channels_last=True)
H = W = 800; focal = 0.5 * W / np.tan(0.5 * 0.6911112)
g = torch.Generator(device=dev).manual_seed(1)
N = 4096
sel = torch.randint(0, H, (N, 2), device=dev, generator=g)
ro, rd = nvsr_amd.training.get_ray_bundle_at(H, W, focal, pose, sel)
rays = nvsr_amd.train_utils.pack_rays(ro, rd, 2.0, 6.0)
sc, keep = mf.native_scene()
out = []
for S in (64, 128):
    z = torch.sort(torch.rand(N, S, device=dev, generator=g) * 4 + 2, -1)[0].contiguous()
    raw = torch.empty(N, S, 4, device=dev); gates = torch.empty(N, S, 32, dtype=torch.int32, device=dev)
    capi.call("nvsr_decode_rays_ex", C.byref(sc), capi.ptr(mf.packed_decoder()), N, S, capi.ptr(rays), capi.ptr(z), capi.ptr(raw), capi.ptr(gates), None, capi.stream())
    g_raw = torch.randn(N, S, 4, device=dev, generator=g) * 1e-3
    if os.environ.get("BWD_ZERO_WEIGHTS") == "1":
        with torch.no_grad():
            for n_, p_ in mf.named_parameters():
                if n_.endswith(".weight") and (n_.startswith("density_dec.") or n_.startswith("rgb_dec.")):
                    p_.zero_()
    gpl = [torch.zeros_like(k) for k in keep]
    gptrs = (C.c_void_p * 4)(*[t.data_ptr() for t in gpl])
    vws = torch.empty(lib.nvsr_view_grad_workspace_floats(N, S), device=dev)
    ts = []
    for i in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        capi.call("nvsr_render_pass_backward_gates", C.byref(sc), capi.ptr(mf.packed_decoder()), capi.ptr(mf.packed_decoder_bwd()), N, S, capi.ptr(rays), capi.ptr(z),
                  capi.ptr(g_raw), capi.ptr(gates), gptrs, capi.ptr(vws), None, capi.stream())
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    out.append("S=%d %.3f" % (S, min(ts[2:])))
print("  ".join(out))
