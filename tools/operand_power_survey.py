"""Which of the matrix kernels run on the power-managed clock?  Each launch is timed on its product operands and again with ZERO weight matrices
(same instruction stream and memory traffic; operands that toggle fewer bits).  A kernel bound by issue / latency / memory takes the same time; one on
the power-managed clock gets faster.  python tools/operand_power_survey.py"""
import sys, os, ctypes as C; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nvsr_amd
from bench import make_synthetic_scene
capi = nvsr_amd.capi; lib = capi.lib(); dev = torch.device("cuda", 0)


def best(fn, reps=6):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts[1:])


# --- SR trunk convolution 256 -> 256, 3 planes of 270^2 (f16x2), forward kernel and weight-gradient kernel
x = torch.randn((3, 256, 270, 270), device=dev)
for tag, w in (("product", torch.randn((256, 256, 3, 3), device=dev) / np.sqrt(9 * 256)), ("zero weights", torch.zeros((256, 256, 3, 3), device=dev))):
    pk = torch.empty(lib.nvsr_conv3x3_packed_floats(256, 256), device=dev)
    capi.call("nvsr_pack_conv3x3", capi.ptr(w), 256, 256, capi.ptr(pk), capi.stream())
    out = torch.empty((3, 256, 268, 268), device=dev)
    t = best(lambda: [capi.call("nvsr_conv3x3_arith", capi.ptr(x[b]), 256, 270, 270, capi.ptr(pk), 256, 1, None, capi.ptr(out[b]), 2, 0, capi.stream()) for b in range(3)])
    print("SR trunk convolution x 3 planes (conv3x3_limb16_kernel, f16x2)      %-13s %.3f ms" % (tag, t))
dy = torch.randn((256, 268, 268), device=dev) * 1e-3
wsf = torch.empty(lib.nvsr_conv3x3_wgrad_workspace_floats(256, 270, 270, 256), device=dev)
dw = torch.zeros((256, 256, 3, 3), device=dev)
for tag, xin in (("product", x[0]), ("zero input", torch.zeros_like(x[0]))):
    t = best(lambda: capi.call("nvsr_conv3x3_wgrad_arith", capi.ptr(dy), capi.ptr(xin), 256, 270, 270, 256, 1.0, capi.ptr(dw), capi.ptr(wsf), 2, capi.stream()))
    print("SR weight gradient, one plane (conv3x3_wgrad_limb_kernel<2>)        %-13s %.3f ms" % (tag, t))

# --- training forward (fine pass of a step: 4096 rays x 128 samples, gates on) and gate-driven backward
mc, mf, sid, pose = make_synthetic_scene(dev, 200, 32, seed=0, channels_last=True)
H = W = 800; focal = 0.5 * W / np.tan(0.5 * 0.6911112)
g = torch.Generator(device=dev).manual_seed(1)
N, S = 4096, 128
sel = torch.randint(0, H, (N, 2), device=dev, generator=g)
ro, rd = nvsr_amd.training.get_ray_bundle_at(H, W, focal, pose, sel)
rays = nvsr_amd.train_utils.pack_rays(ro, rd, 2.0, 6.0)
z = torch.sort(torch.rand(N, S, device=dev, generator=g) * 4 + 2, -1)[0].contiguous()
raw = torch.empty(N, S, 4, device=dev); gates = torch.empty(N, S, 32, dtype=torch.int32, device=dev)
for tag in ("product", "zero weights"):
    if tag == "zero weights":
        with torch.no_grad():
            for n_, p_ in mf.named_parameters():
                if n_.endswith(".weight") and (n_.startswith("density_dec.") or n_.startswith("rgb_dec.")):
                    p_.zero_()
    sc, keep = mf.native_scene()
    packed = mf.packed_decoder()
    t = best(lambda: capi.call("nvsr_decode_rays_ex", C.byref(sc), capi.ptr(packed), N, S, capi.ptr(rays), capi.ptr(z), capi.ptr(raw), capi.ptr(gates), None, capi.stream()))
    print("training forward, fine pass 4096 x 128 (decode_rays_pair_gates_kernel) %-13s %.3f ms" % (tag, t))
