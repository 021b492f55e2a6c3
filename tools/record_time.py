"""times the three record kernels of a decoder-training pass alone (4096 rays x {64, 128} sorted random depths, planes 200^2, f16x2):
   forward with the layer-input half of the weight-gradient record (decode_rays_limb_kernel<true, true, 2>), gate-driven backward with the gradient
   half (render_pass_backward_gates_limb_kernel<true, 2>), the contraction (decoder_wgrad_limb_kernel<4> + head_wgrad_kernel), and the same forward /
   backward WITHOUT the record beside them.  NVSR_HIP_LIB selects a variant.  Prints min ms of 8 launches."""
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
arith = capi.resolve_decoder_arithmetic(None) if hasattr(capi, "resolve_decoder_arithmetic") else -1


def timed(fn, reps=10):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts[2:])


for S in (64, 128):
    z = torch.sort(torch.rand(N, S, device=dev, generator=g) * 4 + 2, -1)[0].contiguous()
    raw = torch.empty(N, S, 4, device=dev); gates = torch.empty(N, S, 32, dtype=torch.int32, device=dev)
    rec = torch.empty(lib.nvsr_decoder_record_floats(N, S), device=dev)
    g_raw = torch.randn(N, S, 4, device=dev, generator=g) * 1e-3
    gpl = [torch.zeros_like(k) for k in keep]
    gptrs = (C.c_void_p * 4)(*[t.data_ptr() for t in gpl])
    vws = torch.empty(lib.nvsr_view_grad_workspace_floats(N, S), device=dev)
    gnat = torch.zeros(capi.DECODER_NATURAL_FLOATS, device=dev)
    fwd = lambda r: capi.call("nvsr_decode_rays_ex", C.byref(sc), capi.ptr(mf.packed_decoder()), N, S, capi.ptr(rays), capi.ptr(z), capi.ptr(raw), capi.ptr(gates), r, capi.stream())
    bwd = lambda r: capi.call("nvsr_render_pass_backward_gates", C.byref(sc), capi.ptr(mf.packed_decoder()), capi.ptr(mf.packed_decoder_bwd()), N, S, capi.ptr(rays), capi.ptr(z),
                              capi.ptr(g_raw), capi.ptr(gates), gptrs, capi.ptr(vws), r, capi.stream())
    t_f, t_fr = timed(lambda: fwd(None)), timed(lambda: fwd(capi.ptr(rec)))
    t_b, t_br = timed(lambda: bwd(None)), timed(lambda: bwd(capi.ptr(rec)))
    t_w = timed(lambda: capi.call("nvsr_decoder_weight_grad_arith", N, S, capi.ptr(rec), capi.ptr(gnat), arith, capi.stream()))
    print("S=%d: forward %.3f, with record %.3f | backward %.3f, with record %.3f | contraction %.3f ms   (record: %.2f GB)"
          % (S, t_f, t_fr, t_b, t_br, t_w, rec.numel() * 4 / 1e9))
