"""Is the fused fine pass bound by instruction issue or by the chip's power-managed clock?  The SAME launch (same instruction stream, same memory
traffic) is timed on three sets of operands:
  product   the bench scene (random smooth planes, calibrated random decoder)
  zero-w    the same planes, every decoder matrix zero (biases kept): the weight fragments of every MFMA are zeros
  zero-all  planes AND decoder matrices zero: both operands of every MFMA are zeros (activations = relu(bias) constants)
A kernel bound by issue / latency takes the same time on all three; a kernel whose clock is power-managed gets faster as the operands toggle
fewer bits.  usage: python tools/fine_pass_operand_power.py [reps]"""
import sys, os, ctypes as C; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nvsr_amd
from bench import make_synthetic_scene
dev = torch.device('cuda', 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
mc, mf, sid, pose = make_synthetic_scene(dev, 800, 32, seed=0)
H = W = 800; focal = 0.5 * W / np.tan(0.5 * 0.6911112)
ro, rd = nvsr_amd.nerf_helpers.get_ray_bundle(H, W, focal, pose)
rays = nvsr_amd.train_utils.pack_rays(ro, rd, 2.0, 6.0); N = rays.shape[0]
rays = rays[nvsr_amd.train_utils.patch_order(N, W, dev)[0]].contiguous()
capi = nvsr_amd.capi
ws = torch.empty(capi.lib().nvsr_render_workspace_floats(N, 64, 128), device=dev)
bufs = [torch.empty((N, 3), device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev), torch.empty((N, 3), device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev)]
sc, keep = mc.native_scene()
capi.call("nvsr_render_rays", C.byref(sc), capi.ptr(mc.packed_decoder()), capi.ptr(mf.packed_decoder()), N, 64, 128, capi.ptr(rays), 0, 0, None, None, None, None,
          *[capi.ptr(b) for b in bufs], capi.ptr(ws), capi.stream())
zf = ws[2 * N * 64:].view(N, 192).clone()            # the frame's own fine depths, kept for all three runs
torch.cuda.synchronize()


def timed(tag):
    sc_, keep_ = mf.native_scene()
    packed = mf.packed_decoder()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        capi.call("nvsr_render_pass", C.byref(sc_), capi.ptr(packed), N, 192, capi.ptr(rays), capi.ptr(zf), None, 0, capi.ptr(bufs[3]), capi.ptr(bufs[4]),
                  capi.ptr(bufs[5]), None, None, capi.stream())
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print("%-9s fine pass: min %.2f ms  med %.2f ms   mean rgb %.4f" % (tag, min(ts), float(np.median(ts)), float(bufs[3].mean())))
    return min(ts)


t0 = timed("product")
with torch.no_grad():
    for n, p in mf.named_parameters():
        if n.endswith(".weight") and (n.startswith("density_dec.") or n.startswith("rgb_dec.")):
            p.zero_()
t1 = timed("zero-w")
with torch.no_grad():
    for p in mf.planes_.values():
        p.zero_()
mf.invalidate()
t2 = timed("zero-all")
print("zero weights: %.1f %% faster; zero weights and planes: %.1f %% faster than the product's operands" % (100 * (t0 / t1 - 1), 100 * (t0 / t2 - 1)))
