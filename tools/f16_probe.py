"""Decoder outputs of the bench scene in every arithmetic against the exact-f32 kernel: max / rms error of raw (rgb logits, sigma) relative
to the range, plus the magnitudes of what the f16 limbs see (weights per layer)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
import nvsr_amd  # noqa: E402

dev = torch.device("cuda:0")
capi = nvsr_amd.capi
mc, mf, sid, pose = bench.make_synthetic_scene(dev)
for name, p in mf.named_parameters():
    if p.dim() == 2:
        print("%-28s max |w| %.3g  rms %.3g" % (name, float(p.abs().max()), float(p.pow(2).mean().sqrt())))
H = W = 800
focal = 0.5 * W / np.tan(0.5 * 0.6911112)
ro, rd = nvsr_amd.nerf_helpers.get_ray_bundle(H, W, focal, pose.float())
rays = nvsr_amd.train_utils.pack_rays(ro, rd, 2.0, 6.0)
N, S = 131072, 16
rays = rays[torch.randperm(rays.shape[0], device=dev)[:N]].contiguous()
z = torch.sort(torch.rand(N, S, device=dev) * 4 + 2, -1).values.contiguous()
sc, keep = mf.native_scene()
res = {}
for mode in ("f32", "bf16x3", "f16x2"):
    o = dict(rgb=torch.empty((N, 3), device=dev), disp=torch.empty((N,), device=dev), acc=torch.empty((N,), device=dev),
             raw=torch.empty((N, S, 4), device=dev))
    capi.call("nvsr_render_pass_arith", C.byref(sc), capi.ptr(mf.packed_decoder()), N, S, capi.ptr(rays), capi.ptr(z), None, 1,
              capi.ptr(o["rgb"]), capi.ptr(o["disp"]), capi.ptr(o["acc"]), None, None, capi.ptr(o["raw"]), capi.ARITHMETIC[mode], capi.stream())
    torch.cuda.synchronize()
    res[mode] = {k: v.double() for k, v in o.items()}
# double-precision decoder of the same points through torch (the module's own forward in float64)
for ch, nm in ((slice(0, 3), "rgb logits"), (slice(3, 4), "sigma")):
    ref = res["f32"]["raw"][..., ch]
    rng = float(ref.abs().max())
    for mode in ("bf16x3", "f16x2"):
        d = res[mode]["raw"][..., ch] - ref
        print("%-10s %-7s vs f32 kernel: max %.3g rms %.3g mean %.3g of range %.3g; nan %d" % (nm, mode, float(d.abs().max()) / rng, float(d.pow(2).mean().sqrt()) / rng,
                                                                              float(d.mean()) / rng, rng, int(torch.isnan(res[mode]["raw"]).sum())))
for mode in ("bf16x3", "f16x2"):
    d = res[mode]["rgb"] - res["f32"]["rgb"]
    print("pixels %-7s max %.3g rms %.3g" % (mode, float(d.abs().max()), float(d.pow(2).mean().sqrt())))

# against the double-precision oracle at the same depths (2048 rays)
from oracle.oracle import Oracle, decoder_blob  # noqa: E402
chk = Oracle(f32=False)
planes = [mf.planes_[nvsr_amd.models.get_plane_name(sid, d)].detach().cpu().numpy() for d in range(4)]
osc = chk.scene(planes, mf.box_coords[sid].numpy())
dec = chk.decoder(decoder_blob({k: v.detach().cpu().numpy() for k, v in mf.state_dict().items()}))
n = 2048
fo = chk.render_given_z(osc, dec, rays[:n].cpu().numpy(), z[:n].cpu().numpy(), white_background=True, want_raw=True)
ref = fo["raw"].astype(np.float64)
for ch, nm in ((slice(0, 3), "rgb logits"), (slice(3, 4), "sigma")):
    rng = float(np.abs(ref[..., ch]).max())
    for mode in ("f32", "bf16x3", "f16x2"):
        d = res[mode]["raw"][:n].cpu().numpy()[..., ch] - ref[..., ch]
        print("%-10s %-7s vs float64 oracle: max %.3g rms %.3g mean %+.3g (of range %.3g)" % (nm, mode, np.abs(d).max() / rng, np.sqrt((d ** 2).mean()) / rng, d.mean() / rng, rng))
