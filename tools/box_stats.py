"""How compact are the texel footprints of a wave tile (32 consecutive rays in patch order, one sample index) on each plane?  Decides the box
of an LDS-staged plane gather: fraction of (tile, sample, plane) whose 2x2 taps fit a BW x BH texel box, and the fraction of lanes inside a
box anchored at the tile's minimum.  Fine pass (importance depths of the bench frame) and coarse pass (uniform depths)."""
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
H = W = 800
R = 800
focal = 0.5 * W / np.tan(0.5 * 0.6911112)
ro, rd = nvsr_amd.nerf_helpers.get_ray_bundle(H, W, focal, pose.float())
rays = nvsr_amd.train_utils.pack_rays(ro, rd, 2.0, 6.0)
N = rays.shape[0]
rays = rays[nvsr_amd.train_utils.patch_order(N, W, dev)[0]].contiguous()
ws = torch.empty(capi.lib().nvsr_render_workspace_floats(N, 64, 128), device=dev)
bufs = [torch.empty((N, 3), device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev), torch.empty((N, 3), device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev)]
sc, keep = mc.native_scene()
capi.call("nvsr_render_rays", C.byref(sc), capi.ptr(mc.packed_decoder()), capi.ptr(mf.packed_decoder()), N, 64, 128, capi.ptr(rays), 0, 0, None, None, None, None,
          *[capi.ptr(b) for b in bufs], capi.ptr(ws), capi.stream())
torch.cuda.synchronize()
zf = ws[2 * N * 64: 2 * N * 64 + N * 192].view(N, 192)
zc = torch.linspace(0, 1, 64, device=dev)[None, :] * 4 + 2 + torch.zeros(N, 1, device=dev)
axes = [(1, 2), (0, 2), (0, 1)]
for name, z in (("coarse (64 uniform)", zc), ("fine (192 importance)", zf)):
    sel = torch.arange(0, N // 32, 7, device=dev)[:, None] * 32 + torch.arange(32, device=dev)[None, :]      # every 7th tile
    r = rays[sel.reshape(-1)]
    zz = z[sel.reshape(-1)]
    pts = r[:, None, 0:3] + r[:, None, 3:6] * zz[..., None]                  # [T*32, S, 3]
    t = ((pts + 4.0) / 8.0).clamp(0, 1) * (R - 1)
    tx = torch.floor(t).clamp(max=R - 2).long().view(-1, 32, z.shape[1], 3)   # [T, 32, S, 3]
    print("== %s: %d tiles x %d samples" % (name, tx.shape[0], tx.shape[2]))
    for d, (a, b) in enumerate(axes):
        x, y = tx[..., a], tx[..., b]
        sx = x.max(1).values - x.min(1).values + 2                           # span incl. the +1 tap
        sy = y.max(1).values - y.min(1).values + 2
        line = "   plane %d: span x median %d p90 %d, y median %d p90 %d;" % (d, sx.float().median(), sx.float().quantile(0.9), sy.float().median(), sy.float().quantile(0.9))
        for bw, bh in ((6, 6), (8, 4), (8, 8), (12, 6), (16, 4)):
            fit = ((sx <= bw) & (sy <= bh)) | ((sx <= bh) & (sy <= bw))
            # per-lane: box anchored at the tile's per-axis 10 % quantile - 0, lanes whose 2x2 taps fall inside
            ax = x.float().quantile(0.05, dim=1, keepdim=True).floor().long()
            ay = y.float().quantile(0.05, dim=1, keepdim=True).floor().long()
            inl = ((x >= ax) & (x + 1 < ax + bw) & (y >= ay) & (y + 1 < ay + bh)).float().mean()
            line += "  %dx%d: tiles %.2f lanes %.2f" % (bw, bh, fit.float().mean(), inl)
        print(line)
