"""does the ORDER of the rays matter?  Renders the 800x800 frame with the rays permuted into pw x ph pixel patches (32 rays = one wave tile)
and times the whole fused render (coarse pass + resampler + fine pass):  python tools/ray_order_time.py"""
import sys, os, ctypes as C; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nvsr_amd
from bench import make_synthetic_scene
dev = torch.device('cuda', 0)
mc, mf, sid, pose = make_synthetic_scene(dev, 800, 32, seed=0)
H = W = 800; focal = 0.5 * W / np.tan(0.5 * 0.6911112)
ro, rd = nvsr_amd.nerf_helpers.get_ray_bundle(H, W, focal, pose)
rays0 = nvsr_amd.train_utils.pack_rays(ro, rd, 2.0, 6.0); N = rays0.shape[0]
capi = nvsr_amd.capi
ws = torch.empty(capi.lib().nvsr_render_workspace_floats(N, 64, 128), device=dev)
sc, keep = mc.native_scene()
pc, pf = mc.packed_decoder(), mf.packed_decoder()

def perm_for(pw, ph, sw=1, sh=1):
    """pixel order: patches of pw x ph pixels, row-major inside a patch; patches grouped into super-patches of sw x sh patches, row-major"""
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    key = (((ys // (ph * sh)) * (-(-W // (pw * sw))) + xs // (pw * sw)) * (sw * sh) + ((ys // ph) % sh) * sw + (xs // pw) % sw) * (pw * ph) + (ys % ph) * pw + xs % pw
    return torch.as_tensor(np.argsort(key.reshape(-1), kind="stable"), device=dev)

ref = None
ONLY = [a for a in sys.argv[1:]]
for name, args in [c for c in [("32x1 (row order)", (32, 1)), ("8x4", (8, 4)), ("4x8", (4, 8)), ("16x2", (16, 2)), ("8x4 in 2x1 super (X/Y tiles side by side)", (8, 4, 2, 1)),
                   ("8x4 in 1x2 super", (8, 4, 1, 2)), ("8x4 in 4x2 super (256 rays = 32x8)", (8, 4, 4, 2)), ("8x4 in 2x4 super (16x16)", (8, 4, 2, 4)),
                   ("8x4 in 16x8 super (128x32 px)", (8, 4, 16, 8)), ("8x4 in 16x32 super (128x128 px)", (8, 4, 16, 32)), ("8x4 in 32x16 super (256x64 px)", (8, 4, 32, 16)),
                   ("8x4 in 25x25 super (200x100 px)", (8, 4, 25, 25)), ("8x4 in 50x25 super (400x100 px)", (8, 4, 50, 25)),
                   ("16x2 in 8x16 super (128x32 px)", (16, 2, 8, 16)), ("8x4 in 16x8 super, again", (8, 4, 16, 8)), ("8x4, again", (8, 4))]
                   if not ONLY or any(o in c[0] for o in ONLY)]:
    perm = perm_for(*args)
    rays = rays0[perm].contiguous()
    bufs = [torch.empty((N, 3), device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev), torch.empty((N, 3), device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev)]
    ts = []
    for rep in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        capi.call("nvsr_render_rays", C.byref(sc), capi.ptr(pc), capi.ptr(pf), N, 64, 128, capi.ptr(rays), 0, 0, None, None, None, None, *[capi.ptr(t) for t in bufs], capi.ptr(ws), capi.stream())
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    img = torch.empty_like(bufs[3]); img[perm] = bufs[3]
    if ref is None:
        ref = img
    print("%-48s frame %.2f ms (min of 3)   pixels identical to row order: %s" % (name, min(ts[1:]), torch.equal(ref, img)))
