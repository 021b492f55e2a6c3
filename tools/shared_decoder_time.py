"""frame time with ONE decoder for both passes (models.fine.type == 'use_same'): the shared path (fine pass on the importance samples only)
against the path that recomputes the coarse samples (NVSR_NO_SHARED_DECODER=1); 800 x 800, 64 + 128, the bench's synthetic scene"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nvsr_amd
from bench import make_synthetic_scene, render_options
dev = torch.device("cuda", 0)
mc, mf, sid, pose = make_synthetic_scene(dev, 800, 32, seed=0)
H = W = 800; focal = 0.5 * W / np.tan(0.5 * 0.6911112)
ro, rd = nvsr_amd.nerf_helpers.get_ray_bundle(H, W, focal, pose)
opts, scfg = render_options(64, 128)
def frame(fine):
    ts = []
    for i in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = nvsr_amd.train_utils.eval_nerf(H, W, focal, mc, fine, ro, rd, opts, scene_id=sid, scene_config=scfg)
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts[1:]), out[3]
t2, _ = frame(mf)
os.environ.pop("NVSR_NO_SHARED_DECODER", None)
ts, img_s = frame(mc)
os.environ["NVSR_NO_SHARED_DECODER"] = "1"
tr, img_r = frame(mc)
print("two decoders %.1f ms | one decoder: recomputed %.1f ms, shared %.1f ms (%.2f x) | max |shared - recomputed| %.2e" % (
    t2, tr, ts, tr / ts, float((img_s - img_r).abs().max())))
