"""bench.frame_error_evidence on the bench's frame (800 x 800, planes 800^2, 64 + 128): where the frame error against the float64 checker sits,
per arithmetic.   python tools/frame_evidence.py [n_rays] [plane_res] [res]"""
import sys, os, json; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nvsr_amd, bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
pr = int(sys.argv[2]) if len(sys.argv) > 2 else 800
res = int(sys.argv[3]) if len(sys.argv) > 3 else 800
dev = torch.device("cuda", 0)
mc, mf, sid, pose = bench.make_synthetic_scene(dev, pr, 32, seed=0, theta=30.0)
focal = 0.5 * res / np.tan(0.5 * bench.CAMERA_ANGLE_X)
ro, rd = nvsr_amd.nerf_helpers.get_ray_bundle(res, res, focal, pose)
rays = nvsr_amd.train_utils.pack_rays(ro, rd, 2.0, 6.0)
print(json.dumps(bench.frame_error_evidence(nvsr_amd, mc, mf, sid, rays, n_rays=n), indent=1, default=float))
