"""fuzz of the batched SR-training path (PlanesSR.forward_many -> nvsr_planes_sr_*_batch_arith: ragged launches) against the plane-by-plane path:
random plane sizes, 2..4 planes, random regions of interest (some touching borders, some tiny), hidden sizes that take the narrow limb kernels (16, 64),
the exact-f32 fallback inside a ragged batch (48 -> 128 input layer) and the 16x16x32 kernels (128, 256), all three arithmetics.
   python tools/sr_batch_fuzz.py [cases] [seed]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nvsr_amd
dev = "cuda:0"
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
M = nvsr_amd.models
worst = {"planes": 0.0, "gw": 0.0, "glr": 0.0}
for case in range(cases):
    hid = int(rng.choice([16, 64, 128, 256]))
    nb = int(rng.integers(1, 3 if hid >= 128 else 4))
    R0, R1 = int(rng.integers(12, 40)), int(rng.integers(12, 40))
    B = int(rng.integers(2, 5))
    mode = str(rng.choice(["f16x2", "bf16x3", "f32"]))
    rois = []
    for _ in range(B):
        a, b = np.sort(rng.uniform(-1, 1, 2)), np.sort(rng.uniform(-1, 1, 2))
        if rng.random() < 0.3: a[0] = -1.0
        if rng.random() < 0.3: b[1] = 1.0
        rois.append([float(a[0]), float(b[0]), float(a[1]), float(b[1])])
    res = {}
    for path in ("batched", "single"):
        torch.manual_seed(1000 + case)
        sr = M.PlanesSR(M.EDSR, 4, 48, 48, {"model": {"hidden_size": hid, "n_blocks": nb}}, "bilinear").to(dev)
        with torch.no_grad():
            for p_ in sr.parameters():
                p_.mul_(10.0)
        sr.inner_model.arithmetic = mode
        sr.train()
        g = torch.Generator(device=dev).manual_seed(2000 + case)
        lrs = [torch.nn.Parameter(torch.randn(1, 48, R0, R1, device=dev, generator=g) * 0.5) for _ in range(B)]
        for k, t in enumerate(lrs):
            sr.set_LR_plane(t, id="p%d" % k, save_interpolated=False)
        if path == "batched":
            outs = sr.forward_many([("p%d" % k, rois[k]) for k in range(B)])
        else:
            outs = [sr(("p%d" % k, torch.tensor(rois[k]).reshape(2, 2))) for k in range(B)]
        gen = torch.Generator(device=dev).manual_seed(7)
        sum((torch.nan_to_num(o) * torch.randn(o.shape, device=dev, generator=gen)).sum() for o in outs).backward()
        gw = torch.cat([(w.grad if w.grad is not None else torch.zeros_like(w)).reshape(-1) for w in sr.inner_model.conv_parameters()])
        res[path] = ([o.detach().clone() for o in outs], gw.clone(), [t.grad.clone() for t in lrs])
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
    for a, b in zip(res["batched"][0], res["single"][0]):
        assert torch.equal(torch.isnan(a), torch.isnan(b)), (case, "nan pattern")
        d = float((torch.nan_to_num(a) - torch.nan_to_num(b)).abs().max())
        worst["planes"] = max(worst["planes"], d)
        assert d == 0.0, (case, hid, nb, R0, R1, B, mode, d)
    e = rel(res["batched"][1], res["single"][1]); worst["gw"] = max(worst["gw"], e)
    assert e <= 2e-5, (case, hid, nb, R0, R1, B, mode, "gw", e)
    for a, b in zip(res["batched"][2], res["single"][2]):
        e = rel(a, b); worst["glr"] = max(worst["glr"], e)
        assert e <= 2e-5, (case, hid, nb, R0, R1, B, mode, "glr", e)
    print("case %2d  hid %3d nb %d  plane %2dx%2d  B %d  %-6s ok" % (case, hid, nb, R0, R1, B, mode), flush=True)
print("all %d cases equal: planes bit for bit, worst relative L2 of the weight gradient %.2e, of an LR gradient %.2e" % (cases, worst["gw"], worst["glr"]))
