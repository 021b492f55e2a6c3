"""time the SR network forward / forward+backward at BASELINE config 3 size (48ch, hidden 256, 32 blocks, x4, LR 200^2 -> 336^2 padded)"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nvsr_amd
dev = torch.device('cuda', 0)
torch.manual_seed(0)
sr = nvsr_amd.models.PlanesSR(nvsr_amd.models.EDSR, 4, 48, 48, {"model": {"hidden_size": 256, "n_blocks": 32}}, "bilinear").to(dev)
lr = torch.randn(1, 48, 200, 200, device=dev) * 0.5
sr.set_LR_plane(lr, id="p", save_interpolated=False)
def run(train):
    sr.train(train)
    sr.clear_SR_planes()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if train:
        sr.zero_grad(set_to_none=True)
        out = sr("p")
        torch.cuda.synchronize(); t1 = time.perf_counter()
        out.sum().backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        return t1 - t0, t2 - t1
    with torch.no_grad():
        out = sr("p")
    torch.cuda.synchronize()
    return time.perf_counter() - t0, 0.0
for i in range(2): run(False)
f, _ = run(False); print("eval forward  %.1f ms  (6.74 TFLOP -> %.1f TFLOP/s)" % (f * 1e3, 6.74 / f))
for i in range(2): a = run(True)
fw, bw = run(True); print("train forward %.1f ms, backward %.1f ms  (13.5 TFLOP -> %.1f TFLOP/s)" % (fw * 1e3, bw * 1e3, 13.48 / bw))
print("max mem %.1f GB" % (torch.cuda.max_memory_allocated() / 1e9))

# three planes: one by one vs one batched pass
names = ["p", "q", "r"]
for n in names[1:]:
    sr.set_LR_plane(torch.randn(1, 48, 200, 200, device=dev) * 0.5, id=n, save_interpolated=False)
sr.eval()
with torch.no_grad():
    for mode in ("one by one", "batched"):
        for rep in range(2):
            sr.clear_SR_planes()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            if mode == "batched":
                sr.super_resolve_many(names)
            else:
                for n in names: sr(n)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("3 planes %-11s %.1f ms  (20.2 TFLOP -> %.1f TFLOP/s = %.1f %% of the fp32 MFMA peak)" % (mode, dt * 1e3, 20.22 / dt, 100 * 20.22 / dt / 157.3))
