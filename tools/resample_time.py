"""importance_resample_kernel alone on the inference shape (640 000 rays, 64 coarse depths -> 128 samples, deterministic u):
   python tools/resample_time.py        (NVSR_RESAMPLE_GENERAL=1: every ray through the general path)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nvsr_amd
capi = nvsr_amd.capi
dev = "cuda:0"
N, Nc, Nf = 640000, 64, 128
torch.manual_seed(0)
rays = torch.zeros((N, 11), device=dev); rays[:, 6] = 2.0; rays[:, 7] = 6.0
w = torch.rand((N, Nc), device=dev) ** 4
zf = torch.empty((N, Nc + Nf), device=dev)
ts = []
for i in range(8):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    capi.call("nvsr_importance_resample_rays", N, Nc, Nf, capi.ptr(rays), 0, capi.ptr(w), None, capi.ptr(zf), capi.stream())
    b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
nbytes = 4 * (Nc + Nc + Nc + Nf) * N          # weights + (computed) depths + output row, as bench.py counts the stage
t = min(ts[2:])
print("importance_resample (640k rays, 64 -> 128, det): %.3f ms = %.2f TB/s of algorithmic bytes (%.0f %% of 8 TB/s)  checksum %.6f"
      % (t, nbytes / t / 1e9, 100 * nbytes / t / 1e9 / 8000, float(zf.double().mean())))
