import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
import nvsr_amd as hip
dev='cuda:0'
torch.manual_seed(0)
sr = hip.models.PlanesSR(hip.models.EDSR, 4, 48, 48, {"model": {"hidden_size": 256, "n_blocks": 32}}, "bilinear").to(dev)
sr.eval()
lr = torch.randn(1,48,200,200,device=dev)*0.5
sr.set_LR_plane(lr, id='p', save_interpolated=False)
for i in range(3):
    sr.clear_SR_planes()
    torch.cuda.synchronize(); t0=time.perf_counter()
    out = sr('p')
    torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print("PlanesSR 200^2->800^2: %.1f ms  -> %.1f TFLOP/s (6.74 TFLOP algorithmic)" % (dt*1e3, 6.74/dt), out.shape, float(out.abs().mean()))
