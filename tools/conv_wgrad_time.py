"""times the weight-gradient contraction of one 3x3 conv layer of the SR network (conv3x3_wgrad[_limb]_kernel + the fixed-order reduction):
   python tools/conv_wgrad_time.py [Cin Cout H W]      (H, W = size of the layer's input; default: the EDSR trunk layer of a 200^2 plane)
   NVSR_WGRAD_PIECES=<n> forces the number of workgroups (pieces of the step range) in a -DWG_TUNE variant build (NVSR_HIP_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nvsr_amd
capi = nvsr_amd.capi
a = [int(v) for v in sys.argv[1:5]] + [256, 256, 270, 270][len(sys.argv) - 1:]
Cin, Cout, H, W = a
dev = "cuda:0"
x = torch.randn((Cin, H, W), device=dev)
dy = torch.randn((Cout, H - 2, W - 2), device=dev)
dw = torch.zeros((Cout, Cin, 3, 3), device=dev)
ws = torch.empty(max(capi.lib().nvsr_conv3x3_wgrad_workspace_floats(Cin, H, W, Cout), 9 * Cin * Cout * 9 * 64), device=dev)
flop = 2.0 * 9 * Cin * Cout * (H - 2) * (W - 2)
for mode, code in (("bf16x3", 3), ("f32", 0)):
    ts = []
    for rep in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dw.zero_()
        e0.record()
        for k in range(3):
            capi.call("nvsr_conv3x3_wgrad_arith", capi.ptr(dy), capi.ptr(x), Cin, H, W, Cout, 1.0, capi.ptr(dw), capi.ptr(ws), code, capi.stream())
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 3)
    t = min(ts[1:])
    if mode == "bf16x3" and os.environ.get("NVSR_WGRAD_STAMPS"):      # -DWG_STAMP=1 variant build: cycles per step and section, wave 0 of every workgroup
        n_wg = int(os.environ["NVSR_WGRAD_STAMPS"])
        st = ws[n_wg * 2 * 9 * 64 * 64: n_wg * 2 * 9 * 64 * 64 + n_wg * 8].reshape(n_wg, 8).cpu().numpy()
        per = st[:, :6] / st[:, 6:7]
        print("cycles per step: barrier-1 wait %.0f, split + LDS writes %.0f, barrier-2 wait %.0f, fetch issue %.0f, MFMA phase %.0f, loop %.0f  (steps per workgroup %.1f)"
              % (tuple(per.mean(0)) + (st[:, 6].mean(),)))
    print("%-7s %dx%d %d->%d: %.3f ms  %.1f TFLOP/s  (pieces=%s) checksum %.6e" % (mode, H, W, Cin, Cout, t, flop / t / 1e9, os.environ.get("NVSR_WGRAD_PIECES", "-"), float(dw.double().abs().sum())))
