"""per-FLOP rate of the EDSR up-sampling convolution (256 -> 1024 + PixelShuffle epilogue) against the trunk convolution (256 -> 256 + ReLU), f16x2:
   python tools/upconv_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nvsr_amd
capi = nvsr_amd.capi
dev = "cuda:0"
for name, Cin, Cout, H, W, epi in (("trunk 256->256 relu 270^2", 256, 256, 270, 270, 1), ("up1 256->1024 shuffle 206^2", 256, 1024, 206, 206, 3),
                                   ("up1 256->1024 NO shuffle 206^2", 256, 1024, 206, 206, 0), ("up2 256->1024 shuffle 408^2", 256, 1024, 408, 408, 3),
                                   ("up2 256->1024 NO shuffle 408^2", 256, 1024, 408, 408, 0)):
    x = torch.randn((Cin, H, W), device=dev)
    w = torch.randn((Cout, Cin, 3, 3), device=dev) / np.sqrt(9 * Cin)
    pk = torch.empty(capi.lib().nvsr_conv3x3_packed_floats(Cin, Cout), device=dev)
    capi.call("nvsr_pack_conv3x3", capi.ptr(w), Cin, Cout, capi.ptr(pk), capi.stream())
    out = torch.empty((Cout * (H - 2) * (W - 2),), device=dev)
    flop = 2.0 * 9 * Cin * Cout * (H - 2) * (W - 2)
    ts = []
    for rep in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        capi.call("nvsr_conv3x3_arith", capi.ptr(x), Cin, H, W, capi.ptr(pk), Cout, epi, None, capi.ptr(out), 2, 0, capi.stream())
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    t = min(ts[1:])
    print("%-34s %.3f ms  %.0f TFLOP/s" % (name, t, flop / t / 1e9))
