"""times one wide 3x3 conv layer of the SR network (the EDSR trunk shape by default) per row-tile variant and arithmetic:
   python tools/conv_time.py [Cin Cout H W batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nvsr_amd
capi = nvsr_amd.capi
a = [int(v) for v in sys.argv[1:6]] + [256, 256, 270, 270, 3][len(sys.argv) - 1:]
Cin, Cout, H, W, B = a
dev = "cuda:0"
x = torch.randn((B, Cin, H, W), device=dev)
w = torch.randn((Cout, Cin, 3, 3), device=dev) / np.sqrt(9 * Cin)
pk = torch.empty(capi.lib().nvsr_conv3x3_packed_floats(Cin, Cout), device=dev)
capi.call("nvsr_pack_conv3x3", capi.ptr(w), Cin, Cout, capi.ptr(pk), capi.stream())
out = torch.empty((B, Cout, H - 2, W - 2), device=dev)
flop = 2.0 * 9 * Cin * Cout * (H - 2) * (W - 2) * B
ref = None
only = os.environ.get("CONV_TIME_MODES", "f16x2,bf16x3,f32").split(",")        # e.g. CONV_TIME_MODES=f16x2 CONV_TIME_ROWS=22 for a counter pass
only_rows = [int(v) for v in os.environ.get("CONV_TIME_ROWS", "").split(",") if v]
for mode, code in (("f16x2", 2), ("bf16x3", 3), ("f32", 0)):
    if mode not in only:
        continue
    for rows in only_rows or ((0, 16, 18, 19, 20, 22) if mode == "f16x2" else (0, 2, 3, 4, 8, 16, 18, 19, 20) if mode == "bf16x3" else (0, 2, 3, 4)):
        ts = []
        for rep in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for b in range(B):     # (the batch rides in the grid inside the EDSR entry points; here: B launches)
                capi.call("nvsr_conv3x3_arith", capi.ptr(x[b]), Cin, H, W, capi.ptr(pk), Cout, 1, None, capi.ptr(out[b]), code, rows, capi.stream())
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        same = ""
        if mode == "bf16x3":
            ref = out.clone() if ref is None else ref
            same = "same=%s" % (torch.equal(ref, out) if rows != 4 else "max|d| vs default %.2e" % float((ref - out).abs().max()))
        t = min(ts[1:])
        print("%-7s rows %d: %.3f ms  %.1f TFLOP/s %s" % (mode, rows, t, flop / t / 1e9, same))
