"""time nvsr_decoder_weight_grad alone on a random record (4096 rays x 128 / 64 samples), both arithmetic modes.
usage: wgrad_time.py"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nvsr_amd
capi = nvsr_amd.capi
dev = torch.device("cuda", 0)
for S in (128, 64):
    N = 4096
    rec = torch.randn(capi.lib().nvsr_decoder_record_floats(N, S), device=dev)
    grad = torch.zeros(capi.DECODER_NATURAL_FLOATS, device=dev)
    for mode in ("f32", "bf16x3"):
        capi.set_decoder_arithmetic(mode)
        ts = []
        for _ in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            capi.call("nvsr_decoder_weight_grad", N, S, capi.ptr(rec), capi.ptr(grad), capi.stream())
            b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        print("S=%d %s: weight-gradient contraction + heads %.3f ms (min of 6)" % (S, mode, min(ts)))
