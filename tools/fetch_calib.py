"""Calibration of FETCH_SIZE for the access widths this library uses (MI355X_MICROARCH.md, HBM: "Other access widths are uncalibrated:
calibrate on a known byte count in your own access pattern").  Run under `tools/pmc.sh cal FETCH_SIZE` with PMC_SCRIPT=fetch_calib.py:
  * to_channel_last_kernel / from_channel_last_kernel read a [48, R, R] plane ONCE with 4-byte-per-lane coalesced loads (global_load_dword,
    256 B per wave) -- the access width of the SR convolutions' input patches (conv3x3_limb_kernel: gload) and of the element-wise helpers;
  * a torch float4 copy of the same tensor reads it with 16-byte-per-lane loads (the width the guide calibrated: FETCH_SIZE = 1/2 bytes).
The plane (R = 1400: 376 MB) is larger than the 256 MB Infinity Cache.  tools/pmc_read.py --kernel <name> cal prints the counters; the bytes to
compare with are printed here."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nvsr_amd
capi = nvsr_amd.capi
R = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1400
dev = "cuda:0"
x = torch.randn(48, R, R, device=dev)
y = torch.empty(R, R, 48, device=dev)
z = torch.empty_like(x)
for _ in range(2):
    capi.call("nvsr_plane_to_channel_last", capi.ptr(x), capi.ptr(y), 48, R, R, capi.stream())
    capi.call("nvsr_plane_from_channel_last", capi.ptr(y), capi.ptr(z), 48, R, R, capi.stream())
    w = x.clone()              # vectorised (16 B per lane) elementwise copy
torch.cuda.synchronize()
assert torch.equal(z, x)
print("calibration: each kernel reads %d bytes = %.1f KB once (plane 48 x %d x %d f32)" % (x.numel() * 4, x.numel() * 4 / 1024, R, R))
