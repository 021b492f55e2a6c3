"""Cycle stamps of render3.hip's step sections.  Run on the GPU box:
   NVSR_EXTRA_HIPCC_FLAGS=-DR3_STAMP=1 python tools/limb_stamp.py   (builds a VARIANT library with the stamps compiled in under
   gpurun_out/variants/ and loads that one; the product library is never overwritten)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("NVSR_EXTRA_HIPCC_FLAGS", "").split() and "NVSR_HIP_LIB" not in os.environ:
    # experiment flags: compile to a separate file and bind THAT library (capi reads NVSR_HIP_LIB at import)
    from importlib import util as _u
    _spec = _u.spec_from_file_location("_nvsr_build", os.path.join(ROOT, "neural-volume-super-resolution_amd", "build.py"))
    _b = _u.module_from_spec(_spec); _spec.loader.exec_module(_b)
    os.makedirs(os.path.join(ROOT, "gpurun_out", "variants"), exist_ok=True)
    os.environ["NVSR_HIP_LIB"] = _b.build_extension(out_path=os.path.join(ROOT, "gpurun_out", "variants", "stamp.so"))
import nvsr_amd as hip
from bench import make_synthetic_scene

capi = hip.capi
dev = "cuda:0"
mc, mf, sid, pose = make_synthetic_scene(dev, plane_res=800, view_res=32, seed=0)
H = W = 800
focal = 0.5 * W / np.tan(0.5 * 0.6911112)
ro, rd = hip.nerf_helpers.get_ray_bundle(H, W, focal, pose)
rays = hip.train_utils.pack_rays(ro, rd, 2.0, 6.0).contiguous()
if not os.environ.get("KBENCH_ROW_ORDER"):      # the bench renders a frame in patch order
    rays = rays[hip.train_utils.patch_order(rays.shape[0], W, dev)[0]].contiguous()
N, S = rays.shape[0], 192
rng = np.random.default_rng(17)
z = torch.as_tensor(np.sort(rng.uniform(2, 6, (N, S)).astype(np.float32), -1), device=dev)
if not os.environ.get("STAMP_UNIFORM_Z"):       # the frame's own importance-sampled fine depths (what the bench's fine pass runs on)
    ws = torch.empty(capi.lib().nvsr_render_workspace_floats(N, 64, 128), device=dev)
    bufs = [torch.empty((N, 3), device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev), torch.empty((N, 3), device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev)]
    sc0, keep0 = mc.native_scene()
    capi.call("nvsr_render_rays", C.byref(sc0), capi.ptr(mc.packed_decoder()), capi.ptr(mf.packed_decoder()), N, 64, 128, capi.ptr(rays), 0, 0, None, None, None, None,
              *[capi.ptr(b) for b in bufs], capi.ptr(ws), capi.stream())
    torch.cuda.synchronize()
    z = ws[2 * N * 64: 2 * N * 64 + N * 192].view(N, 192).clone()
    del ws
sc, keep = mf.native_scene()
packed = mf.packed_decoder()
names = ["top+gather", "ring wait", "rgb0", "rgb1-3", "den0", "den1-3", "epilogue"]
if "R3_STAMP=4" in os.environ.get("NVSR_EXTRA_HIPCC_FLAGS", ""):
    names = ["sync a (x6)", "issue a", "X a (Y relu)", "Y a", "sync+issue b", "X b", "Y b (X relu)"]
if "R3_STAMP=3" in os.environ.get("NVSR_EXTRA_HIPCC_FLAGS", ""):
    names = ["B0 X view", "B1 Y view", "B2 X p0", "B3 Y p0", "B4 X p1", "B5 Y p1", "B6 X p2"]
for mode in sys.argv[1:] or ["bf16x3", "f16x2"]:
    capi.set_decoder_arithmetic(mode)
    o = dict(rgb=torch.empty((N, 3), device=dev), disp=torch.empty((N,), device=dev), acc=torch.empty((N,), device=dev), raw=torch.zeros((N, S, 4), device=dev))
    for rep in range(2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        capi.call("nvsr_render_pass_ex", C.byref(sc), capi.ptr(packed), N, S, capi.ptr(rays), capi.ptr(z), None, 0,
                  capi.ptr(o["rgb"]), capi.ptr(o["disp"]), capi.ptr(o["acc"]), None, None, capi.ptr(o["raw"]), capi.stream())
        b.record(); torch.cuda.synchronize()
    st = o["raw"][:, :2, :].reshape(N, 8)[:, :7].cpu().numpy()
    # one lane per tile X: rays 0..31 of each 64
    sel = (np.arange(N) % 64) < 32
    st = st[sel]
    tot = st.sum(1)
    print("%s: %.2f ms; cycles per sample-step: total mean %.0f (min %.0f max %.0f)" % (mode, a.elapsed_time(b), tot.mean(), tot.min(), tot.max()))
    for i, n in enumerate(names):
        print("   %-12s %8.0f  %5.1f %%" % (n, st[:, i].mean(), 100 * st[:, i].mean() / tot.mean()))
