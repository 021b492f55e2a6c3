#!/usr/bin/env python3
"""Benchmark of the rendering hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N > 1: one rank per GPU under torch.distributed.run -- either launched that way
                                                           by the driver, or, when WORLD_SIZE is not set, by this script itself: the
                                                           parent spawns the N ranks BEFORE touching the GPU and relays their output)

Default workload (BASELINE.json configs[1]): one "step" renders 800x800 views (640 000 rays each) of a synthetic Lego-like scene with
64 coarse + 128 fine samples per ray (256 decoder evaluations per ray) through the tri-plane decoder, planes 3 x 800^2 x 48 +
32^2 x 48, everything resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
Partition of the rays over N > 1 GPUs (--partition; SURVEY.md 8e: contiguous blocks of image rows per GPU, scene replicated, one
all_gather of the finished pixels, no other data-path collective):
  rows   (default for N > 1) a step renders N views, EVERY view sharded by row blocks over the N ranks (distributed.render_views_sharded):
         a rank renders its rows of all N views in one launch = one frame's worth of rays per step whatever N is -> "scaling": "weak";
  frame  a step renders ONE view, its rows sharded over the ranks (distributed.render_image_sharded) -> "scaling": "strong";
  view   every rank renders its own whole view, no collective at all (round 1's mode) -> "weak".

Other workloads of the same path (--workload; same JSON contract, their own metric):
  train   BASELINE configs[3] (Feature_Planes_Only.yml): one optimisation step = 4096 random rays of one view, 64+64 samples, planes
          200^2, nerf.train.what = ['LR_planes'] (--train-what planes+decoder: both decoders train as well, TrainModels.yml), Adam;
          N > 1: every rank draws its own rays, gradients are averaged with one bucketed RCCL all-reduce (planes 23 MB [+ decoders
          1 MB]) before the optimizer steps.
  sr      BASELINE configs[2]'s SR stage: the 3 position planes of a scene 200^2 -> 800^2 through EDSR (hidden 256, 32 blocks, x4)
          in one batched pass (20.2 TFLOP); N > 1: independent replicas (a scene's planes are SR'd once and cached), or with
          --partition bands the stated SR partition of SURVEY 8e: every plane cut into N horizontal bands (68-pixel LR halo), one
          all_gather per plane ("scaling": "strong").
  train --rays-global 4096: the stated training partition of SURVEY 8e (4096 rays -> 4096 / N per GPU, same pixels on all ranks), "strong".
  refine  BASELINE configs[4]'s iteration (config/RefineOnTestScene.yml, TrainModels.yml): 4096 random rays of an 800x800 view, 64+64 samples, LR
          planes 200^2, the fine model samples the three position planes super-resolved by PlanesSR(EDSR 256 x 32) in TRAINING mode on the regions
          of interest of the batch (forward that keeps its activations, data + weight gradients, Adam), the coarse model samples the LR planes;
          --refine-what joint (what = ['LR_planes', 'decoder', 'SR'], the YAMLs' value; default) | sr (what = ['SR']); --refine-scene llff: configs[4] as
          written (LLFF-'fern'-like forward-facing 378x504 view, NDC rays, 64+128 samples; `other_workloads.refine_llff_ndc`).  The line carries the
          split of an iteration (regions of interest / SR forward / render / SR backward / optimizers) and the crops' algorithmic FLOP.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_EVAL = 259072          # 129 536 MAC per decoded point
GATHER_BYTES_PER_EVAL = 3072    # 16 texels x 48 ch x 4 B
PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32
PEAK_BF16_MFMA_TFLOPS = 2516.6  # dense bf16: 256 CUs x 4 SIMDs x 1024 FLOP/clk (v_mfma_f32_32x32x16_bf16 in 32 clk) x 2.4 GHz
# What bare MFMA streams sustain on this chip (profiles/r03_mfma_power_roof.txt, tools/mfma_power_roof.hip; builder-run, NOT measured by this process): at
# 100 % matrix-pipe occupancy the clock is a power limit -- 2.38 GHz on all-zero operands, 1.65 GHz on random bf16 operands.  Informational only:
# `roofline.frac` stays priced against the 2.4 GHz dense peak.
SUSTAINED_BF16_MFMA = {"tflops_random_operands": 1734.9, "tflops_zero_operands": 2493.3, "frac_of_peak_random": 0.689,
                       "tflops_random_operands_16x16x32_two_waves_per_simd": 2050.0,
                       # the f16 instructions of the 2-limb arithmetic sustain less (11-bit multipliers): third run of the same file
                       "f16_tflops_random_operands": 1609.2, "f16_tflops_random_operands_16x16x32_two_waves_per_simd": 1837.2,
                       "source": "profiles/r03_mfma_power_roof.txt (tools/mfma_power_roof.hip: bare v_mfma_f32_32x32x16_bf16 streams, 256 CUs; builder-run "
                                 "on another MI355X box, not measured by this process)"}
# Arithmetic of the fused render pass (include/nvsr.h NVSR_ARITH_*): kernel, executed MFMA work per algorithmic FLOP, pipe peak.
# The roofline peak of a limb mode is the bf16 pipe's dense peak divided by the bf16 products it spends per f32 product.
ARITHMETIC = {
    "f32": {"kernel": "render_pass2_kernel", "products": 1, "pipe_peak": PEAK_F32_MFMA_TFLOPS,
            "dtype": "f32 (v_mfma_f32_32x32x2_f32)"},
    "bf16x3": {"kernel": "render_pass3_kernel<3>", "products": 6, "pipe_peak": PEAK_BF16_MFMA_TFLOPS,
               "dtype": "f32 (GEMM operands split exactly into 3 bf16 limbs, 6 of 9 limb products on v_mfma_f32_32x32x16_bf16, f32 accumulation)"},
    "f16x2": {"kernel": "render_pass3_kernel<2>", "products": 3, "pipe_peak": PEAK_BF16_MFMA_TFLOPS,      # (f16 and bf16 MFMAs run at the same rate)
              "dtype": "f32 (GEMM operands split into 2 round-to-nearest f16 limbs: |x - hi - lo| <= 2^-23 |x|; 3 of 4 limb products on "
                       "v_mfma_f32_32x32x16_f16, f32 accumulation; as close to float64 as the 3-bf16-limb arithmetic or closer, include/nvsr.h)"},
}
CAMERA_ANGLE_X = 0.6911112

# ---- the line the driver parses -------------------------------------------------------------------------------------------------------
# Round 5's line had grown to 28 KB and the driver could not parse it (BENCH_r05.json: parsed null).  The LAST stdout line is now a compact
# record of at most COMPACT_LIMIT characters; the full record goes to bench_full.json next to this script (and to gpurun_out/ when that
# directory exists) and to stderr -- never after the compact line.  tests/test_host.py holds the size and the keys on a maximal record.
COMPACT_LIMIT = 4096
SHORT_DTYPE = {"f32": "f32", "bf16x3": "f32 as 3 bf16 limbs (6 MFMA products, f32 acc)", "f16x2": "f32 as 2 f16 limbs (3 MFMA products, f32 acc)"}


def _short(v, sig=6):
    """floats to `sig` significant digits (integers and everything else untouched; `value` / `ms_per_step` keep every digit: the driver and
    the rehearsal tests re-derive one from the other)"""
    if isinstance(v, float):
        return float("%.*g" % (sig, v)) if np.isfinite(v) else None
    if isinstance(v, dict):
        return {k: (x if k in ("value", "ms_per_step") and isinstance(x, float) and np.isfinite(x) else _short(x, sig)) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_short(x, sig) for x in v]
    return v


def _compact_roofline(r):
    if not r:
        return None
    out = {"kernel": str(r.get("kernel_short") or r.get("kernel", ""))[:72]}
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "algorithmic_flop_per_launch", "algorithmic_flop_per_step",
              "algorithmic_bytes_per_launch", "algorithmic_bytes_per_step"):
        if k in r:
            out[k] = r[k]
    return out


def compact_record(full):
    """The driver's line: the contract's keys + roofline + cpu_baseline + one row per other workload, short strings only."""
    arith = full.get("arithmetic") or full.get("decoder_arithmetic") or full.get("conv_arithmetic")
    cfg = full.get("config", {})
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline")}
    out["metric"] = str(out["metric"])[:120]
    out["dtype"] = SHORT_DTYPE.get(arith, str(full.get("dtype", ""))[:48])
    out["data"] = full.get("data", "synthetic")
    out["config"] = {"workload": str(cfg.get("workload_short") or cfg.get("workload", ""))[:160]}
    for k in ("rays_per_step_per_gpu", "planes_per_step_per_gpu", "partition_short", "partition"):
        if k in cfg and not (k == "partition" and "partition" in out["config"]):
            out["config"]["partition" if k == "partition_short" else k] = cfg[k] if not isinstance(cfg[k], str) else cfg[k][:48]
    out["roofline"] = _compact_roofline(full.get("roofline"))
    rc = full.get("roofline_counters")
    if out["roofline"] and rc:                  # matrix-pipe busy fraction and clock of the same launch (committed rocprofv3 --pmc pass, hash-tied)
        out["roofline"]["mfma_busy_frac"], out["roofline"]["clock_ghz"] = rc.get("mfma_busy_frac"), rc.get("clock_ghz")
    cb = full.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {k: (cb[k] if k != "sample" else str(cb.get("sample_short") or cb[k])[:100])
                               for k in ("value", "unit", "cores", "kind", "sample") if k in cb}
    for k in ("psnr_vs_oracle_db", "sharded_frame_identical_to_one_gpu", "decoder_evals_per_s_per_gpu", "host_issue_ms_per_step"):
        if k in full:
            out[k] = full[k]
    c = full.get("collectives")
    if c:
        out["collectives"] = {"backend": c.get("backend"), "rccl_ranks": c.get("rccl_ranks"), "world_size": c.get("world_size")}
    ow = full.get("other_workloads")
    if ow:
        rows = {}
        for name, r in ow.items():
            if "error" in r:
                rows[name] = {"error": str(r["error"])[:80]}
                continue
            rf = r.get("roofline") or {}
            rows[name] = {"value": r.get("value"), "unit": r.get("unit"), "ms_per_step": r.get("ms_per_step"), "roofline_bound": rf.get("bound"),
                          "roofline_frac": rf.get("frac"), "traffic": rf.get("traffic"),
                          "algorithmic_bytes": rf.get("algorithmic_bytes_per_step", rf.get("algorithmic_bytes_per_launch")),
                          "cpu_baseline_value": (r.get("cpu_baseline") or {}).get("value")}
        out["other_workloads"] = rows
    out["full_record"] = "bench_full.json"
    out = _short(out)
    line = json.dumps(out, separators=(",", ":"))
    if len(line) > COMPACT_LIMIT:            # never the reason a line cannot be parsed: drop the optional parts, widest first
        for k in ("other_workloads", "collectives", "host_issue_ms_per_step"):
            out.pop(k, None)
            line = json.dumps(out, separators=(",", ":"))
            if len(line) <= COMPACT_LIMIT:
                break
    assert len(line) <= COMPACT_LIMIT and "\n" not in line, len(line)
    return line


def emit(full, path=None):
    """full record -> bench_full.json (+ gpurun_out/bench_full.json; + `path` = --full-record) and stderr; compact record -> the last stdout line"""
    text = json.dumps(full)
    for f in [os.path.join(d, "bench_full.json") for d in (ROOT, os.path.join(ROOT, "gpurun_out")) if os.path.isdir(d)] + ([path] if path else []):
        try:
            with open(f, "w") as fh:
                fh.write(text + "\n")
        except OSError as e:
            print("bench.py: could not write %s: %s" % (f, e), file=sys.stderr)
    print("BENCH_FULL_RECORD " + text, file=sys.stderr, flush=True)
    print(compact_record(full), flush=True)


class Opt:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def render_options(nc, nf, perturb=False, noise=0.0, white=False):
    m = Opt(chunksize=131072, perturb=perturb, num_coarse=nc, num_fine=nf, white_background=white,
            radiance_field_noise_std=noise, lindisp=False)
    return Opt(nerf=Opt(use_viewdirs=True, train=m, validation=m)), Opt(near=2.0, far=6.0, no_ndc=True)


def pose_spherical(theta, phi, radius):
    """Blender-style camera-to-world (the reference's load_blender.pose_spherical, load_blender.py:34-39)"""
    t = np.eye(4); t[2, 3] = radius
    p = np.deg2rad(phi)
    rp = np.array([[1, 0, 0, 0], [0, np.cos(p), -np.sin(p), 0], [0, np.sin(p), np.cos(p), 0], [0, 0, 0, 1.0]])
    th = np.deg2rad(theta)
    rt = np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1.0]])
    c2w = rt @ rp @ t
    c2w = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1.0]]) @ c2w
    return c2w.astype(np.float32)


def make_synthetic_scene(dev, plane_res=800, view_res=32, seed=0, theta=30.0, channels_last=False):
    """Random-init decoder pair + random planes (no dataset / checkpoint is reachable), calibrated so that the density is
    neither empty nor saturated (SURVEY.md 7 'degenerate synthetic scenes')."""
    import nvsr_amd
    M = nvsr_amd.models
    torch.manual_seed(seed)
    sid = M.get_scene_id("lego", 1, (plane_res, view_res))
    kw = dict(use_viewdirs=True, skip_connect_every=3, proj_combination="avg", viewdir_proj_combination="concat_pos", align_corners=True)
    mc = M.TwoDimPlanesModel(**kw)
    mf = M.TwoDimPlanesModel(num_planes_or_rot_mats=mc.rot_mats(), **kw)
    # band-limited random planes (random res/8 grids, bilinearly up-sampled): trained feature planes are smooth at texel scale,
    # white noise at 800^2 would make the radiance field a chaotic function of depth.  Values do not change the work done.
    def smooth_plane(res):
        src = max(res // 8, 4)
        low = 0.7 * torch.randn(1, 48, src, src, device=dev)
        p = torch.nn.functional.interpolate(low, size=(res, res), mode="bilinear", align_corners=True).contiguous()
        # channels_last: the same [1,48,R,R] parameter in the kernels' native memory order (models.create_plane(channels_last=True))
        return torch.nn.Parameter(p.contiguous(memory_format=torch.channels_last) if channels_last else p)
    mc, mf = mc.to(dev), mf.to(dev)
    planes = torch.nn.ParameterDict({M.get_plane_name(sid, d): smooth_plane(plane_res if d < 3 else view_res) for d in range(4)})
    box = torch.tensor([[-4.0, -4, -4, -np.pi, -np.pi / 2], [4, 4, 4, np.pi, np.pi / 2]], dtype=torch.float64)
    g = torch.Generator().manual_seed(seed + 1)
    pts = torch.rand(8192, 3, generator=g) * 6 - 3
    d = torch.randn(8192, 3, generator=g)
    x = torch.cat([pts, d / d.norm(dim=-1, keepdim=True)], -1).to(dev)
    for m in (mc, mf):
        m.planes_, m.box_coords = planes, {sid: box}
        m.set_cur_scene_id(sid)
        m.eval()
        with torch.no_grad():
            raw = m(x)[:, 3]
            scale = 1.0 / float(raw.std())
            m.fc_alpha["0"].weight.mul_(scale)
            m.fc_alpha["0"].bias.mul_(scale)
            raw = m(x)[:, 3]
            m.fc_alpha["0"].bias.add_(-float(raw.mean()) - 0.5)
    pose = torch.from_numpy(pose_spherical(theta, -30.0, 4.0)).to(dev)
    return mc, mf, sid, pose


def time_fine_pass_kernel(nvsr_amd, mf, rays, z_fine, reps=3):
    """Average duration of the dominant kernel (fused fine render pass, S = Nc+Nf) measured with HIP events on the launch stream."""
    import ctypes as C
    capi = nvsr_amd.capi
    N, S = z_fine.shape
    dev = rays.device
    rgb = torch.empty((N, 3), device=dev); disp = torch.empty(N, device=dev); acc = torch.empty(N, device=dev)
    sc, keep = mf.native_scene()
    packed = mf.packed_decoder()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        capi.call("nvsr_render_pass", C.byref(sc), capi.ptr(packed), N, S, capi.ptr(rays), capi.ptr(z_fine), None, 0, capi.ptr(rgb),
                  capi.ptr(disp), capi.ptr(acc), None, None, capi.stream())
        b.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in ev])) * 1e-3


PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def csrc_tree_hash():
    """sha256 over the kernel sources (csrc/*.hip, csrc/*.h, include/nvsr.h; names + contents): identifies the kernels a counter pass was
    taken with.  (`git rev-parse HEAD:.../csrc` would do the same, but the GPU box receives a snapshot without .git.)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "neural-volume-super-resolution_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))) + [os.path.join(ROOT, "include", "nvsr.h")]:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def pmc_record():
    """profiles/pmc_latest.json if -- and only if -- its counters were taken with the kernels of THIS tree (csrc_sha256 recorded by
    tools/profile_collect.py): a stale file must not put another kernel's traffic into the line.  Else None."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if not os.path.exists(path):
        return None
    rec = json.load(open(path))
    return rec if rec.get("csrc_sha256") == csrc_tree_hash() else None


def pmc_source():
    """what a reader must know about every number of the line that comes from profiles/pmc_latest.json: NOT measured by this process"""
    rec = pmc_record()
    if rec is None:
        return None
    return ("profiles/pmc_latest.json (round %s): builder-run rocprofv3 --pmc passes of this same command on another MI355X box, not measured by "
            "this process; valid for this tree only -- csrc_sha256 %s... matches the kernel sources that ran" % (rec.get("round", "?"), rec.get("csrc_sha256", "")[:16]))


def pmc_traffic(workload, dtype):
    """HBM bytes of the workload's dominant kernel from the committed rocprofv3 --pmc passes of this same command (profiles/pmc_latest.json,
    FETCH_SIZE x 2 + WRITE_SIZE as the guide prescribes) -- counters cannot be read from inside this process.  None when the passes were
    taken with another arithmetic or with other kernel sources."""
    rec = pmc_record()
    if rec is None:
        return None
    e = rec.get(workload)
    return e["traffic_bytes"] if e and e.get("arithmetic") == dtype else None


def hbm_stage_rates(nvsr_amd, H, W, focal, pose, ro, rd, rays, ws, reps=5):
    """The bandwidth-bound helper kernels of the frame (ray generation, ray packing, coarse depths, importance resampling), each timed
    alone with events on the launch stream: algorithmic bytes (inputs read once + outputs written once) / duration against the HBM peak."""
    capi = nvsr_amd.capi
    N = rays.shape[0]
    z_c, w_c = ws[:N * 64], ws[N * 64: 2 * N * 64]
    z_f = torch.empty(N * 192, device=rays.device)

    def timed(fn):
        fn()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e-3

    stages = {
        "get_ray_bundle (ray_bundle_kernel)": (24 * N, lambda: nvsr_amd.nerf_helpers.get_ray_bundle(H, W, focal, pose)),
        "pack_rays (pack_rays_kernel)": ((24 + 44) * N, lambda: nvsr_amd.train_utils.pack_rays(ro, rd, 2.0, 6.0)),
        # (not part of this frame any more: without stratified jitter the limb passes compute the coarse depths in registers; the kernel remains
        #  for training batches and the exact-f32 arithmetic)
        "coarse depths (coarse_z_kernel; NOT launched by an inference frame)": ((8 + 4 * 64) * N, lambda: capi.call("nvsr_coarse_z", N, 64, capi.ptr(rays), 0, None, capi.ptr(z_c), capi.stream())),
        "sample_pdf + sort (importance_resample_kernel)": (4 * (64 + 64 + 192) * N, lambda: capi.call(
            "nvsr_importance_resample", N, 64, 128, capi.ptr(z_c), capi.ptr(w_c), None, capi.ptr(z_f), capi.stream())),
    }
    out = {}
    for name, (nbytes, fn) in stages.items():
        dt = timed(fn)
        out[name] = {"ms": dt * 1e3, "algorithmic_bytes": nbytes, "GB/s": nbytes / dt / 1e9, "frac_of_hbm_peak": nbytes / dt / 1e9 / PEAK_HBM_GBS}
    return out


def decoder_error_by_arithmetic(nvsr_amd, mf, sid, rays, z_fine, n_rays=16384, n_check=1024):
    """Decoder outputs (rgb logits, sigma) of one fused fine pass in every arithmetic against the float64 checker AT THE SAME DEPTHS, on
    the first n_check of n_rays rays of the frame: max / rms error relative to the output range.  (The frame PSNR regenerates the fine
    depths per arithmetic, so rays whose importance samples flip bins dominate it; this is the arithmetic alone.)"""
    import ctypes as C
    from oracle.oracle import Oracle, decoder_blob
    capi = nvsr_amd.capi
    dev = rays.device
    r, z = rays[:n_rays].contiguous(), z_fine[:n_rays].contiguous()
    N, S = z.shape
    chk = Oracle(f32=False)
    planes = [mf.planes_[nvsr_amd.models.get_plane_name(sid, d)].detach().cpu().numpy() for d in range(4)]
    osc = chk.scene(planes, mf.box_coords[sid].numpy())
    dec = chk.decoder(decoder_blob({k: v.detach().cpu().numpy() for k, v in mf.state_dict().items()}))
    ref = chk.render_given_z(osc, dec, r[:n_check].cpu().numpy(), z[:n_check].cpu().numpy(), want_raw=True)["raw"].astype(np.float64)
    sc, keep = mf.native_scene()
    out = {}
    for mode in ("f32", "bf16x3", "f16x2"):
        o = [torch.empty((N, 3), device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev)]
        raw = torch.empty((N, S, 4), device=dev)
        capi.call("nvsr_render_pass_arith", C.byref(sc), capi.ptr(mf.packed_decoder()), N, S, capi.ptr(r), capi.ptr(z), None, 0,
                  *[capi.ptr(b) for b in o], None, None, capi.ptr(raw), capi.ARITHMETIC[mode], capi.stream())
        torch.cuda.synchronize()
        d = raw[:n_check].cpu().numpy().astype(np.float64) - ref
        e = {}
        for name, ch in (("rgb_logits", slice(0, 3)), ("sigma", slice(3, 4))):
            rng = float(np.abs(ref[..., ch]).max())
            e[name] = {"max": float(np.abs(d[..., ch]).max() / rng), "rms": float(np.sqrt((d[..., ch] ** 2).mean()) / rng),
                       "mean": float(d[..., ch].mean() / rng)}
        out[mode] = e
    out["note"] = ("|raw - float64 checker| / range over %d rays x %d samples of the frame's fine pass, same depths for every arithmetic "
                   "(render_pass3 / render_pass2 kernels through nvsr_render_pass_arith)" % (n_check, S))
    return out


FLIP_DEPTH_TOL = 1e-3      # a ray whose fine depths differ from the checker's by more than this (1.6 % of a coarse bin of the 2..6 range) "flipped"


def frame_error_evidence(nvsr_amd, mc, mf, sid, rays, n_rays=16384, nc=64, nf=128, seed=0, budget_s=None):
    """Where a frame's error against the float64 checker comes from, per arithmetic (the default arithmetic's frame PSNR
    sat 3.8 dB under the exact-f32 kernels' although its decoder outputs are closer to float64).  n_rays rays of the frame are rendered pass
    by pass (the kernels and depths of the frame path) in every arithmetic and by the checker (C oracle, double accumulation); per arithmetic:
      flipped rays   rays with a fine depth more than FLIP_DEPTH_TOL from the checker's: an importance sample landed in another coarse bin
                     (inverse-CDF sampling is discontinuous in the coarse weights: nerf_helpers.py:688-700; a rounding-level change of a
                     weight moves a sample by a bin where u meets a knot of the cdf) -- such a ray's pixel is a DIFFERENT correct render,
                     its error is the field's variation along the ray, not the arithmetic's;
      |rgb error|    50 / 99 / 99.9-th percentile and maximum over the checked rays (max over the 3 channels);
      psnr           over all checked rays, over the rays that did not flip, and over the rays that flipped in NO arithmetic (same set for all)."""
    from oracle.oracle import Oracle, decoder_blob
    nv = torch.ops.nvsr
    capi = nvsr_amd.capi
    N = rays.shape[0]
    rng = np.random.default_rng(seed)
    ids = np.sort(rng.choice(N, size=min(N, n_rays), replace=False))
    r = rays[torch.from_numpy(ids).to(rays.device)].contiguous()
    chk = Oracle(f32=False)
    planes = [mc.planes_[nvsr_amd.models.get_plane_name(sid, d)].detach().cpu().numpy() for d in range(4)]
    osc = chk.scene(planes, mc.box_coords[sid].numpy())
    dc = chk.decoder(decoder_blob({k: v.detach().cpu().numpy() for k, v in mc.state_dict().items()}))
    df = chk.decoder(decoder_blob({k: v.detach().cpu().numpy() for k, v in mf.state_dict().items()}))
    if budget_s is not None and len(ids) > 2048:
        # bounded checker time (the default bench run must stay within minutes on any host): probe 1 024 rays, keep what fits the budget
        t0 = time.perf_counter()
        chk.render_rays(osc, dc, df, r[:1024].cpu().numpy(), nc, nf)
        keep = int(min(len(ids), max(2048, 1024 * budget_s / max(time.perf_counter() - t0, 1e-3)))) // 256 * 256
        sel = np.linspace(0, len(ids) - 1, keep).astype(np.int64)            # (evenly over the sorted sample: no image region preferred)
        ids, r = ids[sel], r[torch.from_numpy(sel).to(r.device)].contiguous()
    t0 = time.perf_counter()
    ref = chk.render_rays(osc, dc, df, r.cpu().numpy(), nc, nf, want_aux=True)
    t_ref = time.perf_counter() - t0
    ref_rgb, ref_z = ref["rgb_fine"].astype(np.float64), ref["z_fine"].astype(np.float64)
    planes_c, consts = mc.scene_args()
    planes_f, _ = mf.scene_args()
    res, flipped = {}, {}
    for mode in ("f32", "bf16x3", "f16x2"):
        a = capi.ARITHMETIC[mode]
        z_c = nv.coarse_z(r, nc, False, None)
        _, _, _, w_c = nv.render_pass(planes_c, consts, mc.packed_decoder(), r, z_c, None, False, True, a)
        z_f = nv.importance_resample(z_c, w_c, nf, None)
        rgb_f = nv.render_pass(planes_f, consts, mf.packed_decoder(), r, z_f, None, False, False, a)[0]
        torch.cuda.synchronize()
        dz = np.abs(z_f.cpu().numpy().astype(np.float64) - ref_z).max(-1)
        err = np.abs(rgb_f.cpu().numpy().astype(np.float64) - ref_rgb)
        flipped[mode] = dz > FLIP_DEPTH_TOL
        res[mode] = (err, dz, np.abs(w_c.cpu().numpy().astype(np.float64) - ref["weights_coarse"]).max())
    none_flipped = ~(flipped["f32"] | flipped["bf16x3"] | flipped["f16x2"])

    def psnr(err, keep):
        e = err[keep]
        mse = float(np.mean(e ** 2)) if e.size else 0.0
        return 200.0 if mse == 0 else -10.0 * np.log10(mse)

    out = {"rays_checked": int(len(ids)), "flip_depth_tolerance": FLIP_DEPTH_TOL, "rays_flipped_in_no_arithmetic": int(none_flipped.sum()),
           "checker": "C oracle, double accumulation, %d rays in %.1f s" % (len(ids), t_ref)}
    for mode, (err, dz, dw) in res.items():
        e1 = err.max(-1)
        fl = flipped[mode]
        out[mode] = {"flipped_rays": int(fl.sum()), "flipped_fraction": float(fl.mean()),
                     "coarse_weight_max_abs_error": float(dw),
                     "rgb_abs_error_percentiles": {"p50": float(np.percentile(e1, 50)), "p99": float(np.percentile(e1, 99)),
                                                   "p99.9": float(np.percentile(e1, 99.9)), "max": float(e1.max())},
                     "rgb_abs_error_max_over_non_flipped": float(e1[~fl].max()) if (~fl).any() else 0.0,
                     "psnr_db_all": psnr(err, np.ones_like(fl)), "psnr_db_non_flipped": psnr(err, ~fl),
                     "psnr_db_rays_flipped_in_no_arithmetic": psnr(err, none_flipped),
                     "share_of_squared_error_in_flipped_rays": float((err[fl] ** 2).sum() / max((err ** 2).sum(), 1e-300))}
    return out


def cpu_baseline(nvsr_amd, mc, mf, sid, rays, rgb_fine_gpu, budget_s=15.0):
    """The oracle (plain-C port of the reference algorithm, fp32, OpenMP over all host cores) timed on a bounded sample of the same rays.
    (The PSNR of the GPU pixels against the double-accumulating checker comes from frame_error_evidence's larger sample.)"""
    from oracle.oracle import Oracle, decoder_blob
    fast = Oracle(f32=True)
    planes = [mc.planes_[nvsr_amd.models.get_plane_name(sid, d)].detach().cpu().numpy() for d in range(4)]
    box = mc.box_coords[sid].numpy()
    sdc = {k: v.detach().cpu().numpy() for k, v in mc.state_dict().items()}
    sdf = {k: v.detach().cpu().numpy() for k, v in mf.state_dict().items()}
    N = rays.shape[0]
    rng = np.random.default_rng(0)
    ids = np.sort(rng.choice(N, size=min(N, 262144), replace=False))
    rays_np = rays[torch.from_numpy(ids).to(rays.device)].cpu().numpy()

    def run(o, n):
        sc = o.scene(planes, box)
        t0 = time.perf_counter()
        out = o.render_rays(sc, o.decoder(decoder_blob(sdc)), o.decoder(decoder_blob(sdf)), rays_np[:n], 64, 128)
        return time.perf_counter() - t0, out

    run(fast, 256)                                  # warm-up (OpenMP team, page faults)
    t_probe, _ = run(fast, 2048)
    n = int(min(len(ids), max(2048, 2048 * budget_s / max(t_probe, 1e-3))))
    t, _ = run(fast, n)
    cores = os.cpu_count() or 1
    return {"value": n / t, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": "%d rays of the same 800x800 / 64+128 / planes 800^2 frame, %.1f s, C oracle fp32 -Ofast OpenMP (%d threads)" % (n, t, cores),
            "sample_short": "%d rays of the same frame, %.1f s, C oracle fp32 OpenMP %d threads" % (n, t, cores),
            "evals_per_s": n * 256 / t,
            # the reference ITSELF never travels to the GPU box; its only CPU figure is the survey's
            "reference_on_cpu": {"value": 10000 / 9.79, "unit": "rays/s", "cores": 8, "kind": "reference",
                                 "sample": "the reference's own PyTorch path, 100x100 rays of this configuration (64+128 samples, planes 800^2) in 9.79 s on the "
                                           "8 Xeon cores of the build container (BASELINE.md section 2 / SURVEY.md 8d) -- quoted, not measured by this process"}}


def _sync_time(dist, dev, fn, warmup, steps):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    _sync_time.issue_s = time.perf_counter() - t0        # the host's share: enqueueing the K steps (the GPU is still working on them)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def train_partition_check(nvsr_amd, dist, dev, rank, world, mc, mf, sid, scfg, opts, pose, H, W, focal, N, Nc, Nf, planes, dec, what):
    """Rehearsal only (NVSR_BENCH_REHEARSAL=1, tests/test_hip_round3.py): the 8e training partition is exact -- the gradients of a step on
    this rank's N rays, averaged over the ranks by distributed.allreduce_gradients, equal the one-rank gradients of the step on all
    world x N rays (every rank computes that reference itself).  Tolerance: the order of the float atomics of the plane scatter
    (test_plane_gradients_vs_oracle_larger: relative L2 2e-3; here the same kernels on both sides: 1e-5) -- decoders: 1e-4."""
    G = N * world
    g = torch.Generator(device=dev).manual_seed(4242)
    sel = torch.randint(0, H, (G, 2), device=dev, generator=g)
    target = torch.rand(G, 3, device=dev, generator=g)
    rnd = dict(t_rand=torch.rand(G, Nc, device=dev, generator=g), u=torch.rand(G, Nf, device=dev, generator=g),
               noise_coarse=0.2 * torch.randn(G, Nc, device=dev, generator=g), noise_fine=0.2 * torch.randn(G, Nc + Nf, device=dev, generator=g))
    params = planes + (dec if "decoder" in what else [])

    def grads_of(lo, hi):
        for p in params:
            p.grad = None
        ro, rd = nvsr_amd.training.get_ray_bundle_at(H, W, focal, pose, sel[lo:hi])
        out = nvsr_amd.train_utils.run_one_iter_of_nerf(H, W, focal, mc, mf, torch.stack([ro, rd], 0), opts, sid, mode="train", scene_config=scfg,
                                                        randoms={k: v[lo:hi].contiguous() for k, v in rnd.items()})
        loss = torch.nn.functional.mse_loss(out[0], target[lo:hi]) + torch.nn.functional.mse_loss(out[3], target[lo:hi])
        loss.backward()
        return [p.grad for p in params]

    ref = [t.clone() for t in grads_of(0, G)]
    mine = grads_of(rank * N, (rank + 1) * N)
    nvsr_amd.distributed.allreduce_gradients(mine)
    worst = 0.0
    for i, (a, b) in enumerate(zip(mine, ref)):
        rel = float((a - b).norm() / b.norm().clamp_min(1e-30))
        worst = max(worst, rel)
        tol = 1e-5 if i < len(planes) else 1e-4
        if not rel <= tol:
            print("TRAIN_GRADIENTS_DIFFER rank %d tensor %d rel %.3e" % (rank, i, rel), file=sys.stderr, flush=True)
            sys.exit(3)
    for p in params:
        p.grad = None
    print("TRAIN_GRADIENTS_MATCH rank %d of %d: %d tensors, worst relative L2 difference %.2e" % (rank, world, len(params), worst), file=sys.stderr, flush=True)


def bench_train(args, nvsr_amd, dist, dev, rank, world):
    """4096 rays / 64+64 / planes 200^2: forward + backward (planes + both decoders) + Adam"""
    import ctypes as C
    capi = nvsr_amd.capi
    R, N, Nc, Nf = 200, 4096, 64, 64
    # --rays-global G (SURVEY.md 8e "Training partition": 4096 rays -> 4096 / N per GPU, same scene and same pixels on all ranks): every
    # rank draws the SAME G pixels and random numbers (a generator seeded alike) and keeps its contiguous G / world share -> strong scaling.
    # Default: every rank draws its own 4096 rays (global batch world x 4096) -> weak scaling.
    strong = args.rays_global is not None
    if strong:
        G = int(args.rays_global)
        if G % world:
            sys.exit("bench.py: --rays-global %d is not divisible by %d ranks (the loss is a mean over the local rays)" % (G, world))
        N = G // world
    what = {"planes": {"LR_planes"}, "planes+decoder": {"LR_planes", "decoder"}}[args.train_what]
    mc, mf, sid, pose = make_synthetic_scene(dev, R, 32, seed=0, theta=30.0, channels_last=not args.nchw_planes)
    for m in (mc, mf):
        for n, p in m.named_parameters():
            p.requires_grad_("rot_mats" not in n and ("planes_" in n or "decoder" in what))
        m.train()
    H = W = 800
    focal = 0.5 * W / np.tan(0.5 * CAMERA_ANGLE_X)
    opts, scfg = render_options(Nc, Nf, perturb=True, noise=0.2)
    g = torch.Generator(device=dev).manual_seed(100 + (0 if strong else rank))
    target = torch.rand(H, W, 3, device=dev, generator=g)
    lo = rank * N if strong else 0                    # this rank's share of a global draw of n_draw rays
    n_draw = N * world if strong else N
    dec = list({id(p): p for m in (mc, mf) for p in m.decoder_parameters()}.values())
    planes = list(mc.planes_.values())
    # (fused=True: one kernel per parameter group instead of five multi-tensor passes over the 23 MB of planes)
    # one rank: the iteration is captured into a HIP graph and replayed (training.GraphedTrainStep); --no-graph and multi-rank runs launch it
    # kernel by kernel (the RCCL all-reduce of a multi-rank step has never run on this pool: it stays out of a capture)
    graphed = world == 1 and not args.no_graph
    opt, popt = (torch.optim.Adam(dec, lr=5e-4, fused=True, capturable=graphed) if "decoder" in what else None), \
        torch.optim.Adam(planes, lr=4e-3, fused=True, capturable=graphed)
    sync = (lambda: nvsr_amd.distributed.allreduce_gradients([p.grad for p in planes + dec if p.grad is not None])) if world > 1 else None
    # uniform without replacement like the reference's np.random.choice(H*W, n, replace=False) (train_nerf.py:836-838), drawn on the device
    # by the library's sampler (one kernel: pixels + targets): the host permutation of 640 000 indices costs more than the whole GPU step
    device_sampler = nvsr_amd.training.DevicePixelSampler(seed=100 + (0 if strong else rank), n_draw=n_draw if strong else None, lo=lo)

    step = nvsr_amd.training.TrainStep(mc, mf, opts, what, optimizer=opt, planes_optimizer=popt, grad_sync=sync, pixel_sampler=device_sampler)
    np.random.seed(rank)
    it = [0]

    def draw():
        # the random draws of the train mode come from the device generator here: the reference draws them on the host, which on
        # this box costs more than the whole GPU step
        rnd = dict(t_rand=torch.rand(n_draw, Nc, device=dev, generator=g), u=torch.rand(n_draw, Nf, device=dev, generator=g),
                   noise_coarse=torch.empty(n_draw, Nc, device=dev).normal_(0.0, 0.2, generator=g),
                   noise_fine=torch.empty(n_draw, Nc + Nf, device=dev).normal_(0.0, 0.2, generator=g))
        if strong:
            rnd = {k: v[lo: lo + N].contiguous() for k, v in rnd.items()}
        return rnd

    def one():
        step(it[0], target, pose, H, W, focal, 1, sid, scfg, N, randoms=draw())
        it[0] += 1

    launch = "kernel by kernel"
    if graphed:
        # one HIP-graph replay per iteration takes the host out of the step (0.03 ms instead of ~1 ms) but starts every kernel node ~1 us later
        # than a launch that was already queued: with a host that keeps up the eager iteration is 2-5 % FASTER.  Both are probed here, outside
        # the timed region, and the faster one runs it; the line says which, with both probe times.
        eager_one = one
        replay = nvsr_amd.training.GraphedTrainStep(step, target, pose, H, W, focal, 1, sid, scfg, N, randoms_fn=draw, generators=(g,))

        def probe(fn, k=25):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k):
                fn()
            t_issue = time.perf_counter() - t0
            torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t0) / k, 1e3 * t_issue / k

        probes = {"eager": probe(eager_one), "graph": probe(replay)}
        one = replay if probes["graph"][0] <= probes["eager"][0] else eager_one
        launch = ("one HIP graph replay per iteration (training.GraphedTrainStep)" if one is replay else "kernel by kernel (the eager iteration)") + \
                 "; probes before the timed region, ms per iteration (host time to enqueue it): eager %.3f (%.3f), graph replay %.3f (%.3f)" \
                 % (probes["eager"] + probes["graph"])

    if strong and world > 1 and os.environ.get("NVSR_BENCH_REHEARSAL", "0") == "1":
        train_partition_check(nvsr_amd, dist, dev, rank, world, mc, mf, sid, scfg, opts, pose, H, W, focal, N, Nc, Nf, planes, dec, what)

    elapsed = _sync_time(dist, dev, one, args.warmup, args.steps)
    value = world * N * args.steps / elapsed
    host_issue_ms = 1e3 * _sync_time.issue_s / args.steps
    label = "planes + decoder gradients" if "decoder" in what else "plane gradients (Feature_Planes_Only.yml: what = ['LR_planes'])"
    par = ("%d rays per iteration split %d per rank (same pixels and random numbers on every rank: the step equals the one-rank step on the "
           "whole batch)" % (N * world, N)) if strong else "every rank draws its own %d rays (global batch %d)" % (N, N * world)
    result = {"metric": "training rays/sec (4096 rays/iter, 64+64 samples, planes 200^2, %s, Adam)" % label, "value": value,
              "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
              "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
              # time the host needed to ENQUEUE a step (python + launches; the GPU runs behind): a value near ms_per_step = the step waits for the host
              "host_issue_ms_per_step": host_issue_ms, "arithmetic": capi.get_decoder_arithmetic(),
              "launch": launch,
              "dtype": {"f32": ARITHMETIC["f32"]["dtype"], "bf16x3": ARITHMETIC["bf16x3"]["dtype"],
                        "f16x2": "f32 (forward -- with or without the weight-gradient record -- and gate-driven backward of every pass: GEMM operands split "
                                 "into 2 round-to-nearest f16 limbs, 3 products, the backward with a power-of-two scale per point and chain; the "
                                 "weight-gradient contraction: 3 exact bf16 limbs, 6 products; f32 accumulation)"}[capi.get_decoder_arithmetic()],
              "data": "synthetic",
              "config": {"workload": "train step: 4096 random rays of an 800x800 view, 64 coarse + 64 fine samples, 3x200^2x48 + 32^2x48 planes, "
                                     "what = %s, Adam" % sorted(what), "rays_per_step_per_gpu": N, "train_what": args.train_what,
                         "plane_memory_format": "NCHW" if args.nchw_planes else "channels_last (same [1,48,R,R] parameters, native [H][W][C] memory)",
                         "partition": par,
                         "parallelism": "rays sharded by rank; in-place asynchronous all-reduces of the %s (one per tensor, one wait) per step"
                                        % ("plane + decoder gradients" if "decoder" in what else "plane gradients (23 MB)")}}
    if rank == 0:
        # dominant kernel: gate-driven backward of the fine pass (transposed layers + plane scatter + gradient half of the record), S = 128
        batch = torch.stack(nvsr_amd.training.get_ray_bundle_at(H, W, focal, pose, torch.randint(0, H, (N, 2), device=dev)), 0)
        out = nvsr_amd.train_utils.run_one_iter_of_nerf(H, W, focal, mc, mf, batch, opts, sid, mode="train", scene_config=scfg, randoms={})
        sv = out[3].grad_fn.saved
        S = Nc + Nf
        rays = nvsr_amd.train_utils.pack_rays(batch[0], batch[1], 2.0, 6.0)
        g_raw = torch.randn(N, S, 4, device=dev) * 1e-3
        sc, keep = mf.native_scene()
        gpl = [torch.zeros_like(k) for k in keep]
        gptrs = (C.c_void_p * 4)(*[t.data_ptr() for t in gpl])
        vws = torch.empty(capi.lib().nvsr_view_grad_workspace_floats(N, S), device=dev)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
        for a, b in ev:
            a.record()
            capi.call("nvsr_render_pass_backward_gates", C.byref(sc), capi.ptr(mf.packed_decoder()), capi.ptr(mf.packed_decoder_bwd()), N, S,
                      capi.ptr(rays), capi.ptr(sv["z_f"]), capi.ptr(g_raw), capi.ptr(sv["gates_f"]), gptrs, capi.ptr(vws), capi.ptr(sv.get("rec_f")),
                      capi.stream())
            b.record()
        torch.cuda.synchronize()
        dt = float(np.mean([a.elapsed_time(b) for a, b in ev])) * 1e-3
        flops = FLOP_PER_EVAL * N * S              # the transposed layers move exactly the forward's 129 536 MAC per point
        ach = flops / dt / 1e12
        limb = capi.get_decoder_arithmetic() != "f32"
        # products per f32 product of the launch just timed: 2 f16 limbs (3) under f16x2 (with or without the gradient half of the record), else 3 bf16 limbs (6)
        nprod = 3 if capi.get_decoder_arithmetic() == "f16x2" else 6
        peak = PEAK_BF16_MFMA_TFLOPS / nprod if limb else PEAK_F32_MFMA_TFLOPS
        result["roofline"] = {"kernel": "render_pass_backward_gates_%skernel<%s%s> (fine pass, S=128; incl. its view-plane reduce)"
                                        % ("limb_" if limb else "", "record" if sv.get("rec_f") is not None else "no record",
                                           (", %d limbs" % (2 if nprod == 3 else 3)) if limb else ""), "bound": "mfma",
                              "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                              "traffic": pmc_traffic("train_dec" if "decoder" in what else "train", result["dtype"]),
                              "kernel_ms": dt * 1e3, "algorithmic_flop_per_launch": flops,
                              "peak_note": ("algorithmic f32 FLOP; peak = %.1f TFLOP/s dense on the 16-bit pipe / %d MFMA products per f32 product" % (PEAK_BF16_MFMA_TFLOPS, nprod))
                                           if limb else "v_mfma_f32_32x32x2_f32 dense peak",
                              "vs_f32_mfma_peak": ach / PEAK_F32_MFMA_TFLOPS}
        result["roofline"]["traffic_source"] = None if result["roofline"]["traffic"] is None else pmc_source()
        if sv.get("rec_f") is not None:
            # with the weight-gradient record the launch also moves, per point: the 8 gradient rows of the record it writes (8 x 128 floats) + the
            # 16-byte head row, and the gate words (128 B), dL/draw (16 B) and depth (4 B) it reads = 4 260 B (the plane scatter not counted).
            # Whichever roof it stands closer to is reported as the bound; the other fraction stays beside it.
            nbytes = 4260.0 * N * S
            hbm = nbytes / dt / 1e9
            mfma = dict(achieved=ach, peak=peak, unit="TFLOP/s", frac=ach / peak)
            if hbm / PEAK_HBM_GBS > ach / peak:
                result["roofline"].update({"bound": "hbm", "achieved": hbm, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": hbm / PEAK_HBM_GBS,
                                           "algorithmic_bytes_per_launch": nbytes, "mfma": mfma,
                                           "peak_note": "algorithmic record / gate / gradient bytes of the launch over the 8 TB/s HBM peak (the matrix-pipe fraction of the "
                                                        "same launch is under 'mfma': " + result["roofline"]["peak_note"] + ")"})
            else:
                result["roofline"]["hbm"] = dict(achieved=hbm, peak=PEAK_HBM_GBS, unit="GB/s", frac=hbm / PEAK_HBM_GBS, algorithmic_bytes_per_launch=nbytes)
        if world == 1 and not args.no_cpu_baseline:
            from oracle.oracle import Oracle, decoder_blob
            o = Oracle(f32=False)
            pl = [mc.planes_[nvsr_amd.models.get_plane_name(sid, d)].detach().cpu().numpy() for d in range(4)]
            osc = o.scene(pl, mc.box_coords[sid].numpy())
            dc = o.decoder(decoder_blob({k: v.detach().cpu().numpy() for k, v in mc.state_dict().items()}))
            df = o.decoder(decoder_blob({k: v.detach().cpu().numpy() for k, v in mf.state_dict().items()}))
            n = 96 if getattr(args, "cpu_cache", None) is not None else 256      # ~10 s of single-threaded CPU work (~4 s as a side workload)
            rn = rays[:n].cpu().numpy()
            gg = np.full((n, 3), 1e-3, np.float32)
            t0 = time.perf_counter()
            o.render_rays(osc, dc, df, rn, Nc, Nf)
            if "decoder" in what:
                o.render_backward_decoder(osc, dc, df, rn, Nc, Nf, gg, gg)
            else:
                o.render_backward(osc, [p_.shape for p_ in pl], dc, df, rn, Nc, Nf, gg, gg)
            t = time.perf_counter() - t0
            result["cpu_baseline"] = {"value": n / t, "unit": "rays/s", "cores": 1, "kind": "port",
                                      "sample": "%d rays of the same step (forward + analytic backward: %s), %.1f s, C "
                                                "oracle, double accumulation, single thread" % (n, label, t)}
        return result


def bench_sr(args, nvsr_amd, dist, dev, rank, world):
    """3 planes 48 x 200^2 -> 48 x 800^2 through PlanesSR(EDSR hidden 256, 32 blocks, x4), one batched pass"""
    torch.manual_seed(0)
    M = nvsr_amd.models
    sr = M.PlanesSR(M.EDSR, 4, 48, 48, {"model": {"hidden_size": 256, "n_blocks": 32}}, "bilinear").to(dev)
    sr.eval()
    names = ["p0", "p1", "p2"]
    for n in names:
        sr.set_LR_plane(torch.randn(1, 48, 200, 200, device=dev) * 0.5, id=n, save_interpolated=False)
    flop_scene = 3 * 6.74e12

    bands = world > 1 and args.partition == "bands"

    def one():
        sr.clear_SR_planes()
        with torch.no_grad():
            if bands:          # every rank one horizontal band of every plane (+ the 68-pixel LR halo), one all_gather per plane
                nvsr_amd.distributed.super_resolve_planes_sharded(sr, names)
            else:
                sr.super_resolve_many(names)

    if bands and os.environ.get("NVSR_BENCH_REHEARSAL", "0") == "1":
        # rehearsal (tests/test_hip_round3.py): the planes assembled from the ranks' bands are the single-rank planes, bit for bit
        one()
        got = [sr.SR_planes[n].clone() for n in names]
        sr.clear_SR_planes()
        with torch.no_grad():
            sr.super_resolve_many(names)
        same = all(torch.equal(a, sr.SR_planes[n]) for a, n in zip(got, names))
        print("SR_BANDS_%s rank %d of %d" % ("IDENTICAL" if same else "DIFFER", rank, world), file=sys.stderr, flush=True)
        if not same:
            sys.exit(3)
    elapsed = _sync_time(dist, dev, one, args.warmup, args.steps)
    value = (1 if bands else world) * 3 * args.steps / elapsed
    mode = nvsr_amd.capi.get_conv_arithmetic()
    arith = ARITHMETIC[mode]
    result = {"metric": "super-resolved feature planes/sec (48ch 200^2 -> 800^2, EDSR hidden 256 x 32 blocks)", "value": value, "unit": "planes/s",
              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
              "scaling": "strong" if bands else "weak", "vs_baseline": None, "dtype": arith["dtype"], "data": "synthetic", "conv_arithmetic": mode,
              "config": {"workload": "PlanesSR full-plane pass over the 3 position planes of one scene (batched), LR 200^2 + 68 px replicate "
                                     "padding -> HR 800^2", "planes_per_step_per_gpu": 3.0 / world if bands else 3,
                         "partition": "bands" if bands else "replicas",
                         "parallelism": ("every plane split into %d horizontal bands (one per rank, 68-pixel LR halo from real rows), one all_gather per "
                                         "plane (123 MB)" % world) if bands else "replicas only"}}
    if rank == 0:
        ach = flop_scene / (elapsed / args.steps) / 1e12
        peak = arith["pipe_peak"] / arith["products"]
        kname = "conv3x3_kernel" if mode == "f32" else "conv3x3_limb_kernel"
        result["roofline"] = {"kernel": "%s (70 launches per step)" % kname, "bound": "mfma", "achieved": ach, "peak": peak,
                              "unit": "TFLOP/s", "frac": ach / peak, "traffic": pmc_traffic("sr", arith["dtype"]), "algorithmic_flop_per_step": flop_scene,
                              "algorithmic_bytes_per_step": 3 * (173e6 + 2 * 4 * 256 * 270 * 270 * 66),     # weights once per plane + activations in / out of the 66 wide layers (approx.)
                              "peak_note": "algorithmic f32 FLOP over the whole step; peak = %.1f TFLOP/s dense on the pipe used / %d MFMA products per f32 product"
                                           % (arith["pipe_peak"], arith["products"]),
                              "vs_f32_mfma_peak": ach / PEAK_F32_MFMA_TFLOPS}
        result["roofline"]["traffic_source"] = None if result["roofline"]["traffic"] is None else pmc_source()
        if mode != "f32":
            result["roofline"]["kernel"] = ("conv3x3_limb16_kernel<., ., %d> (v_mfma_f32_16x16x32_%s; 67 of the 70 launches per step) + conv3x3_limb_kernel (3 launches, 3 bf16 limbs)"
                                            % ((2, "f16") if mode == "f16x2" else (3, "bf16")))
            result["roofline"]["sustained_pipe_rate"] = dict(SUSTAINED_BF16_MFMA, executed_over_sustained_16x16x32=ach * arith["products"] /
                                                             SUSTAINED_BF16_MFMA[("f16_" if mode == "f16x2" else "") + "tflops_random_operands_16x16x32_two_waves_per_simd"])
        if world == 1 and not args.no_modes:
            modes = {}
            for m2 in ("f32", "bf16x3", "f16x2"):
                nvsr_amd.capi.set_conv_arithmetic(m2)
                one(); torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(2):
                    one()
                torch.cuda.synchronize()
                modes[m2] = {"planes_per_s": 3 * 2 / (time.perf_counter() - t1), "ms_per_step": 1e3 * (time.perf_counter() - t1) / 2}
            nvsr_amd.capi.set_conv_arithmetic(mode)
            result["arithmetic_modes"] = modes
            # the SR *training* iteration of one plane (BASELINE configs[4]'s refinement: forward that keeps its activations, backward
            # with data + weight gradients), full-plane region of interest
            sr.train()
            tf = tb = 0.0
            for rep in range(3):
                sr.clear_SR_planes()
                sr.zero_grad(set_to_none=True)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                out_t = sr("p0")
                torch.cuda.synchronize(); t1 = time.perf_counter()
                out_t.sum().backward()
                torch.cuda.synchronize(); t2 = time.perf_counter()
                if rep:
                    tf += (t1 - t0) / 2; tb += (t2 - t1) / 2
            sr.eval()
            sr.zero_grad(set_to_none=True)
            result["sr_training_per_plane"] = {"forward_ms": 1e3 * tf, "backward_ms": 1e3 * tb,
                                               "algorithmic_tflop": {"forward": 6.74, "backward": 13.48},
                                               "achieved_tflops": {"forward": 6.74 / tf if tf else None, "backward": 13.48 / tb if tb else None}}
        if world == 1 and not args.no_cpu_baseline:
            cv = cpu_conv_rate(args, 10.0)
            result["cpu_baseline"] = {"value": cv["flops"] / (flop_scene / 3), "unit": "planes/s", "cores": cv["cores"], "kind": "port",
                                      "sample": "%d x one 256->256 3x3 conv on a 66x66 tile (%.2f GFLOP each, %.1f s in total, C oracle fp32 OpenMP "
                                                "%d threads), scaled by FLOPs to a whole plane" % (cv["reps"], cv["gflop_each"], cv["seconds"], cv["cores"])}
        return result


def cpu_conv_rate(args, budget_s):
    """FLOP/s of the C oracle's 256 -> 256 3 x 3 convolution on a 66 x 66 tile (fp32, OpenMP over all host cores), timed for ~budget_s; measured once
    per process when the side workloads of the default line share it (args.cpu_cache)."""
    cache = getattr(args, "cpu_cache", None)
    if cache is not None and "conv" in cache:
        return cache["conv"]
    from oracle.oracle import Oracle
    o = Oracle(f32=True)
    x = np.random.default_rng(0).standard_normal((256, 66, 66), dtype=np.float32)
    w = np.random.default_rng(1).standard_normal((256, 256, 3, 3), dtype=np.float32) * 0.01
    o.conv3x3(x[:, :10, :10], w)
    o.conv3x3(x, w)
    fl = 2 * 256 * 256 * 9 * 64 * 64
    reps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s and reps < 20000:
        o.conv3x3(x, w)
        reps += 1
    t = time.perf_counter() - t0
    out = {"flops": reps * fl / t, "reps": reps, "seconds": t, "gflop_each": fl / 1e9, "cores": os.cpu_count() or 1}
    if cache is not None:
        cache["conv"] = out
    return out


def edsr_flops(Hp, Wp, cin=48, cout=48, hid=256, nb=32, n_up=2):
    """algorithmic FLOP (2 per multiply-add) of one EDSR forward on a [cin, Hp, Wp] input: un-padded 3 x 3 convolutions, models.py:789-822"""
    f, h, w = 0, Hp - 2, Wp - 2
    f += 2 * 9 * cin * hid * h * w                        # conv_input
    for _ in range(2 * nb + 1):                           # residual convolutions + conv_mid
        h, w = h - 2, w - 2
        f += 2 * 9 * hid * hid * h * w
    for _ in range(n_up):                                 # hid -> 4 hid, PixelShuffle(2)
        h, w = h - 2, w - 2
        f += 2 * 9 * hid * 4 * hid * h * w
        h, w = 2 * h, 2 * w
    h, w = h - 2, w - 2
    return f + 2 * 9 * hid * cout * h * w                 # conv_output


def sr_roi_pixels(R, roi):
    """LR texels [lo, hi) per axis of a PlanesSR region of interest (models.py:898-906 as csrc/sr_core.h sr_roi restates it)"""
    out = []
    for a in range(2):
        mn, mx = R * (1.0 + roi[a]) / 2.0, R * (1.0 + roi[2 + a]) / 2.0
        out.append((max(int(np.floor(mn)) - 1, 0), min(int(np.ceil(mx)) + 1, R)))
    return out


def bench_refine(args, nvsr_amd, dist, dev, rank, world):
    """BASELINE configs[4]'s iteration (config/RefineOnTestScene.yml, config/TrainModels.yml; train_nerf.py:554-561,790-923): 4096 random rays of
    an 800 x 800 view, 64 + 64 samples, LR planes 200^2; the FINE model samples the three position planes super-resolved by PlanesSR(EDSR 256 x 32)
    in training mode on the region of interest the batch covers (models.py:270-284,884-926), the coarse model samples the LR planes
    (super_resolution.apply_2_coarse False); loss on the fine output (super_resolution.training.loss: fine); Adam.
      --refine-what sr      nerf.train.what = ['SR']: the coarse pass under torch.no_grad, only the EDSR weights receive gradients
      --refine-what joint   what = ['LR_planes', 'decoder', 'SR'] (both YAMLs): the LR planes (through the SR network AND the coarse pass), both
                            decoders and the SR network train."""
    capi, M, T = nvsr_amd.capi, nvsr_amd.models, nvsr_amd.training
    R, N, Nc, Nf = 200, 4096, 64, 64
    llff = getattr(args, "refine_scene", "blender") == "llff"
    if llff:
        Nf = 128                                            # BASELINE configs[4] as written: LLFF 'fern', NDC rays, 64 + 128 samples
    joint = args.refine_what == "joint"
    what = {"LR_planes", "decoder", "SR"} if joint else {"SR"}
    mc, mf, sid, pose = make_synthetic_scene(dev, R, 32, seed=0, theta=30.0, channels_last=True)
    torch.manual_seed(1)
    sr = M.PlanesSR(M.EDSR, 4, 48, 48, {"model": {"hidden_size": 256, "n_blocks": 32}}, "bilinear").to(dev)
    for m in (mc, mf):
        for n, p in m.named_parameters():
            p.requires_grad_(joint and "rot_mats" not in n)
        m.train()
    sr.train()
    mf.detach_LR_planes = False
    mf.assign_SR_model(sr, SR_viewdir=False)
    mf.assign_LR_planes()
    if not joint:
        mc.optional_no_grad = torch.no_grad                 # train_nerf.py:560
    H = W = 800
    focal = 0.5 * W / np.tan(0.5 * CAMERA_ANGLE_X)
    opts, scfg = render_options(Nc, Nf, perturb=True, noise=0.2)
    if llff:
        # a forward-facing view at 1/8 of the LLFF resolution (tests/test_hip_round2.py: the same scene against the oracle): NDC rays
        # (scene_config.no_ndc False, train_utils.py:215-218), near 0, far 1, the box spans the NDC cube
        H, W, focal = 378, 504, 407.6
        pose = torch.tensor([[1.0, 0.0, 0.0, 0.03], [0.0, 1.0, 0.0, -0.02], [0.0, 0.0, 1.0, 0.1], [0.0, 0.0, 0.0, 1.0]], device=dev)
        scfg = Opt(near=0.0, far=1.0, no_ndc=False)
        for m in (mc, mf):
            m.box_coords = {sid: torch.tensor([[-1.5, -1.5, -1.5, -np.pi, -np.pi / 2], [1.5, 1.5, 1.5, np.pi, np.pi / 2]], dtype=torch.float64)}
            m.invalidate()
    g = torch.Generator(device=dev).manual_seed(100 + int(os.environ.get("NVSR_BENCH_SEED_RANK", rank)))
    target = torch.rand(H, W, 3, device=dev, generator=g)
    dec = list({id(p): p for m in (mc, mf) for p in m.decoder_parameters()}.values())
    planes = list(mc.planes_.values())
    srp = list(sr.parameters())
    opt = torch.optim.Adam(dec, lr=5e-4, fused=True) if joint else None
    popt = torch.optim.Adam(planes, lr=5e-4, fused=True) if joint else None
    sropt = torch.optim.Adam(srp, lr=5e-5, fused=True)
    trained = srp + (planes + dec if joint else [])
    # N > 1: the EDSR gradient is all-reduced bucket by bucket WHILE the SR backward runs (distributed.OverlappedSRGradSync: the batched backward records an
    # event per bucket of its gradient blob), the planes' and decoders' gradients behind the backward; NVSR_BENCH_SYNC=after: everything behind the backward
    if world > 1 and os.environ.get("NVSR_BENCH_SYNC", "overlapped") == "overlapped":
        sync = nvsr_amd.distributed.OverlappedSRGradSync(sr, other_parameters=(planes + dec if joint else []))
    else:
        sync = (lambda: nvsr_amd.distributed.allreduce_gradients([p.grad for p in trained if p.grad is not None])) if world > 1 else None
    if os.environ.get("NVSR_BENCH_DEBUG_NAN") == "1":
        # diagnostics (round 5's backward-prologue race): are this rank's gradients finite before / after the all-reduce, per iteration?
        # (host reads: the timing of such a run means nothing)
        inner = sync

        def sync():
            gs = [p.grad for p in trained if p.grad is not None]
            pre = [n_ for n_, g_ in enumerate(gs) if not bool(torch.isfinite(g_).all())]
            if inner is not None:
                inner()
            post = [n_ for n_, g_ in enumerate(gs) if not bool(torch.isfinite(g_).all())]
            if pre or post:
                print("NAN_DEBUG rank %d iteration %d: non-finite gradient tensors before the all-reduce %s, after %s (of %d)"
                      % (rank, it[0], pre[:6], post[:6], len(gs)), file=sys.stderr, flush=True)
    seed_rank = int(os.environ.get("NVSR_BENCH_SEED_RANK", rank))          # (a one-process run on another rank's pixels and random numbers)
    sampler = T.DevicePixelSampler(seed=100 + seed_rank)
    mk = lambda sync_, sampler_: T.TrainStep(mc, mf, opts, what, optimizer=opt, SR_optimizer=sropt, planes_optimizer=popt, SR_model=sr, sr_loss="fine",
                                             grad_sync=sync_, pixel_sampler=sampler_)
    step = mk(sync, sampler)
    it = [0]

    def draw():
        return dict(t_rand=torch.rand(N, Nc, device=dev, generator=g), u=torch.rand(N, Nf, device=dev, generator=g),
                    noise_coarse=torch.empty(N, Nc, device=dev).normal_(0.0, 0.2, generator=g),
                    noise_fine=torch.empty(N, Nc + Nf, device=dev).normal_(0.0, 0.2, generator=g))

    def one(s=step):
        m = s(it[0], target, pose, H, W, focal, 1, sid, scfg, N, sr_iter=True, randoms=draw())
        it[0] += 1
        return m

    elapsed = _sync_time(dist, dev, one, args.warmup, args.steps)
    host_issue_ms = 1e3 * _sync_time.issue_s / args.steps
    if world > 1 and os.environ.get("NVSR_BENCH_REHEARSAL", "0") == "1":
        # rehearsal (tests/test_hip_round5.py): every rank started from the same parameters and stepped on the same averaged gradients -- a sample of
        # every parameter group (the SR network's first and last layers, a plane, a decoder matrix) is bit-identical on all ranks after the timed steps
        probe_ = [srp[0], srp[-1]] + ([planes[0], dec[0]] if joint else [])
        mine = torch.cat([p_.detach().reshape(-1)[:4096].float().cpu() for p_ in probe_])
        both = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        same = all(torch.equal(both[0], b_) for b_ in both[1:]) and bool(torch.isfinite(mine).all())
        overlapped = isinstance(sync, nvsr_amd.distributed.OverlappedSRGradSync)
        print("REFINE_PARAMS_%s rank %d of %d (%s gradient sync%s)" % ("IDENTICAL" if same else "DIFFER", rank, world, "overlapped" if overlapped else "post-backward",
              ", %d buckets" % sync.stats["buckets"] if overlapped else ""), file=sys.stderr, flush=True)
        if not same:
            sys.exit(3)
    if world > 1 and isinstance(sync, nvsr_amd.distributed.OverlappedSRGradSync):
        sync.detach()          # (the probe iterations below run on rank 0 ALONE: the SR backward must not start collectives there)
    mode = capi.get_conv_arithmetic()
    arith = ARITHMETIC[mode]
    label = "what = ['LR_planes', 'decoder', 'SR']" if joint else "what = ['SR']"
    view = "LLFF-'fern'-like forward-facing %dx%d view, NDC rays" % (H, W) if llff else "800x800 view"
    result = {"metric": "SR-refinement training rays/sec (4096 rays/iter of %s, %d+%d samples, LR planes 200^2 -> ROI x4 by PlanesSR(EDSR 256x32) in training "
                        "mode on the fine model, %s, Adam)" % (view if llff else "an 800x800 view", Nc, Nf, label),
              "value": world * N * args.steps / elapsed, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
              "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
              "host_issue_ms_per_step": host_issue_ms, "dtype": arith["dtype"], "conv_arithmetic": mode, "data": "synthetic",
              "config": {"workload": "SR refinement iteration (BASELINE configs[4]; config/RefineOnTestScene.yml / TrainModels.yml): 4096 random rays of %s, "
                                     "%d coarse + %d fine samples, 3x200^2x48 + 32^2x48 LR planes, EDSR(hidden 256, 32 blocks, x4) on the "
                                     "regions of interest of the three position planes, SR model on the fine model only, loss on the fine output, %s, Adam"
                                     % ("a " + view if llff else "an 800x800 view", Nc, Nf, label), "rays_per_step_per_gpu": N, "refine_what": args.refine_what,
                         "refine_scene": "llff" if llff else "blender",
                         "parallelism": "every rank draws its own %d rays; EDSR gradient (173 MB) all-reduced in ~48 MB buckets during the SR backward%s"
                                        % (N, ", planes 23 MB + decoders 1 MB behind it" if joint else "")}}
    if rank != 0:
        return None
    if getattr(args, "no_split", False):          # counter passes: the timed iteration only
        return result
    # ---- the split of one iteration: events on the launch stream at the phase boundaries of one more (eager) iteration
    marks = {}

    def mark(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.setdefault(name, []).append(e)

    rois = {}
    real_many = sr.forward_many

    def timed_many(requests):
        mark("sr_fwd_begin")
        outs = real_many(requests)
        mark("sr_fwd_end")
        for (name, roi), out in zip(requests, outs):
            rois[name] = [float(v) for v in roi]
            if out.requires_grad:
                out.register_hook(lambda g_: (mark("sr_bwd_begin"), g_)[1])
        return outs

    split = None
    probe = mk(lambda: mark("bwd_end"), sampler)
    sr.forward_many = timed_many
    try:
        reps = []
        for _ in range(3):
            marks.clear()
            mark("start")
            try:
                one(probe)
            except capi.NvsrError:
                # which operands are non-finite when the range flag comes up: parameters = the previous iteration's update wrote them, none = this
                # iteration's forward met something else (how round 5's backward-prologue race was cornered: DESIGN.md section 6)
                bad = [n_ for n_, p_ in list(mc.named_parameters()) + list(mf.named_parameters()) + list(sr.named_parameters())
                       if not bool(torch.isfinite(p_.detach()).all())]
                print("RANGE_FLAG_DIAGNOSTICS iteration %d: non-finite parameters: %s" % (it[0], bad[:8] if bad else "none"), file=sys.stderr, flush=True)
                raise
            mark("end")
            torch.cuda.synchronize()
            ms = lambda a, b: marks[a][0].elapsed_time(marks[b][-1])
            reps.append({"random inputs + pixels + rays + regions of interest (side stream) + weight packing": ms("start", "sr_fwd_begin"),
                         "PlanesSR forward (3 ROI crops, keeps activations)": ms("sr_fwd_begin", "sr_fwd_end"),
                         "render forward + loss + render backward (coarse on LR planes, fine on the SR planes)": marks["sr_fwd_end"][-1].elapsed_time(marks["sr_bwd_begin"][0]),
                         "PlanesSR backward (3 ROI crops: data + weight gradients)": marks["sr_bwd_begin"][0].elapsed_time(marks["bwd_end"][0]),
                         "optimizers (Adam: EDSR 43.3 M parameters%s)" % (" + planes + decoders" if joint else ""): ms("bwd_end", "end"),
                         "whole iteration": ms("start", "end")})
        split = {k: float(np.median([r[k] for r in reps])) for k in reps[0]}
    finally:
        sr.forward_many = real_many
    pad = int(sr.inner_model.required_padding)
    crops, fwd_flop, pass_bytes = {}, 0.0, 0.0
    for name, roi in rois.items():
        (l0, h0), (l1, h1) = sr_roi_pixels(R, roi)
        crops[name] = {"lr_rows": [l0, h0], "lr_cols": [l1, h1], "network_input": [h0 - l0 + 2 * pad, h1 - l1 + 2 * pad]}
        fwd_flop += edsr_flops(h0 - l0 + 2 * pad, h1 - l1 + 2 * pad)
        # one pass over a crop, counted like the sr workload counts a plane: the weights once + the activations in / out of the 66 wide layers at the
        # crop's mid-network size (network input - 66)
        pass_bytes += 173e6 + 2 * 4 * 256 * (h0 - l0 + 2 * pad - 66) * (h1 - l1 + 2 * pad - 66) * 66
    t_fwd = split["PlanesSR forward (3 ROI crops, keeps activations)"] * 1e-3
    t_bwd = split["PlanesSR backward (3 ROI crops: data + weight gradients)"] * 1e-3
    sr_flop = 3.0 * fwd_flop                                  # forward + data gradient + weight gradient (dx of conv_input is 0.3 % and counted)
    ach = sr_flop / (t_fwd + t_bwd) / 1e12
    peak = arith["pipe_peak"] / arith["products"]
    # decoder evaluations: forward, transposed layers (data gradient), weight-gradient contraction = 1 + 1 + 1 forward's worth each
    render_flop = FLOP_PER_EVAL * N * (Nc + (Nc + Nf)) * 3.0 if joint else FLOP_PER_EVAL * N * (Nc + 2.0 * (Nc + Nf))
    result["split_ms"] = split
    result["roi_crops"] = crops
    result["roofline"] = {"kernel": "conv3x3_limb16_kernel (forward + data gradients) + conv3x3_wgrad_limb_kernel: every convolution launch of the three "
                                    "PlanesSR forward + backward passes of one iteration",
                          "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                          "traffic": None if llff else pmc_traffic("refine_" + args.refine_what, arith["dtype"]),      # (the counter passes ran the blender scene)
                          "kernel_ms": 1e3 * (t_fwd + t_bwd), "kernel_ms_source": "HIP events on the launch stream around the PlanesSR phases of an iteration, this process",
                          "algorithmic_flop_per_step": sr_flop, "algorithmic_flop_forward": fwd_flop,
                          "algorithmic_bytes_per_step": 3.0 * pass_bytes,                 # forward, data-gradient and weight-gradient pass over every crop
                          "forward": {"ms": 1e3 * t_fwd, "achieved": fwd_flop / t_fwd / 1e12, "frac": fwd_flop / t_fwd / 1e12 / peak},
                          "backward": {"ms": 1e3 * t_bwd, "achieved": 2 * fwd_flop / t_bwd / 1e12, "frac": 2 * fwd_flop / t_bwd / 1e12 / peak},
                          "render_algorithmic_flop_per_step": render_flop,
                          "whole_step": {"algorithmic_flop": sr_flop + render_flop, "achieved": (sr_flop + render_flop) / (elapsed / args.steps) / 1e12,
                                         "frac": (sr_flop + render_flop) / (elapsed / args.steps) / 1e12 / peak},
                          "peak_note": "algorithmic f32 FLOP of the ROI crops actually processed (3 x the forward: forward, data gradient, weight gradient); "
                                       "peak = %.1f TFLOP/s dense on the pipe used / %d MFMA products per f32 product" % (arith["pipe_peak"], arith["products"])}
    result["roofline"]["traffic_source"] = None if result["roofline"]["traffic"] is None else pmc_source()
    rec = pmc_record()
    if rec is not None and not llff and rec.get("refine_" + args.refine_what, {}).get("sr_backward_split_ms"):
        result["sr_backward_split_ms"] = dict(rec["refine_" + args.refine_what]["sr_backward_split_ms"], source=pmc_source())
    if world == 1 and not args.no_cpu_baseline:
        from oracle.oracle import Oracle, decoder_blob
        cv = cpu_conv_rate(args, 8.0)
        cache = getattr(args, "cpu_cache", None)
        key = ("rays", Nc, Nf)
        if cache is not None and key in cache:
            n, t_rays = cache[key]
        else:
            # (Blender-style rays of the same sample counts also for the LLFF scene: the oracle's cost per ray depends on the sample counts only)
            o64 = Oracle(f32=False)
            pl = [mc.planes_[M.get_plane_name(sid, d)].detach().cpu().numpy() for d in range(4)]
            osc = o64.scene(pl, mc.box_coords[sid].numpy())
            dc = o64.decoder(decoder_blob({k: v.detach().cpu().numpy() for k, v in mc.state_dict().items()}))
            df = o64.decoder(decoder_blob({k: v.detach().cpu().numpy() for k, v in mf.state_dict().items()}))
            n = 96 if cache is not None else 192
            batch = torch.stack(T.get_ray_bundle_at(800, 800, 0.5 * 800 / np.tan(0.5 * CAMERA_ANGLE_X), torch.from_numpy(pose_spherical(30.0, -30.0, 4.0)).to(dev),
                                                    torch.randint(0, 800, (n, 2), device=dev)), 0)
            rn = nvsr_amd.train_utils.pack_rays(batch[0], batch[1], 2.0, 6.0).cpu().numpy()
            gg = np.full((n, 3), 1e-3, np.float32)
            t0 = time.perf_counter()
            o64.render_rays(osc, dc, df, rn, Nc, Nf)
            o64.render_backward(osc, [p_.shape for p_ in pl], dc, df, rn, Nc, Nf, gg, gg)
            t_rays = time.perf_counter() - t0
            if cache is not None:
                cache[key] = (n, t_rays)
        # one iteration on the host = the convolutions' FLOP at the measured conv rate + 4096 rays at the measured render rate
        t_iter = sr_flop / cv["flops"] + N * t_rays / n
        result["cpu_baseline"] = {"value": N / t_iter, "unit": "rays/s", "cores": cv["cores"], "kind": "port",
                                  "sample": "C oracle: %d x one 256->256 3x3 convolution on a 66x66 tile (fp32, OpenMP %d threads, %.1f s) scaled by FLOP to the "
                                            "%.1f TFLOP of the iteration's three PlanesSR forward + backward passes, + %d rays of the render step (forward + "
                                            "analytic plane backward, double accumulation, single thread, %.1f s) scaled to 4096 rays"
                                            % (cv["reps"], cv["cores"], cv["seconds"], sr_flop / 1e12, n, t_rays)}
    return result


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=["render", "train", "sr", "refine"], default="render")
    ap.add_argument("--refine-what", choices=["sr", "joint"], default="joint",
                    help="--workload refine: nerf.train.what = ['SR'] (sr) or ['LR_planes', 'decoder', 'SR'] (joint; the value of RefineOnTestScene.yml and "
                         "TrainModels.yml)")
    ap.add_argument("--train-what", choices=["planes", "planes+decoder"], default="planes",
                    help="--workload train: nerf.train.what (default = Feature_Planes_Only.yml, BASELINE configs[3])")
    ap.add_argument("--nchw-planes", action="store_true",
                    help="--workload train: keep the plane parameters in the reference's NCHW memory order (a re-layout kernel per plane and step) "
                         "instead of torch.channels_last, whose memory is the kernels' native [H][W][C] layout")
    ap.add_argument("--rays-global", type=int, default=None,
                    help="--workload train: total rays per iteration over ALL ranks, rank r takes its 1/N share of the same draw (SURVEY 8e: 4096 -> "
                         "4096 / N per GPU) -> \"scaling\": \"strong\"; default: 4096 rays per rank, drawn per rank (weak)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--partition", choices=["rows", "frame", "view", "bands"], default=None,
                    help="--workload render, N > 1: how the rays are sharded (see the module docstring); default rows.  --workload sr, N > 1: "
                         "bands = every plane split into one horizontal band per rank + one all_gather (strong scaling); default replicas")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--res", type=int, default=800, help="image side (default 800 = BASELINE config)")
    ap.add_argument("--plane-res", type=int, default=800)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="train workload: launch the iteration kernel by kernel instead of replaying its HIP graph")
    ap.add_argument("--no-modes", action="store_true", help="skip the per-arithmetic-mode frames (profiling passes)")
    ap.add_argument("--refine-scene", choices=["blender", "llff"], default="blender",
                    help="--workload refine: blender = 4096 rays of an 800x800 Lego-like view, 64+64 samples (the YAMLs' values); llff = BASELINE configs[4] as "
                         "written: an LLFF-'fern'-like forward-facing 378x504 view, NDC rays, 64+128 samples")
    ap.add_argument("--no-split", action="store_true", help="--workload refine: skip the three probe iterations of the phase split (counter passes)")
    ap.add_argument("--full-record", default=None, metavar="PATH",
                    help="also write the full record (every key; the stdout line is its compact form, <= 4 KB) to PATH")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="--workload render, N = 1: do not append the short train / sr runs (`other_workloads` of the line)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher.  Nothing in this process has touched the GPU yet (no HIP call, no
        # torch.cuda.is_available()); the ranks are CHILD processes (never an exec from a process that has initialised the GPU).
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.call(cmd, env=env))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d -- launch one rank per GPU (or run without torch.distributed.run: --gpus N self-launches)"
                 % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # NVSR_BENCH_REHEARSAL=1: every rank on cuda:0 over gloo -- the N > 1 code path on a one-GPU box (tests/test_hip_parity.py);
    # its numbers mean nothing.  The driver's runs use one GPU per rank over RCCL.
    rehearsal = os.environ.get("NVSR_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import nvsr_amd
    nvsr_amd.capi.lib()  # fail loudly if the HIP library is not built

    # what the collectives of this run really are: backend, its world size and every rank's device, gathered THROUGH the process group
    # (a SCALE line carries them: "rccl_ranks": N with N distinct devices, or the rehearsal's gloo / shared cuda:0)
    comm = {"backend": "none", "rccl_ranks": 0, "world_size": 1, "rank_devices": [str(dev)]}
    if dist is not None:
        devs = [None] * world
        dist.all_gather_object(devs, "%s (%s)" % (dev, torch.cuda.get_device_properties(dev).name))
        comm = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                "rccl_ranks": dist.get_world_size() if dist.get_backend() == "nccl" else 0, "rank_devices": devs}

    if args.workload != "render":
        res = {"train": bench_train, "sr": bench_sr, "refine": bench_refine}[args.workload](args, nvsr_amd, dist, dev, rank, world)
        if res is not None:          # rank 0
            res["collectives"] = comm
            emit(res, args.full_record)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    H = W = args.res
    focal = 0.5 * W / np.tan(0.5 * CAMERA_ANGLE_X)
    partition = args.partition or ("rows" if world > 1 else "view")
    if partition == "bands":
        sys.exit("bench.py: --partition bands belongs to --workload sr")
    if world == 1:
        partition = "view"            # one rank: every partition is the same single-GPU frame
    # same (replicated) scene on every rank; "view": a different view per rank, "frame": one view, "rows": one view per rank, all shared
    mc, mf, sid, pose = make_synthetic_scene(dev, args.plane_res, 32, seed=0, theta=30.0 + (45.0 * rank if partition == "view" else 0.0))
    opts, scfg = render_options(64, 128)
    ro, rd = nvsr_amd.nerf_helpers.get_ray_bundle(H, W, focal, pose)
    D = nvsr_amd.distributed
    poses = [torch.from_numpy(pose_spherical(30.0 + 45.0 * v, -30.0, 4.0)).to(dev) for v in range(world)]

    def step():
        if partition == "view":
            r, d = nvsr_amd.nerf_helpers.get_ray_bundle(H, W, focal, pose)   # ray generation is part of the path
            return nvsr_amd.train_utils.eval_nerf(H, W, focal, mc, mf, r, d, opts, scene_id=sid, scene_config=scfg)
        if partition == "frame":
            r, d = nvsr_amd.nerf_helpers.get_ray_bundle(H, W, focal, pose)
            return D.render_image_sharded(H, W, focal, mc, mf, r, d, opts, sid, scfg)
        return D.render_views_sharded(H, W, focal, mc, mf, poses, opts, sid, scfg)

    if rehearsal and world > 1 and partition != "view":
        # correctness of the sharded partitions with the HIP renderer (asserted by tests/test_hip_round2.py through the exit code): every
        # rank's assembled frames equal, bit for bit, the frames one rank renders alone
        got = step()
        if partition == "frame":
            alone = nvsr_amd.train_utils.eval_nerf(H, W, focal, mc, mf, ro, rd, opts, scene_id=sid, scene_config=scfg)
            same = torch.equal(got[0], alone[0]) and torch.equal(got[1], alone[3])
        else:
            same = True
            for v, pv in enumerate(poses):
                a = nvsr_amd.train_utils.eval_nerf(H, W, focal, mc, mf, *nvsr_amd.nerf_helpers.get_ray_bundle(H, W, focal, pv), opts, scene_id=sid,
                                                   scene_config=scfg)
                same = same and torch.equal(got[0][v], a[0]) and torch.equal(got[1][v], a[3])
        print("SHARDED_RENDER_%s rank %d partition %s" % ("IDENTICAL" if same else "DIFFERS", rank, partition), file=sys.stderr, flush=True)
        if not same:
            sys.exit(3)

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    rays_per_step = H * W                                            # per GPU and step ("frame": of the whole job)
    frames_per_step = 1 if partition == "frame" else world
    value = frames_per_step * rays_per_step * args.steps / elapsed
    mode = nvsr_amd.capi.get_decoder_arithmetic()
    arith = ARITHMETIC[mode]
    result = {
        "metric": "rendered rays/sec (64+128 samples) at 800x800 Lego-like view",
        "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong" if (partition == "frame" and world > 1) else "weak",
        "vs_baseline": None,
        "dtype": arith["dtype"], "data": "synthetic", "decoder_arithmetic": mode,
        "config": {"workload": "Blender-'lego'-like %dx%d view, 64 coarse + 128 fine samples, tri-plane decoder (3x%d^2x48 + 32^2x48 planes, "
                               "4+4x128 MLP), %s" % (H, W, args.plane_res,
                                                     {"view": "1 view per GPU per step", "frame": "1 view per step, its rows sharded over the GPUs",
                                                      "rows": "%d views per step, every view's rows sharded over the GPUs" % world}[partition]),
                   "rays_per_step_per_gpu": rays_per_step // (world if partition == "frame" else 1), "decoder_evals_per_ray": 256,
                   "partition": partition,
                   "ray_order": "row-major (NVSR_ROW_ORDER)" if os.environ.get("NVSR_ROW_ORDER") else
                                "16x2 pixel patches in 128x32 pixel blocks (train_utils.patch_order; same pixels bit for bit, returned in row-major order)",
                   "parallelism": {"view": "rays sharded by view, no collective",
                                   "frame": "rays of one frame sharded by row blocks, one all_gather of the pixels per frame",
                                   "rows": "rays of every frame sharded by row blocks (a rank renders its rows of all views in one launch), "
                                           "one all_gather of the pixels per step"}[partition]},
        "decoder_evals_per_s_per_gpu": value * 256 / world,
    }
    if rank == 0 and world > 1 and partition != "view":
        # BASELINE's metric asks for "PSNR vs ref at 1/2/4/8 MI355X": a sharded frame is the one-GPU frame BIT FOR BIT (same kernels on the same rays, only
        # the launch a ray sits in differs), so its PSNR against the checker is the N = 1 line's figure.  Checked here, outside the timed region and without
        # a collective: rank 0 renders view 0 alone and compares it with the frame the ranks assembled in the last timed step.
        alone = nvsr_amd.train_utils.eval_nerf(H, W, focal, mc, mf, *nvsr_amd.nerf_helpers.get_ray_bundle(H, W, focal, poses[0] if partition == "rows" else pose),
                                               opts, scene_id=sid, scene_config=scfg)
        got_c, got_f = (out[0][0], out[1][0]) if partition == "rows" else (out[0], out[1])
        result["sharded_frame_identical_to_one_gpu"] = bool(torch.equal(got_c, alone[0]) and torch.equal(got_f, alone[3]))
        result["psnr_note"] = "sharded frames equal the one-GPU frame bit for bit: psnr_vs_oracle_db is the N = 1 line's"
    if rank == 0:
        # dominant kernel: fused fine render pass (192 of the 256 evaluations per ray), render2.hip
        rays = nvsr_amd.train_utils.pack_rays(ro, rd, 2.0, 6.0)
        N = rays.shape[0]
        rays_row, inv = rays, None                        # (row-major: what the CPU baseline / PSNR check pairs with row-major images)
        if not os.environ.get("NVSR_ROW_ORDER") and N >= nvsr_amd.train_utils.PATCH_ORDER_MIN_RAYS:
            perm, inv = nvsr_amd.train_utils.patch_order(N, W, dev)
            rays = rays.index_select(0, perm)             # the order eval_nerf renders a frame in: the kernel is timed on what the steps launch
        import ctypes as C
        capi = nvsr_amd.capi
        ws = torch.empty(capi.lib().nvsr_render_workspace_floats(N, 64, 128), device=dev)
        bufs = [torch.empty((N, 3), device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev),
                torch.empty((N, 3), device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev)]
        sc, keep = mc.native_scene()
        capi.call("nvsr_render_rays", C.byref(sc), capi.ptr(mc.packed_decoder()), capi.ptr(mf.packed_decoder()), N, 64, 128, capi.ptr(rays),
                  0, 0, None, None, None, None, *[capi.ptr(b) for b in bufs], capi.ptr(ws), capi.stream())
        z_fine = ws[2 * N * 64: 2 * N * 64 + N * 192].view(N, 192)     # [z_c | w_c | z_f | (raw of the un-fused small-N path)]
        dt = time_fine_pass_kernel(nvsr_amd, mf, rays, z_fine)
        flops = FLOP_PER_EVAL * N * 192
        achieved = flops / dt / 1e12
        # HBM-side traffic of that launch: PMC counters cannot be read from inside this process; the value comes from the
        # committed rocprofv3 --pmc passes of this same command (profiles/pmc_latest.json), corrected as the guide prescribes
        traffic = None
        rec = pmc_record()                            # None unless the counter passes were taken with these kernel sources
        if rec is not None and H == 800 and args.plane_res == 800 and rec.get("decoder_arithmetic", "f32") == mode:
            traffic = rec.get("traffic_bytes")
        peak = arith["pipe_peak"] / arith["products"]
        # matrix-pipe busy fraction / clock of this launch as the counters saw them (same guard as `traffic`)
        mfma_pmc = rec.get("mfma") if (rec is not None and traffic is not None) else None
        result["roofline_counters"] = None if mfma_pmc is None else {
            "mfma_busy_frac": mfma_pmc["mfma_busy_frac"], "clock_ghz": mfma_pmc["clock_ghz"], "SQ_VALU_MFMA_BUSY_CYCLES": mfma_pmc["SQ_VALU_MFMA_BUSY_CYCLES"],
            "GRBM_GUI_ACTIVE": mfma_pmc["GRBM_GUI_ACTIVE"], "note": "busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); the dense peak "
            "assumes 2.4 GHz: frac ~= busy x clock / 2.4", "source": pmc_source()}
        result["roofline"] = {"kernel": "%s (fine pass, S=192)" % arith["kernel"], "bound": "mfma", "achieved": achieved,
                              "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                              "traffic_source": None if traffic is None else pmc_source(),
                              "kernel_ms": dt * 1e3, "kernel_ms_source": "HIP events on the launch stream, this process",
                              "algorithmic_flop_per_launch": flops,
                              "algorithmic_gather_bytes_per_launch": GATHER_BYTES_PER_EVAL * N * 192,
                              "peak_note": "algorithmic f32 FLOP; peak = %.1f TFLOP/s dense on the pipe used / %d MFMA products per f32 product"
                                           % (arith["pipe_peak"], arith["products"]),
                              "executed_mfma_tflops": achieved * arith["products"], "vs_f32_mfma_peak": achieved / PEAK_F32_MFMA_TFLOPS}
        if arith["pipe_peak"] == PEAK_BF16_MFMA_TFLOPS:
            result["roofline"]["sustained_pipe_rate"] = dict(SUSTAINED_BF16_MFMA, executed_over_sustained_random=achieved * arith["products"] /
                                                             SUSTAINED_BF16_MFMA[("f16_" if arith["products"] == 3 else "") + "tflops_random_operands"])
        # bandwidth-bound stages: the helper kernels alone, and the fused pass's plane sampling + compositing as the HBM traffic the
        # counters saw (profiles/pmc_latest.json) over the live kernel time
        result["hbm_stages"] = hbm_stage_rates(nvsr_amd, H, W, focal, pose, ro, rd, rays, ws)
        if traffic is not None:
            result["hbm_stages"]["plane sampling + compositing inside %s" % arith["kernel"]] = {
                "ms": dt * 1e3, "hbm_bytes_pmc": traffic, "GB/s": traffic / dt / 1e9, "frac_of_hbm_peak": traffic / dt / 1e9 / PEAK_HBM_GBS,
                "algorithmic_gather_bytes": GATHER_BYTES_PER_EVAL * N * 192}
        if world == 1 and not args.no_modes:
            # the same frame in the other arithmetic modes (2 steps each), so that every number of this line can be re-based
            modes = {}
            for m2 in ("f32", "bf16x3", "f16x2"):
                nvsr_amd.capi.set_decoder_arithmetic(m2)
                step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(2):
                    step()
                torch.cuda.synchronize()
                t_frame = (time.perf_counter() - t1) / 2
                modes[m2] = {"rays_per_s": rays_per_step / t_frame, "ms_per_step": 1e3 * t_frame,
                             "fine_pass_kernel_ms": 1e3 * time_fine_pass_kernel(nvsr_amd, mf, rays, z_fine, reps=2)}
            nvsr_amd.capi.set_decoder_arithmetic(mode)
            result["arithmetic_modes"] = modes
            # models.fine.type: use_same (train_nerf.py:353-355: ONE model for both passes; NOT the configuration of this line's `value`): the
            # fine pass evaluates the 128 importance samples only and reuses the coarse pass's outputs for the 64 coarse depths
            # (nvsr_render_rays_shared_arith) against recomputing them like the reference does
            def one_model_frame():
                r, d = nvsr_amd.nerf_helpers.get_ray_bundle(H, W, focal, pose)
                return nvsr_amd.train_utils.eval_nerf(H, W, focal, mc, mc, r, d, opts, scene_id=sid, scene_config=scfg)
            one = {}
            for tag, env in (("shared", None), ("recomputed", "1")):
                if env is None:
                    os.environ.pop("NVSR_NO_SHARED_DECODER", None)
                else:
                    os.environ["NVSR_NO_SHARED_DECODER"] = env
                img = one_model_frame()[3]
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(2):
                    one_model_frame()
                torch.cuda.synchronize()
                one[tag] = {"ms_per_step": 1e3 * (time.perf_counter() - t1) / 2, "frame": img}
            os.environ.pop("NVSR_NO_SHARED_DECODER", None)
            result["one_decoder_for_both_passes"] = {
                "config": "models.fine.type: use_same -- not the headline configuration (two models)",
                "shared_ms_per_step": one["shared"]["ms_per_step"], "recomputed_ms_per_step": one["recomputed"]["ms_per_step"],
                "rays_per_s_shared": rays_per_step / one["shared"]["ms_per_step"] * 1e3,
                "max_abs_rgb_difference": float((one["shared"]["frame"] - one["recomputed"]["frame"]).abs().max())}
            del one
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(nvsr_amd, mc, mf, sid, rays_row, None)
            # PSNR against the float64 checker on the LARGE sample (16 384 rays where the host's cores allow it within the budget): round 4 showed
            # that a 2 048-ray figure is decided by which handful of rays had an importance sample land in another coarse bin -- ~5 % of the rays
            # in EVERY arithmetic, exact f32 included, holding > 90 % of the squared error -- so the line carries the all-ray figure AND the one
            # over the rays whose fine depths are the checker's, per arithmetic, with the evidence they come from
            ev = frame_error_evidence(nvsr_amd, mc, mf, sid, rays_row, budget_s=25.0)
            result["frame_error_evidence"] = ev
            result["psnr_vs_oracle_db"] = ev[mode]["psnr_db_all"]
            result["psnr_vs_oracle"] = {"rays_checked": ev["rays_checked"], "all_rays_db": ev[mode]["psnr_db_all"],
                                        "rays_with_the_checkers_depths_db": ev[mode]["psnr_db_non_flipped"],
                                        "rays_flipped_in_no_arithmetic_db": ev[mode]["psnr_db_rays_flipped_in_no_arithmetic"],
                                        "flipped_fraction": ev[mode]["flipped_fraction"], "checker": ev["checker"]}
            result["psnr_vs_oracle_db_by_arithmetic"] = {m2: ev[m2]["psnr_db_all"] for m2 in ("f32", "bf16x3", "f16x2")}
            if "arithmetic_modes" in result:
                for m2 in result["arithmetic_modes"]:
                    result["arithmetic_modes"][m2]["psnr_vs_oracle_db"] = ev[m2]["psnr_db_all"]
                result["decoder_error_vs_float64_by_arithmetic"] = decoder_error_by_arithmetic(nvsr_amd, mf, sid, rays, z_fine)
        if world == 1 and not args.no_modes and not args.no_other_workloads and H == 800 and args.plane_res == 800:
            # The other BASELINE configurations of the same path, measured in this same driver-timed process (short runs; each is also its
            # own `--workload`): configs[3] = the 4 096-ray Feature_Planes_Only optimisation step, configs[2]'s SR stage = EDSR 256 x 32 on
            # the three 200^2 planes of a scene (+ its training forward / backward per plane).  Same JSON contract per entry.
            del out
            sub = argparse.Namespace(**vars(args))
            # their CPU legs share one convolution probe and one render probe per sample count (sub.cpu_cache), ~25 s of host time in all
            sub.no_cpu_baseline, sub.no_modes, sub.cpu_cache = args.no_cpu_baseline, False, {}
            other = {}
            for wl, fn, steps, warm, extra in (("train", bench_train, 30, 5, {"train_what": "planes"}),
                                               ("train_decoder", bench_train, 20, 3, {"train_what": "planes+decoder"}),
                                               ("sr", bench_sr, 3, 1, {}),
                                               ("refine", bench_refine, 5, 2, {"refine_what": "joint"}),
                                               ("refine_sr_only", bench_refine, 5, 2, {"refine_what": "sr"}),
                                               ("refine_llff_ndc", bench_refine, 5, 2, {"refine_what": "joint", "refine_scene": "llff"})):
                sub.steps, sub.warmup = steps, warm
                for k, v in extra.items():
                    setattr(sub, k, v)
                try:
                    other[wl] = fn(sub, nvsr_amd, None, dev, 0, 1)
                except Exception as e:          # the headline line must not depend on the side runs
                    other[wl] = {"error": "%s: %s" % (type(e).__name__, e)}
                torch.cuda.empty_cache()
            result["other_workloads"] = other
        result["collectives"] = comm
        emit(result, args.full_record)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
