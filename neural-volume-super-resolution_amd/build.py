"""Builds csrc/*.hip into libnvsr_hip.so (in-tree) with hipcc for gfx950.  Cross-compiles without a GPU."""
import glob
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libnvsr_hip.so")
# -ffp-contract=off: a*b+c stays two roundings like the reference's separate ATen ops; FMAs are written explicitly (fmaf / MFMA)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(PKG_DIR, "..", "include", "*.h")) + [os.path.abspath(__file__)]
    return any(os.path.getmtime(p) > t for p in deps)


# Per-file flags.  render3.hip: the SLP vectorizer packs pairs of f32 operations out of different MFMA gaps into v_pk_* and, to do so,
# re-schedules the whole block at IR level -- every pure instruction (the MFMAs included) sinks below the block's loads and
# sched_barriers, the hand-placed interleave is gone and hundreds of registers spill.
# decode_limb.hip: same blocks, same reason (250 spilled registers with the vectorizer, none without).
PER_FILE_FLAGS = {"render3.hip": ["-fno-slp-vectorize"], "decode_limb.hip": ["-fno-slp-vectorize"], "render_bwd_limb.hip": ["-fno-slp-vectorize"]}
OBJ_DIR = os.path.join(CSRC, "_obj")


def _compile_one(hipcc, src, verbose):
    obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
    deps = [src] + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(PKG_DIR, "..", "include", "*.h")) + [os.path.abspath(__file__)]
    if os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps):
        return obj
    flags = [f for f in HIPCC_FLAGS if f != "-shared"] + PER_FILE_FLAGS.get(os.path.basename(src), [])
    flags += os.environ.get("NVSR_EXTRA_HIPCC_FLAGS", "").split()       # experiments (e.g. -DR3_STAMP=1); use with build_extension(force=True)
    cmd = [hipcc] + flags + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return obj


def build_extension(force=False, verbose=False):
    """hipcc --offload-arch=gfx950: csrc/*.hip -> one object each (in parallel) -> libnvsr_hip.so.  Returns the library path."""
    if not force and not _stale():
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(OBJ_DIR, exist_ok=True)
    if force:
        for o in glob.glob(os.path.join(OBJ_DIR, "*.o")):
            os.remove(o)
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(lambda src: _compile_one(hipcc, src, verbose), sources()))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH
