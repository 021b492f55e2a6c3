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


STAMP_PATH = LIB_PATH + ".flags"


def _flag_stamp():
    """what the library at LIB_PATH must have been built with: the compiler flags of every file + the experiment flags of the
    environment.  Stored next to the library; a library whose stamp differs (e.g. an interrupted `NVSR_EXTRA_HIPCC_FLAGS=-DBL_ABLATE=...`
    experiment, which produces WRONG results by design) is stale no matter how new it is."""
    extra = os.environ.get("NVSR_EXTRA_HIPCC_FLAGS", "").split()
    return " ".join(HIPCC_FLAGS) + " | " + " ".join("%s:%s" % kv for kv in sorted((k, ",".join(v)) for k, v in PER_FILE_FLAGS.items())) + " | " + " ".join(extra)


def _stamp_matches():
    try:
        with open(STAMP_PATH) as f:
            return f.read() == _flag_stamp()
    except OSError:
        return False


def _stale():
    if not os.path.exists(LIB_PATH) or not _stamp_matches():
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(PKG_DIR, "..", "include", "*.h")) + [os.path.abspath(__file__)]
    return any(os.path.getmtime(p) > t for p in deps)


# Per-file flags.  render3.hip: the SLP vectorizer packs pairs of f32 operations out of different MFMA gaps into v_pk_* and, to do so,
# re-schedules the whole block at IR level -- every pure instruction (the MFMAs included) sinks below the block's loads and
# sched_barriers, the hand-placed interleave is gone and hundreds of registers spill.
# decode_limb.hip: same blocks, same reason (250 spilled registers with the vectorizer, none without).
# -pragma-unroll-threshold: the blocks of the tile-pair kernels are `#pragma unroll` loops whose side work is selected by the (constant) slot
# index; before the selection folds away the body counts every alternative, and beyond 16 k instructions LLVM ignores the pragma -- the slot
# index then reaches an asm immediate ("constraint 'n' expects an integer constant expression") or a register index as a run-time value.
_PAIR_FLAGS = ["-fno-slp-vectorize", "-mllvm", "-pragma-unroll-threshold=1000000"]
# sr.hip / sr_bwd.hip: without the post-RA machine scheduler (round 5).  The convolution kernels place their fragment loads, LDS reads and MFMA groups
# by hand between sched_barriers; the pre-RA scheduler respects those, the post-RA pass still reorders inside the regions.  Same box, alternating
# (bench.py --workload sr / refine): SR stage 49.2 -> 48.0 ms, refine iteration 86.4 -> 84.9 ms, same bits (the SR tests).  The render / training
# kernels do not gain (fine pass 77.1 -> 77.5 ms, planes-only backward 0.58 -> 0.595 ms with the flag): they keep the default.  Other strategies
# tried on these two files (max-ilp, max-memory-clause, no pre-RA scheduler): none better (scratch timings of the session, profiles/r05_README.md).
_SR_FLAGS = ["-mllvm", "-enable-post-misched=0"]
PER_FILE_FLAGS = {"render3.hip": _PAIR_FLAGS, "decode_pair.hip": _PAIR_FLAGS, "decode_limb.hip": ["-fno-slp-vectorize"], "render_bwd_limb.hip": ["-fno-slp-vectorize"],
                  "sr.hip": _SR_FLAGS, "sr_bwd.hip": _SR_FLAGS}
OBJ_DIR = os.path.join(CSRC, "_obj")      # (of the product build; experiment variants use <out_path>.obj)


def _compile_one(hipcc, src, verbose):
    obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
    deps = [src] + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(PKG_DIR, "..", "include", "*.h")) + [os.path.abspath(__file__)]
    stamp = obj + ".flags"
    same_flags = os.path.exists(stamp) and open(stamp).read() == _flag_stamp()
    if same_flags and os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps):
        return obj
    flags = [f for f in HIPCC_FLAGS if f != "-shared"] + PER_FILE_FLAGS.get(os.path.basename(src), [])
    flags += os.environ.get("NVSR_EXTRA_HIPCC_FLAGS", "").split()       # experiments (e.g. -DR3_STAMP=1); use with build_extension(force=True)
    cmd = [hipcc] + flags + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(obj + ".flags", "w") as f:
        f.write(_flag_stamp())
    return obj


def build_extension(force=False, verbose=False, out_path=None):
    """hipcc --offload-arch=gfx950: csrc/*.hip -> one object each (in parallel) -> libnvsr_hip.so.  Returns the library path.
    out_path: build an EXPERIMENT variant (NVSR_EXTRA_HIPCC_FLAGS) to another file, with its own object directory, and leave the product
    library alone -- load it with NVSR_HIP_LIB=<out_path> (capi.py); tools/*.sh do this."""
    global OBJ_DIR
    if out_path is not None:
        return _build_variant(out_path, verbose)
    if os.environ.get("NVSR_EXTRA_HIPCC_FLAGS", "").split() and not os.environ.get("NVSR_ALLOW_INPLACE_EXPERIMENT"):
        raise RuntimeError("NVSR_EXTRA_HIPCC_FLAGS is set: experiment builds go to a separate file (build_extension(out_path=...) + "
                           "NVSR_HIP_LIB), never over the product library")
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
    with open(STAMP_PATH, "w") as f:
        f.write(_flag_stamp())
    return LIB_PATH


def _build_variant(out_path, verbose):
    """all sources with the current NVSR_EXTRA_HIPCC_FLAGS -> out_path (objects under <out_path>.obj/).
    NVSR_VARIANT_ONLY="a.hip b.hip": only these files see the extra flags and are recompiled; the other objects are the product build's
    (which must be current) -- a one-file switch builds in seconds instead of a minute."""
    global OBJ_DIR
    from concurrent.futures import ThreadPoolExecutor

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    saved = OBJ_DIR
    only = os.environ.get("NVSR_VARIANT_ONLY", "").split()
    if os.path.exists(out_path):
        os.remove(out_path)          # a failed build must not leave an older variant (other flags) behind to be timed under this name
    try:
        objs = []
        if only:
            extra = os.environ.pop("NVSR_EXTRA_HIPCC_FLAGS", "")
            try:
                objs = [_compile_one(hipcc, src, verbose) for src in sources() if os.path.basename(src) not in only]      # product objects
            finally:
                os.environ["NVSR_EXTRA_HIPCC_FLAGS"] = extra
        OBJ_DIR = os.path.abspath(out_path) + ".obj"
        os.makedirs(OBJ_DIR, exist_ok=True)
        mine = [src for src in sources() if not only or os.path.basename(src) in only]
        with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
            objs += list(pool.map(lambda src: _compile_one(hipcc, src, verbose), mine))
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out_path])
    finally:
        OBJ_DIR = saved
    return out_path
