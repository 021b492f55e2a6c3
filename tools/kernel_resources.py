#!/usr/bin/env python3
"""Register / scratch / LDS budget of every kernel in libnvsr_hip.so, read from the gfx950 code objects' metadata.

    python tools/kernel_resources.py [--lib PATH] [--json] [--check]

The code objects are unbundled with `llvm-objdump --offloading` into a TEMPORARY directory (the tool drops them into its working
directory -- never run it inside the source tree) and their `amdhsa.kernels` notes are parsed.  --check applies the spill gate of
tests/test_host.py::test_no_spills_on_benchmarked_kernels to EVERY kernel: no scratch, no spilled vector registers, except the listed
exact-f32 opt-in kernels and ALLOW (name substring -> max spilled VGPRs)."""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, "neural-volume-super-resolution_amd", "libnvsr_hip.so")
LLVM_BIN = "/opt/rocm/lib/llvm/bin"

# Kernels that bench.py times (render / train / sr workloads).  Demangled-name substrings.
BENCHMARKED = [
    "render_pass3_kernel", "render_pass3_coarse", "importance_resample", "ray_bundle_kernel", "pack_rays_kernel",
    "decode_rays_limb_kernel", "render_pass_backward_gates_limb_kernel", "composite_kernel", "composite_backward_kernel",
    "view_reduce_scatter", "decoder_wgrad_limb_kernel", "head_wgrad_kernel",
    "conv3x3_limb_kernel", "conv3x3_limb16_kernel", "conv3x3_wgrad_limb_kernel", "sr_prepare_kernel", "sr_finish_kernel",
]
# The gate covers EVERY kernel of the library (round 4; round 3 covered the benchmarked ones): no scratch memory, no spilled VGPRs -- except:
#   ALLOW  name substring -> spilled VGPRs tolerated.  What is listed is debt, with the round that recorded it:
#     decoder_wgrad_limb_kernel<4>: 1 (8 B of scratch, one store before the row loop and three loads outside it); rounds 2-3 carried 35 in the
#       flush of its 256 accumulators -- the flush loop was unrolled four times with every accumulator copied out of its AGPR ahead of the copies,
#       and its LDS addresses were computed as 64-bit generic pointers (round 4: one copy, LDS-typed pointer with immediate offsets)
#   (rounds 1-4 exempted the exact-f32 MFMA kernels of render.hip / render_bwd.hip by name -- 23 to 88 spilled VGPRs, 96 to 352 B of scratch.
#    Round 5: they take one wave per SIMD (up to 512 registers; the recording gate-driven backward runs 4-wave workgroups) and spill nothing:
#    the exemption list is gone, every kernel of the library passes the same gate.)
#   Spilled VGPRs with 0 bytes of scratch are copies into free AGPRs (v_accvgpr_write): no memory traffic; tolerated up to 16.
ALLOW = {"decoder_wgrad_limb_kernelILi4": 1}


def _tool(name):
    p = os.path.join(LLVM_BIN, name)
    return p if os.path.exists(p) else shutil.which(name)


def kernel_table(lib=DEFAULT_LIB):
    """[{name, vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds, wg_max}] for every kernel of every gfx950 code object in lib"""
    objdump, readelf, filt = _tool("llvm-objdump"), _tool("llvm-readelf"), _tool("llvm-cxxfilt")
    if not (objdump and readelf):
        raise RuntimeError("llvm-objdump / llvm-readelf not found")
    import yaml

    rows = []
    with tempfile.TemporaryDirectory(prefix="nvsr_co_") as tmp:
        # llvm-objdump --offloading writes the bundles NEXT TO ITS INPUT: work on a copy in the temporary directory
        copy = os.path.join(tmp, "lib.so")
        shutil.copyfile(os.path.abspath(lib), copy)
        subprocess.run([objdump, "--offloading", copy], cwd=tmp, check=True, stdout=subprocess.DEVNULL)
        for f in sorted(os.listdir(tmp)):
            if "gfx950" not in f:
                continue
            notes = subprocess.run([readelf, "--notes", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            m = re.search(r"^\s*---\s*$(.*?)^\s*\.\.\.\s*$", notes, re.S | re.M)
            if not m:
                continue
            meta = yaml.safe_load(m.group(1))
            for k in meta.get("amdhsa.kernels", []):
                rows.append({key.lstrip("."): val for key, val in k.items() if key != ".args"})
    out = []
    names = [r.get("name", "") for r in rows if "vgpr_count" in r]
    dem = names
    if filt and names:
        dem = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    for r, d in zip([r for r in rows if "vgpr_count" in r], dem):
        out.append({"name": d, "mangled": r.get("name", ""), "vgpr": int(r.get("vgpr_count", 0)), "agpr": int(r.get("agpr_count", 0)), "sgpr": int(r.get("sgpr_count", 0)),
                    "vgpr_spill": int(r.get("vgpr_spill_count", 0)), "sgpr_spill": int(r.get("sgpr_spill_count", 0)),
                    "scratch": int(r.get("private_segment_fixed_size", 0)), "lds": int(r.get("group_segment_fixed_size", 0)),
                    "wg_max": int(r.get("max_flat_workgroup_size", 0))})
    return out


def violations(table, raw_names=None):
    """kernels that break the gate.  table rows carry demangled names; ALLOW matches mangled-name substrings (kept in `mangled`)."""
    bad = []
    for k in table:
        key = k.get("mangled", k["name"])
        allow = max([v for s, v in ALLOW.items() if s in key] or [0])
        if allow:
            if k["vgpr_spill"] > allow:
                bad.append(k)
        elif k["scratch"] > 0 or k["vgpr_spill"] > 16:
            bad.append(k)
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=DEFAULT_LIB)
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--all", action="store_true", help="list every kernel, not only those that spill or are benchmarked")
    a = ap.parse_args()
    t = kernel_table(a.lib)
    if a.json:
        print(json.dumps(t, indent=1))
    else:
        print("%-6s %-6s %-6s %-8s %-8s %-8s %-8s %s" % ("vgpr", "agpr", "sgpr", "v-spill", "s-spill", "scratch", "lds", "kernel"))
        for k in sorted(t, key=lambda k: -k["vgpr_spill"]):
            if a.all or k["vgpr_spill"] or k["sgpr_spill"] or any(b in k["name"] for b in BENCHMARKED):
                print("%-6d %-6d %-6d %-8d %-8d %-8d %-8d %s" % (k["vgpr"], k["agpr"], k["sgpr"], k["vgpr_spill"], k["sgpr_spill"], k["scratch"],
                                                             k["lds"], k["name"][:150]))
    if a.check:
        bad = violations(t)
        for k in bad:
            print("SPILL GATE: %s spills %d VGPRs (%d B scratch)" % (k["name"][:120], k["vgpr_spill"], k["scratch"]), file=sys.stderr)
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
