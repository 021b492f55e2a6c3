#!/usr/bin/env python3
"""Instruction histogram of one kernel of a hipcc object / library (gfx950 code object unbundled into a temporary directory).
   python tools/isa_hist.py <object or .so> <mangled-name substring> [top N] [--dump FILE]"""
import collections
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
obj, sub = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 24
dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
with tempfile.TemporaryDirectory(prefix="nvsr_isa_") as tmp:
    copy = os.path.join(tmp, "in.o")
    shutil.copyfile(obj, copy)
    subprocess.run([LLVM + "/llvm-objdump", "--offloading", copy], cwd=tmp, check=True, stdout=subprocess.DEVNULL)
    # (a library holds one code object per translation unit)
    text = "".join(subprocess.run([LLVM + "/llvm-objdump", "-d", os.path.join(tmp, co)], check=True, capture_output=True, text=True).stdout
                   for co in sorted(os.listdir(tmp)) if "gfx950" in co)
cur, bodies = None, {}
for line in text.splitlines():
    m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
    if m:
        cur = m.group(1)
        bodies[cur] = []
    elif cur and line.startswith("\t"):
        bodies[cur].append(line)
for name, body in bodies.items():
    if sub in name:
        h = collections.Counter(l.split()[0] for l in body)
        valu = sum(n for k, n in h.items() if k.startswith("v_") and "mfma" not in k)
        print("%s: %d instructions, %d VALU, %d MFMA" % (name, len(body), valu, sum(n for k, n in h.items() if "mfma" in k)))
        for k, n in h.most_common(top):
            print("   %6d %s" % (n, k))
        if dump:
            open(dump, "w").write("\n".join(body))
