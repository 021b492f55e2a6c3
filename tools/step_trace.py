"""Kernel sequence of ONE training step from a rocprofv3 kernel trace (which small kernels sit between the render passes):
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_train -- python3 bench.py --workload train --steps 4 --warmup 2 --no-cpu-baseline
   python tools/step_trace.py gpurun_out/trace_train [marker-substring, default FusedOptimizer]"""
import csv
import glob
import sys

d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "FusedOptimizer"
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
if len(idx) < 2:
    sys.exit("fewer than two '%s' kernels in %s" % (marker, f))
a, b = idx[-2] + 1, idx[-1] + 1
t0 = int(rows[a - 1]["End_Timestamp"])
busy = 0
prev_end = t0
print("%8s %8s %8s  kernel" % ("start us", "dur us", "gap us"))
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    print("%8.1f %8.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r["Kernel_Name"][:150]))
    prev_end = e
print("step: %d kernels, %.1f us from marker to marker, %.1f us busy" % (b - a, (prev_end - t0) / 1e3, busy / 1e3))
