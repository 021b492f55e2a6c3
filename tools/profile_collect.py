"""condense gpurun_out/<tag>/ (tools/profile_round.sh) into profiles/<tag>_*: bench lines, kernel-stat tables, HBM traffic of the
dominant kernel (FETCH_SIZE / WRITE_SIZE from separate --pmc passes; gfx950 correction per MI355X_MICROARCH.md)"""
import csv, glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")
out = {}
for w in ("render", "train", "train_dec", "sr", "refine_joint", "refine_sr"):
    f = os.path.join(src, "bench_%s.json" % w)
    if os.path.exists(f):
        line = [l for l in open(f).read().splitlines() if l.startswith("{")]
        if line:
            out[w] = json.loads(line[-1])
            json.dump(out[w], open(os.path.join(dst, "%s_bench_%s.json" % (tag, w)), "w"), indent=1)
    st = glob.glob(os.path.join(src, "stats_%s" % w, "*", "*kernel_stats.csv"))
    if st:
        rows = list(csv.DictReader(open(st[0])))
        with open(os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, w)), "w") as fo:
            wr = csv.writer(fo)
            wr.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows[:14]:
                wr.writerow([r["Name"][:120], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
CORRECTION = "gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide (16 B/lane) loads -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact"


def counters(prefix, pick):
    """{counter: (value_kb, kernel_ms)} of the dispatch(es) `pick` selects: pick(rows of one counter) -> (value, ms)"""
    pm = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in sorted(glob.glob(os.path.join(src, "%s%s" % (prefix, c), "*", "*counter_collection.csv")), key=os.path.getmtime)[-1:]:   # newest pass only
            rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c]
            got = pick(rows)
            if got:
                pm[c] = got
    return pm if len(pm) == 2 else None


def dur_ms(r):
    return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6


def longest(match):
    def pick(rows):
        rows = [r for r in rows if match(r["Kernel_Name"])]
        if not rows:
            return None
        r = max(rows, key=dur_ms)
        return float(r["Counter_Value"]), dur_ms(r)
    return pick


def summed(match):
    def pick(rows):
        rows = [r for r in rows if match(r["Kernel_Name"])]
        return (sum(float(r["Counter_Value"]) for r in rows), sum(dur_ms(r) for r in rows)) if rows else None
    return pick


sys.path.insert(0, root)
from bench import csrc_tree_hash  # noqa: E402
# the kernels these counters belong to: bench.py prints `traffic` only while the tree still hashes to this value
latest = {"round": tag, "correction": CORRECTION, "csrc_sha256": csrc_tree_hash()}
# render: the longest fused-pass launch = the fine pass (S = 192)
pm = counters("pmc_", longest(lambda k: "render_pass" in k and "backward" not in k))
if pm:
    traffic = (2.0 * pm["FETCH_SIZE"][0] + pm["WRITE_SIZE"][0]) * 1024.0
    latest.update({"kernel": "%s (fine pass, S=192)" % out.get("render", {}).get("roofline", {}).get("kernel", "render_pass kernel").split(" (")[0],
                   "decoder_arithmetic": out.get("render", {}).get("decoder_arithmetic", "f32"),
                   "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-modes (two separate passes)",
                   "fetch_size_kb": pm["FETCH_SIZE"][0], "write_size_kb": pm["WRITE_SIZE"][0],
                   "kernel_ms_under_pmc": [pm["FETCH_SIZE"][1], pm["WRITE_SIZE"][1]], "traffic_bytes": traffic})
    print("render: traffic per fine launch %.1f GB" % (traffic / 1e9))
# matrix-pipe busy fraction and clock of the same launch: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); clock =
# GRBM_GUI_ACTIVE / 8 / kernel time (MI355X_MICROARCH.md)
mf = {}
for f in sorted(glob.glob(os.path.join(src, "pmc_mfma", "*", "*counter_collection.csv")), key=os.path.getmtime)[-1:]:
    rows = [r for r in csv.DictReader(open(f)) if "render_pass" in r["Kernel_Name"] and "backward" not in r["Kernel_Name"]]
    for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_MFMA"):
        rc = [r for r in rows if r["Counter_Name"] == c]
        if rc:
            r = max(rc, key=dur_ms)
            mf[c] = (float(r["Counter_Value"]), dur_ms(r))
if len(mf) == 3 and "traffic_bytes" in latest:
    cyc = mf["GRBM_GUI_ACTIVE"][0] / 8.0
    latest["mfma"] = {"SQ_VALU_MFMA_BUSY_CYCLES": mf["SQ_VALU_MFMA_BUSY_CYCLES"][0], "GRBM_GUI_ACTIVE": mf["GRBM_GUI_ACTIVE"][0],
                      "SQ_INSTS_MFMA": mf["SQ_INSTS_MFMA"][0], "kernel_ms_under_pmc": mf["GRBM_GUI_ACTIVE"][1],
                      "mfma_busy_frac": mf["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (1024.0 * cyc), "clock_ghz": cyc / (mf["GRBM_GUI_ACTIVE"][1] * 1e6),
                      "command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA --kernel-trace -- python3 bench.py --steps 1 --warmup 0 "
                                 "--no-cpu-baseline --no-modes"}
    print("render: matrix pipe %.1f %% busy at %.2f GHz" % (100 * latest["mfma"]["mfma_busy_frac"], latest["mfma"]["clock_ghz"]))
# train: the longest gate-driven backward launch = the fine pass (S = 128); sr: every convolution of one step
for w, prefix, pick, what in (("train", "pmc_train_", longest(lambda k: "backward_gates" in k), "longest render_pass_backward_gates launch (fine pass)"),
                              ("train_dec", "pmc_train_dec_", longest(lambda k: "backward_gates" in k), "longest render_pass_backward_gates launch (fine pass)"),
                              ("sr", "pmc_sr_", summed(lambda k: "conv3x3" in k), "all conv3x3 launches of one step")):
    pm = counters(prefix, pick)
    if pm:
        traffic = (2.0 * pm["FETCH_SIZE"][0] + pm["WRITE_SIZE"][0]) * 1024.0
        latest[w] = {"what": what, "fetch_size_kb": pm["FETCH_SIZE"][0], "write_size_kb": pm["WRITE_SIZE"][0],
                     "kernel_ms_under_pmc": [pm["FETCH_SIZE"][1], pm["WRITE_SIZE"][1]], "traffic_bytes": traffic,
                     "arithmetic": out.get(w, {}).get("dtype", "")}
        print("%s: traffic %.2f GB (%s)" % (w, traffic / 1e9, what))
# refine: every convolution launch (forward, data gradient, weight gradient + its reduction) of the ONE timed iteration of the counter pass (the
# warm-up iteration's launches come first in the trace: the second half of the matching dispatches), and the split of the SR backward from the
# kernel-stat run: weight-gradient kernels by name; the data gradients share their kernels with the forward, so they are the backward's rest
def second_half(match):
    def pick(rows):
        rows = sorted([r for r in rows if match(r["Kernel_Name"])], key=lambda r: int(r["Start_Timestamp"]))
        rows = rows[len(rows) // 2:]
        return (sum(float(r["Counter_Value"]) for r in rows), sum(dur_ms(r) for r in rows)) if rows else None
    return pick


is_sr_conv = lambda k: "conv3x3" in k or "wgrad_reduce" in k
for w in ("refine_joint", "refine_sr"):
    pm = counters("pmc_%s_" % w, second_half(is_sr_conv))
    if pm:
        traffic = (2.0 * pm["FETCH_SIZE"][0] + pm["WRITE_SIZE"][0]) * 1024.0
        latest[w] = {"what": "all conv3x3 / weight-gradient launches of one iteration (3 ROI crops: forward + data gradients + weight gradients)",
                     "fetch_size_kb": pm["FETCH_SIZE"][0], "write_size_kb": pm["WRITE_SIZE"][0],
                     "kernel_ms_under_pmc": [pm["FETCH_SIZE"][1], pm["WRITE_SIZE"][1]], "traffic_bytes": traffic,
                     "arithmetic": out.get(w, {}).get("dtype", "")}
        print("%s: traffic %.2f GB" % (w, traffic / 1e9))
    st = glob.glob(os.path.join(src, "stats_%s" % w, "*", "*kernel_stats.csv"))
    line = os.path.join(src, "stats_%s.json" % w)
    if st and os.path.exists(line) and w in latest:
        try:
            b = json.loads([l for l in open(line).read().splitlines() if l.startswith("{")][-1])
            iters = b["steps"] + b["warmup"] + 3                       # timed + warm-up + the three probe iterations of the split
            rows = list(csv.DictReader(open(st[0])))
            tot = lambda pred: sum(int(r["TotalDurationNs"]) for r in rows if pred(r["Name"])) / 1e6 / iters
            wg = tot(lambda k: "wgrad" in k)
            conv = tot(lambda k: "conv3x3" in k and "wgrad" not in k)
            bw = b["split_ms"]["PlanesSR backward (3 ROI crops: data + weight gradients)"]
            fw = b["split_ms"]["PlanesSR forward (3 ROI crops, keeps activations)"]
            latest[w]["sr_backward_split_ms"] = {"weight gradients (conv3x3_wgrad_limb_kernel + wgrad_reduce_pieces_kernel)": wg,
                                                 "data gradients + the rest of the backward (same kernels as the forward: the backward's time minus the weight gradients)": bw - wg,
                                                 "all conv3x3 forward + data-gradient launches": conv, "forward (events)": fw, "backward (events)": bw,
                                                 "source": "rocprofv3 --kernel-trace --stats of bench.py --workload refine (per-iteration means over %d iterations)" % iters}
        except Exception as e:      # a condensed profile must not fail on a missing field
            print("refine split:", e)
# executed matrix instructions of the SR workloads against the algorithmic count (tile rounding on ragged crops): one more counter
# pass each; algorithmic = FLOP of the line x 3 limb products / 16 384 FLOP per v_mfma_f32_16x16x32 (the three launches per step on the narrow
# kernels execute 32x32x16 instructions of twice the FLOP: < 1 % of the count); a refine iteration's weight gradients -- a third of its FLOP --
# run conv3x3_wgrad_limb_kernel on v_mfma_f32_32x32x16 (32 768 FLOP per instruction): (2/3 + 1/6) of the 16x16x32 count
for w, pick in (("sr", summed(lambda k: "conv3x3" in k)), ("refine_joint", second_half(is_sr_conv))):
    for f in sorted(glob.glob(os.path.join(src, "pmc_%s_SQ_INSTS_MFMA" % w, "*", "*counter_collection.csv")), key=os.path.getmtime)[-1:]:
        got = pick([r for r in csv.DictReader(open(f)) if r["Counter_Name"] == "SQ_INSTS_MFMA"])
        roof = out.get(w, {}).get("roofline", {})
        flop = roof.get("algorithmic_flop_per_step")
        if got and w in latest and flop:
            alg = flop * 3.0 / 16384.0 * ((2.0 / 3.0 + 1.0 / 6.0) if w == "refine_joint" else 1.0)
            latest[w]["SQ_INSTS_MFMA"] = got[0]
            latest[w]["algorithmic_mfma_instructions"] = alg
            latest[w]["executed_over_algorithmic_mfma"] = got[0] / alg
            print("%s: SQ_INSTS_MFMA %.4g executed / %.4g algorithmic = %.3f" % (w, got[0], alg, got[0] / alg))
if "traffic_bytes" in latest:
    json.dump(latest, open(os.path.join(dst, "%s_pmc.json" % tag), "w"), indent=1)
    json.dump(latest, open(os.path.join(dst, "pmc_latest.json"), "w"), indent=1)
for w, r in out.items():
    print(w, "%.4g %s" % (r["value"], r["unit"]), "ms/step %.2f" % r["ms_per_step"], "roofline frac %.3f" % r["roofline"]["frac"])
