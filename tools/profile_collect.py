"""condense gpurun_out/<tag>/ (tools/profile_round.sh) into profiles/<tag>_*: bench lines, kernel-stat tables, HBM traffic of the
dominant kernel (FETCH_SIZE / WRITE_SIZE from separate --pmc passes; gfx950 correction per MI355X_MICROARCH.md)"""
import csv, glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")
out = {}
for w in ("render", "train", "train_dec", "sr"):
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
pm = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(src, "pmc_%s" % c, "*", "*counter_collection.csv")):
        best = None
        for r in csv.DictReader(open(f)):
            if "render_pass" not in r["Kernel_Name"] or "backward" in r["Kernel_Name"] or r["Counter_Name"] != c:
                continue
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            if best is None or dur > best[1]:
                best = (float(r["Counter_Value"]), dur)      # the longest launch = the fine pass (S = 192)
        if best:
            pm[c] = best
if len(pm) == 2:
    fetch_kb, write_kb = pm["FETCH_SIZE"][0], pm["WRITE_SIZE"][0]
    traffic = (2.0 * fetch_kb + write_kb) * 1024.0
    mode = out.get("render", {}).get("decoder_arithmetic", "f32")
    d = {"kernel": "%s (fine pass, S=192)" % out.get("render", {}).get("roofline", {}).get("kernel", "render_pass kernel").split(" (")[0],
         "decoder_arithmetic": mode,
         "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-modes (two separate passes)",
         "fetch_size_kb": fetch_kb, "write_size_kb": write_kb, "kernel_ms_under_pmc": [pm["FETCH_SIZE"][1], pm["WRITE_SIZE"][1]],
         "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide (16 B/lane) loads -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
         "traffic_bytes": traffic, "round": tag}
    json.dump(d, open(os.path.join(dst, "%s_pmc.json" % tag), "w"), indent=1)
    json.dump(d, open(os.path.join(dst, "pmc_latest.json"), "w"), indent=1)
    print("traffic per fine launch: %.1f GB" % (traffic / 1e9))
for w, r in out.items():
    print(w, "%.4g %s" % (r["value"], r["unit"]), "ms/step %.2f" % r["ms_per_step"], "roofline frac %.3f" % r["roofline"]["frac"])
