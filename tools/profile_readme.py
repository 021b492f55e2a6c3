"""regenerate the `## <tag>` section of profiles/r01_README.md from the condensed files of that tag (tools/profile_collect.py <tag>)
usage: profile_readme.py <tag> <title>"""
import csv, json, os, sys
tag, title = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda n: os.path.join(root, "profiles", n)
R, T, TD, SR = (json.load(open(P("%s_bench_%s.json" % (tag, w)))) for w in ("render", "train", "train_dec", "sr"))
PM = json.load(open(P("%s_pmc.json" % tag)))


def kst(w, key):
    for r in csv.DictReader(open(P("%s_%s_kernel_stats.csv" % (tag, w)))):
        if key in r["Name"]:
            return "%.2f ms (%s calls, min %.2f / max %.2f)" % (float(r["AverageNs"]) / 1e6, r["Calls"], float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6)
    return "n/a"


def st(k):
    v = [x for n, x in R["hbm_stages"].items() if k in n][0]
    return "%.0f GB/s (%.3f ms)" % (v["GB/s"], v["ms"])


gb = lambda w, k: PM[w][k] * 1024 / 1e9
sec = """## %s (%s)
`tools/profile_round.sh %s`: per workload the bench line (`%s_bench_{render,train,train_dec,sr}.json`; `train` = BASELINE configs[3]
exactly, `Feature_Planes_Only.yml`; `train_dec` = `--train-what planes+decoder`), `rocprofv3 --kernel-trace --stats` of the same command
(`%s_*_kernel_stats.csv`, top 14 kernels) and two `--pmc` passes (FETCH_SIZE, WRITE_SIZE) per workload (`%s_pmc.json` =
`pmc_latest.json`, which `bench.py` reads for `roofline.traffic`).  `tools/profile_readme.py` writes this section.

| workload | value | ms / step | dominant kernel (live, events) | rocprofv3 average of that kernel | achieved | of its roof | HBM-side traffic (PMC) |
|---|---|---|---|---|---|---|---|
| render 800x800, 64+128 (3-limb arithmetic) | %.3g rays/s | %.1f | `render_pass3_kernel<3>` (fine pass) %.1f ms | %s | %.1f TFLOP/s of f32 work | %.1f %% of 2 516.6 / 6 | %.1f GB per fine launch (377 GB of algorithmic gathers) |
| train, planes only (C4) | %.3g rays/s | %.2f | `render_pass_backward_gates_limb_kernel<false>` fine pass %.2f ms | %s over coarse + fine launches | %.0f TFLOP/s | %.0f %% of 2 516.6 / 6 (%.0f %% of the f32 peak) | %.2f GB per fine launch, %.2f GB of it written: the float atomics, resolved memory-side (plain stores: 0.1 GB of view rows) |
| train, planes + decoders | %.3g rays/s | %.2f | `render_pass_backward_gates_limb_kernel<true>` fine pass %.2f ms | %s | %.0f TFLOP/s | %.0f %% | %.2f GB (%.2f GB written: atomics + 2.1 GB of gradient-record rows) |
| SR 3 planes 200^2 -> 800^2 | %.1f planes/s | %.1f | `conv3x3_limb_kernel` x 70 | see `%s_sr_kernel_stats.csv` | %.0f TFLOP/s over the step | %.1f %% of 2 516.6 / 6 | %.0f GB per step = %.2f TB/s (about 30 GB algorithmic) |

The coarse and the fine render pass are separate kernel symbols (`render_pass3_coarse_kernel<3>`, `render_pass3_kernel<3>`), so the
`AverageNs` of `render_pass3_kernel<3>` in `%s_render_kernel_stats.csv` IS the fine-pass duration that `roofline.kernel_ms` reports (the
training kernels are launched for both passes: min = coarse S = 64, max = fine S = 128).
`hbm_stages` of the render line (bandwidth-bound helpers, algorithmic bytes / event time): ray generation %s, ray packing %s, coarse
depths %s, inverse-CDF resampling + merge %s (2.75 ms with the O(n^2) rank sort of `r01d`); plane sampling + compositing inside the fused
pass: %s of HBM traffic.
Train steps over the round: planes + decoders 10.15 ms (`r01d`) -> 8.70 (`r01e`: no host wait inside the step, limb weight gradient) ->
%.2f ms (limb forward / backward, run-merged atomics, ray-major record); planes only 4.89 -> %.2f ms.  Kernel split of the planes + decoder
step (`%s_train_dec_kernel_stats.csv`): backward %s, forward %s, contraction %s, heads %s.

""" % (tag, title, tag, tag, tag, tag,
       R["value"], R["ms_per_step"], R["roofline"]["kernel_ms"], kst("render", "render_pass3_kernel"), R["roofline"]["achieved"], 100 * R["roofline"]["frac"], PM["traffic_bytes"] / 1e9,
       T["value"], T["ms_per_step"], T["roofline"]["kernel_ms"], kst("train", "backward_gates"), T["roofline"]["achieved"], 100 * T["roofline"]["frac"],
       100 * T["roofline"]["vs_f32_mfma_peak"], PM["train"]["traffic_bytes"] / 1e9, gb("train", "write_size_kb"),
       TD["value"], TD["ms_per_step"], TD["roofline"]["kernel_ms"], kst("train_dec", "backward_gates"), TD["roofline"]["achieved"], 100 * TD["roofline"]["frac"],
       PM["train_dec"]["traffic_bytes"] / 1e9, gb("train_dec", "write_size_kb"),
       SR["value"], SR["ms_per_step"], tag, SR["roofline"]["achieved"], 100 * SR["roofline"]["frac"], PM["sr"]["traffic_bytes"] / 1e9,
       PM["sr"]["traffic_bytes"] / 1e9 / SR["ms_per_step"],
       tag, st("get_ray_bundle"), st("pack_rays"), st("coarse depths"), st("sample_pdf"), st("plane sampling"),
       TD["ms_per_step"], T["ms_per_step"], tag,
       kst("train_dec", "backward_gates"), kst("train_dec", "decode_rays"), kst("train_dec", "wgrad_limb_kernel<4>"), kst("train_dec", "head_wgrad"))
path = P("r01_README.md")
s = open(path).read()
a = s.index("## %s (" % tag)
b = s.index("\n## ", a + 4) + 1
open(path, "w").write(s[:a] + sec + s[b:])
print(sec[:300])
