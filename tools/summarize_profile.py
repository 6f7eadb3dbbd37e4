"""Condense a rocprofv3 --kernel-trace --stats run (csv) into profiles/<tag>_summary.md.

    python tools/summarize_profile.py gpurun_out/prof_r01/cfg4 profiles/r01_cfg4 [--dominant "<kernel substring>" <grid_y> "<api @ label>" <flop>]
                                      [--call <kernel_trace.csv> <calls> "<api @ label>" <reference TFLOP> <executed TFLOP> "<excluded kernels>"]

--dominant adds a per-dispatch table of the call bench.py names in `roofline.kernel`: its dispatches are the ones of that kernel
template whose grid has `grid_y` workgroup columns (the decode5 data gradient is the only launch of its template with 3 output tiles), so
`roofline.frac` = flop / duration / peak can be recomputed from this file alone.
"""
import collections
import csv
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
dom = None
if "--dominant" in sys.argv:
    i = sys.argv.index("--dominant")
    dom = (sys.argv[i + 1], sys.argv[i + 2], sys.argv[i + 3], float(sys.argv[i + 4]))
stats = list(csv.DictReader(open(src + "_kernel_stats.csv")))
trace = list(csv.DictReader(open(src + "_kernel_trace.csv")))
shutil.copy(src + "_kernel_stats.csv", dst + "_kernel_stats.csv")
tot = sum(int(r["TotalDurationNs"]) for r in stats)
with open(dst + "_summary.md", "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats summary ({src})\n\n")
    f.write(f"total kernel time {tot / 1e6:.2f} ms over the profiled process\n\n| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for r in stats[:25]:
        f.write(f"| `{r['Name'][:90]}` | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {r['Percentage']} |\n")
    f.write("\n## per-grid average duration of the MFMA kernels (one row per layer shape)\n\n| kernel | grid (x,y,z) | dispatches | avg ms |\n|---|---|---|---|\n")
    d = collections.defaultdict(list)
    for r in trace:
        n = r["Kernel_Name"]
        if "k_conv_pipe" in n or "k_wgrad_pipe" in n or "k_wgrad_roll" in n or "k_conv16_pipe" in n:
            base = "k_wgrad_roll" if "k_wgrad_roll" in n else ("k_wgrad_pipe" if "wgrad" in n else ("k_conv16_pipe" if "conv16" in n else "k_conv_pipe"))
            short = base + "<" + n.split("<")[-1][:40] if "<" in n else n[:40]
            d[(short, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for (k, gx, gy, gz), v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        f.write(f"| `{k}` | {gx},{gy},{gz} | {len(v)} | {sum(v) / len(v):.4f} |\n")
if dom:
    sub, gy, label, flop = dom
    rows = [r for r in trace if sub in r["Kernel_Name"] and r["Grid_Size_Y"] == gy]
    with open(dst + "_summary.md", "a") as f:
        f.write(f"\n## per-dispatch table of the dominant call: {label}\n\nkernel `{sub}`, grid y = {gy}; algorithmic work {flop / 1e12:.4f} TFLOP per dispatch "
                "(2 * voxels * taps * Cin * Cout, SURVEY 8d)\n\n| # | kernel | grid (threads x, y, z) | workgroup | start (ms into the trace) | duration us | TFLOP/s | frac of 2500 |\n|---|---|---|---|---|---|---|---|\n")
        t0 = min(int(r["Start_Timestamp"]) for r in trace)
        durs = []
        for i, r in enumerate(rows):
            du = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            durs.append(du)
            f.write(f"| {i} | `{r['Kernel_Name'][:60]}` | {r['Grid_Size_X']},{r['Grid_Size_Y']},{r['Grid_Size_Z']} | {r['Workgroup_Size_X']} | "
                    f"{(int(r['Start_Timestamp']) - t0) / 1e6:.3f} | {du:.1f} | {flop / du / 1e6:.0f} | {flop / du / 1e6 / 2500:.3f} |\n")
        if durs:
            durs.sort()
            med = durs[len(durs) // 2]
            f.write(f"\n{len(durs)} dispatches: mean {sum(durs) / len(durs):.1f} us, median {med:.1f} us -> {flop / med / 1e6:.0f} TFLOP/s = {flop / med / 1e6 / 2500:.3f} of the bf16 MFMA peak "
                    "(under the profiler every dispatch is serialised and runs a few per cent slower than inside bench.py's timed region).\n")
if "--call" in sys.argv:
    # --call <kernel_trace.csv of an isolated run of ONE C-ABI call (tools/bench_foldt.py with BENCH_LEGS=<leg>)> <calls in that run> "<api @ label>"
    #        <TFLOP of the reference ops the call stands for> <TFLOP executed> "<'|'-separated substrings of kernels that are NOT part of the call>"
    i = sys.argv.index("--call")
    ctrace, ncalls, label, ref_tf, exe_tf, excl = sys.argv[i + 1], int(sys.argv[i + 2]), sys.argv[i + 3], float(sys.argv[i + 4]), float(sys.argv[i + 5]), sys.argv[i + 6].split("|")
    rows = [r for r in csv.DictReader(open(ctrace)) if not r["Kernel_Name"].startswith("void at::") and not any(x and x in r["Kernel_Name"] for x in excl)]
    order, agg = [], collections.OrderedDict()
    for r in rows:
        k = (r["Kernel_Name"], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
        agg.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    with open(dst + "_summary.md", "a") as f:
        f.write(f"\n## the dominant call, kernel by kernel: {label}\n\n`{ctrace.split('gpurun_out/')[-1]}`: {ncalls} calls of the one C-ABI entry point in isolation "
                f"(`tools/bench_foldt.py bf16 decode5`), rocprofv3 kernel trace of the FETCH_SIZE pass (every dispatch serialised).  Work of the reference ops the call "
                f"stands for (SURVEY 8d): {ref_tf:.4f} TFLOP per call; the folded kernels execute {exe_tf:.4f} TFLOP.  HBM traffic of the call: `profiles/pmc_traffic_cfg4.json`.\n\n"
                "| kernel | grid (threads x,y,z) | dispatches per call | avg us | us per call |\n|---|---|---|---|---|\n")
        tot = 0.0
        for (k, gx, gy, gz), v in agg.items():
            per = sum(v) / ncalls
            tot += per
            f.write(f"| `{k[:100]}` | {gx},{gy},{gz} | {len(v) / ncalls:.2f} | {sum(v) / len(v):.1f} | {per:.1f} |\n")
        f.write(f"\nsum: {tot:.1f} us per call -> {ref_tf / tot * 1e6:.0f} TFLOP/s of reference work = {ref_tf / tot * 1e6 / 2500:.3f} of the bf16 MFMA peak "
                f"({exe_tf / tot * 1e6:.0f} TFLOP/s executed); bench.py's `roofline.launch_ms` is the same call timed with HIP events inside the step.\n")
print("wrote", dst + "_summary.md")
