"""Condense a rocprofv3 --kernel-trace --stats run (csv) into profiles/<tag>_summary.md.

    python tools/summarize_profile.py gpurun_out/prof_r01/cfg4 profiles/r01_cfg4
"""
import collections
import csv
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
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
        if "k_conv_pipe" in n or "k_wgrad_pipe" in n:
            short = ("k_wgrad_pipe" if "wgrad" in n else "k_conv_pipe") + "<" + n.split("<")[-1][:40] if "<" in n else n[:40]
            d[(short, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for (k, gx, gy, gz), v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        f.write(f"| `{k}` | {gx},{gy},{gz} | {len(v)} | {sum(v) / len(v):.4f} |\n")
print("wrote", dst + "_summary.md")
