"""Turn bench.py's per-call breakdown (profiles/<tag>_bench_breakdown.txt) into a per-call roofline table (markdown):
each C-ABI call of one step with its HIP-event time, algorithmic TFLOP/s and GB/s (SURVEY 8d convention) and the fraction of
the roofline that bounds it (bf16 MFMA 2.5 PFLOP/s or HBM 8 TB/s, whichever gives the longer ideal time).

    python tools/roofline_table.py profiles/r01_cfg4_bench_breakdown.txt profiles/r01_cfg4_roofline.md [bf16|f32]"""
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
dt = sys.argv[3] if len(sys.argv) > 3 else "bf16"
MFMA = {"bf16": 2500.0, "f32": 157.3}[dt]
HBM = 8000.0
rows, head = [], ""
for line in open(src):
    if line.startswith("#"):
        head = line[1:].strip()
        continue
    m = re.match(r"\s*([\d.]+) ms\s+(\S+)\s+(\S+)\s+([\d.]+) TFLOP/s\s+([\d.]+) GB/s", line)
    if m:
        rows.append((float(m.group(1)), m.group(2), m.group(3), float(m.group(4)), float(m.group(5))))
tot = sum(r[0] for r in rows)
with open(dst, "w") as f:
    f.write(f"# Per-call roofline of one training step ({head})\n\n")
    f.write("Times are HIP events around each C-ABI call; FLOP and bytes are the algorithmic figures of SURVEY.md 8d for that call "
            "(fused BatchNorm / transform traffic is not counted, so fused calls are priced against the bare operator).\n\n")
    f.write("| call | layer | ms | % of step | TFLOP/s | GB/s | bound | fraction of that roofline |\n|---|---|---|---|---|---|---|---|\n")
    for ms, api, label, tf, gb in rows:
        if ms < 0.04:
            continue
        t_m, t_h = tf / MFMA, gb / HBM
        bound, frac = ("mfma", t_m) if t_m >= t_h else ("hbm", t_h)
        if tf == 0 and gb == 0:
            bound, frac = "-", 0.0
        f.write(f"| `{api}` | {label} | {ms:.3f} | {100 * ms / tot:.1f} | {tf:.0f} | {gb:.0f} | {bound} | {frac:.2f} |\n")
    f.write(f"\nsum of calls: {tot:.2f} ms\n")
print("wrote", dst)
