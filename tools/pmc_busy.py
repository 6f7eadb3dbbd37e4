"""MFMA-busy / wait / LDS counters per kernel of an isolated run (one rocprofv3 --pmc pass, --kernel-trace only) -> a markdown table.

    python tools/pmc_busy.py <out.md> <title> <dir of the pass> [<dir> ...]

Units (MI355X_MICROARCH.md, cycle constants): SQ_VALU_MFMA_BUSY_CYCLES is summed over SIMDs; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are
quad-cycles summed over waves; GRBM_GUI_ACTIVE is summed over the 8 XCDs.  The table reports ratios inside one family, and MFMA busy per
SIMD-cycle = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)."""
import collections
import csv
import glob
import sys

dst, title, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
with open(dst, "w") as f:
    f.write(f"# {title}\n\n`rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES "
            "GRBM_GUI_ACTIVE` (one pass) and `--pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE` (another) of `tools/bench_foldt.py` / `tools/bench_conv.py`; "
            "kernels under 20 us per dispatch left out.\n")
    for d in dirs:
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.Counter()
        for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(fn)):
                k = r["Kernel_Name"]
                if k.startswith("void at::"):
                    continue
                key = (k[:72], r.get("Grid_Size", r.get("Grid_Size_X", "?")))
                per[key][r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_LDS_IDX_ACTIVE"):
                    cnt[(key, r["Counter_Name"])] += 1
        f.write(f"\n## {d.rstrip('/').split('/')[-1]}\n\n| kernel | grid | dispatches | GUI_ACTIVE cycles per dispatch | MFMA busy per SIMD-cycle | WAIT_ANY / WAVE | WAIT_INST_ANY / WAVE | "
                "ACTIVE_INST_ANY / WAVE | ACTIVE_INST_LDS / WAVE | LDS bank conflict / LDS active |\n|---|---|---|---|---|---|---|---|---|---|\n")
        for key, c in sorted(per.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
            n = max(cnt[(key, "SQ_WAVE_CYCLES")], cnt[(key, "SQ_LDS_IDX_ACTIVE")], 1)
            gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
            if gui / n < 20e-6 * 1.4e9 and c.get("SQ_LDS_IDX_ACTIVE", 0) / n < 1e6:
                continue
            w = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
            busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(gui * 256 * 4, 1.0)
            conf = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0)
            f.write(f"| `{key[0]}` | {key[1]} | {n} | {gui / n:.0f} | {busy:.3f} | {c.get('SQ_WAIT_ANY', 0) / w:.3f} | {c.get('SQ_WAIT_INST_ANY', 0) / w:.3f} | "
                    f"{c.get('SQ_ACTIVE_INST_ANY', 0) / w:.3f} | {c.get('SQ_ACTIVE_INST_LDS', 0) / w:.3f} | {conf:.2f} |\n")
print("wrote", dst)
