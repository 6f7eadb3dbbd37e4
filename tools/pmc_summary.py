"""Summarise tools/pmc_round.sh output into a markdown table: per kernel dispatch group (layer x leg) the MFMA-busy share of the
busy cycles, the wait shares and the effective clock.

    python tools/pmc_summary.py gpurun_out/pmc_r02 profiles/r02_cfg4_mfma_busy.md

Units (MI355X_MICROARCH.md, cycle constants): SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; SQ_BUSY_CYCLES counts per
shader engine; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; GRBM_GUI_ACTIVE is summed over the
8 XCDs.  The table therefore reports RATIOS inside one counter family plus MFMA-busy per SIMD-cycle = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 *
256 CUs * 4 SIMDs)."""
import collections, csv, glob, os, sys

src, dst = sys.argv[1], sys.argv[2]
rows = []
for d in sorted(glob.glob(os.path.join(src, "sq_*"))):
    if not os.path.isdir(d):
        continue
    layer = os.path.basename(d)[3:]
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f:
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(int)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "k_conv_pipe" not in k and "k_wgrad_pipe" not in k:
            continue
        key = ("wgrad" if "wgrad" in k else "conv") + " grid " + r.get("Grid_Size", r.get("Grid_Size_X", "?"))
        per[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES":
            cnt[key] += 1
    for key, c in per.items():
        n = max(cnt[key], 1)
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0                      # cycles of one XCD clock domain, summed over dispatches
        simd_cycles = gui * 256 * 4
        rows.append((layer, key, n, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(simd_cycles, 1), c.get("SQ_WAIT_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1),
                     c.get("SQ_WAIT_INST_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1), c.get("SQ_ACTIVE_INST_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1),
                     c.get("SQ_ACTIVE_INST_LDS", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1), gui / n))
with open(dst, "w") as f:
    f.write("# MFMA-busy and wait counters of the dominant kernels (rocprofv3 --pmc, tools/pmc_round.sh)\n\n")
    f.write("`tools/bench_conv.py cfg4 bf16 <layer>` (forward+statistics, data gradient, weight gradient with fused BatchNorm backward) under\n"
            "`rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS\n"
            "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`; dispatches grouped by kernel and grid.\n\n")
    f.write("| layer | kernel / grid | dispatches | MFMA busy (share of SIMD cycles) | WAIT_ANY / WAVE_CYCLES | WAIT_INST_ANY / WAVE | ACTIVE_INST_ANY / WAVE | ACTIVE_INST_LDS / WAVE | GUI_ACTIVE cycles per dispatch |\n|---|---|---|---|---|---|---|---|---|\n")
    for r in rows:
        f.write(f"| {r[0]} | {r[1]} | {r[2]} | {r[3]:.3f} | {r[4]:.3f} | {r[5]:.3f} | {r[6]:.3f} | {r[7]:.3f} | {r[8]:.0f} |\n")
    for kind, name in (("rd", "FETCH_SIZE (x2: gfx950 wide-read correction)"), ("wr", "WRITE_SIZE")):
        for d in sorted(glob.glob(os.path.join(src, kind + "_*"))):
            if not os.path.isdir(d):
                continue
            fcsv = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
            if not fcsv:
                continue
            tot = collections.defaultdict(float); n = collections.defaultdict(int)
            for r in csv.DictReader(open(fcsv[0])):
                k = r["Kernel_Name"]
                if ("k_conv_pipe" in k or "k_wgrad_pipe" in k) and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                    key = ("wgrad" if "wgrad" in k else "conv") + " grid " + r.get("Grid_Size", "?")
                    tot[key] += float(r["Counter_Value"]) * 1024.0 * (2.0 if kind == "rd" else 1.0); n[key] += 1
            f.write(f"\n**{os.path.basename(d)}** {name}: " + "; ".join(f"{k}: {v / max(n[k], 1) / 1e9:.3f} GB per dispatch ({n[k]} dispatches)" for k, v in tot.items()) + "\n")
print("wrote", dst)
