"""Stability run: hundreds of steps of the bench workloads (eager, graph replay; fp32 in the default bf16x6 products, the opt-in bf16x3 in a
child process); memory must not grow and the parameters must stay finite.    python tools/soak.py"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.getcwd())
import torch

import bench

MODE = sys.argv[1] if len(sys.argv) > 1 else "all"
if MODE == "x3":
    import bio_image_unet_amd
    bio_image_unet_amd.set_fp32_products("bf16x3")
    runs = [("cfg1", False, 600), ("cfg2", False, 30)]
else:
    runs = [("cfg4", False, 300), ("cfg1", False, 600), ("cfg1", True, 600), ("cfg3", True, 100), ("cfg2", False, 30), ("cfg5", False, 20)]
for W, graph, n in runs:
    wl = bench.WORKLOADS[W]
    model, step, fwd, nvox, _ = bench.make_step(wl, torch.device("cuda", 0), graph=graph)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    m0, r0 = torch.cuda.memory_allocated(), torch.cuda.memory_reserved()
    free0, total = torch.cuda.mem_get_info()
    t0 = time.time()
    for i in range(n):
        step()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    ok = all(torch.isfinite(p).all().item() for p in model.parameters())
    print(W, "graph" if graph else "eager", "bf16x3" if MODE == "x3" else "",
          f"{n} steps in {time.time() - t0:.1f}s; torch allocated {m0 / 2 ** 20:.0f} -> {torch.cuda.memory_allocated() / 2 ** 20:.0f} MiB, "
          f"reserved {r0 / 2 ** 20:.0f} -> {torch.cuda.memory_reserved() / 2 ** 20:.0f} MiB, device free {free0 / 2 ** 20:.0f} -> {free1 / 2 ** 20:.0f} MiB, "
          f"params finite: {ok}", flush=True)
    del model, step, fwd
if MODE == "all":              # the product mode is process-wide: its own process
    sys.exit(subprocess.run([sys.executable, os.path.abspath(__file__), "x3"]).returncode)
