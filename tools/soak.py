import sys, os, time, torch
sys.path.insert(0, os.getcwd())
import bench
for W in ("cfg4", "cfg1"):
    wl = bench.WORKLOADS[W]
    model, step, fwd, nvox, _ = bench.make_step(wl, torch.device("cuda", 0))
    for _ in range(5): step()
    torch.cuda.synchronize()
    m0 = torch.cuda.memory_allocated(); r0 = torch.cuda.memory_reserved()
    free0, total = torch.cuda.mem_get_info()
    t0 = time.time()
    n = 300 if W == "cfg4" else 600
    for i in range(n): step()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    ok = all(torch.isfinite(p).all().item() for p in model.parameters())
    print(W, f"{n} steps in {time.time()-t0:.1f}s; torch allocated {m0/2**20:.0f} -> {torch.cuda.memory_allocated()/2**20:.0f} MiB, reserved {r0/2**20:.0f} -> {torch.cuda.memory_reserved()/2**20:.0f} MiB, device free {free0/2**20:.0f} -> {free1/2**20:.0f} MiB, params finite: {ok}", flush=True)
    del model, step, fwd
