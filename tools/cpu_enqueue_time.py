import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg4"]
model, step, fwd, nvox, _avg = bench.make_step(wl, torch.device("cuda", 0))
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/10:.2f} ms/step, total {1e3*(t2-t0)/10:.2f} ms/step")
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(); t1 = time.perf_counter()
    ts.append(1e3 * (t1 - t0))
print("host time to enqueue ONE step on an idle queue (ms):", ", ".join(f"{t:.2f}" for t in ts))
