"""cProfile of the host side of a training step (where does the enqueue time go?):  python tools/host_profile.py cfg1"""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg1"]
model, step, fwd, nvox, _avg = bench.make_step(wl, torch.device("cuda", 0))
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20): step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print("\n".join(l[:150] for l in s.getvalue().splitlines()))
