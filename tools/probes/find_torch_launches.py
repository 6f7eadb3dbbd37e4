"""Which Python lines still launch torch kernels (fills, copies, elementwise) inside one train step of a bench workload: torch.profiler with
stacks over two steps, aggregated by the innermost repo frame.      python tools/probes/find_torch_launches.py [cfg4|cfg1|...]"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
model, step, fwd, nvox, avg = bench.make_step(bench.WORKLOADS[wl], torch.device("cuda:0"))
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(2):
        step()
    torch.cuda.synchronize()
agg = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue
    if not ev.kernels:
        continue
    frame = next((f for f in ev.stack if ROOT in f or "bench.py" in f), ev.stack[0] if ev.stack else "?")
    agg[(ev.name, frame.replace(ROOT + "/", ""))] += len(ev.kernels) or 1
for (name, frame), n in agg.most_common(40):
    print(f"{n / 2:6.1f} launches/step  {name:28s} {frame}")
