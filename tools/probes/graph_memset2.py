"""Second pass of graph_memset.py: which byte counts of a captured hipMemsetAsync replay wrongly, what the wrong elements hold, and where they lie."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bio_image_unet_amd.graph import _loaded_hip_runtime
hip = _loaded_hip_runtime()

def run(nbytes, value=0, pre_fill=5.0, replays=4, async_fn="hipMemsetAsync"):
    numel = (nbytes + 3) // 4
    ws = torch.full((numel,), pre_fill, device="cuda")
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.stream(side):
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            assert getattr(hip, async_fn)(C.c_void_p(ws.data_ptr()), value, C.c_size_t(nbytes), st) == 0
    g.instantiate()
    res = []
    for r in range(replays):
        ws.fill_(pre_fill); torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        b = ws.view(torch.uint8)[:nbytes]
        wrong = (b != value).nonzero().flatten()
        if wrong.numel():
            vals = b[wrong].unique().tolist()[:6]
            res.append(f"r{r}: {wrong.numel()} wrong bytes in [{int(wrong.min())}, {int(wrong.max())}], values {vals}")
        else:
            res.append(f"r{r}: ok")
    print(f"{async_fn} bytes {nbytes:9d} value {value:3d}: " + " | ".join(res), flush=True)

for nb in (4096, 65536, 110592, 110592 + 4, 131072, 262144, 524288, 1 << 20, 4 << 20):
    run(nb)
run(110592, value=0x5A)
run(110592, pre_fill=0.0)

# device-to-device copy nodes (what the select-backward of an indexed loss puts into a step): the source changes between replays
def run_copy(nbytes, replays=4):
    numel = nbytes // 4
    src = torch.zeros(numel, device="cuda"); dst = torch.full((numel,), -1.0, device="cuda")
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.stream(side):
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            assert hip.hipMemcpyAsync(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), C.c_size_t(nbytes), 3, st) == 0     # hipMemcpyDeviceToDevice
    g.instantiate()
    res = []
    for r in range(replays):
        src.fill_(float(r + 1)); dst.fill_(-1.0); torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        res.append("ok" if bool((dst == float(r + 1)).all()) else f"WRONG {dst.unique().tolist()[:4]}")
    print(f"hipMemcpyAsync D2D bytes {nbytes:9d}: " + " | ".join(f"r{i}: {v}" for i, v in enumerate(res)), flush=True)

for nb in (4096, 110592, 1 << 20, 16 << 20):
    run_copy(nb)
for nb in (4096, 110592, 1 << 20):
    run(nb // 4, async_fn="hipMemsetD32Async")      # count is in elements here: nb / 4 dwords
    run(nb // 2, async_fn="hipMemsetD16Async")
