"""Probe for the round-2 observation 'a captured hipMemsetAsync lost its order against the kernels around it once eager work ran between
two replays' (bio_image_unet_amd/graph.py): the smallest graph with that shape, outside the engine.

    capture:   memset(ws, 0)  ->  ws += 1 (kernel)  ->  out = ws + 0 (kernel)
    between replays: eager work that dirties ws on the default stream (what an eager validation step between two graphed train steps does)

Every replay must leave out == 1.  Prints, per memset flavour and size, the graph's node list with each node's dependencies
(hipGraphNodeGetDependencies) and the first replay whose result is wrong, if any.      python tools/probes/graph_memset.py
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bio_image_unet_amd.graph import _NODE_KINDS, _loaded_hip_runtime  # noqa: E402

hip = _loaded_hip_runtime()


def dump(graph):
    raw = C.c_void_p(graph.raw_cuda_graph())
    n = C.c_size_t(0)
    assert hip.hipGraphGetNodes(raw, None, C.byref(n)) == 0
    nodes = (C.c_void_p * n.value)()
    assert hip.hipGraphGetNodes(raw, nodes, C.byref(n)) == 0
    idx = {nodes[i]: i for i in range(n.value)}
    out = []
    for i in range(n.value):
        t = C.c_int(-1)
        hip.hipGraphNodeGetType(C.c_void_p(nodes[i]), C.byref(t))
        nd = C.c_size_t(0)
        hip.hipGraphNodeGetDependencies(C.c_void_p(nodes[i]), None, C.byref(nd))
        deps = (C.c_void_p * max(nd.value, 1))()
        if nd.value:
            hip.hipGraphNodeGetDependencies(C.c_void_p(nodes[i]), deps, C.byref(nd))
        out.append(f"{i}:{_NODE_KINDS.get(t.value, t.value)}<-{[idx.get(deps[j], '?') for j in range(nd.value)]}")
    return " ".join(out)


def run(flavour, numel, dirty, replays=6):
    ws = torch.full((numel,), 5.0, device="cuda")
    out = torch.empty_like(ws)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.stream(side):
        for _ in range(2):                                   # warm-up on the capture stream
            ws.zero_(); ws.add_(1.0); torch.add(ws, 0.0, out=out)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            if flavour == "memset8":
                rc = hip.hipMemsetAsync(C.c_void_p(ws.data_ptr()), 0, C.c_size_t(numel * 4), st)
            elif flavour == "memset32":
                rc = hip.hipMemsetD32Async(C.c_void_p(ws.data_ptr()), 0, C.c_size_t(numel), st)
            else:
                ws.zero_(); rc = 0                           # torch's fill kernel (what the library's k_zero_f32 amounts to)
            assert rc == 0, rc
            ws.add_(1.0)
            torch.add(ws, 0.0, out=out)
    g.instantiate()
    nodes = dump(g)
    bad = None
    for r in range(replays):
        if dirty == "eager_same_stream":
            ws.fill_(7.0)                                    # default stream, no sync: stream order is all that separates it from the replay
        elif dirty == "eager_other_stream":
            with torch.cuda.stream(side):
                ws.fill_(7.0)
            torch.cuda.current_stream().wait_stream(side)
        elif dirty == "eager_sync":
            ws.fill_(7.0); torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        v = out.unique().tolist()
        if v != [1.0] and bad is None:
            bad = (r, v[:4])
    print(f"{flavour:9s} numel {numel:9d} dirty {dirty:18s}: {'OK' if bad is None else 'WRONG at replay %d: out holds %s' % bad} | {nodes}", flush=True)


if __name__ == "__main__":
    print(torch.__version__, torch.version.hip)
    for flavour in ("memset8", "memset32", "fill_kernel"):
        for numel in (27 * 32 * 32, 1 << 20, (1 << 20) + 3):
            for dirty in ("none", "eager_same_stream", "eager_other_stream", "eager_sync"):
                run(flavour, numel, dirty)
