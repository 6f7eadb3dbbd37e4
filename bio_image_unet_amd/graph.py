"""A whole training step -- forward, loss, backward, Adam -- captured once in a hipGraph and replayed per batch.

The engine is a static graph over preallocated buffers and every kernel of a step is enqueued through the C ABI on torch's current stream,
so the ~280 launches of a 2-D U-Net step can be recorded by ``torch.cuda.CUDAGraph`` (hipGraph on ROCm) and replayed with one call: the
host cost of a step drops from ~4 ms of Python / ctypes / launch calls to the replay.  That matters where a step is short
(``Unet(1, 1, 32)`` on 2 x 256 x 256: GPU time < host time); a step of the 3-D workloads is GPU-bound either way.

What differs from the eager step, and how it is kept equal to it:
  * inputs and targets are copied into static tensors before every replay;
  * Adam's scalars (learning rate, bias correction of THIS step) are read from device memory, refreshed by a one-thread kernel in front
    of every replay (``Adam.refresh_hyper``), so ``ReduceLROnPlateau`` (``unet/train.py:103``) keeps working;
  * the warm-up steps that precede the capture run on copies: parameters, BatchNorm buffers and optimizer state are restored afterwards;
  * parameter version counters are bumped after every replay, so an eager forward in between (validation) re-packs the weights.
Single process only: the gradient all-reduce of ``ddp.GradAverager`` is not captured.

The graph must hold no MEMSET nodes.  Round 2 saw weight gradients of 1e35 at the third replay of a step whose workspace was zeroed with
``hipMemsetAsync`` and took it for lost ordering.  Round 3 found the cause (``tools/probes/graph_memset.py``, ``graph_memset2.py``;
``tests/test_gpu_graph.py``): on this stack (ROCm 7.0 runtime under torch 2.10) a memset NODE writes the right pattern at the first launch
of the instantiated graph and a corrupted 16-byte pattern (zeros except for a few bytes that look like launch parameters, the byte count
among them) at every later launch -- a graph of ONE memset node shows it, all three widths (``hipMemsetAsync`` / ``D16`` / ``D32``), any
size from 4 KiB to 4 MiB, with or without eager work in between; the node's dependencies (``hipGraphNodeGetDependencies``) are captured
correctly, so it never was an ordering problem.  Device-to-device copy nodes replay correctly (4 KiB - 16 MiB, source changed between
replays), so steps that contain them -- the select-backward of an indexed loss, the stacked heads of ``MultiOutputUnet3D`` -- are captured;
the library zero-fills with a kernel of its own (``k_zero_f32``) and Adam's pointer table is uploaded after the capture.
"""
from __future__ import annotations

import copy
from typing import Callable, Sequence

import torch

from .optim import Adam


_NODE_KINDS = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "wait_event", 7: "event_record"}


def _loaded_hip_runtime():
    """The HIP runtime THIS process already runs on (the copy torch loaded), never a second one: a box can hold another
    ``libamdhip64.so`` (/opt/rocm next to torch's own), and a ``hipGraph_t`` handed to a different runtime is undefined behaviour.
    The path comes from /proc/self/maps and is opened with RTLD_NOLOAD, which fails instead of loading anything new."""
    import ctypes as C
    import os
    path = None
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64.so" in line:
                path = line.split()[-1]
                break
    if path is None:
        raise OSError("no libamdhip64.so is mapped into this process")
    return C.CDLL(path, mode=getattr(os, "RTLD_NOLOAD", 4) | getattr(os, "RTLD_NOW", 2))


def graph_node_kinds(graph: torch.cuda.CUDAGraph) -> dict:
    """Node kinds of a captured graph (``hipGraphGetNodes`` / ``hipGraphNodeGetType`` on the runtime torch already loaded); the graph must
    have been created with ``keep_graph=True``.  Empty dict when the node kinds cannot be determined (the caller decides what that means:
    ``GraphedTrainStep`` warns, because its no-memset-node check is then skipped)."""
    import ctypes as C
    try:
        hip = _loaded_hip_runtime()
        raw = C.c_void_p(graph.raw_cuda_graph())
        n = C.c_size_t(0)
        if hip.hipGraphGetNodes(raw, None, C.byref(n)) != 0:
            return {}
        nodes = (C.c_void_p * max(n.value, 1))()
        if hip.hipGraphGetNodes(raw, nodes, C.byref(n)) != 0:
            return {}
        kinds: dict = {}
        for i in range(n.value):
            t = C.c_int(-1)
            if hip.hipGraphNodeGetType(C.c_void_p(nodes[i]), C.byref(t)) != 0:
                return {}
            k = _NODE_KINDS.get(t.value, f"type{t.value}")
            kinds[k] = kinds.get(k, 0) + 1
        return kinds
    except (OSError, AttributeError, RuntimeError):
        return {}


class GraphedTrainStep:
    """``step = GraphedTrainStep(model, loss_fn, optimizer, example_inputs, example_targets)``; then ``loss = step(inputs, targets)``.

    ``loss_fn(outputs, *targets)`` receives what ``model(*inputs)`` returns.  The returned loss is a static tensor that the next replay
    overwrites (``.item()`` / ``.clone()`` it to keep a value)."""

    def __init__(self, model: torch.nn.Module, loss_fn: Callable, optimizer: Adam, example_inputs: Sequence[torch.Tensor],
                 example_targets: Sequence[torch.Tensor], warmup: int = 2, after_backward: Callable[[], None] | None = None):
        if not isinstance(optimizer, Adam):
            raise TypeError("GraphedTrainStep needs bio_image_unet_amd.optim.Adam (its step reads lr / bias correction from device memory)")
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            raise NotImplementedError("GraphedTrainStep is single-process: the gradient all-reduce is not captured")
        self.model, self.loss_fn, self.opt, self.after_backward = model, loss_fn, optimizer, after_backward
        self.static_in = [t.detach().clone() for t in example_inputs]
        self.static_tgt = [t.detach().clone() for t in example_targets]
        self.params = [p for g in optimizer.param_groups for p in g["params"] if p.requires_grad]

        # the warm-up steps (first-use allocations of the library on the capture stream, autograd's buffers) must not train the model:
        # snapshot parameters / BatchNorm buffers / moments, restore them IN PLACE afterwards (the capture holds their addresses)
        model_state = copy.deepcopy(model.state_dict())
        moments = {p: (st["exp_avg"].clone(), st["exp_avg_sq"].clone(), int(st["step"])) for p, st in optimizer.state.items() if "exp_avg" in st}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 1)):
                self._body()
                optimizer.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        optimizer.prepare_for_capture()
        self.graph = torch.cuda.CUDAGraph(keep_graph=True)
        with torch.cuda.graph(self.graph, stream=side):
            self.loss = self._body()
            optimizer.step()                         # -> Adam._captured_step
        optimizer.finish_capture()
        # what the capture holds: memset nodes are refused (this runtime replays them with a corrupted fill pattern: module docstring); copy
        # nodes -- the select-backward of a loss that indexes the logits as unet/train.py:133-134 does, the stacked heads of
        # MultiOutputUnet3D -- replay correctly and are only reported in node_kinds
        self.node_kinds = graph_node_kinds(self.graph)
        if not self.node_kinds:
            import warnings
            warnings.warn("GraphedTrainStep: the node kinds of the captured step could not be read (hipGraphGetNodes on the loaded runtime): "
                          "the check that it holds no memset nodes was SKIPPED", stacklevel=2)
        if self.node_kinds.get("memset", 0):
            raise RuntimeError(f"GraphedTrainStep: the captured step holds memset nodes {self.node_kinds}: from its second launch on this "
                               "runtime replays a memset node with a corrupted fill pattern (zero tensors with a kernel: tensor.zero_() / "
                               "torch.zeros, not hipMemsetAsync)")
        self.graph.instantiate()
        # the captured launches write the buffers of exactly these engines (activations, statistics, workspace): hold them here --
        # not only through self.loss.grad_fn -- so that an engine-cache eviction or set_compute_dtype cannot free what a replay writes
        self.engines = [e for lst in getattr(model, "_engines", {}).values() for e in lst]
        # nothing of the capture ran; undo the warm-up
        model.load_state_dict(model_state)
        with torch.no_grad():
            for p, st in optimizer.state.items():
                if "exp_avg" not in st:
                    continue
                if p in moments:
                    st["exp_avg"].copy_(moments[p][0]); st["exp_avg_sq"].copy_(moments[p][1]); st["step"] = moments[p][2]
                else:
                    st["exp_avg"].zero_(); st["exp_avg_sq"].zero_(); st["step"] = 0
        torch.autograd.graph.increment_version(self.params)

    def _body(self):
        outs = self.model(*self.static_in)
        loss = self.loss_fn(outs, *self.static_tgt)
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        if self.after_backward is not None:
            self.after_backward()
        return loss

    def __call__(self, inputs: Sequence[torch.Tensor], targets: Sequence[torch.Tensor]) -> torch.Tensor:
        for e in self.engines:
            # a replay overwrites the engine's saved activations exactly like a forward does: an eager forward still waiting for its
            # backward on the same engine must fail loudly there (generation check of _NetFn.backward), not be corrupted silently.
            # (The capture's own autograd node -- alive through self.loss -- does not count: its backward is part of the replay.)
            if e.busy():
                raise RuntimeError("GraphedTrainStep: an eager forward on this model still awaits its backward; a replay would overwrite "
                                   "its saved activations (run that backward, or drop its outputs, first)")
            e.generation += 1
        for dst, src in zip(self.static_in, inputs):
            dst.copy_(src, non_blocking=True)
        for dst, src in zip(self.static_tgt, targets):
            dst.copy_(src, non_blocking=True)
        self.opt.refresh_hyper()
        self.graph.replay()
        torch.autograd.graph.increment_version(self.params)        # the replay updated the parameters behind autograd's back
        return self.loss
