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

The graph must hold KERNEL nodes only.  A ``hipMemsetAsync`` captured in it was seen to lose its order against the kernels around it as soon
as eager work ran between two replays (weight gradients accumulated onto stale workspace contents); the library therefore zero-fills with a
kernel of its own, Adam's pointer table is uploaded after the capture, and networks whose step contains ``Tensor.copy_`` between device
tensors (the stacked heads of ``MultiOutputUnet3D``) are refused.
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
        if len(getattr(model, "output_heads", None) or {}) > 1:
            # the stacked-head backward copies weights / d logits with Tensor.copy_: device-to-device copy NODES in the graph (12 per step
            # of the bench's cfg5, rocprofv3 kernel trace) -- the kind of node that lost its order (module docstring); single-head steps
            # are kernel nodes only (cfg4: 176 kernels, 0 copy / fill nodes per replay)
            raise NotImplementedError("GraphedTrainStep: multi-head networks are not captured (their step holds device-to-device copy nodes)")
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
        # what the capture holds: memset nodes are refused (see the module docstring), copy nodes -- e.g. the select-backward of a loss that
        # indexes the logits, as unet/train.py:133-134 does -- are reported: they replayed in order in every test run so far
        self.node_kinds = graph_node_kinds(self.graph)
        if not self.node_kinds:
            import warnings
            warnings.warn("GraphedTrainStep: the node kinds of the captured step could not be read (hipGraphGetNodes on the loaded runtime): "
                          "the check that it holds no memset nodes was SKIPPED", stacklevel=2)
        if self.node_kinds.get("memset", 0):
            raise RuntimeError(f"GraphedTrainStep: the captured step holds memset nodes {self.node_kinds}: they are not replayed in order "
                               "on this runtime (zero tensors with a kernel: tensor.zero_() / torch.zeros, not hipMemsetAsync)")
        if self.node_kinds.get("memcpy", 0):
            import warnings
            warnings.warn(f"GraphedTrainStep: the captured step holds device copy nodes {self.node_kinds}; a step of kernel nodes only is the "
                          "verified configuration", stacklevel=2)
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
