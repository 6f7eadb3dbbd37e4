"""Fused multi-tensor Adam on the HIP side (``biu_adam_step``): one launch updates every parameter; gradient-norm clipping over the same
tables (``biu_grad_clip``).

Replaces ``torch.optim.Adam(model.parameters(), lr=lr)`` of the reference trainers (``unet/train.py:102``; defaults
betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False).  ``state_dict`` / ``param_groups`` keep the torch
layout so ``ReduceLROnPlateau`` (``unet/train.py:103``) can drive ``param_groups[0]['lr']`` unchanged.
"""
from __future__ import annotations

import ctypes as C

import torch

from ._lib import check, lib


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._tables = {}

    def load_state_dict(self, state_dict):
        """Moments are replaced by new tensors: drop the device pointer tables built over the old ones.  Checkpoints of
        ``torch.optim.Adam`` carry ``step`` as a tensor; the kernel takes a C int."""
        super().load_state_dict(state_dict)
        self._tables = {}
        for st in self.state.values():
            if "step" in st:
                st["step"] = int(st["step"])
            for k in ("exp_avg", "exp_avg_sq"):
                if k in st:
                    st[k] = st[k].to(dtype=torch.float32).contiguous()

    def _group_tables(self, gi, group):
        ps = [p for p in group["params"] if p.requires_grad]
        for p in ps:
            st = self.state[p]
            if "exp_avg" not in st:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p, dtype=torch.float32)
                st["exp_avg_sq"] = torch.zeros_like(p, dtype=torch.float32)
        # the kernel writes through raw pointers: the tables are valid only for exactly these parameter AND moment tensors
        key = (gi, tuple(p.data_ptr() for p in ps), tuple(self.state[p]["exp_avg"].data_ptr() for p in ps),
               tuple(self.state[p]["exp_avg_sq"].data_ptr() for p in ps))
        t = self._tables.get(gi)
        if t is None or t["key"] != key:
            dev = ps[0].device
            mk = lambda ts: torch.tensor([x.data_ptr() for x in ts], dtype=torch.int64, device=dev)
            t = {"key": key, "ps": ps, "p": mk(ps), "m": mk([self.state[p]["exp_avg"] for p in ps]),
                 "v": mk([self.state[p]["exp_avg_sq"] for p in ps]),
                 "n": torch.tensor([p.numel() for p in ps], dtype=torch.int64, device=dev), "g": None, "gkey": None}
            self._tables[gi] = t
        return t

    # ---- a step inside a captured hipGraph (bio_image_unet_amd/graph.py) -------------------------------------------------------
    def prepare_for_capture(self):
        """Allocates what a captured step needs BEFORE the capture starts (pinned and device memory cannot be allocated inside one):
        per group the scalars {lr, beta1, beta2, eps, grad_scale, step} in device memory and a pinned / device pair for the table of
        gradient pointers."""
        for gi, group in enumerate(self.param_groups):
            t = self._group_tables(gi, group)
            if "hyper" not in t:
                n, dev = len(t["ps"]), t["ps"][0].device
                t["hyper"] = torch.zeros(6, dtype=torch.float32, device=dev)
                t["gcap_host"] = torch.empty(n, dtype=torch.int64).pin_memory()
                t["gcap_dev"] = torch.empty(n, dtype=torch.int64, device=dev)
                t["clip"] = torch.empty(int(lib.biu_grad_clip_scratch_floats(n)) + 1, dtype=torch.float32, device=dev)

    def refresh_hyper(self, grad_scale: float = 1.0):
        """In front of every replay of a captured step: advance the step count and hand the step's scalars (current learning rate
        included) to the captured Adam launch.  One tiny kernel, arguments by value -- no host buffer the GPU could read late."""
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for gi, group in enumerate(self.param_groups):
            t = self._group_tables(gi, group)
            step = int(self.state[t["ps"][0]]["step"]) + 1
            for p in t["ps"]:
                self.state[p]["step"] = step
            b1, b2 = group["betas"]
            check(lib.biu_adam_set_hyper(C.c_void_p(t["hyper"].data_ptr()), float(group["lr"]), b1, b2, group["eps"], step, float(grad_scale), st),
                  "adam_set_hyper")

    def _captured_step(self, st):
        for gi, group in enumerate(self.param_groups):
            t = self._group_tables(gi, group)
            if "hyper" not in t:
                raise RuntimeError("Adam.step() inside a graph capture needs prepare_for_capture() before the capture starts")
            ps = t["ps"]
            for p in ps:
                assert p.grad is not None and p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()
            # the gradients of a captured backward live in the graph's private pool at fixed addresses; the captured launch only refers to
            # the table, finish_capture() fills it once the capture has ended (no copy node in the graph: a captured hipMemsetAsync was
            # seen to lose its order against the kernels around it, and a copy node is the same kind of thing)
            t["gcap_host"].copy_(torch.tensor([p.grad.data_ptr() for p in ps], dtype=torch.int64))
            check(lib.biu_adam_step_hyper(len(ps), C.c_void_p(t["p"].data_ptr()), C.c_void_p(t["gcap_dev"].data_ptr()),
                                          C.c_void_p(t["m"].data_ptr()), C.c_void_p(t["v"].data_ptr()), C.c_void_p(t["n"].data_ptr()),
                                          C.c_void_p(t["hyper"].data_ptr()), st), "adam_step_hyper")
            torch.autograd.graph.increment_version(ps)

    def finish_capture(self):
        """After the capture has ended: upload the gradient-pointer tables the captured Adam launches read."""
        for t in self._tables.values():
            if "gcap_host" in t:
                t["gcap_dev"].copy_(t["gcap_host"])
        torch.cuda.synchronize()

    def _upload_grad_table(self, t):
        """Device table of the gradient pointers of ``t['ps']`` in ``t['g']`` (shared by ``clip_grad_norm_`` and ``step``)."""
        ps = t["ps"]
        gkey = tuple(p.grad.data_ptr() for p in ps)
        if t["gkey"] != gkey:
            # The gradient tensors are new allocations every step, so this table changes every step: upload it from
            # pinned memory without blocking (a pageable torch.tensor(..., device=) copy waits for the whole stream and
            # would keep the host from ever running ahead of the GPU).  Four rotating buffers, each guarded by the event
            # of its previous upload.
            if "g_host" not in t:
                n = len(ps)
                t["g_host"] = [torch.empty(n, dtype=torch.int64).pin_memory() for _ in range(4)]
                t["g_dev"] = [torch.empty(n, dtype=torch.int64, device=ps[0].device) for _ in range(4)]
                t["g_ev"] = [None] * 4
                t["rot"] = -1
            r = t["rot"] = (t["rot"] + 1) % 4
            if t["g_ev"][r] is not None:
                t["g_ev"][r].synchronize()
            t["g_host"][r].copy_(torch.tensor(gkey, dtype=torch.int64))
            t["g_dev"][r].copy_(t["g_host"][r], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            t["g_ev"][r] = ev
            t["g"] = t["g_dev"][r]
            t["gkey"] = gkey

    @torch.no_grad()
    def clip_grad_norm_(self, max_norm: float = 1.0) -> torch.Tensor:
        """``torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)`` of ``multi_output_unet3d/train.py:201`` over this optimizer's
        parameters, as three launches whatever their number (``biu_grad_clip``): returns the total norm before clipping (0-dim device
        tensor) and scales every ``p.grad`` in place by ``min(1, max_norm / (total + 1e-6))``.  Usable inside a captured step (after
        ``prepare_for_capture``).  More than one parameter group: falls back to the torch function (a joint norm over groups)."""
        if len(self.param_groups) != 1:
            return torch.nn.utils.clip_grad_norm_([p for g in self.param_groups for p in g["params"] if p.grad is not None], max_norm)
        t = self._group_tables(0, self.param_groups[0])
        ps = t["ps"]
        for p in ps:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            assert p.is_cuda and p.grad.dtype == torch.float32 and p.grad.is_contiguous()
        n = len(ps)
        need = int(lib.biu_grad_clip_scratch_floats(n))
        if "clip" not in t or t["clip"].numel() < need + 1:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("Adam.clip_grad_norm_() inside a graph capture needs one eager call (or prepare_for_capture) first")
            t["clip"] = torch.empty(need + 1, dtype=torch.float32, device=ps[0].device)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if torch.cuda.is_current_stream_capturing():
            if "hyper" not in t:
                raise RuntimeError("Adam.clip_grad_norm_() inside a graph capture needs prepare_for_capture() before the capture starts")
            t["gcap_host"].copy_(torch.tensor([p.grad.data_ptr() for p in ps], dtype=torch.int64))     # (finish_capture uploads it)
            table = t["gcap_dev"]
        else:
            self._upload_grad_table(t)
            table = t["g"]
        total = t["clip"][need:need + 1]
        check(lib.biu_grad_clip(n, C.c_void_p(table.data_ptr()), C.c_void_p(t["n"].data_ptr()), float(max_norm), C.c_void_p(t["clip"].data_ptr()),
                                need, C.c_void_p(total.data_ptr()), st), "grad_clip")
        return total[0]

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        loss = closure() if closure is not None else None
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if torch.cuda.is_current_stream_capturing():
            self._captured_step(st)
            return loss
        for gi, group in enumerate(self.param_groups):
            t = self._group_tables(gi, group)
            ps = t["ps"]
            for p in ps:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                assert p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()
            self._upload_grad_table(t)
            step = int(self.state[ps[0]]["step"]) + 1
            for p in ps:
                self.state[p]["step"] = step
            b1, b2 = group["betas"]
            check(lib.biu_adam_step(len(ps), C.c_void_p(t["p"].data_ptr()), C.c_void_p(t["g"].data_ptr()),
                                    C.c_void_p(t["m"].data_ptr()), C.c_void_p(t["v"].data_ptr()),
                                    C.c_void_p(t["n"].data_ptr()), float(group["lr"]), b1, b2, group["eps"], step,
                                    float(grad_scale), st), "adam_step")
            # the kernel wrote through raw pointers: bump the version counters so autograd and the engine's
            # packed-weight cache (keyed on Tensor._version) see the in-place update
            torch.autograd.graph.increment_version(ps)
        return loss
