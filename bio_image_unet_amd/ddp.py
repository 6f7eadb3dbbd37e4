"""Data parallelism for the hot path: one process per GPU, gradients averaged with one RCCL all-reduce per step.

The reference has no distributed code at all; its BatchNorm layers are plain per-process BatchNorm,
so per-rank batch statistics ARE the reference semantics.  Parameters are broadcast once from rank 0; each step the
fp32 gradients (19 MB for UNet3D F=32 ... 124 MB for Unet F=64) are all-reduced in decoder -> encoder buckets that go
out while the backward pass is still running (``backend='nccl'`` is RCCL over xGMI on ROCm; 'gloo' on CPU for the
tests) and averaged; BN running statistics are left rank-local (rank 0's are the ones a checkpoint saves).
"""
from __future__ import annotations

import os
from typing import Iterable, List

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None):
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun contract); returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


class _Bucket:
    __slots__ = ("flat", "params", "views", "pending", "work", "late")

    def __init__(self, params, device):
        # every slot starts on a 16-byte boundary: the gradient kernels write straight into the slots (GradAverager._alloc)
        n = sum((p.numel() + 3) // 4 * 4 for p in params)
        self.flat = torch.zeros(n, dtype=torch.float32, device=device)
        self.params, self.views, o = list(params), [], 0
        for p in params:
            self.views.append(self.flat[o:o + p.numel()].view_as(p))
            o += (p.numel() + 3) // 4 * 4
        self.pending, self.work, self.late = set(), None, False


class GradAverager:
    """Bucketed gradient all-reduce (mean), overlapped with the backward pass, + initial parameter/buffer broadcast.

    Parameters are grouped into buckets of ``bucket_mb`` MiB in the order their gradients become final during backward -- the
    REVERSE of registration order, i.e. decoder first, encoder last.  Networks of this package report each finished
    parameter gradient from inside ``Engine.backward`` (``register_grad_ready_hook``); the engine has WRITTEN it into its bucket slot
    already (``register_grad_alloc``: the bucket view is the gradient tensor the kernels fill -- no per-parameter copy) and, when the
    bucket is complete, ``all_reduce(async_op=True)`` is issued at once -- RCCL's stream then runs beside the remaining backward
    kernels (xGMI is point-to-point: a 19 MB UNet3D gradient is ~0.2 ms of ring time against a >= 10 ms step, so three or four
    buckets hide all of it behind the encoder's backward).  ``average()`` after backward launches whatever is still incomplete
    (parameters without a gradient count as zeros), waits, scales by 1/world and leaves ``p.grad`` = views of the flat buckets.
    Any other ``nn.Module`` (no hook) is bucketed the same way, launched back to back from ``average()``.

    Contract: between ``backward()`` and ``average()`` nobody may modify ``p.grad`` -- a bucket that went out during backward
    is not re-read, so e.g. ``clip_grad_norm_`` belongs AFTER ``average()`` (it then clips the global-batch gradient, which is what
    torch DDP users get too).  Gradient ACCUMULATION is supported: when a hook fires for a parameter whose ``p.grad`` already
    holds something (second ``backward()`` before ``average()``, or ``zero_grad(set_to_none=False)`` / last step's bucket views
    still installed), autograd is about to add into ``p.grad`` in place, so that bucket is not launched early (an in-flight
    all-reduce of it is waited for: the bucket -- which IS ``p.grad``'s memory -- then holds the mean of the first gradients, the new local
    gradient is added on top, and the second all-reduce of ``average()`` turns the sum into mean + mean)."""

    def __init__(self, module: torch.nn.Module, bucket_mb: float = 8.0):
        self.params: List[torch.nn.Parameter] = [p for p in module.parameters() if p.requires_grad]
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        dev = self.params[0].device
        cap = max(int(bucket_mb * (1 << 20) / 4), 1)
        self.buckets: List[_Bucket] = []
        cur, cur_n = [], 0
        for p in reversed(self.params):                  # decoder -> encoder: the order gradients land in backward
            if cur and cur_n + p.numel() > cap:
                self.buckets.append(_Bucket(cur, dev))
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            self.buckets.append(_Bucket(cur, dev))
        self._where = {p: (b, i) for b in self.buckets for i, p in enumerate(b.params)}
        self._reset()
        if self.world > 1:
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    # through a detached alias, NOT .data: the alias shares the version counter, so the in-place receive
                    # bumps Tensor._version and the engines' cached MFMA weight packings are rebuilt on ranks != 0
                    dist.broadcast(t.detach(), src=0)
            if hasattr(module, "invalidate_packed"):
                module.invalidate_packed()
        self.hooked = hasattr(module, "register_grad_ready_hook")
        if self.hooked and self.world > 1:
            module.register_grad_ready_hook(self._on_grad_ready)
            if hasattr(module, "register_grad_alloc"):
                # the engine writes each parameter gradient STRAIGHT INTO its bucket (no per-parameter device copy: ~70 launches per
                # UNet3D step otherwise): Engine.new_grad asks here for the destination
                module.register_grad_alloc(self._alloc)
        # RCCL averages in the collective itself (ReduceOp.AVG); gloo (the CPU tests) sums and average() scales
        self._avg_op = self.world > 1 and dist.get_backend() == "nccl"
        self.launched_in_backward = 0                    # buckets whose all-reduce went out before backward returned (last step)
        self.copies_in_backward = 0                      # gradients that still had to be copied into their bucket (last step)
        self.average_wait_ms = 0.0                       # host time the last average() spent launching the rest and waiting

    _copies = 0

    def _reset(self):
        for b in self.buckets:
            b.pending, b.work, b.late = set(range(len(b.params))), None, False

    def _launch(self, b: _Bucket):
        b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.AVG if self._avg_op else dist.ReduceOp.SUM, async_op=True)

    def _wait(self, b: _Bucket):
        """Completes the bucket's all-reduce: the bucket then holds the MEAN over the ranks (RCCL averages in the collective, gloo sums and is
        scaled here).  p.grad aliases the bucket, so a gradient accumulated on top of an already reduced bucket (second backward before
        average()) stays right: mean(g1) + g2_local is reduced again to mean(g1) + mean(g2)."""
        b.work.wait()
        b.work = None
        if not self._avg_op:
            b.flat.mul_(1.0 / self.world)

    def _alloc(self, p: torch.nn.Parameter):
        """Destination of the gradient the engine is about to compute for ``p``: a FRESH view of its bucket slot (a fresh tensor object, so
        that autograd's AccumulateGrad takes it as ``p.grad`` instead of cloning it), or None when the gradient cannot live there --
        ``p.grad`` already holds something (accumulation: autograd will add in place, possibly into this very memory) or the bucket is
        no longer collecting."""
        hit = self._where.get(p)
        if hit is None or p.grad is not None:
            return None
        b, i = hit
        if b.late or i not in b.pending or b.work is not None:
            return None
        v = b.views[i]
        return b.flat[v.storage_offset():v.storage_offset() + v.numel()].view_as(p)

    @torch.no_grad()
    def _on_grad_ready(self, p: torch.nn.Parameter, g: torch.Tensor):
        """Called by the engine, on the compute stream, the moment the gradient of ``p`` is final."""
        hit = self._where.get(p)
        if hit is None:
            return
        b, i = hit
        if b.late or i not in b.pending or p.grad is not None:
            # the gradient is being accumulated: autograd adds ``g`` into the existing p.grad right after this hook, so the
            # sum exists only there.  Nothing of this bucket may be on the wire while that happens (p.grad can BE a view of
            # b.flat): wait out an early launch, and let average() read the whole bucket from p.grad.
            b.late = True
            if b.work is not None:
                self._wait(b)
            return
        if g.data_ptr() != b.views[i].data_ptr():        # (a gradient the engine could not write in place: shared weights, the zero conv biases)
            b.views[i].copy_(g)
            self._copies += 1
        b.pending.discard(i)
        if not b.pending and b.work is None:
            self._launch(b)
            self.launched_in_backward += 1

    @torch.no_grad()
    def average(self):
        """Call after backward(): leaves p.grad = mean over ranks (views of the flat fp32 buckets)."""
        if self.world == 1:
            return
        import time
        t0 = time.perf_counter()
        early = sum(1 for b in self.buckets if b.work is not None)
        for b in self.buckets:
            if b.work is None:
                for i in (range(len(b.params)) if b.late else b.pending):    # not reported during backward (or accumulated): take p.grad (or zeros)
                    g = b.params[i].grad
                    if g is None:
                        b.views[i].zero_()
                    elif g.data_ptr() != b.views[i].data_ptr():      # (p.grad may still be last step's view of this bucket)
                        b.views[i].copy_(g)
                self._launch(b)
        for b in self.buckets:
            self._wait(b)
            for p, v in zip(b.params, b.views):
                if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                    p.grad = v
        self.launched_in_backward = early
        self.copies_in_backward, self._copies = self._copies, 0
        self.average_wait_ms = (time.perf_counter() - t0) * 1e3
        self._reset()
