"""Data feed for the hot path: a memory-mapped uint8 tile store and an asynchronous host-to-device feeder.

The reference keeps every training tile as its own uint8 TIFF and ``DataProcess.__getitem__`` re-reads, decodes and divides it by
255 for every access (``unet/data.py:253-266``), with ``num_workers=0`` (``unet/train.py:92-93``): fine for a CPU that needs a
third of a second per step, a starvation hazard for eight GPUs that take ~10 ms.  Here the tiles of a data set live in ONE
flat uint8 file per field, mapped into memory:

* ``TileStore`` -- ``[N, *shape]`` uint8 arrays (``image``, ``mask``, ``prev_image``, ``volume`` ... -- the reference's item keys).  It
  is a ``torch.utils.data.Dataset`` with the reference's item contract (``float32`` in [0, 1]), so every Trainer takes it as is;
  ``TileStore.from_dataset`` converts any data set that yields such items (values are multiples of 1/255 there, so the round
  trip is exact).
* ``DeviceFeeder`` -- what the Trainers iterate when handed a ``TileStore``: a background thread gathers the next batches out of
  the page cache into pinned buffers, a side HIP stream copies them to the device as BYTES (a quarter of the float32 traffic over
  PCIe) while the current step computes, and the batch reaches the network as uint8 -- the 1/255 scaling rides in the input-layout
  kernel (``biu_from_nchw_u8``), targets are widened by ``biu_u8_to_f32``.

TIFF decoding, augmentation and tiling stay out of scope (``DataProcess``): they run once, offline, and their output is what this
store holds.
"""
from __future__ import annotations

import json
import os
import queue
import threading
from typing import Dict, Iterable, Optional, Sequence

import numpy as np
import torch

_MAGIC = "biu-tilestore-1"


class TileStore(torch.utils.data.Dataset):
    """Flat uint8 files ``<path>.<field>.u8`` + ``<path>.json``; fields are ``[N, *shape]``."""

    def __init__(self, path: str, mode: str = "r"):
        with open(path + ".json") as f:
            hdr = json.load(f)
        if hdr.get("magic") != _MAGIC:
            raise ValueError(f"{path}.json is not a tile store header")
        self.path, self.n, self.fields = path, int(hdr["n"]), {k: tuple(v) for k, v in hdr["fields"].items()}
        self.attrs = hdr.get("attrs", {})
        self.dim_out = tuple(self.attrs["dim_out"]) if "dim_out" in self.attrs else next(iter(self.fields.values()))
        for k, v in self.attrs.items():                      # aug_factor, clip_threshold, ...: the Trainers record them in checkpoints
            if k != "dim_out" and not hasattr(self, k):
                setattr(self, k, v)
        self.maps = {k: np.memmap(f"{path}.{k}.u8", dtype=np.uint8, mode=mode, shape=(self.n,) + shp) for k, shp in self.fields.items()}

    # ---- construction ----------------------------------------------------------------------------------------------
    @classmethod
    def create(cls, path: str, n: int, fields: Dict[str, Sequence[int]], attrs: Optional[dict] = None) -> "TileStore":
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        for k, shp in fields.items():
            np.memmap(f"{path}.{k}.u8", dtype=np.uint8, mode="w+", shape=(n,) + tuple(shp)).flush()
        with open(path + ".json", "w") as f:
            json.dump({"magic": _MAGIC, "n": n, "fields": {k: list(v) for k, v in fields.items()}, "attrs": attrs or {}}, f)
        return cls(path, mode="r+")

    @classmethod
    def from_dataset(cls, path: str, dataset: Iterable, keys: Optional[Sequence[str]] = None) -> "TileStore":
        """Convert a data set with the reference's item contract (dict of float32 tensors in [0, 1]) into a store."""
        first = dataset[0]
        keys = list(keys) if keys is not None else [k for k, v in first.items() if torch.is_tensor(v) or isinstance(v, np.ndarray)]
        fields = {k: tuple(np.asarray(first[k]).shape) for k in keys}
        attrs = {}
        for a in ("dim_out", "aug_factor", "clip_threshold", "noise_lims", "noise_amp", "brightness_contrast", "shiftscalerotate"):
            if hasattr(dataset, a):
                v = getattr(dataset, a)
                attrs[a] = list(v) if isinstance(v, (tuple, list)) else v
        st = cls.create(path, len(dataset), fields, attrs)
        for i in range(len(dataset)):
            item = dataset[i]
            for k in keys:
                st.maps[k][i] = np.clip(np.rint(np.asarray(item[k], dtype=np.float64) * 255.0), 0, 255).astype(np.uint8)
        st.flush()
        return st

    def flush(self):
        for m in self.maps.values():
            m.flush()

    # ---- Dataset contract of the reference: float32 in [0, 1] ---------------------------------------------------------
    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return {k: torch.from_numpy(np.asarray(m[i], dtype=np.float32) / 255.0) for k, m in self.maps.items()}

    def batch_u8(self, indices, out: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
        """uint8 batch ``{field: [B, *shape]}`` gathered from the map (into ``out``'s tensors when given, e.g. pinned buffers)."""
        idx = np.asarray(indices, dtype=np.int64)
        res = {}
        for k, m in self.maps.items():
            if out is not None:
                dst = out[k][:len(idx)]
                np.take(m, idx, axis=0, out=dst.numpy())
                res[k] = dst
            else:
                res[k] = torch.from_numpy(np.take(m, idx, axis=0))
        return res


class DeviceFeeder:
    """Iterable over uint8 device batches of a ``TileStore`` (one epoch per ``iter()``), ``depth`` batches in flight."""

    def __init__(self, store: TileStore, indices: Sequence[int], batch_size: int, device, drop_last: bool = True, depth: int = 3):
        self.store, self.indices, self.batch_size = store, list(indices), batch_size
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError(f"DeviceFeeder copies to a GPU; got device '{device}'")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.drop_last, self.depth = drop_last, max(2, depth)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        mk = lambda shp, pin: (torch.empty((batch_size,) + shp, dtype=torch.uint8).pin_memory() if pin
                               else torch.empty((batch_size,) + shp, dtype=torch.uint8, device=self.device))
        self.slots = [{"host": {k: mk(s, True) for k, s in store.fields.items()}, "dev": {k: mk(s, False) for k, s in store.fields.items()},
                       "ready": None, "free": None, "released": threading.Event()} for _ in range(self.depth)]

    def __len__(self):
        n = len(self.indices)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        batches = [self.indices[i:i + self.batch_size] for i in range(0, len(self.indices), self.batch_size)]
        if self.drop_last:
            batches = [b for b in batches if len(b) == self.batch_size]
        filled: "queue.Queue" = queue.Queue(maxsize=self.depth - 1)
        stop = threading.Event()
        for slot in self.slots:
            slot["released"].set()
            slot["free"] = None

        def producer():
            try:
                produce()
            except BaseException as e:                       # surface a failure of the feeder thread in the training thread
                filled.put(e)

        def produce():
            # gather (page cache -> pinned memory) and enqueue the asynchronous upload; never blocks the training thread
            torch.cuda.set_device(self.device)
            for bi, idx in enumerate(batches):
                slot = self.slots[bi % self.depth]
                slot["released"].wait()                     # the consumer has handed the slot back ...
                slot["released"].clear()
                if stop.is_set():
                    break
                if slot["free"] is not None:
                    slot["free"].synchronize()              # ... and the step that read its device buffers has finished on the GPU
                self.store.batch_u8(idx, out=slot["host"])
                with torch.cuda.stream(self.copy_stream):
                    for k in slot["host"]:
                        slot["dev"][k][:len(idx)].copy_(slot["host"][k][:len(idx)], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self.copy_stream)
                slot["ready"] = ev
                filled.put((bi, len(idx)))
            filled.put(None)

        th = threading.Thread(target=producer, daemon=True)
        th.start()
        try:
            while True:
                item = filled.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise RuntimeError("DeviceFeeder: the feeder thread failed") from item
                bi, nb = item
                slot = self.slots[bi % self.depth]
                torch.cuda.current_stream(self.device).wait_event(slot["ready"])     # device-side wait: the host does not block
                yield {k: v[:nb] for k, v in slot["dev"].items()}
                done = torch.cuda.Event()
                done.record(torch.cuda.current_stream(self.device))                  # everything enqueued so far read the slot
                slot["free"] = done
                slot["released"].set()
        finally:
            stop.set()
            for slot in self.slots:
                slot["released"].set()
            while th.is_alive():
                try:
                    filled.get(timeout=0.05)
                except queue.Empty:
                    pass
            th.join()


def u8_to_float(t: torch.Tensor, scale: float = 1.0 / 255.0) -> torch.Tensor:
    """uint8 device tensor -> float32 * scale through ``biu_u8_to_f32`` (targets of the fused losses)."""
    import ctypes as C
    from ._lib import check, lib
    t = t.contiguous()
    out = torch.empty(t.shape, dtype=torch.float32, device=t.device)
    check(lib.biu_u8_to_f32(C.c_void_p(t.data_ptr()), float(scale), C.c_void_p(out.data_ptr()), t.numel(),
                            C.c_void_p(torch.cuda.current_stream().cuda_stream)), "u8_to_f32")
    return out
