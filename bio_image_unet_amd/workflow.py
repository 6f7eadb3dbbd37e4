"""Trainer / Predict counterparts: the host-side callers of the hot path.

They keep the reference's constructor signatures, attribute names, checkpoint dictionary and -- deliberately -- its
numerics-relevant quirks, so a training run is step-for-step comparable:

* 2-D loss indexes the *batch* axis with the channel index and validation only counts the last batch
  (``unet/train.py:133-134,151-153``); the mask ``.view`` uses ``dim[0]`` twice (``:128``).
* 3-D adds ``SmoothL1(y_logits[1:], y_logits[:-1]) * time_loss_weight`` across the *batch* axis; validation hard-codes 0.1
  (``unet3d/train.py:140-145,163-169``).
* No ``model.eval()`` during validation: BatchNorm keeps using (and updating) batch statistics under ``no_grad``.
* ``DataLoader(shuffle=False, drop_last=True, num_workers=0, pin_memory=True)``; unseeded ``random_split``.
* The saved ``'optimizer'`` entry is the construction-time state (2-D / 3-D), never refreshed.

What differs by design: the network class comes from this package (every layer a HIP kernel), the optimizer is the fused
``biu_adam_step``, and data sets are any ``torch.utils.data.Dataset`` yielding the reference's dict items (TIFF I/O and
augmentation are out of scope).
"""
from __future__ import annotations

import os
from typing import Optional, Union

import numpy as np
import torch
from torch import nn, optim
from torch.utils.data import DataLoader, random_split

from .losses import BCEDiceLoss, BCEDiceLossSiam, BCEDiceTemporalLoss, TverskyLoss, logcoshTverskyLoss, weightedBCELoss
from .models import AttentionUnet, MultiOutputUnet3D, Siam_UNet, UNet3D, Unet, Unet_v0
from .optim import Adam
from .utils import get_device, init_weights

try:                                    # progress bars are cosmetic
    from tqdm import tqdm
except Exception:                       # pragma: no cover
    def tqdm(it, **_):
        return it


def _pick_device(device):
    return get_device() if device == "auto" else torch.device(device)


def _make_criterion(name, params, extra=None):
    table = {"BCEDice": BCEDiceLoss, "Tversky": TverskyLoss, "logcoshTversky": logcoshTverskyLoss}
    if extra:
        table.update(extra)
    if name not in table:
        raise ValueError(f'Loss "{name}" not defined!')
    return table[name](params[0], params[1])


def _loaders(dataset, train_data, val_data, batch_size, device):
    """The reference's loaders (``DataLoader(shuffle=False, drop_last=True, num_workers=0, pin_memory=True)``, unet/train.py:92-93)
    -- or, for a memory-mapped ``feed.TileStore``, asynchronous uint8 feeders over the same index split (same order, same
    batches; the tiles cross PCIe as bytes while the previous step computes)."""
    from .feed import DeviceFeeder, TileStore
    if isinstance(dataset, TileStore) and torch.device(device).type == "cuda":
        return (DeviceFeeder(dataset, train_data.indices, batch_size, device), DeviceFeeder(dataset, val_data.indices, batch_size, device))
    return (DataLoader(train_data, batch_size=batch_size, pin_memory=True, drop_last=True),
            DataLoader(val_data, batch_size=batch_size, pin_memory=True, drop_last=True))


def _target(t: torch.Tensor) -> torch.Tensor:
    """Targets arrive as float32 in [0, 1] (reference items) or as uint8 0..255 from a DeviceFeeder."""
    if t.dtype == torch.uint8:
        from .feed import u8_to_float
        return u8_to_float(t)
    return t


class _gc_paused:
    """The batch loop of an epoch runs with Python's cyclic garbage collector paused (restored, and run once, at the end): the host
    enqueues a step in 3-4 ms and runs only a few steps ahead of the GPU, while a generation-2 collection -- it walks every object torch
    has imported -- stalls it for 60-140 ms, i.e. the GPU idles for the length of 10-20 short steps (measured on ``Unet(1,1,32)`` at 2 x 256^2:
    one step in twenty took 60-115 ms instead of 4.9).  A training step creates no reference cycles; reference counting frees what it allocates."""

    def __enter__(self):
        import gc
        self.was = gc.isenabled()
        gc.disable()
        return self

    def __exit__(self, *exc):
        import gc
        if self.was:
            gc.enable()
            gc.collect()
        return False


class _EpochLoop:
    """Shared skeleton: split, loaders, Adam + ReduceLROnPlateau, best-validation checkpointing."""
    item_key = "image"

    def _setup(self, dataset, num_epochs, batch_size, lr, val_split, save_dir, save_name, save_iter):
        self.data, self.num_epochs, self.batch_size, self.lr = dataset, num_epochs, batch_size, lr
        self.best_loss = torch.tensor(float("inf"))
        self.save_iter, self.save_dir, self.save_name = save_iter, save_dir, save_name
        n_val = int(len(dataset) * val_split)
        self.dim = dataset.dim_out
        train_data, val_data = random_split(dataset, [len(dataset) - n_val, n_val])
        self.train_loader, self.val_loader = _loaders(dataset, train_data, val_data, batch_size, self.device)
        self.optimizer = Adam(self.model.parameters(), lr=lr)
        self.scheduler = optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, mode="min", patience=4, factor=0.1)
        os.makedirs(save_dir, exist_ok=True)

    def _data_attr(self, *names):
        return {n: getattr(self.data, n, None) for n in names}

    # subclasses: _forward_loss(batch, validating) -> loss
    def _train_epoch(self, epoch):
        print("\nStarting training epoch %s ..." % epoch)
        with _gc_paused():
            for batch in tqdm(self.train_loader, total=len(self.train_loader), unit="batch"):
                loss = self._forward_loss(batch, validating=False)
                self.optimizer.zero_grad()
                loss.backward()
                self.optimizer.step()

    def _save(self, name):
        torch.save(self.state, self.save_dir + "/" + name)

    def _after_validation(self, epoch, val_loss):
        if val_loss < self.best_loss:
            print("\nValidation loss improved from %s to %s - saving model state"
                  % (round(self.best_loss.item(), 5), round(val_loss.item(), 5)))
            self.state["best_loss"] = self.best_loss = val_loss
            self._save(self.save_name)
        if self.save_iter:
            self._save(f"model_epoch_{epoch}.pt")


class Trainer2D(_EpochLoop):
    """``bio_image_unet.unet.Trainer`` counterpart (``unet/train.py:16-198``)."""

    def __init__(self, dataset, num_epochs, network=Unet, batch_size=4, lr=1e-3, in_channels=1, out_channels=1,
                 channel_weights=None, n_filter=64, dilation=1, val_split=0.2, save_dir="./", save_name="model.pt",
                 save_iter=False, load_weights=False, loss_function="BCEDice", loss_params=(0.5, 0.5),
                 device: Union[torch.device, str] = "auto", fp32_products: Optional[str] = None):
        """``fp32_products`` (not in the reference): ``"exact"`` | ``"bf16x3"`` | ``"bf16x6"`` -- how the fp32 tensors of this trainer are
        multiplied (``bio_image_unet_amd.set_fp32_products``; process-wide).  ``"bf16x3"`` plays the role ``torch.backends.cudnn.allow_tf32``
        plays for the reference; ``None`` leaves the process's mode alone (default: bf16x6, fp32-grade split products)."""
        if fp32_products is not None:
            from . import set_fp32_products
            set_fp32_products(fp32_products)
        self.device = _pick_device(device)
        self.network = network
        self.model = network(n_filter=n_filter, in_channels=in_channels, out_channels=out_channels, dilation=dilation).to(self.device)
        self.model.apply(init_weights)
        self.loss_function, self.loss_params = loss_function, loss_params
        self.n_filter, self.in_channels, self.out_channels = n_filter, in_channels, out_channels
        self.channel_weights = torch.ones(out_channels) if channel_weights is None else torch.tensor(channel_weights)
        self.criterion = _make_criterion(loss_function, loss_params)
        self._setup(dataset, num_epochs, batch_size, lr, val_split, save_dir, save_name, save_iter)
        self.params = {"optimizer": self.optimizer.state_dict(), "lr": lr, "loss_function": loss_function,
                       "loss_params": loss_params, "n_filter": n_filter, "dilation": dilation, "batch_size": batch_size,
                       "augmentation": getattr(dataset, "aug_factor", None), "in_channels": in_channels,
                       "out_channels": out_channels,
                       **self._data_attr("clip_threshold", "noise_lims", "brightness_contrast", "shiftscalerotate")}
        if load_weights:
            self.state = torch.load(save_dir + "/" + save_name)
            self.model.load_state_dict(self.state["state_dict"])

    def _forward_loss(self, batch, validating):
        d = self.dim
        x = batch["image"].view(self.batch_size, self.in_channels, d[0], d[1]).to(self.device)
        # the training branch reshapes the mask with dim[0] twice (square tiles assumed), the validation one does not
        y = _target(batch["mask"].view(self.batch_size, self.out_channels, d[0], d[1] if validating else d[0]).to(self.device))
        _, logits = self.model(x)
        cw = self.channel_weights
        # NOTE the reference indexes the BATCH axis with the channel index
        return sum(self.criterion(logits[ch], y[ch]) * cw[j] for j, ch in enumerate(range(self.out_channels))) / sum(cw)

    def _validate(self, epoch):
        print("\nStarting validation epoch %s ..." % epoch)
        losses = []
        loss = None
        with torch.no_grad():
            for batch in tqdm(self.val_loader, total=len(self.val_loader), unit="batch"):
                loss = self._forward_loss(batch, validating=True)
        losses.append(loss.detach())            # reference: appended once, after the loop -> last batch only
        return torch.stack(losses).mean()

    def start(self, test_data_path=None, result_path=None, test_resize_dim=(512, 512)):
        for epoch in range(self.num_epochs):
            self._train_epoch(epoch)
            self.state = {"epoch": epoch, "best_loss": self.best_loss, "state_dict": self.model.state_dict()}
            self.state.update(self.params)
            with torch.no_grad():
                val_loss = self._validate(epoch)
                self.scheduler.step(val_loss)
            self._after_validation(epoch, val_loss)
            if test_data_path is not None:
                raise NotImplementedError("per-epoch prediction of TIFF test folders is outside the hot path; call "
                                          "Predict on arrays instead")


class Trainer3D(_EpochLoop):
    """``bio_image_unet.unet3d.Trainer`` counterpart (``unet3d/train.py:18-217``)."""

    def __init__(self, dataset, num_epochs, network=UNet3D, use_interpolation=False, batch_size=4, lr=1e-3,
                 in_channels=1, out_channels=1, channel_weights=None, n_filter=64, dilation=1, val_split=0.2,
                 save_dir="./", save_name="model.pt", save_iter=False, load_weights=False, loss_function="BCEDice",
                 loss_params=(0.5, 0.5), time_loss_weight=0.1, device: Union[torch.device, str] = "auto"):
        self.device = _pick_device(device)
        self.network = network
        self.model = network(n_filter=n_filter, in_channels=in_channels, out_channels=out_channels,
                             use_interpolation=use_interpolation).to(self.device)
        self.model.apply(init_weights)          # a no-op on Conv3d layers, as in the reference
        self.loss_function, self.loss_params, self.time_loss_weight = loss_function, loss_params, time_loss_weight
        self.n_filter, self.in_channels, self.out_channels = n_filter, in_channels, out_channels
        self.use_interpolation = use_interpolation
        self.channel_weights = torch.ones(out_channels) if channel_weights is None else torch.tensor(channel_weights)
        self.criterion = _make_criterion(loss_function, loss_params)
        self.criterion_time = nn.SmoothL1Loss()
        self._setup(dataset, num_epochs, batch_size, lr, val_split, save_dir, save_name, save_iter)
        self.params = {"optimizer": self.optimizer.state_dict(), "lr": lr, "loss_function": loss_function,
                       "loss_params": loss_params, "time_loss_weight": time_loss_weight, "n_filter": n_filter,
                       "use_interpolation": use_interpolation, "dilation": dilation, "batch_size": batch_size,
                       "augmentation": getattr(dataset, "aug_factor", None), "in_channels": in_channels,
                       "out_channels": out_channels,
                       **self._data_attr("clip_threshold", "noise_amp", "brightness_contrast", "shiftscalerotate")}
        if load_weights:
            self.state = torch.load(save_dir + "/" + save_name)
            self.model.load_state_dict(self.state["state_dict"])

    def _forward_loss(self, batch, validating):
        d = self.dim
        x = batch["volume"].view(self.batch_size, self.in_channels, d[0], d[1], d[2]).to(self.device)
        y = _target(batch["mask"].view(self.batch_size, self.out_channels, d[0], d[1], d[2]).to(self.device))
        _, logits = self.model(x)
        w = 0.1 if validating else self.time_loss_weight         # validation hard-codes 0.1
        # criterion(y_logits, y_i) + SmoothL1(y_logits[1:], y_logits[:-1]) * w (unet3d/train.py:140-145), one fused pass each way
        return self.criterion(logits, y, time_weight=w)

    def _validate(self, epoch):
        print("\nStarting validation epoch %s ..." % epoch)
        losses = []
        with torch.no_grad():
            for batch in tqdm(self.val_loader, total=len(self.val_loader), unit="batch"):
                losses.append(self._forward_loss(batch, validating=True).detach())
        return torch.stack(losses).mean()

    def start(self, test_data_path=None, result_path=None, test_resize_dim=(512, 512)):
        for epoch in range(self.num_epochs):
            self._train_epoch(epoch)
            with torch.no_grad():
                val_loss = self._validate(epoch)
                self.state = {"val_loss": val_loss, "epoch": epoch, "best_loss": self.best_loss,
                              "state_dict": self.model.state_dict()}
                self.state.update(self.params)
                self.scheduler.step(val_loss)
            self._after_validation(epoch, val_loss)
            if test_data_path is not None:
                raise NotImplementedError("per-epoch prediction of TIFF test folders is outside the hot path")


class TrainerSiam(_EpochLoop):
    """``bio_image_unet.siam_unet.Trainer`` counterpart (``siam_unet/train.py:16-172``); the model class is fixed."""

    def __init__(self, dataset, num_epochs, batch_size=4, lr=1e-3, n_filter=32, mode="max", val_split=0.2,
                 save_dir="./", save_name="model.pt", save_iter=False, loss_function="BCEDice", loss_params=(1, 1),
                 load_weights=None, device: Union[torch.device, str] = "auto", fp32_products: Optional[str] = None):
        if fp32_products is not None:                 # see Trainer2D
            from . import set_fp32_products
            set_fp32_products(fp32_products)
        self.device = _pick_device(device)
        self.model = Siam_UNet(n_filter=n_filter, mode=mode).to(self.device)      # no init_weights here (reference :61)
        self.n_filter, self.mode = n_filter, mode
        self.loss_function, self.loss_params = loss_function, loss_params
        # the Siam package's own criteria: its BCEDice takes nn.BCELoss on sigmoid(logits) (siam_unet/losses.py:5-39,73-105)
        self.criterion = _make_criterion(loss_function, loss_params, extra={"BCEDice": BCEDiceLossSiam, "weightedBCELoss": weightedBCELoss})
        self._setup(dataset, num_epochs, batch_size, lr, val_split, save_dir, save_name, save_iter)
        if load_weights is not None:
            self.state = torch.load(load_weights)
            self.model.load_state_dict(self.state["state_dict"])

    def _forward_loss(self, batch, validating):
        d = self.dim
        shape = (self.batch_size, 1, d[0], d[1])
        x = batch["image"].view(shape).to(self.device)
        px = batch["prev_image"].view(shape).to(self.device)
        y = _target(batch["mask"].view(shape).to(self.device))
        _, logits = self.model(x, px)
        return self.criterion(logits, y)

    def iterate(self, epoch, mode):
        if mode == "train":
            self._train_epoch(epoch)
            return None
        print("\nStarting validation epoch %s ..." % epoch)
        losses = []
        with torch.no_grad():
            for batch in tqdm(self.val_loader, total=len(self.val_loader), unit="batch"):
                losses.append(self._forward_loss(batch, validating=True).detach())
        return torch.stack(losses).mean()

    def start(self, test_data_path=None, result_path=None, test_resize_dim=(512, 512)):
        for epoch in range(self.num_epochs):
            self.iterate(epoch, "train")
            self.state = {"epoch": epoch, "best_loss": self.best_loss, "state_dict": self.model.state_dict(),
                          "optimizer": self.optimizer.state_dict(), "lr": self.lr, "loss": self.loss_function,
                          "loss_params": self.loss_params, "n_filter": self.n_filter, "mode": self.mode,
                          "augmentation": getattr(self.data, "aug_factor", None),
                          **self._data_attr("clip_threshold", "noise_amp", "brightness_contrast", "shiftscalerotate")}
            with torch.no_grad():
                val_loss = self.iterate(epoch, "val")
                self.scheduler.step(val_loss)
            if val_loss < self.best_loss:
                print(f"\nEpoch {epoch}: Validation loss improved from {round(self.best_loss.item(), 5)} to "
                      f"{round(val_loss.item(), 5)} - saving model state")
                self.state["best_loss"] = self.best_loss = val_loss
                self._save(self.save_name)
            else:
                print(f"\nEpoch {epoch}: Validation loss did not improve from {round(self.best_loss.item(), 5)}")
            if self.save_iter:
                self._save(f"model_epoch_{epoch}.pt")
            if test_data_path is not None:
                raise NotImplementedError("per-epoch prediction of TIFF test folders is outside the hot path")


class TrainerMo3d:
    """``bio_image_unet.multi_output_unet3d.Trainer`` counterpart (``multi_output_unet3d/train.py:17-292``).

    Per-head loss ``output_heads[name]['loss']`` applied to the model's (already activated) output, summed with the heads'
    weights; ``clip_grad_norm_(1.0)`` before the step; validation applies the head activation once more before the loss
    (``:224``); scheduler ``ReduceLROnPlateau(patience=5, factor=0.2)``; the model stays in train mode during validation."""

    def __init__(self, dataset, output_heads, num_epochs, network=MultiOutputUnet3D, use_interpolation=False, batch_size=4,
                 lr=1e-3, in_channels=1, n_filter=64, dilation=1, val_split=0.2, save_dir="./", save_name="model.pt",
                 save_iter=False, load_weights=False, loss_function="BCEDice", loss_params=(0.5, 0.5), time_loss_weight=0.1,
                 device: Union[torch.device, str] = "auto"):
        self.device = _pick_device(device)
        self.network = network
        self.model = network(n_filter=n_filter, in_channels=in_channels, output_heads=output_heads,
                             use_interpolation=use_interpolation).to(self.device)
        self.model.apply(init_weights)          # a no-op on Conv3d layers, as in the reference
        self.data, self.output_heads, self.num_epochs, self.batch_size, self.lr = dataset, output_heads, num_epochs, batch_size, lr
        self.best_loss = torch.tensor(float("inf"))
        self.save_iter, self.save_dir, self.save_name = save_iter, save_dir, save_name
        self.loss_function, self.loss_params, self.time_loss_weight = loss_function, loss_params, time_loss_weight
        self.n_filter, self.in_channels, self.use_interpolation = n_filter, in_channels, use_interpolation
        self.loss_functions = {name: self._get_loss_function(cfg["loss"]) for name, cfg in output_heads.items()}
        self.activations = {name: cfg.get("activation", None) for name, cfg in output_heads.items()}
        self.loss_weights = {name: cfg.get("weight", 1.0) for name, cfg in output_heads.items()}
        n_val = int(len(dataset) * val_split)
        self.dim = dataset.dim_out
        train_data, val_data = random_split(dataset, [len(dataset) - n_val, n_val])
        self.train_loader, self.val_loader = _loaders(dataset, train_data, val_data, batch_size, self.device)
        if loss_function == "BCEDiceTemporalLoss":
            self.criterion = BCEDiceTemporalLoss(loss_params=loss_params)
        else:
            self.criterion = _make_criterion(loss_function, loss_params)       # built but unused by the loop, as upstream
        self.criterion_time = nn.SmoothL1Loss()
        self.optimizer = Adam(self.model.parameters(), lr=lr)
        self.scheduler = optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, mode="min", patience=5, factor=0.2)
        os.makedirs(save_dir, exist_ok=True)
        keys = ("clip_threshold", "scale_limit", "rotate_limit", "gauss_noise_lims", "shot_noise_lims", "blur_limit",
                "random_rotate", "brightness_contrast")
        self.params = {"optimizer": self.optimizer.state_dict(), "lr": lr, "loss_function": loss_function,
                       "loss_params": loss_params, "time_loss_weight": time_loss_weight, "n_filter": n_filter,
                       "use_interpolation": use_interpolation, "dilation": dilation, "batch_size": batch_size,
                       "augmentation": getattr(dataset, "aug_factor", None),
                       **{k: getattr(dataset, k, None) for k in keys}, "in_channels": in_channels, "output_heads": output_heads}
        if load_weights:
            self.state = torch.load(os.path.join(save_dir, save_name))
            self.model.load_state_dict(self.state["state_dict"])
            self.epoch_start = self.state["epoch"]
        else:
            self.epoch_start = 0

    @staticmethod
    def _get_loss_function(loss_name):
        table = {"BCEDiceLoss": lambda: BCEDiceLoss(1, 1), "DiceLoss": lambda: BCEDiceLoss(0, 1), "TverskyLoss": TverskyLoss,
                 "logcoshTverskyLoss": logcoshTverskyLoss, "BCEDiceTemporalLoss": BCEDiceTemporalLoss}
        if loss_name not in table:
            raise ValueError(f'Loss "{loss_name}" not defined!')
        return table[loss_name]()

    @staticmethod
    def _apply_activation(x, activation):
        fn = {"sigmoid": torch.sigmoid, "tanh": torch.tanh, "relu": torch.relu, "softmax": lambda t: torch.softmax(t, dim=1)}
        return fn[activation](x) if activation in fn else x

    def _total_loss(self, batch, validating):
        x = batch["volume"].to(self.device, non_blocking=True)
        y = {key: _target(batch[key].to(self.device, non_blocking=True)) for key in self.output_heads}
        if x.dim() == 4:
            x = x.unsqueeze(1)
        pred = self.model(x)
        total = 0
        for name in self.output_heads:
            target = y[name]
            if target.dim() == 4:
                target = target.unsqueeze(1)
            p = self._apply_activation(pred[name], self.activations.get(name)) if validating else pred[name]
            total = total + self.loss_weights[name] * self.loss_functions[name](p, target)
        return total

    def iterate(self, epoch, mode):
        if mode == "train":
            with _gc_paused():
                for batch in tqdm(self.train_loader, total=len(self.train_loader), unit="batch"):
                    loss = self._total_loss(batch, validating=False)
                    self.optimizer.zero_grad()
                    loss.backward()
                    self.optimizer.clip_grad_norm_(1.0)          # multi_output_unet3d/train.py:201, as three launches (biu_grad_clip)
                    self.optimizer.step()
            return None
        losses = []
        with torch.no_grad():
            for batch in tqdm(self.val_loader, total=len(self.val_loader), unit="batch"):
                losses.append(self._total_loss(batch, validating=True).detach())
        return torch.stack(losses).mean()

    def start(self):
        for epoch in range(self.num_epochs):
            self.iterate(epoch, "train")
            self.state = {"epoch": epoch + self.epoch_start, "epoch_start": self.epoch_start, "best_loss": self.best_loss,
                          "state_dict": self.model.state_dict()}
            self.state.update(self.params)
            with torch.no_grad():
                val_loss = self.iterate(epoch, "val")
                self.scheduler.step(val_loss)
            if val_loss < self.best_loss:
                print(f"\nValidation loss improved from {self.best_loss.item():.5f} to {val_loss.item():.5f} - saving model state")
                self.state["best_loss"] = self.best_loss = val_loss
                torch.save(self.state, os.path.join(self.save_dir, self.save_name))
            else:
                print(f"\nValidation loss did not improve from {self.best_loss.item():.5f}")
            if self.save_iter:
                torch.save(self.state, os.path.join(self.save_dir, f"model_epoch_{epoch + self.epoch_start}.pt"))


# ----------------------------------------------------------------------------------------------------------------------
# 2-D prediction: normalise -> tile -> forward (eval) -> uint8 -> stitch (mean of overlaps)
# ----------------------------------------------------------------------------------------------------------------------
def normalise_stack(imgs: np.ndarray, mode: str, clip, invert: bool) -> np.ndarray:
    """Percentile clip and rescale to [0, 255] (``unet/predict.py:122-150``): per image ('single'), by the first image's
    histogram ('first') or the whole stack's ('all').  Lower bound ``nanpercentile``, upper ``percentile`` as upstream."""
    def scale(a, lo, hi):
        a = np.clip(a, a_min=lo, a_max=hi)
        a = a - np.min(a)
        a = a / np.max(a) * 255
        return 255 - a if invert else a

    if mode == "single":
        for i, img in enumerate(imgs):
            imgs[i] = scale(img, np.nanpercentile(img, clip[0]), np.percentile(img, clip[1]))
        return imgs
    if mode == "first":
        return scale(imgs, np.nanpercentile(imgs[0], clip[0]), np.percentile(imgs[0], clip[1]))
    if mode == "all":
        return scale(imgs, np.nanpercentile(imgs, clip[0]), np.percentile(imgs, clip[1]))
    raise ValueError(f"normalization_mode {mode} not valid!")


def tile_starts(extent: int, tile: int, n: int) -> np.ndarray:
    """Evenly spaced tile origins, truncated to uint16 like the reference (``unet/predict.py:170-171``)."""
    return np.linspace(0, extent - tile, n).astype("uint16")


class _Stitcher:
    """Stitched result volume kept in HBM: patches are added where the network wrote them (``biu_stitch_add``) and the volume is
    normalised once (``biu_stitch_finish``) -- no per-patch device-to-host copy, no host numpy accumulation.  ``layers`` > 1 keeps
    separate overwrite-mode layers (the three-layer buffer of ``unet3d/predict.py:173-195``)."""

    def __init__(self, device, channels, shape3, layers=1):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError(f"Predict stitches on the GPU (biu_stitch_add / biu_stitch_finish); got device '{device}'. There is no CPU path.")
        self.device, self.channels, self.shape, self.layers = device, channels, tuple(shape3), layers
        d, h, w = self.shape
        self.acc = torch.zeros((layers, channels, d, h, w), dtype=torch.float32, device=device)
        self.wsum = torch.zeros((layers, d, h, w), dtype=torch.float32, device=device)

    def reset(self):
        self.acc.zero_()
        self.wsum.zero_()

    def add(self, patch: torch.Tensor, origin, weight: torch.Tensor = None, layer: int = 0, overwrite: bool = False):
        """patch: device tensor [channels, pd, ph, pw] (uint8 or float32); origin (z0, y0, x0); weight [pd, ph, pw] float32 or None."""
        import ctypes as C
        from ._lib import check, lib
        patch = patch.contiguous()
        assert patch.dim() == 4 and patch.shape[0] == self.channels and patch.dtype in (torch.uint8, torch.float32)
        c, pd, ph, pw = patch.shape
        d, h, w = self.shape
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(lib.biu_stitch_add(C.c_void_p(patch.data_ptr()), int(patch.dtype == torch.uint8),
                                 C.c_void_p(weight.data_ptr()) if weight is not None else None, c, pd, ph, pw,
                                 C.c_void_p(self.acc[layer].data_ptr()), C.c_void_p(self.wsum[layer].data_ptr()), d, h, w,
                                 int(origin[0]), int(origin[1]), int(origin[2]), int(overwrite), st), "stitch_add")

    def finish(self, as_uint8: bool) -> torch.Tensor:
        import ctypes as C
        from ._lib import check, lib
        d, h, w = self.shape
        out = torch.empty((self.channels, d, h, w), dtype=torch.uint8 if as_uint8 else torch.float32, device=self.device)
        check(lib.biu_stitch_finish(C.c_void_p(self.acc.data_ptr()), C.c_void_p(self.wsum.data_ptr()), self.layers, self.channels,
                                    d * h * w, C.c_void_p(out.data_ptr()), int(as_uint8),
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)), "stitch_finish")
        return out


def _quantize_u8(prob: torch.Tensor) -> torch.Tensor:
    """``(res * 255).astype('uint8')`` (unet/predict.py:200) on the device."""
    import ctypes as C
    from ._lib import check, lib
    prob = prob.contiguous().float()
    out = torch.empty(prob.shape, dtype=torch.uint8, device=prob.device)
    check(lib.biu_quantize_u8(C.c_void_p(prob.data_ptr()), 255.0, C.c_void_p(out.data_ptr()), prob.numel(),
                              C.c_void_p(torch.cuda.current_stream().cuda_stream)), "quantize_u8")
    return out


class Predict2D:
    """``bio_image_unet.unet.Predict`` counterpart (``unet/predict.py:14-229``) for in-memory arrays.

    Same preprocessing, tiling, uint8 re-quantisation ``(p*255).astype('uint8')`` and nan-mean stitching; patches are
    pushed through the network in batches (eval-mode BatchNorm makes that identical to the reference's batch of 1) so the
    device is not synchronised once per patch.  The result is kept in ``self.imgs_result`` and written to
    ``result_name`` (TIFF via tifffile when importable -- float16 like ``save_as_tif`` -- otherwise ``.npy``)."""

    def __init__(self, imgs, result_name, model_params, network="Unet", resize_dim=(512, 512), invert=False,
                 normalization_mode="single", clip_threshold=(0., 99.8), add_tile=0, normalize_result=False,
                 show_progress=True, device: Union[torch.device, str] = "auto", progress_notifier=None, batch_size=8):
        self.device = _pick_device(device)
        if isinstance(imgs, str):
            import tifffile                      # only needed for file input; not a dependency of the hot path
            imgs = tifffile.imread(imgs)
        imgs = np.array(imgs, dtype=np.float64 if np.asarray(imgs).dtype.kind != "f" else np.asarray(imgs).dtype)
        self.resize_dim, self.add_tile = tuple(resize_dim), add_tile
        if imgs.ndim == 2:
            imgs = imgs[None]
        self.imgs_shape = imgs.shape
        imgs = normalise_stack(imgs, normalization_mode, clip_threshold, invert)
        patches = self._split(imgs)

        self.model_params = torch.load(model_params, map_location=self.device) if isinstance(model_params, str) else model_params
        if network is None:
            network = self.model_params.get("network")
            if network is None:
                raise ValueError("network is not defined")
        if network == "Unet":
            network = Unet
        elif network == "AttentionUnet":
            network = AttentionUnet
        elif network == "Unet_v0":                       # legacy checkpoints carry no channel counts (unet/predict.py:93-97)
            network = Unet_v0
            if "in_channels" not in self.model_params:
                self.model_params["in_channels"] = 1
                self.model_params["out_channels"] = 1
        elif isinstance(network, str):
            raise ValueError(f"network '{network}' is not one of 'Unet', 'AttentionUnet', 'Unet_v0' (or pass a class)")
        mp = self.model_params
        # (the reference ignores the checkpoint's 'dilation' here, unet/predict.py:98-99 -- so does this)
        self.model = network(n_filter=mp["n_filter"], in_channels=mp["in_channels"], out_channels=mp["out_channels"]).to(self.device)
        self.model.load_state_dict(mp["state_dict"])
        self.model.eval()
        self.imgs_result = self._predict_and_stitch(patches, batch_size)
        self._save(result_name, normalize_result)

    def _split(self, imgs):
        n_img, h, w = self.imgs_shape
        th, tw = self.resize_dim
        self.N_x = int(np.ceil(h / th)) + self.add_tile
        self.N_y = int(np.ceil(w / tw)) + self.add_tile
        self.N_per_img = self.N_x * self.N_y
        if th > h:
            imgs = np.pad(imgs, ((0, 0), (0, th - h), (0, 0)), "reflect")
        if tw > w:
            imgs = np.pad(imgs, ((0, 0), (0, 0), (0, tw - w)), "reflect")
        self.X_start, self.Y_start = tile_starts(h, th, self.N_x), tile_starts(w, tw, self.N_y)
        patches = np.zeros((n_img * self.N_per_img, 1, th, tw), dtype="uint8")
        k = 0
        for img in imgs:
            for xs in self.X_start:
                for ys in self.Y_start:
                    patches[k, 0] = img[xs:xs + th, ys:ys + tw]        # float -> uint8 truncation, as upstream
                    k += 1
        return patches

    def _predict_and_stitch(self, patches, batch_size):
        """uint8 patches up (scaled by 1/255 in the input-layout kernel), eval-mode forward in batches, ``(p * 255)`` truncated to
        uint8 and added into the stitched image on the device; nan-mean of the overlapping uint8 tiles followed by the uint8
        cast == floor(sum / count) (``unet/predict.py:184-229``).  One download per stack."""
        n_img, h, w = self.imgs_shape
        th, tw = self.resize_dim
        oc = self.model_params["out_channels"]
        H, W = max(th, h), max(tw, w)
        origins = [(0, int(xs), int(ys)) for xs in self.X_start for ys in self.Y_start]
        st = _Stitcher(self.device, oc, (1, H, W))
        done, cur = [], 0
        with torch.no_grad():
            for i in range(0, patches.shape[0], batch_size):
                x = torch.from_numpy(patches[i:i + batch_size]).to(self.device, non_blocking=True)      # uint8 across PCIe
                prob, _ = self.model(x)
                q = _quantize_u8(prob)
                for j in range(q.shape[0]):
                    k = i + j
                    if k // self.N_per_img != cur:
                        done.append(st.finish(True))
                        st.reset()
                        cur = k // self.N_per_img
                    st.add(q[j].unsqueeze(1), origins[k % self.N_per_img])
            done.append(st.finish(True))
        res = torch.stack(done).cpu().numpy()[:, :, 0]                       # (n_img, oc, H, W)
        return np.squeeze(res[:, :, :h, :w])

    def _save(self, result_name, normalize):
        if result_name is None:
            return
        img = self.imgs_result
        if normalize:
            img = img - np.nanmin(img)
            img = img / np.nanpercentile(img, 99.8)
            img = np.clip(img, 0, 1)
        try:
            import tifffile
            tifffile.imwrite(result_name, img.astype("float16") if normalize else img)
        except ImportError:
            np.save(result_name + ".npy" if not result_name.endswith(".npy") else result_name, img)


def _load_params(model_params, device):
    return torch.load(model_params, map_location=device) if isinstance(model_params, str) else model_params


def _write(result_name, arr):
    if result_name is None:
        return
    try:
        import tifffile
        tifffile.imwrite(result_name, arr)
    except ImportError:
        np.save(result_name if result_name.endswith(".npy") else result_name + ".npy", arr)


class Predict3D:
    """``bio_image_unet.unet3d.Predict`` counterpart (``unet3d/predict.py:12-195``) for in-memory volumes.

    Whole-volume percentile clip to [0, 255]; patches of ``resize_dim`` (z, x, y) at ``linspace`` origins (reflect padding
    when the volume is smaller); one eval-mode forward per patch; uint8 re-quantisation; stitching through the
    reference's three-layer float16 buffer (patch ``n`` goes to layer ``n % 3``, later patches overwrite earlier ones of
    the same layer, then ``nanmean`` over the layers)."""

    def __init__(self, vol, result_name, model_params, network=UNet3D, resize_dim=(64, 128, 128), invert=False,
                 normalization_mode="single", clip_threshold=(0., 99.8), add_patch=0, normalize_result=False,
                 progress_bar=True, device: Union[torch.device, str] = "auto", progress_notifier=None):
        self.device = _pick_device(device)
        if isinstance(vol, str):
            import tifffile
            vol = tifffile.imread(vol)
        vol = np.asarray(vol)
        self.resize_dim, self.add_patch = tuple(resize_dim), add_patch
        self.vol_shape = vol.shape
        if vol.ndim == 2:
            vol = vol[None]
            self.vol_shape = vol.shape
        lo, hi = np.nanpercentile(vol, clip_threshold[0]), np.percentile(vol, clip_threshold[1])
        vol = np.clip(vol, lo, hi)
        vol = vol - np.min(vol)
        vol = vol / np.max(vol) * 255
        if invert:
            vol = 255 - vol
        patches = self._split(vol)
        mp = self.model_params = _load_params(model_params, self.device)
        self.model = network(n_filter=mp["n_filter"], in_channels=mp["in_channels"], out_channels=mp["out_channels"],
                             use_interpolation=mp.get("use_interpolation", False)).to(self.device)
        self.model.load_state_dict(mp["state_dict"])
        self.model.eval()
        # one eval-mode forward per patch as upstream (unet3d/predict.py:155-171); quantisation and the three-layer stitch buffer
        # (patch n overwrites layer n % 3, then the nan-mean over the layers, :173-195) stay on the device
        vs, rd = self.vol_shape, self.resize_dim
        st = _Stitcher(self.device, 1, tuple(max(vs[a], rd[a]) for a in range(3)), layers=3)
        origins = [(int(z), int(x), int(y)) for z in self.Z_start for x in self.X_start for y in self.Y_start]
        with torch.no_grad():
            for i, p in enumerate(patches):
                x = torch.from_numpy(p).to(self.device, non_blocking=True).view((1, 1) + self.resize_dim)      # uint8; /255 in the layout kernel
                prob, _ = self.model(x)
                st.add(_quantize_u8(prob).view((1,) + self.resize_dim), origins[i], layer=i % 3, overwrite=True)
        out = st.finish(True)[0].cpu().numpy()
        self.vol_result = np.squeeze(out[:vs[0], :vs[1], :vs[2]])
        out = self.vol_result
        if normalize_result:
            out = out - np.nanmin(out)
            out = np.clip(out / np.nanpercentile(out, 99.8), 0, 1).astype("float16")
        _write(result_name, out)

    def _split(self, vol):
        vs, rd, ap = self.vol_shape, self.resize_dim, self.add_patch
        self.N_z = int(np.ceil(vs[0] / rd[0])) + ap
        self.N_x = int(np.ceil(vs[1] / rd[1])) + ap
        self.N_y = int(np.ceil(vs[2] / rd[2])) + ap
        self.N_x += ap if self.N_z > 1 else 0          # (sic) the reference bumps N_x twice (unet3d/predict.py:124-126)
        self.N_x += ap if self.N_x > 1 else 0
        self.N_y += ap if self.N_y > 1 else 0
        self.N = self.N_x * self.N_y * self.N_z
        vol = np.pad(vol, [(0, max(0, rd[a] - vs[a])) for a in range(3)], "reflect")
        self.Z_start = tile_starts(vs[0], rd[0], self.N_z)
        self.X_start = tile_starts(vs[1], rd[1], self.N_x)
        self.Y_start = tile_starts(vs[2], rd[2], self.N_y)
        patches = np.zeros((self.N,) + rd, dtype="uint8")
        n = 0
        for z in self.Z_start:
            for x in self.X_start:
                for y in self.Y_start:
                    patches[n] = vol[z:z + rd[0], x:x + rd[1], y:y + rd[2]]
                    n += 1
        return patches


class PredictSiam:
    """``bio_image_unet.siam_unet.Predict`` counterpart (``siam_unet/predict.py:16-250``) for an in-memory movie (T, H, W).

    Frame i is predicted together with its predecessor (frame 0 with frame 1, or with itself in a one-frame movie); every
    pair is normalised on its own (``normalization_mode`` over the two-frame stack), zero-padded (``'constant'``) up to
    ``resize_dim``, tiled, predicted and stitched by the nan-mean of overlapping uint8 tiles."""

    def __init__(self, movie, result_name, model_params, resize_dim=None, invert=False, normalization_mode="single",
                 clip_threshold=(0., 99.8), add_tile=0, normalize_result=False, show_progress=True,
                 device: Union[torch.device, str] = "auto", progress_notifier=None, batch_size=8):
        self.device = _pick_device(device)
        if isinstance(movie, str):
            import tifffile
            movie = tifffile.imread(movie)
        movie = np.asarray(movie)
        if movie.ndim == 2:
            movie = movie[None]
        mp = self.model_params = _load_params(model_params, self.device)
        self.model = Siam_UNet(n_filter=mp["n_filter"], mode=mp["mode"]).to(self.device)
        self.model.load_state_dict(mp["state_dict"])
        self.model.eval()
        self.tif_len = movie.shape[0]
        self.imgs_shape = [self.tif_len, movie.shape[1], movie.shape[2]]
        self.resize_dim = tuple(resize_dim) if resize_dim is not None else (movie.shape[1], movie.shape[2])
        th, tw = self.resize_dim
        self.N_x = int(np.ceil(self.imgs_shape[1] / th)) + add_tile
        self.N_y = int(np.ceil(self.imgs_shape[2] / tw)) + add_tile
        self.N_per_img = self.N = self.N_x * self.N_y
        self.X_start = tile_starts(self.imgs_shape[1], th, self.N_x)
        self.Y_start = tile_starts(self.imgs_shape[2], tw, self.N_y)
        frames = []
        cur = None
        h, w = self.imgs_shape[1], self.imgs_shape[2]
        stitch = _Stitcher(self.device, 1, (1, max(th, h), max(tw, w)))
        origins = [(0, int(xs), int(ys)) for xs in self.X_start for ys in self.Y_start]
        for i in range(self.tif_len):
            prev = (movie[0] if self.tif_len == 1 else movie[1]) if i == 0 else cur
            cur = movie[i]
            pair = normalise_stack(np.array([prev, cur], dtype=np.float64), normalization_mode, clip_threshold, invert).astype("uint8")
            patches = self._split(pair)
            stitch.reset()
            with torch.no_grad():
                for b in range(0, self.N, batch_size):
                    both = torch.from_numpy(patches[b:b + batch_size]).to(self.device, non_blocking=True)       # uint8 (cur, prev)
                    q = _quantize_u8(self.model(both[:, 0:1].contiguous(), both[:, 1:2].contiguous())[0])
                    for j in range(q.shape[0]):
                        stitch.add(q[j].unsqueeze(1), origins[b + j])
            frames.append(stitch.finish(True)[0, 0, :h, :w])                 # nan-mean of uint8 tiles, truncated like astype
        self.imgs_result = torch.stack(frames).cpu().numpy()
        _write(result_name, self.imgs_result)

    def _split(self, pair):
        th, tw = self.resize_dim
        h, w = self.imgs_shape[1], self.imgs_shape[2]
        pair = np.pad(pair, ((0, 0), (0, max(0, th - h)), (0, max(0, tw - w))), "constant")
        patches = np.zeros((self.N, 2, th, tw), dtype="uint8")
        n = 0
        for xs in self.X_start:
            for ys in self.Y_start:
                patches[n, 0] = pair[1][xs:xs + th, ys:ys + tw]          # current frame
                patches[n, 1] = pair[0][xs:xs + th, ys:ys + tw]          # previous frame
                n += 1
        return patches


class PredictMo3d:
    """``bio_image_unet.multi_output_unet3d.Predict`` counterpart (``multi_output_unet3d/predict.py:13-307``).

    Float normalisation to [0, 1] (no uint8 step), patches of ``min(volume, max_patch_size)`` on a stride of
    ``patch * (1 - overlap_factor)`` plus a last patch flush with the border, batches of ``batch_size``, outputs kept as
    float32 per head, weighted blending with the reference's linear ramps over ``blend_margin`` voxels at interior patch
    faces (built exactly as upstream assigns them: plane by plane, later axes overwriting earlier ones)."""

    def __init__(self, imgs, model_params, result_path=None, network=MultiOutputUnet3D, max_patch_size=(64, 256, 256),
                 overlap_factor=0.1, batch_size=1, normalization_mode="single", clip_threshold=(0., 99.98), add_tile=0,
                 compress_tif=False, show_progress=True, device: Union[torch.device, str] = "auto", progress_notifier=None):
        self.device = _pick_device(device)
        if isinstance(imgs, str):
            import tifffile
            imgs = tifffile.imread(imgs)
        self.max_patch_size, self.overlap_factor, self.batch_size = max_patch_size, overlap_factor, batch_size
        imgs = np.array(imgs, dtype="float32")
        if imgs.ndim == 3:
            imgs = imgs[None]
        elif imgs.ndim != 4:
            raise ValueError(f"Unsupported input shape: {imgs.shape}")
        self.imgs_shape = imgs.shape
        imgs = self._preprocess(imgs, normalization_mode, clip_threshold)
        patches = self._split(imgs)
        mp = self.model_params = _load_params(model_params, self.device)
        self.model = network(in_channels=mp["in_channels"], n_filter=mp["n_filter"], output_heads=mp["output_heads"],
                             use_interpolation=mp.get("use_interpolation", True)).to(self.device)
        self.model.load_state_dict(mp["state_dict"])
        self.model.eval()
        self.target_keys = list(mp["output_heads"].keys())
        result = self._predict_and_blend(patches, batch_size)
        if result_path is not None:
            for k in self.target_keys:
                _write((result_path + k + ".tif") if os.path.exists(result_path) else (result_path + "_" + k + ".tif"), result[k])
            self.result = None
        else:
            self.result = result

    @staticmethod
    def _preprocess(imgs, mode, clip):
        if mode == "single":
            for i in range(len(imgs)):
                c = np.clip(imgs[i], np.percentile(imgs[i], clip[0]), np.percentile(imgs[i], clip[1]))
                imgs[i] = (c - c.min()) / (np.ptp(c) + 1e-8)
            return imgs
        if mode in ("first", "all"):
            lo, hi = np.percentile(imgs[0] if mode == "first" else imgs, [clip[0], clip[1]])
            imgs[:] = (np.clip(imgs, lo, hi) - lo) / (hi - lo + 1e-8)
            return imgs
        raise ValueError(f"Invalid normalization mode: {mode}")

    def _split(self, imgs):
        nvol, D, H, W = imgs.shape
        self.patch_size = ps = tuple(min(a, b) for a, b in zip((D, H, W), self.max_patch_size))
        stride = [max(1, int(s * (1 - self.overlap_factor))) for s in ps]

        def starts(extent, patch, st):
            v = list(range(0, max(extent - patch + 1, 1), st))
            if v[-1] + patch < extent:
                v.append(extent - patch)
            return v
        self.Z_start, self.Y_start, self.X_start = starts(D, ps[0], stride[0]), starts(H, ps[1], stride[1]), starts(W, ps[2], stride[2])
        self.N_z, self.N_y, self.N_x = len(self.Z_start), len(self.Y_start), len(self.X_start)
        self.N_per_vol = self.N_z * self.N_y * self.N_x
        out = [imgs[v, z:z + ps[0], y:y + ps[1], x:x + ps[2]] for v in range(nvol) for z in self.Z_start for y in self.Y_start
               for x in self.X_start]
        return np.stack(out)[:, None]

    def _weights(self, flags, shape, blend_margin):
        """Blend mask of one patch; ``flags`` = (front, back, top, bottom, left, right) interior faces.  Restates the
        assignment order of ``predict.py:246-268`` including its index arithmetic: the 'far' ramps all land on plane 0
        (``max(-(i+1), 0) == 0``) and the depth ramps run over ``min(blend_margin, N_z)`` planes."""
        w = np.ones(shape, dtype="float32")
        nz = min(blend_margin, self.N_z)
        if flags[0]:
            for i in range(nz):
                w[:, i, :, :] = i / blend_margin
        if flags[1]:
            for i in range(nz):
                w[:, 0, :, :] = i / blend_margin
        if flags[2]:
            for i in range(blend_margin):
                w[:, :, i, :] = i / blend_margin
        if flags[3]:
            for i in range(blend_margin):
                w[:, :, 0, :] = i / blend_margin
        if flags[4]:
            for i in range(blend_margin):
                w[:, :, :, i] = i / blend_margin
        if flags[5]:
            for i in range(blend_margin):
                w[:, :, :, 0] = i / blend_margin
        return w

    def _predict_and_blend(self, patches, batch_size, blend_margin=16):
        """Batches of patches through the network; every head's float32 patch is multiplied by its blend mask and added into
        the head's volume on the device, then ``vol / wsum`` where ``wsum > 0`` (``multi_output_unet3d/predict.py:203-307``)."""
        nvol, D, H, W = self.imgs_shape
        ps = self.patch_size
        heads = self.model_params["output_heads"]
        stitch = {k: [_Stitcher(self.device, heads[k]["channels"], (D, H, W)) for _ in range(nvol)] for k in self.target_keys}
        wcache = {}

        def weight_of(zi, yi, xi):
            flags = (zi > 0, zi < self.N_z - 1, yi > 0, yi < self.N_y - 1, xi > 0, xi < self.N_x - 1)
            if flags not in wcache:
                wcache[flags] = torch.from_numpy(self._weights(flags, (1,) + ps, blend_margin)[0]).to(self.device).contiguous()
            return wcache[flags]
        index = [(v, zi, yi, xi) for v in range(nvol) for zi in range(self.N_z) for yi in range(self.N_y) for xi in range(self.N_x)]
        with torch.no_grad():
            for b in range(0, len(patches), batch_size):
                preds = self.model(torch.from_numpy(np.ascontiguousarray(patches[b:b + batch_size])).to(self.device, non_blocking=True))
                for j in range(min(batch_size, len(patches) - b)):
                    v, zi, yi, xi = index[b + j]
                    wt = weight_of(zi, yi, xi)
                    for k in self.target_keys:
                        stitch[k][v].add(preds[k][j].float(), (self.Z_start[zi], self.Y_start[yi], self.X_start[xi]), weight=wt)
        return {k: np.squeeze(torch.stack([s_.finish(False) for s_ in stitch[k]]).cpu().numpy()) for k in self.target_keys}
