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
from typing import Union

import numpy as np
import torch
from torch import nn, optim
from torch.utils.data import DataLoader, random_split

from .losses import BCEDiceLoss, TverskyLoss, logcoshTverskyLoss
from .models import Siam_UNet, UNet3D, Unet
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
        self.train_loader = DataLoader(train_data, batch_size=batch_size, pin_memory=True, drop_last=True)
        self.val_loader = DataLoader(val_data, batch_size=batch_size, pin_memory=True, drop_last=True)
        self.optimizer = Adam(self.model.parameters(), lr=lr)
        self.scheduler = optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, mode="min", patience=4, factor=0.1)
        os.makedirs(save_dir, exist_ok=True)

    def _data_attr(self, *names):
        return {n: getattr(self.data, n, None) for n in names}

    # subclasses: _forward_loss(batch, validating) -> loss
    def _train_epoch(self, epoch):
        print("\nStarting training epoch %s ..." % epoch)
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
                 device: Union[torch.device, str] = "auto"):
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
        y = batch["mask"].view(self.batch_size, self.out_channels, d[0], d[1] if validating else d[0]).to(self.device)
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
        y = batch["mask"].view(self.batch_size, self.out_channels, d[0], d[1], d[2]).to(self.device)
        _, logits = self.model(x)
        w = 0.1 if validating else self.time_loss_weight         # validation hard-codes 0.1
        return self.criterion(logits, y) + self.criterion_time(logits[1:, :, :], logits[:-1, :, :]) * w

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
                 load_weights=None, device: Union[torch.device, str] = "auto"):
        self.device = _pick_device(device)
        self.model = Siam_UNet(n_filter=n_filter, mode=mode).to(self.device)      # no init_weights here (reference :61)
        self.n_filter, self.mode = n_filter, mode
        self.loss_function, self.loss_params = loss_function, loss_params
        if loss_function == "weightedBCELoss":
            raise NotImplementedError("weightedBCELoss (siam_unet/losses.py:109-148) is not restated yet")
        self.criterion = _make_criterion(loss_function, loss_params)
        self._setup(dataset, num_epochs, batch_size, lr, val_split, save_dir, save_name, save_iter)
        if load_weights is not None:
            self.state = torch.load(load_weights)
            self.model.load_state_dict(self.state["state_dict"])

    def _forward_loss(self, batch, validating):
        d = self.dim
        shape = (self.batch_size, 1, d[0], d[1])
        x = batch["image"].view(shape).to(self.device)
        px = batch["prev_image"].view(shape).to(self.device)
        y = batch["mask"].view(shape).to(self.device)
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
        elif isinstance(network, str):
            raise NotImplementedError(f"network '{network}' is outside the hot path of this package")
        mp = self.model_params
        # (the reference ignores the checkpoint's 'dilation' here, unet/predict.py:98-99 -- so does this)
        self.model = network(n_filter=mp["n_filter"], in_channels=mp["in_channels"], out_channels=mp["out_channels"]).to(self.device)
        self.model.load_state_dict(mp["state_dict"])
        self.model.eval()
        result_patches = self._predict(patches, batch_size)
        self.imgs_result = self._stitch(result_patches)
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

    def _predict(self, patches, batch_size):
        oc = self.model_params["out_channels"]
        out = np.zeros((patches.shape[0], oc) + patches.shape[2:], dtype="uint8")
        with torch.no_grad():
            for i in range(0, patches.shape[0], batch_size):
                x = torch.from_numpy(patches[i:i + batch_size].astype("float32") / 255).to(self.device)
                prob, _ = self.model(x)
                out[i:i + batch_size] = (prob * 255).to(torch.uint8).cpu().numpy()      # truncation == astype('uint8')
        return out

    def _stitch(self, result_patches):
        n_img, h, w = self.imgs_shape
        th, tw = self.resize_dim
        oc = self.model_params["out_channels"]
        H, W = max(th, h), max(tw, w)
        res = np.zeros((n_img, oc, H, W), dtype="uint8")
        for i in range(n_img):
            acc = np.zeros((oc, H, W), dtype=np.int64)
            cnt = np.zeros((1, H, W), dtype=np.int64)
            k = 0
            for xs in self.X_start:
                for ys in self.Y_start:
                    acc[:, xs:xs + th, ys:ys + tw] += result_patches[i * self.N_per_img + k]
                    cnt[:, xs:xs + th, ys:ys + tw] += 1
                    k += 1
            # nan-mean of the overlapping uint8 tiles followed by the uint8 cast == floor(sum / count)
            res[i] = (acc // np.maximum(cnt, 1)).astype("uint8")
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
