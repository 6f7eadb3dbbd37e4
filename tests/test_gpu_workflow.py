"""Trainer / Predict counterparts on the GPU: one optimisation step against the oracle (loss expression quirks included),
a short end-to-end run mirroring the reference's smoke script (utils/test.py:18-46, synthetic tensors instead of TIFFs),
and tiled prediction against an oracle-side restatement of tile -> forward -> uint8 -> nan-mean stitch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import bio_image_unet_amd.siam_unet as siam  # noqa: E402
import bio_image_unet_amd.unet as unet  # noqa: E402
import bio_image_unet_amd.unet3d as unet3d  # noqa: E402
from oracle import unet_oracle as O  # noqa: E402


class Tiles(torch.utils.data.Dataset):
    """The reference data sets' item contract: dict of float32 tensors in [0,1] + a few attributes the Trainer records."""
    aug_factor, clip_threshold, noise_lims, noise_amp, brightness_contrast, shiftscalerotate = 1, (0.2, 99.8), None, None, None, None

    def __init__(self, n, dim, keys, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.dim_out = dim
        self.items = []
        for _ in range(n):
            it = {k: torch.rand(dim, generator=g) for k in keys}
            it["mask"] = (torch.rand(dim, generator=g) > 0.5).float()
            self.items.append(it)

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def test_trainer2d_one_step_matches_oracle(tmp_path):
    torch.manual_seed(0)
    ds = Tiles(8, (32, 32), ["image"])
    tr = unet.Trainer(ds, 1, batch_size=2, n_filter=8, in_channels=1, out_channels=1, save_dir=str(tmp_path), device="cuda")
    sd0 = {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()}
    batch = next(iter(tr.train_loader))
    # oracle side: same weights, same batch, the reference's loss expression, torch Adam
    osd = O.clone_state(sd0, requires_grad=True)
    x = batch["image"].view(2, 1, 32, 32)
    y = batch["mask"].view(2, 1, 32, 32)
    _, ol = O.unet2d_forward(osd, x, training=True)
    oloss = O.trainer2d_loss(ol, y, 1)
    params = [v for v in osd.values() if v.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-3)
    oloss.backward()
    opt.step()
    # product side
    loss = tr._forward_loss(batch, validating=False)
    assert abs(float(loss) - float(oloss)) < 1e-4
    tr.optimizer.zero_grad()
    loss.backward()
    tr.optimizer.step()
    torch.cuda.synchronize()
    new = tr.model.state_dict()
    worst = 0.0
    for k, v in osd.items():
        # conv biases in front of a BatchNorm have an exactly-zero gradient here and rounding noise (~1e-9) in the
        # reference, which Adam's first step turns into a random +-lr walk of a parameter the network is invariant to
        is_dead_bias = k.endswith(".0.bias") and not k.startswith("final")
        if v.requires_grad and not is_dead_bias:
            # Adam's first step moves every weight by ~lr*sign(g): compare the *update*
            du_ref = v.detach() - sd0[k]
            du = new[k].cpu() - sd0[k]
            big = du_ref.abs() > 0.5e-3          # entries whose gradient is well away from zero
            if big.any():
                worst = max(worst, float((du - du_ref)[big].abs().max()))
    assert worst < 2e-4, worst


def test_trainers_run_and_checkpoint(tmp_path):
    torch.manual_seed(1)
    tr = unet.Trainer(Tiles(10, (32, 32), ["image"]), 2, batch_size=2, n_filter=4, save_dir=str(tmp_path / "a"), device="cuda")
    tr.start()
    ck = torch.load(str(tmp_path / "a" / "model.pt"), weights_only=False)
    assert {"epoch", "best_loss", "state_dict", "optimizer", "lr", "loss_function", "loss_params", "n_filter", "dilation",
            "batch_size", "in_channels", "out_channels"} <= set(ck)
    assert len(ck["state_dict"]) == 136
    tr3 = unet3d.Trainer(Tiles(10, (8, 16, 16), ["volume"]), 1, batch_size=2, n_filter=8, save_dir=str(tmp_path / "b"), device="cuda")
    tr3.start()
    assert "use_interpolation" in torch.load(str(tmp_path / "b" / "model.pt"), weights_only=False)
    trs = siam.Trainer(Tiles(10, (32, 32), ["image", "prev_image"]), 1, batch_size=2, n_filter=4, mode="max",
                       save_dir=str(tmp_path / "c"), device="cuda")
    trs.start()
    assert torch.load(str(tmp_path / "c" / "model.pt"), weights_only=False)["mode"] == "max"


def test_predict2d_matches_oracle_tiling(tmp_path):
    torch.manual_seed(2)
    sd = O.init_unet2d(1, 1, 4, seed=5)
    # give the BatchNorm buffers non-trivial running statistics
    for k in sd:
        if k.endswith("running_mean"):
            sd[k] = torch.randn_like(sd[k]) * 0.1
        if k.endswith("running_var"):
            sd[k] = torch.rand_like(sd[k]) + 0.5
    ck = {"n_filter": 4, "in_channels": 1, "out_channels": 1, "state_dict": sd}
    imgs = (np.random.RandomState(0).rand(2, 70, 100) * 1000).astype("float32")
    p = unet.Predict(imgs.copy(), str(tmp_path / "res.tif"), ck, network="Unet", resize_dim=(32, 48), add_tile=1,
                     show_progress=False, device="cuda")
    got = p.imgs_result
    assert got.shape == (2, 70, 100) and got.dtype == np.uint8
    # oracle-side restatement of the same pipeline with numpy nan-mean stitching
    from bio_image_unet_amd.workflow import normalise_stack, tile_starts
    norm = normalise_stack(imgs.astype("float64"), "single", (0., 99.8), False)
    xs, ys = tile_starts(70, 32, 4), tile_starts(100, 48, 4)
    want = np.zeros((2, 70, 100), dtype="uint8")
    for i in range(2):
        stack = np.full((16, 70, 100), np.nan)
        k = 0
        for a in xs:
            for b in ys:
                patch = norm[i, a:a + 32, b:b + 48].astype("uint8").astype("float32") / 255
                with torch.no_grad():
                    prob, _ = O.unet2d_forward(sd, torch.from_numpy(patch)[None, None], training=False)
                stack[k, a:a + 32, b:b + 48] = (prob[0, 0].numpy() * 255).astype("uint8")
                k += 1
        want[i] = np.nanmean(stack, axis=0)
    diff = np.abs(got.astype(int) - want.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 0.02      # uint8 truncation may flip at exact .0 boundaries
