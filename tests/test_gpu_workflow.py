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


def _engine_decisions(model):
    """LeakyReLU / max-pool / max-join decisions of the model's last forward (tests/insitu.py): the oracle step below is taken on
    the same branch of the piecewise-linear network, so its Adam update is comparable entry by entry."""
    from tests import insitu
    return insitu.extract_decisions(list(model._engines.values())[-1][-1])


def test_trainer2d_one_step_matches_oracle(tmp_path):
    torch.manual_seed(0)
    ds = Tiles(8, (32, 32), ["image"])
    tr = unet.Trainer(ds, 1, batch_size=2, n_filter=8, in_channels=1, out_channels=1, save_dir=str(tmp_path), device="cuda")
    sd0 = {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()}
    batch = next(iter(tr.train_loader))
    # product side
    loss = tr._forward_loss(batch, validating=False)
    tr.optimizer.zero_grad()
    loss.backward()
    q = _engine_decisions(tr.model)
    tr.optimizer.step()
    torch.cuda.synchronize()
    # oracle side: same weights, same batch, the reference's loss expression, torch Adam
    osd = O.clone_state(sd0, requires_grad=True)
    x = batch["image"].view(2, 1, 32, 32)
    y = batch["mask"].view(2, 1, 32, 32)
    with O.forced_decisions(q):
        _, ol = O.unet2d_forward(osd, x, training=True)
    oloss = O.trainer2d_loss(ol, y, 1)
    opt = torch.optim.Adam([v for v in osd.values() if v.requires_grad], lr=1e-3)
    oloss.backward()
    opt.step()
    assert abs(float(loss) - float(oloss)) < 1e-4
    _adam_update_matches(tr.model.state_dict(), osd, sd0)


def _adam_update_matches(new, osd, sd0, tol=2e-4):
    """After one Adam step every weight moved by lr * g / (|g| + eps) ~ lr * sign(g): compare the UPDATES of the entries whose
    oracle gradient is well away from zero (an entry with |g| ~ eps moves by a rounding-dependent fraction of lr in the
    reference too).  The oracle step is taken on the engine's branch of the LeakyReLU / max decisions (forced_decisions): at these
    tiny extents (4 x 4 bottleneck maps) one decision within fp32 rounding of its boundary flips the SIGN of weight-gradient
    entries of 8 % of the tensor's largest.  Conv biases in front of a BatchNorm have true gradient 0: a random +-lr walk in the
    reference, exact 0 here."""
    worst, where = 0.0, ""
    for k, v in osd.items():
        is_dead_bias = k.endswith(".0.bias") and not k.startswith("final")
        if v.requires_grad and not is_dead_bias and v.grad is not None:
            du_ref, du = v.detach() - sd0[k], new[k].cpu() - sd0[k]
            big = v.grad.abs() > 1e-3 * float(v.grad.abs().max())
            if big.any():
                e = (du - du_ref).abs() * big
                if float(e.max()) > worst:
                    i = int(e.argmax())
                    worst = float(e.max())
                    where = f"{k}[{i}]: update {float(du.flatten()[i]):.3e} vs {float(du_ref.flatten()[i]):.3e}, oracle grad {float(v.grad.flatten()[i]):.3e} (max {float(v.grad.abs().max()):.3e})"
    assert worst < tol, where


def test_trainer3d_one_step_matches_oracle(tmp_path):
    """unet3d/train.py:129-150: BCEDice + SmoothL1 between neighbouring BATCH entries * time_loss_weight, Adam."""
    torch.manual_seed(0)
    ds = Tiles(8, (8, 16, 16), ["volume"])
    tr = unet3d.Trainer(ds, 1, batch_size=2, n_filter=8, save_dir=str(tmp_path), time_loss_weight=0.1, device="cuda")
    sd0 = {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()}
    batch = next(iter(tr.train_loader))
    loss = tr._forward_loss(batch, validating=False)
    tr.optimizer.zero_grad()
    loss.backward()
    q = _engine_decisions(tr.model)
    tr.optimizer.step()
    torch.cuda.synchronize()
    osd = O.clone_state(sd0, requires_grad=True)
    x, y = batch["volume"].view(2, 1, 8, 16, 16), batch["mask"].view(2, 1, 8, 16, 16)
    with O.forced_decisions(q):
        _, ol = O.unet3d_forward(osd, x, training=True)
    oloss = O.trainer3d_loss(ol, y, 0.1)
    opt = torch.optim.Adam([v for v in osd.values() if v.requires_grad], lr=1e-3)
    oloss.backward()
    opt.step()
    assert abs(float(loss) - float(oloss)) < 1e-4
    _adam_update_matches(tr.model.state_dict(), osd, sd0)
    # validation hard-codes the time weight 0.1 whatever the trainer was given (unet3d/train.py:163-169)
    tr.time_loss_weight = 0.7
    with torch.no_grad():
        v = tr._forward_loss(batch, validating=True)
        t = tr._forward_loss(batch, validating=False)
    assert float(t) > float(v)


@pytest.mark.parametrize("mode", ["max", "concat"])
def test_trainer_siam_one_step_matches_oracle(tmp_path, mode):
    """siam_unet/train.py:100-114 with the Siam package's own BCEDice (BCELoss on sigmoid(logits), loss_params (1, 1)); no
    init_weights (reference :61)."""
    torch.manual_seed(0)
    ds = Tiles(8, (32, 32), ["image", "prev_image"])
    tr = siam.Trainer(ds, 1, batch_size=2, n_filter=8, mode=mode, save_dir=str(tmp_path), device="cuda")
    assert type(tr.criterion).__name__ == "BCEDiceLossSiam" and siam.BCEDiceLoss is type(tr.criterion)
    sd0 = {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()}
    batch = next(iter(tr.train_loader))
    loss = tr._forward_loss(batch, validating=False)
    tr.optimizer.zero_grad()
    loss.backward()
    q = _engine_decisions(tr.model)
    tr.optimizer.step()
    torch.cuda.synchronize()
    osd = O.clone_state(sd0, requires_grad=True)
    x, px, y = (batch[k].view(2, 1, 32, 32) for k in ("image", "prev_image", "mask"))
    with O.forced_decisions(q):
        _, ol = O.siam_forward(osd, x, px, mode=mode, training=True)
    oloss = O.siam_bce_dice_loss(ol, y, 1.0, 1.0)
    opt = torch.optim.Adam([v for v in osd.values() if v.requires_grad], lr=1e-3)
    oloss.backward()
    opt.step()
    assert abs(float(loss) - float(oloss)) < 1e-4
    _adam_update_matches(tr.model.state_dict(), osd, sd0)
    # BatchNorm buffers of the weight-shared encoder were updated twice (x, then prev_x)
    assert int(tr.model.state_dict()["encode1.1.num_batches_tracked"]) == 2


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


def _randomise_bn(sd):
    for k in sd:
        if k.endswith("running_mean"):
            sd[k] = torch.randn_like(sd[k]) * 0.1
        if k.endswith("running_var"):
            sd[k] = torch.rand_like(sd[k]) + 0.5
    return sd


def test_predict3d_matches_oracle_tiling(tmp_path):
    """unet3d.Predict: whole-volume normalisation, linspace patches, three-layer float16 stitch buffer (n % 3)."""
    torch.manual_seed(3)
    sd = _randomise_bn(O.init_unet3d(1, 1, 4, seed=6))
    ck = {"n_filter": 4, "in_channels": 1, "out_channels": 1, "state_dict": sd}
    vol = (np.random.RandomState(1).rand(20, 40, 36) * 500).astype("float32")
    rd = (8, 16, 16)
    p = unet3d.Predict(vol.copy(), None, ck, resize_dim=rd, add_patch=0, progress_bar=False, device="cuda")
    got = p.vol_result
    assert got.shape == vol.shape and got.dtype == np.uint8
    v = np.clip(vol, np.nanpercentile(vol, 0.), np.percentile(vol, 99.8))
    v = v - v.min()
    v = v / v.max() * 255
    from bio_image_unet_amd.workflow import tile_starts
    zs, xs, ys = tile_starts(20, 8, 3), tile_starts(40, 16, 3), tile_starts(36, 16, 3)
    buf = np.full((3, 20, 40, 36), np.nan, dtype="float16")
    n = 0
    for z in zs:
        for x in xs:
            for y in ys:
                patch = v[z:z + 8, x:x + 16, y:y + 16].astype("uint8").astype("float32") / 255
                with torch.no_grad():
                    prob, _ = O.unet3d_forward(sd, torch.from_numpy(patch)[None, None], training=False)
                buf[n % 3, z:z + 8, x:x + 16, y:y + 16] = (prob[0, 0].numpy() * 255).astype("uint8")
                n += 1
    want = np.nanmean(buf, axis=0).astype("uint8")
    diff = np.abs(got.astype(int) - want.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 0.02


def test_predict_siam_matches_oracle(tmp_path):
    torch.manual_seed(4)
    sd = _randomise_bn(O.init_unet2d(1, 1, 4, seed=7, siam_mode="max"))
    ck = {"n_filter": 4, "mode": "max", "state_dict": sd}
    movie = (np.random.RandomState(2).rand(3, 40, 48) * 300).astype("float32")
    p = siam.Predict(movie.copy(), None, ck, resize_dim=(32, 32), add_tile=0, show_progress=False, device="cuda")
    got = p.imgs_result
    assert got.shape == movie.shape and got.dtype == np.uint8
    from bio_image_unet_amd.workflow import normalise_stack, tile_starts
    xs, ys = tile_starts(40, 32, 2), tile_starts(48, 32, 2)
    for i in range(3):
        prev = movie[1] if i == 0 else movie[i - 1]
        pair = normalise_stack(np.array([prev, movie[i]], dtype=np.float64), "single", (0., 99.8), False).astype("uint8")
        stack = np.full((4, 40, 48), np.nan)
        k = 0
        for a in xs:
            for b in ys:
                cur_t = torch.from_numpy(pair[1][a:a + 32, b:b + 32].astype("float32") / 255)[None, None]
                prv_t = torch.from_numpy(pair[0][a:a + 32, b:b + 32].astype("float32") / 255)[None, None]
                with torch.no_grad():
                    prob, _ = O.siam_forward(sd, cur_t, prv_t, mode="max", training=False)
                stack[k, a:a + 32, b:b + 32] = (prob[0, 0].numpy() * 255).astype("uint8")
                k += 1
        want = np.nanmean(stack, axis=0).astype("uint8")
        diff = np.abs(got[i].astype(int) - want.astype(int))
        assert diff.max() <= 1 and (diff > 0).mean() < 0.02


HEADS = {"mask": {"channels": 1, "activation": "sigmoid", "loss": "BCEDiceLoss", "weight": 1.0},
         "flow": {"channels": 2, "activation": "tanh", "loss": "DiceLoss", "weight": 0.5}}


def test_mo3d_trainer_and_predict(tmp_path):
    import bio_image_unet_amd.multi_output_unet3d as mo
    torch.manual_seed(5)

    class Vols(torch.utils.data.Dataset):
        aug_factor = 1
        dim_out = (8, 16, 16)

        def __init__(self):
            g = torch.Generator().manual_seed(0)
            self.items = [{"volume": torch.rand(8, 16, 16, generator=g), "mask": (torch.rand(1, 8, 16, 16, generator=g) > 0.5).float(),
                           "flow": (torch.rand(2, 8, 16, 16, generator=g) > 0.5).float()} for _ in range(12)]

        def __len__(self):
            return len(self.items)

        def __getitem__(self, i):
            return self.items[i]

    tr = mo.Trainer(Vols(), HEADS, 1, use_interpolation=True, batch_size=2, n_filter=4, save_dir=str(tmp_path / "m"), device="cuda")
    # one optimisation step against the oracle: same loss expression, gradients clipped to norm 1, Adam
    sd0 = {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()}
    batch = next(iter(tr.train_loader))
    loss = tr._total_loss(batch, validating=False)
    osd = O.clone_state(sd0, requires_grad=True)
    pred = O.mo3d_forward(osd, batch["volume"].unsqueeze(1), HEADS, use_interpolation=True, training=True)
    want = 1.0 * O.bce_dice_loss(pred["mask"], batch["mask"], 1, 1) + 0.5 * O.bce_dice_loss(pred["flow"], batch["flow"], 0, 1)
    assert abs(float(loss) - float(want)) < 1e-3 * max(1.0, abs(float(want)))
    # ... and the rest of multi_output_unet3d/train.py:198-202: zero_grad, backward, clip_grad_norm_(1.0), Adam -- the clipped
    # gradient's norm and the parameter update against the oracle stepped on the engine's branch of the discrete decisions
    tr.optimizer.zero_grad()
    loss.backward()
    q = _engine_decisions(tr.model)
    norm = tr.optimizer.clip_grad_norm_(1.0)             # (what Trainer.iterate calls: biu_grad_clip)
    clipped = {k: p.grad.detach().cpu().clone() for k, p in tr.model.named_parameters()}
    tr.optimizer.step()
    torch.cuda.synchronize()
    with O.forced_decisions(q):
        pred = O.mo3d_forward(osd, batch["volume"].unsqueeze(1), HEADS, use_interpolation=True, training=True)
    oloss = O.trainer_mo3d_loss(pred, {k: batch[k] for k in HEADS}, HEADS)
    assert abs(float(oloss) - float(want)) < 1e-5 * max(1.0, abs(float(want)))          # decisions replayed: same value
    og = O.grads_of(oloss, osd)
    onorm = torch.sqrt(sum((v.double() ** 2).sum() for v in og.values()))
    assert abs(float(norm) - float(onorm)) < 1e-3 * float(onorm), (float(norm), float(onorm))
    scale = min(1.0, 1.0 / (float(onorm) + 1e-6))                                          # clip_grad_norm_'s coefficient
    gmax = max(float(v.abs().max()) for v in og.values()) * scale
    for k, v in og.items():
        if not (k.endswith(".0.bias") and not k.startswith("output_layers")):              # (conv bias in front of BatchNorm: true gradient 0)
            assert float((clipped[k] - v * scale).abs().max()) <= 1e-3 * float(v.abs().max()) * scale + 1e-5 * gmax, f"clipped grad {k}"
    for k, v in osd.items():
        if v.requires_grad:
            v.grad = og[k].clone()
    torch.nn.utils.clip_grad_norm_([v for v in osd.values() if v.requires_grad], max_norm=1.0)
    torch.optim.Adam([v for v in osd.values() if v.requires_grad], lr=1e-3).step()
    _adam_update_matches(tr.model.state_dict(), osd, sd0)
    tr.start()
    ck = torch.load(str(tmp_path / "m" / "model.pt"), weights_only=False)
    assert ck["output_heads"] == HEADS and ck["use_interpolation"] is True and len(ck["state_dict"]) > 100
    # prediction: overlapping patches, float outputs, weighted blend
    vol = np.random.RandomState(3).rand(12, 40, 24).astype("float32") * 100
    ck["state_dict"] = _randomise_bn({k: v.cpu() for k, v in ck["state_dict"].items()})
    p = mo.Predict(vol.copy(), ck, result_path=None, max_patch_size=(8, 16, 16), overlap_factor=0.25, batch_size=3,
                   show_progress=False, device="cuda")
    assert set(p.result) == {"mask", "flow"} and p.result["mask"].shape == (12, 40, 24) and p.result["flow"].shape == (2, 12, 40, 24)
    # where a single patch with weight 1 covers a voxel the blend returns that patch's value: check one interior patch centre
    c = np.clip(vol, np.percentile(vol, 0.), np.percentile(vol, 99.98))
    c = (c - c.min()) / (np.ptp(c) + 1e-8)
    z0, y0, x0 = p.Z_start[0], p.Y_start[0], p.X_start[0]
    with torch.no_grad():
        out = O.mo3d_forward(ck["state_dict"], torch.from_numpy(c[z0:z0 + 8, y0:y0 + 16, x0:x0 + 16])[None, None], HEADS,
                             use_interpolation=True, training=False)
    # voxel (2, 3, 3) lies only in the first patch along every axis (strides 6, 12, 12)
    assert abs(float(out["mask"][0, 0, 2, 3, 3]) - float(p.result["mask"][2, 3, 3])) < 2e-3
    assert np.isfinite(p.result["flow"]).all() and float(np.abs(p.result["flow"]).max()) <= 1.0 + 1e-5


# ---------------------------------------------------------------------------------------------------------------
# tiling / quantisation / device stitching of the Predict counterparts with point-wise stub networks (the numpy restatements below are the checkers)
# ---------------------------------------------------------------------------------------------------------------
class _Stub2D(torch.nn.Module):
    """prob = a fixed smooth function of the input patch; accepts the reference constructor kwargs."""

    def __init__(self, **_):
        super().__init__()
        self.dummy = torch.nn.Parameter(torch.zeros(1))

    def forward(self, x, prev=None):
        f = lambda t: t.float() / 255 if t.dtype == torch.uint8 else t          # Predict hands uint8 patches over (scaled by the engine)
        x, prev = f(x), (f(prev) if prev is not None else None)
        p = 0.25 + 0.5 * x if prev is None else 0.2 + 0.3 * x + 0.3 * prev
        return p, p


class _StubHeads(torch.nn.Module):
    def __init__(self, in_channels=1, n_filter=4, output_heads=None, use_interpolation=True):
        super().__init__()
        self.heads = output_heads
        self.dummy = torch.nn.Parameter(torch.zeros(1))

    def forward(self, x):
        return {k: (0.1 * (i + 1) + 0.5 * x).repeat(1, v["channels"], 1, 1, 1) for i, (k, v) in enumerate(self.heads.items())}


def _nanmean_stitch(shape, tiles, starts, tile):
    import numpy as np
    stack = np.full((len(tiles),) + shape, np.nan)
    for k, (st, t) in enumerate(zip(starts, tiles)):
        sl = tuple(slice(s, s + e) for s, e in zip(st, tile))
        stack[(k,) + sl] = t
    return np.nanmean(stack, axis=0).astype("uint8")


def test_predict2d_tiling_and_stitch_stub(tmp_path):
    import numpy as np
    from bio_image_unet_amd.workflow import Predict2D, normalise_stack, tile_starts
    imgs = (np.random.RandomState(0).rand(2, 50, 70) * 900).astype("float32")
    net = _Stub2D()
    ck = {"n_filter": 4, "in_channels": 1, "out_channels": 1, "state_dict": net.state_dict()}
    p = Predict2D(imgs.copy(), None, ck, network=_Stub2D, resize_dim=(32, 32), add_tile=1, show_progress=False, device="cuda")
    norm = normalise_stack(imgs.astype("float64"), "single", (0., 99.8), False)
    xs, ys = tile_starts(50, 32, 3), tile_starts(70, 32, 4)
    for i in range(2):
        tiles, starts = [], []
        for a in xs:
            for b in ys:
                patch = norm[i, a:a + 32, b:b + 32].astype("uint8").astype("float32") / 255
                tiles.append(((0.25 + 0.5 * patch) * 255).astype("uint8"))
                starts.append((a, b))
        want = _nanmean_stitch((50, 70), tiles, starts, (32, 32))
        assert np.abs(p.imgs_result[i].astype(int) - want.astype(int)).max() <= 1


def test_predict3d_three_layer_stitch_stub():
    import numpy as np
    from bio_image_unet_amd.workflow import Predict3D, tile_starts
    vol = (np.random.RandomState(1).rand(10, 40, 36) * 500).astype("float32")

    class Net(_Stub2D):
        def __init__(self, n_filter=4, in_channels=1, out_channels=1, use_interpolation=False):
            super().__init__()

    ck = {"n_filter": 4, "in_channels": 1, "out_channels": 1, "state_dict": Net().state_dict()}
    p = Predict3D(vol.copy(), None, ck, network=Net, resize_dim=(8, 16, 16), add_patch=0, progress_bar=False, device="cuda")
    v = np.clip(vol, np.nanpercentile(vol, 0.), np.percentile(vol, 99.8))
    v = v - v.min()
    v = v / v.max() * 255
    zs, xs, ys = tile_starts(10, 8, 2), tile_starts(40, 16, 3), tile_starts(36, 16, 3)
    buf = np.full((3, 10, 40, 36), np.nan, dtype="float16")
    n = 0
    for z in zs:
        for x in xs:
            for y in ys:
                patch = v[z:z + 8, x:x + 16, y:y + 16].astype("uint8").astype("float32") / 255
                buf[n % 3, z:z + 8, x:x + 16, y:y + 16] = ((0.25 + 0.5 * patch) * 255).astype("uint8")
                n += 1
    want = np.nanmean(buf, axis=0).astype("uint8")
    assert p.N == 18 and np.abs(p.vol_result.astype(int) - want.astype(int)).max() <= 1


def test_predict_siam_pairs_stub(monkeypatch):
    import numpy as np
    import bio_image_unet_amd.workflow as W
    monkeypatch.setattr(W, "Siam_UNet", lambda n_filter, mode: _Stub2D())
    movie = (np.random.RandomState(2).rand(3, 20, 24) * 300).astype("float32")
    ck = {"n_filter": 4, "mode": "max", "state_dict": _Stub2D().state_dict()}
    p = W.PredictSiam(movie.copy(), None, ck, resize_dim=(32, 32), show_progress=False, device="cuda")     # tiles larger than frames: zero padding
    assert p.imgs_result.shape == movie.shape
    for i in range(3):
        prev = movie[1] if i == 0 else movie[i - 1]
        pair = W.normalise_stack(np.array([prev, movie[i]], dtype=np.float64), "single", (0., 99.8), False).astype("uint8")
        want = ((0.2 + 0.3 * pair[1].astype("float32") / 255 + 0.3 * pair[0].astype("float32") / 255) * 255).astype("uint8")
        assert np.abs(p.imgs_result[i].astype(int) - want.astype(int)).max() <= 1


def test_predict_mo3d_blend_stub():
    import numpy as np
    from bio_image_unet_amd.workflow import PredictMo3d
    heads = {"a": {"channels": 1, "activation": "sigmoid", "loss": "BCEDiceLoss"}, "b": {"channels": 2, "activation": None, "loss": "DiceLoss"}}
    vol = np.random.RandomState(3).rand(12, 40, 24).astype("float32") * 50
    ck = {"in_channels": 1, "n_filter": 4, "output_heads": heads, "use_interpolation": True, "state_dict": _StubHeads(output_heads=heads).state_dict()}
    p = PredictMo3d(vol.copy(), ck, network=_StubHeads, max_patch_size=(8, 16, 16), overlap_factor=0.25, batch_size=4, show_progress=False,
                    device="cuda")
    c = np.clip(vol, np.percentile(vol, 0.), np.percentile(vol, 99.98))
    c = (c - c.min()) / (np.ptp(c) + 1e-8)
    # the stub is point-wise, so every patch predicts the same value for a voxel and any convex blend returns it
    assert p.Z_start == [0, 4] and p.Y_start == [0, 12, 24] and p.X_start == [0, 8]
    np.testing.assert_allclose(p.result["a"], 0.1 + 0.5 * c, rtol=1e-5, atol=1e-6)
    assert p.result["b"].shape == (2, 12, 40, 24)
    np.testing.assert_allclose(p.result["b"][1], 0.2 + 0.5 * c, rtol=1e-5, atol=1e-6)




@pytest.mark.timeout(180)
def test_tile_store_feeder_and_trainer(tmp_path):
    """feed.TileStore + DeviceFeeder: uint8 batches arrive on the device unchanged and in the reference loader's order; a Trainer
    fed from the store computes the same first-step loss as the one fed from the float data set (the 1/255 scaling rides in the
    input-layout kernel, masks are widened on the device)."""
    from bio_image_unet_amd.feed import DeviceFeeder, TileStore
    torch.manual_seed(3)

    class U8Tiles(Tiles):
        def __init__(self, n, dim, keys, seed=0):
            super().__init__(n, dim, keys, seed)
            for it in self.items:                       # the reference's tiles are uint8 / 255
                for k in it:
                    it[k] = torch.round(it[k] * 255) / 255
    ds = U8Tiles(12, (32, 32), ["image"])
    st = TileStore.from_dataset(str(tmp_path / "tiles"), ds)
    fd = DeviceFeeder(st, list(range(12)), 4, "cuda", depth=2)
    assert len(fd) == 3
    for epoch in range(2):
        seen = []
        for b, batch in enumerate(fd):
            assert batch["image"].dtype == torch.uint8 and batch["image"].is_cuda and batch["image"].shape == (4, 32, 32)
            want = torch.stack([torch.round(ds[i]["image"] * 255) for i in range(4 * b, 4 * b + 4)]).to(torch.uint8)
            assert torch.equal(batch["image"].cpu(), want)
            seen.append(b)
        assert seen == [0, 1, 2]
    torch.manual_seed(5)
    tr_a = unet.Trainer(ds, 1, batch_size=2, n_filter=4, save_dir=str(tmp_path / "a"), device="cuda")
    torch.manual_seed(5)
    tr_b = unet.Trainer(st, 1, batch_size=2, n_filter=4, save_dir=str(tmp_path / "b"), device="cuda")
    assert isinstance(tr_b.train_loader, DeviceFeeder) and tr_b.train_loader.indices == list(tr_a.train_loader.dataset.indices)
    la = tr_a._forward_loss(next(iter(tr_a.train_loader)), validating=False)
    lb = tr_b._forward_loss(next(iter(tr_b.train_loader)), validating=False)
    assert abs(float(la) - float(lb)) < 1e-6
    tr_b.start()                                         # a whole epoch + validation + checkpoint through the feeder
    assert torch.load(str(tmp_path / "b" / "model.pt"), weights_only=False)["n_filter"] == 4
