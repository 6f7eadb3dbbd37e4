"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol of include/biu.h, the model classes carry
the reference's state_dict schema, the product refuses CPU tensors loudly, losses equal the oracle's, and the gradient
averager works across two gloo ranks."""
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import bio_image_unet_amd._lib as L
    hdr = open(os.path.join(ROOT, "include", "biu.h")).read()
    declared = set(re.findall(r"\b(biu_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"biu_stream"}
    assert declared, "no declarations parsed"
    missing = [s for s in sorted(declared) if not hasattr(L.lib._c, s)]
    assert not missing, f"declared in biu.h but not exported: {missing}"
    unbound = [s for s in sorted(declared) if s not in L.SIGNATURES]
    assert not unbound, f"declared in biu.h but not bound in _lib.SIGNATURES: {unbound}"
    assert L.lib.biu_version() >= 100


@pytest.mark.parametrize("case", ["unet2d_f4", "unet2d_f4_o2_dil2", "unet3d_f4", "siam_f4_concat", "siam_f4_max", "mo3d_f4_interp", "mo3d_f4_convT"])
def test_state_dict_schema_and_checkpoint_loading(case):
    import bio_image_unet_amd as B
    from tests.golden_util import load_case
    g = load_case(case)
    cls = {"Unet": B.Unet, "UNet3D": B.UNet3D, "Siam_UNet": B.Siam_UNet, "MultiOutputUnet3D": B.MultiOutputUnet3D}[g["meta"]["model"]]
    m = cls(**g["meta"]["ctor"])
    sd = m.state_dict()
    assert list(sd.keys()) == list(g["sd"].keys())            # same keys in the same registration order
    for k in sd:
        assert tuple(sd[k].shape) == tuple(g["sd"][k].shape), k
    m.load_state_dict(g["sd"])                                # a reference checkpoint loads as is


def test_no_cpu_fallback():
    import bio_image_unet_amd as B
    with pytest.raises(RuntimeError, match="no CPU path"):
        B.Unet(1, 1, 4)(torch.rand(1, 1, 32, 32))
    with pytest.raises(RuntimeError, match="no CPU path"):
        B.UNet3D(1, 1, 4)(torch.rand(1, 1, 8, 8, 8))


def test_losses_match_oracle():
    from bio_image_unet_amd import losses as L
    from oracle import unet_oracle as O
    torch.manual_seed(0)
    x, t = torch.randn(3, 1, 16, 16), (torch.rand(3, 1, 16, 16) > 0.5).float()
    torch.testing.assert_close(L.BCEDiceLoss(0.3, 0.7)(x, t), O.bce_dice_loss(x, t, 0.3, 0.7))
    torch.testing.assert_close(L.TverskyLoss(0.4, 0.6)(x, t), O.tversky_loss(x, t, 0.4, 0.6))
    torch.testing.assert_close(L.logcoshTverskyLoss(0.4, 0.6)(x, t), O.logcosh_tversky_loss(x, t, 0.4, 0.6))


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from bio_image_unet_amd import ddp
rank, local, world = ddp.init_from_env("gloo")
torch.manual_seed(100 + rank)                      # different init per rank: broadcast must equalise
m = torch.nn.Sequential(torch.nn.Conv2d(1, 4, 3), torch.nn.BatchNorm2d(4), torch.nn.Conv2d(4, 2, 1))
avg = ddp.GradAverager(m)
ref = [p.detach().clone() for p in m.parameters()]
gathered = [torch.zeros_like(ref[0]) for _ in range(world)]
dist.all_gather(gathered, ref[0])
assert all(torch.equal(gathered[0], g) for g in gathered), "parameters not broadcast"
x = torch.full((2, 1, 8, 8), float(rank + 1))
m(x).sum().backward()
local_g = [p.grad.clone() for p in m.parameters()]
avg.average()
for p, lg in zip(m.parameters(), local_g):
    both = [torch.zeros_like(lg) for _ in range(world)]
    dist.all_gather(both, lg)
    torch.testing.assert_close(p.grad, sum(both) / world)
dist.barrier()
print("OK", rank)
'''


def test_gradient_averager_two_gloo_ranks(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert all("OK" in o for o in outs)


# ---------------------------------------------------------------------------------------------------------------
# host logic of the Predict counterparts (tiling, quantisation, stitching) with a stub network on the CPU
# ---------------------------------------------------------------------------------------------------------------
class _Stub2D(torch.nn.Module):
    """prob = a fixed smooth function of the input patch; accepts the reference constructor kwargs."""

    def __init__(self, **_):
        super().__init__()
        self.dummy = torch.nn.Parameter(torch.zeros(1))

    def forward(self, x, prev=None):
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


def test_predict2d_tiling_and_stitch_cpu(tmp_path):
    import numpy as np
    from bio_image_unet_amd.workflow import Predict2D, normalise_stack, tile_starts
    imgs = (np.random.RandomState(0).rand(2, 50, 70) * 900).astype("float32")
    net = _Stub2D()
    ck = {"n_filter": 4, "in_channels": 1, "out_channels": 1, "state_dict": net.state_dict()}
    p = Predict2D(imgs.copy(), None, ck, network=_Stub2D, resize_dim=(32, 32), add_tile=1, show_progress=False, device="cpu")
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


def test_predict3d_three_layer_stitch_cpu():
    import numpy as np
    from bio_image_unet_amd.workflow import Predict3D, tile_starts
    vol = (np.random.RandomState(1).rand(10, 40, 36) * 500).astype("float32")

    class Net(_Stub2D):
        def __init__(self, n_filter=4, in_channels=1, out_channels=1, use_interpolation=False):
            super().__init__()

    ck = {"n_filter": 4, "in_channels": 1, "out_channels": 1, "state_dict": Net().state_dict()}
    p = Predict3D(vol.copy(), None, ck, network=Net, resize_dim=(8, 16, 16), add_patch=0, progress_bar=False, device="cpu")
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


def test_predict_siam_pairs_cpu(monkeypatch):
    import numpy as np
    import bio_image_unet_amd.workflow as W
    monkeypatch.setattr(W, "Siam_UNet", lambda n_filter, mode: _Stub2D())
    movie = (np.random.RandomState(2).rand(3, 20, 24) * 300).astype("float32")
    ck = {"n_filter": 4, "mode": "max", "state_dict": _Stub2D().state_dict()}
    p = W.PredictSiam(movie.copy(), None, ck, resize_dim=(32, 32), show_progress=False, device="cpu")     # tiles larger than frames: zero padding
    assert p.imgs_result.shape == movie.shape
    for i in range(3):
        prev = movie[1] if i == 0 else movie[i - 1]
        pair = W.normalise_stack(np.array([prev, movie[i]], dtype=np.float64), "single", (0., 99.8), False).astype("uint8")
        want = ((0.2 + 0.3 * pair[1].astype("float32") / 255 + 0.3 * pair[0].astype("float32") / 255) * 255).astype("uint8")
        assert np.abs(p.imgs_result[i].astype(int) - want.astype(int)).max() <= 1


def test_predict_mo3d_blend_cpu():
    import numpy as np
    from bio_image_unet_amd.workflow import PredictMo3d
    heads = {"a": {"channels": 1, "activation": "sigmoid", "loss": "BCEDiceLoss"}, "b": {"channels": 2, "activation": None, "loss": "DiceLoss"}}
    vol = np.random.RandomState(3).rand(12, 40, 24).astype("float32") * 50
    ck = {"in_channels": 1, "n_filter": 4, "output_heads": heads, "use_interpolation": True, "state_dict": _StubHeads(output_heads=heads).state_dict()}
    p = PredictMo3d(vol.copy(), ck, network=_StubHeads, max_patch_size=(8, 16, 16), overlap_factor=0.25, batch_size=4, show_progress=False,
                    device="cpu")
    c = np.clip(vol, np.percentile(vol, 0.), np.percentile(vol, 99.98))
    c = (c - c.min()) / (np.ptp(c) + 1e-8)
    # the stub is point-wise, so every patch predicts the same value for a voxel and any convex blend returns it
    assert p.Z_start == [0, 4] and p.Y_start == [0, 12, 24] and p.X_start == [0, 8]
    np.testing.assert_allclose(p.result["a"], 0.1 + 0.5 * c, rtol=1e-5, atol=1e-6)
    assert p.result["b"].shape == (2, 12, 40, 24)
    np.testing.assert_allclose(p.result["b"][1], 0.2 + 0.5 * c, rtol=1e-5, atol=1e-6)


def test_init_weights_kaiming_normal_on_conv2d_only():
    """utils/utils.py:76-78: kaiming_normal_(nonlinearity='leaky_relu') (a = 0 => gain sqrt(2), fan_in) on nn.Conv2d weights
    and nothing else: biases, Conv3d, ConvTranspose and BatchNorm keep PyTorch's defaults (Trainer.apply(init_weights))."""
    import math
    from torch import nn
    from bio_image_unet_amd.utils import init_weights
    torch.manual_seed(0)
    mods = dict(c2=nn.Conv2d(64, 96, 3), c3=nn.Conv3d(8, 8, 3), t2=nn.ConvTranspose2d(8, 8, 2, 2), t3=nn.ConvTranspose3d(8, 8, 2, 2),
                bn=nn.BatchNorm2d(8), c1=nn.Conv1d(4, 4, 3), lin=nn.Linear(4, 4))
    before = {k: {n: p.detach().clone() for n, p in m.named_parameters()} for k, m in mods.items()}
    for m in mods.values():
        init_weights(m)
    w = mods["c2"].weight.detach()
    std = math.sqrt(2.0 / (64 * 9))
    assert abs(float(w.std()) / std - 1) < 0.02 and abs(float(w.mean())) < 0.02 * std
    kurt = float(((w - w.mean()) ** 4).mean() / w.var() ** 2)
    assert 2.8 < kurt < 3.2, f"normal, not uniform (kurtosis {kurt})"
    assert torch.equal(mods["c2"].bias, before["c2"]["bias"])
    for k in ("c3", "t2", "t3", "bn", "c1", "lin"):
        for n, p in mods[k].named_parameters():
            assert torch.equal(p, before[k][n]), f"{k}.{n} was touched"
    # through the module tree, as the trainers do: a 2-D net gets new conv weights (its 1x1 head included), a 3-D net none
    import bio_image_unet_amd as B
    torch.manual_seed(1)
    m2, m3 = B.Unet(1, 1, 4), B.UNet3D(1, 1, 4)
    b2 = {k: v.clone() for k, v in m2.state_dict().items()}
    b3 = {k: v.clone() for k, v in m3.state_dict().items()}
    m2.apply(init_weights)
    m3.apply(init_weights)
    changed = {k for k, v in m2.state_dict().items() if not torch.equal(v, b2[k])}
    assert changed == {k for k in b2 if k.endswith(".0.weight")}, changed          # encode*/middle*/decode* convs and final.0
    assert all(torch.equal(v, b3[k]) for k, v in m3.state_dict().items())


def test_get_device_rule(monkeypatch, capsys):
    """utils/utils.py:56-73: cuda:0 whenever torch was BUILT with CUDA/ROCm (even with no GPU visible), else mps, else cpu
    with a warning."""
    from bio_image_unet_amd.utils import get_device
    monkeypatch.setattr(torch.backends.cuda, "is_built", lambda: True)
    assert get_device() == torch.device("cuda:0")
    get_device(print_device=True)
    assert "Using device: cuda:0" in capsys.readouterr().out
    monkeypatch.setattr(torch.backends.cuda, "is_built", lambda: False)
    monkeypatch.setattr(torch.backends.mps, "is_built", lambda: True)
    assert get_device() == torch.device("mps")
    monkeypatch.setattr(torch.backends.mps, "is_built", lambda: False)
    assert get_device() == torch.device("cpu")
    assert "Warning" in capsys.readouterr().out


def test_engine_cache_is_not_copied_or_pickled():
    import copy, io
    import bio_image_unet_amd as B
    m = B.Unet(1, 1, 4)
    m._engines[("fake",)] = [object()]
    assert len(copy.deepcopy(m)._engines) == 0
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    assert len(torch.load(buf, weights_only=False)._engines) == 0
