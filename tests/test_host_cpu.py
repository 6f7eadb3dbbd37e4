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
    # ... and the other way round: the library exports nothing with C linkage that the header does not declare (INTEGRATION.md section 2:
    # "exports exactly include/biu.h"; cross-translation-unit helpers carry hidden visibility)
    import subprocess
    so = os.path.join(ROOT, "bio_image_unet_amd", "libbiu_hip.so")
    nm = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in nm.splitlines() if ln.split()}
    extra = sorted(s for s in exported if s.startswith("biu_") and s not in declared)
    assert not extra, f"exported with C linkage but not declared in biu.h: {extra}"


def test_fp32_product_mode_switch_validates_its_argument():
    """include/biu.h: biu_set_fp32_products(0 | 1 | 2); anything else is refused with a message (no GPU involved)."""
    import bio_image_unet_amd as B
    import bio_image_unet_amd._lib as L
    assert L.lib.biu_set_fp32_products(2) == 0
    assert L.lib.biu_set_fp32_products(0) == 0
    assert L.lib.biu_set_fp32_products(7) != 0
    assert b"mode" in L.lib.biu_last_error()
    with pytest.raises(ValueError):
        B.set_fp32_products("tf32")
    B.set_fp32_products("exact")


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
class Net(torch.nn.Sequential):
    """Reports finished gradients from inside backward like the package's networks do (register_grad_ready_hook)."""
    def register_grad_ready_hook(self, fn):
        for p in self.parameters():
            p.register_hook(lambda g, p=p: (fn(p, g), None)[1])
m = (Net if os.environ.get("HOOKED") == "1" else torch.nn.Sequential)(torch.nn.Conv2d(1, 4, 3), torch.nn.BatchNorm2d(4), torch.nn.Conv2d(4, 2, 1))
avg = ddp.GradAverager(m, bucket_mb=float(os.environ.get("BUCKET_MB", "8")))
assert len(avg.buckets) == int(os.environ.get("EXPECT_BUCKETS", "1")), len(avg.buckets)
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
if os.environ.get("HOOKED") == "1":
    assert avg.launched_in_backward == len(avg.buckets), "every bucket must go out from inside backward"
# second step: bucket state must have been reset
m.zero_grad(set_to_none=True)
m(x * 2).sum().backward()
local_g = [p.grad.clone() for p in m.parameters()]
avg.average()
for p, lg in zip(m.parameters(), local_g):
    both = [torch.zeros_like(lg) for _ in range(world)]
    dist.all_gather(both, lg)
    torch.testing.assert_close(p.grad, sum(both) / world)
# gradient accumulation: two backwards before one average() -- the hooked path must not ship the first micro-batch alone
def mean_over_ranks(t):
    both = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(both, t)
    return sum(both) / world
m.zero_grad(set_to_none=True)
m(x).sum().backward()
m(x * 3).sum().backward()
local_g = [p.grad.clone() for p in m.parameters()]
avg.average()
for p, lg in zip(m.parameters(), local_g):
    torch.testing.assert_close(p.grad, mean_over_ranks(lg))
# p.grad kept across steps (zero_grad(set_to_none=False)): p.grad IS a view of the bucket now; autograd accumulates in place
m.zero_grad(set_to_none=False)
assert all(p.grad is not None and float(p.grad.abs().sum()) == 0.0 for p in m.parameters())
m(x * 0.5).sum().backward()
local_g = [p.grad.clone() for p in m.parameters()]
avg.average()
for p, lg in zip(m.parameters(), local_g):
    torch.testing.assert_close(p.grad, mean_over_ranks(lg))
# clipping: AFTER average() (GradAverager contract) == clip of the mean gradient, identical on every rank
m.zero_grad(set_to_none=True)
m(x * 2).sum().backward()
local_g = [p.grad.clone() for p in m.parameters()]
avg.average()
torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=0.01)
want = [mean_over_ranks(lg) for lg in local_g]
norm = torch.sqrt(sum((w.double() ** 2).sum() for w in want))
assert float(norm) > 0.01
for p, w in zip(m.parameters(), want):
    torch.testing.assert_close(p.grad, (w * (0.01 / (norm + 1e-6))).float())
    torch.testing.assert_close(p.grad, mean_over_ranks(p.grad))
dist.barrier()
print("OK", rank)
'''


@pytest.mark.parametrize("variant", ["one_bucket", "many_buckets", "overlapped"])
def test_gradient_averager_two_gloo_ranks(tmp_path, variant):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    port = {"one_bucket": "29617", "many_buckets": "29618", "overlapped": "29619"}[variant]
    extra = {"one_bucket": {}, "many_buckets": {"BUCKET_MB": "0.0001", "EXPECT_BUCKETS": "2"},
             "overlapped": {"BUCKET_MB": "0.0001", "EXPECT_BUCKETS": "2", "HOOKED": "1"}}[variant]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, WORLD_SIZE="2", **extra)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert all("OK" in o for o in outs)


# ---------------------------------------------------------------------------------------------------------------
# host arithmetic of the Predict counterparts (the tiled prediction itself runs on the GPU: tests/test_gpu_workflow.py)
# ---------------------------------------------------------------------------------------------------------------
def test_tile_origins_and_normalisation_cpu():
    import numpy as np
    from bio_image_unet_amd.workflow import normalise_stack, tile_starts
    assert tile_starts(50, 32, 3).tolist() == [0, 9, 18] and tile_starts(70, 32, 4).tolist() == [0, 12, 25, 38]       # linspace, uint16 truncation
    assert tile_starts(32, 32, 1).tolist() == [0]
    imgs = (np.random.RandomState(0).rand(2, 20, 30) * 900).astype("float64")
    n = normalise_stack(imgs.copy(), "single", (0., 99.8), False)
    for i in range(2):
        c = np.clip(imgs[i], np.nanpercentile(imgs[i], 0.), np.percentile(imgs[i], 99.8))
        c = c - c.min()
        np.testing.assert_allclose(n[i], c / c.max() * 255, rtol=1e-12)
    inv = normalise_stack(imgs.copy(), "single", (0., 99.8), True)
    np.testing.assert_allclose(inv, 255 - n, rtol=1e-12, atol=1e-9)
    with pytest.raises(ValueError):
        normalise_stack(imgs.copy(), "bogus", (0., 99.8), False)


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


def test_tile_store_round_trip_cpu(tmp_path):
    """The memory-mapped uint8 store keeps the reference's item contract (float32 in [0, 1], unet/data.py:253-266): a data set of
    uint8-derived tiles converts without loss, indices gather into caller-provided (pinned) buffers."""
    import numpy as np
    from bio_image_unet_amd.feed import TileStore

    class DS(torch.utils.data.Dataset):
        dim_out, aug_factor, clip_threshold = (12, 20), 3, (0.2, 99.8)

        def __init__(self):
            r = np.random.RandomState(0)
            self.img = r.randint(0, 256, (7, 12, 20)).astype(np.uint8)
            self.msk = (r.rand(7, 12, 20) > 0.5).astype(np.uint8) * 255

        def __len__(self):
            return 7

        def __getitem__(self, i):
            return {"image": torch.from_numpy(self.img[i].astype(np.float32) / 255), "mask": torch.from_numpy(self.msk[i].astype(np.float32) / 255)}

    ds = DS()
    st = TileStore.from_dataset(str(tmp_path / "tiles"), ds)
    st2 = TileStore(str(tmp_path / "tiles"))                 # re-open read-only
    assert len(st2) == 7 and st2.dim_out == (12, 20) and st2.aug_factor == 3 and st2.fields == {"image": (12, 20), "mask": (12, 20)}
    for i in (0, 3, 6):
        a, b = st2[i], ds[i]
        assert torch.equal(a["image"], b["image"]) and torch.equal(a["mask"], b["mask"]) and a["image"].dtype == torch.float32
    out = {k: torch.empty((4,) + shp, dtype=torch.uint8) for k, shp in st2.fields.items()}
    got = st2.batch_u8([5, 1, 1], out=out)
    assert np.array_equal(got["image"].numpy(), ds.img[[5, 1, 1]]) and got["mask"].shape[0] == 3
    with pytest.raises(ValueError):
        (tmp_path / "bad.json").write_text("{}")
        TileStore(str(tmp_path / "bad"))
