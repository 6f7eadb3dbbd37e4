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
