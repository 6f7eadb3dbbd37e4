"""Data parallelism on the engine itself: two ranks share the one GPU of the test box (gloo carries the collectives;
on the 8-GPU node the same code runs with backend 'nccl' = RCCL).  Per-rank BatchNorm is the reference semantics, so the averaged
gradient must equal the mean of the two ranks' own gradients -- bucket by bucket, with every bucket's all-reduce issued from
INSIDE the backward pass (the engine's grad-ready hook)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import bio_image_unet_amd as B
from bio_image_unet_amd import ddp
from oracle import unet_oracle as O
rank, local, world = ddp.init_from_env("gloo")
torch.cuda.set_device(0)
torch.manual_seed(100 + rank)                          # different init per rank: the broadcast must equalise
m = B.UNet3D(1, 1, 16).cuda()
m.train()
avg = ddp.GradAverager(m, bucket_mb=0.25)
assert len(avg.buckets) >= 3 and avg.hooked
w0 = m.decode5[0].weight.detach().clone()
both = [torch.zeros_like(w0) for _ in range(world)]
dist.all_gather(both, w0)
assert torch.equal(both[0], both[1]), "parameters not broadcast"
g = torch.Generator().manual_seed(7 + rank)           # every rank its own half of the global batch
x = torch.rand(2, 1, 16, 32, 32, generator=g).cuda()
y = (torch.rand(2, 1, 16, 32, 32, generator=g) > 0.5).float().cuda()
# the rank's OWN gradients: p.grad aliases the bucket a gradient was computed into, and a complete bucket is on the wire (being summed in
# place) before backward returns -- so they are cloned inside the grad-ready hook, in front of the averager's
names = {p: k for k, p in m.named_parameters()}
spy = {}
def _spy(p, g, _orig=avg._on_grad_ready):
    spy[names[p]] = spy[names[p]] + g if names[p] in spy else g.clone()
    _orig(p, g)
m.register_grad_ready_hook(_spy)
for step in range(2):
    m.zero_grad(set_to_none=True)
    spy.clear()
    _, logits = m(x)
    # an engine that had packed its MFMA weights before the broadcast would still be multiplying rank-local weights here
    loss = O.bce_dice_loss(logits, y)
    loss.backward()
    local_g = dict(spy)
    avg.average()
    assert avg.launched_in_backward == len(avg.buckets), (avg.launched_in_backward, len(avg.buckets))
    # the engine computed every weight / BatchNorm gradient straight into its bucket slot: only the all-zero conv biases in front of a
    # train-mode BatchNorm (views of one zero buffer) were copied in
    n_bias = sum(1 for k, _ in m.named_parameters() if k.endswith(".0.bias") and not k.startswith("final"))
    assert avg.copies_in_backward <= n_bias, (avg.copies_in_backward, n_bias)
    assert all(p.grad.data_ptr() == avg._where[p][0].views[avg._where[p][1]].data_ptr() for p in m.parameters())
    for k, p in m.named_parameters():
        parts = [torch.zeros_like(local_g[k]) for _ in range(world)]
        dist.all_gather(parts, local_g[k])
        torch.testing.assert_close(p.grad, sum(parts) / world, rtol=1e-6, atol=1e-12)
        if k == "encode2.0.weight":
            assert not torch.equal(parts[0], parts[1]), "the two ranks must have seen different data"
# gradient accumulation on the engine: two backwards before one average() (each forward/backward pair on its own turn of the engine)
m.zero_grad(set_to_none=True)
spy.clear()
for xx in (x, x.flip(0)):
    O.bce_dice_loss(m(xx)[1], y).backward()
local_g = dict(spy)                                    # (the hook saw both backwards: their sum)
avg.average()
for k, p in m.named_parameters():
    parts = [torch.zeros_like(local_g[k]) for _ in range(world)]
    dist.all_gather(parts, local_g[k])
    want = sum(parts) / world
    # (mean(g1) + mean(g2) against mean(g1 + g2): two fp32 summation orders)
    torch.testing.assert_close(p.grad, want, rtol=1e-5, atol=1e-6 * float(want.abs().max()) + 1e-12)
# the multi-head step of bench.py (cfg5): average, THEN clip (multi_output_unet3d/train.py:201 on the global-batch gradient)
m.zero_grad(set_to_none=True)
spy.clear()
O.bce_dice_loss(m(x)[1], y).backward()
local_g = dict(spy)
avg.average()
from bio_image_unet_amd.optim import Adam
Adam(m.parameters(), lr=1e-3).clip_grad_norm_(1e-3)        # (biu_grad_clip on the bucket views average() left in p.grad: what bench.py's step does)
mean_g = {}
for k in local_g:
    parts = [torch.zeros_like(local_g[k]) for _ in range(world)]
    dist.all_gather(parts, local_g[k])
    mean_g[k] = sum(parts) / world
norm = torch.sqrt(sum((v.double() ** 2).sum() for v in mean_g.values()))
assert float(norm) > 1e-3
for k, p in m.named_parameters():
    torch.testing.assert_close(p.grad, (mean_g[k] * (1e-3 / (norm + 1e-6))).float(), rtol=1e-5, atol=1e-12)
# BatchNorm running statistics stay rank-local (plain nn.BatchNorm in the reference: no SyncBN)
rm = m.encode1[1].running_mean.detach().clone()
parts = [torch.zeros_like(rm) for _ in range(world)]
dist.all_gather(parts, rm)
assert not torch.equal(parts[0], parts[1])
dist.barrier()
print("OK", rank)
'''


@pytest.mark.timeout(300)
def test_two_ranks_on_the_engine_overlapped_buckets(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r), LOCAL_RANK="0"),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert all("OK" in o for o in outs)


@pytest.mark.timeout(400)
def test_bench_two_rank_path(tmp_path):
    """bench.py's N = 2 path exactly as the driver launches it (torch.distributed.run, one JSON line from rank 0), rehearsed on
    one GPU: both ranks on cuda:0, gloo instead of RCCL."""
    env = dict(os.environ, BIU_DDP_BACKEND="gloo", BIU_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29633", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "2", "--workload", "cfg1"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=360)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["config"]["parallelism"] == "dp2" and "cpu_baseline" not in r
    assert r["value"] > 0 and abs(r["value"] - 2 * 2 * 256 * 256 / (r["ms_per_step"] * 1e-3)) < 1e-3 * r["value"]
    assert r["ddp"]["buckets"] >= 1 and r["ddp"]["launched_in_backward"] == r["ddp"]["buckets"]
    assert r["ddp"]["ranks_seen"] == 2
    # per-rank step time and the host time spent inside average(): what a measured scaling curve is attributed with
    assert len(r["ddp"]["per_rank"]["rows"]) == 2 and all(row[0] > 0 for row in r["ddp"]["per_rank"]["rows"])
    assert r["ddp"]["step_ms_min_max_over_ranks"][0] <= r["ms_per_step"] * 1.0001


@pytest.mark.timeout(400)
def test_bench_self_launches_its_ranks(tmp_path):
    """Plain `python bench.py --gpus 2` with no launcher and no WORLD_SIZE: the script starts its own two ranks (before it
    touches the GPU), one JSON line comes out, and the line shows that two ranks took part."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(BIU_DDP_BACKEND="gloo", BIU_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--workload", "cfg1"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=360)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["parallelism"] == "dp2" and r["ddp"]["ranks_seen"] == 2
    assert r["median_ms_per_step"] > 0 and r["steps"] == 3
    # ranks that fail take the job down with a non-zero exit code and no JSON line (here: an unknown collective backend)
    bad = subprocess.run(cmd, env=dict(env, BIU_DDP_BACKEND="no-such-backend"), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=240)
    assert bad.returncode != 0 and not [l for l in bad.stdout.splitlines() if l.startswith("{")]


_NCCL_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import bio_image_unet_amd as B
from bio_image_unet_amd import ddp
from oracle import unet_oracle as O
rank, local, world = ddp.init_from_env("nccl")
torch.cuda.set_device(local)
assert dist.get_backend() == "nccl"
torch.manual_seed(100 + rank)
m = B.UNet3D(1, 1, 16).cuda()
m.train()
avg = ddp.GradAverager(m, bucket_mb=0.25)
g = torch.Generator().manual_seed(7 + rank)
x = torch.rand(2, 1, 16, 32, 32, generator=g).cuda()
y = (torch.rand(2, 1, 16, 32, 32, generator=g) > 0.5).float().cuda()
for step in range(2):
    m.zero_grad(set_to_none=True)
    O.bce_dice_loss(m(x)[1], y).backward()
    torch.cuda.synchronize()
    local_g = {k: p.grad.clone() for k, p in m.named_parameters()}      # (buckets launched in backward already hold sums on their way: compare against a 1-rank recomputation instead)
    avg.average()
    torch.cuda.synchronize()
    assert avg.launched_in_backward == len(avg.buckets)
# 1-rank reference of BOTH halves on this rank: same (broadcast) weights, rank r's data -> the mean of the two gradients
ref = B.UNet3D(1, 1, 16).cuda()
ref.load_state_dict({k: v for k, v in m.state_dict().items()})
ref.train()
tot = None
for r in range(world):
    gg = torch.Generator().manual_seed(7 + r)
    xr = torch.rand(2, 1, 16, 32, 32, generator=gg).cuda()
    yr = (torch.rand(2, 1, 16, 32, 32, generator=gg) > 0.5).float().cuda()
    ref.zero_grad(set_to_none=True)
    # (BatchNorm buffers of m moved during its two steps; gradients do not depend on the running statistics in train mode)
    O.bce_dice_loss(ref(xr)[1], yr).backward()
    gr = {k: p.grad.clone() for k, p in ref.named_parameters()}
    tot = gr if tot is None else {k: tot[k] + gr[k] for k in gr}
for k, p in m.named_parameters():
    want = tot[k] / world
    torch.testing.assert_close(p.grad, want, rtol=2e-3, atol=1e-5 * float(want.abs().max()) + 1e-9, msg=lambda s: f"{k}: {s}")
dist.barrier()
print("OK", rank)
'''


@pytest.mark.timeout(400)
def test_two_ranks_rccl(tmp_path):
    """The same engine-level data parallelism over RCCL (backend 'nccl'), one rank per device: averaged gradients equal a 1-rank recomputation
    of both ranks' batches, every bucket leaves during backward.  Runs wherever two GPUs are visible (the one-GPU test box skips it)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL between processes on one device is not a supported configuration)")
    script = tmp_path / "w_nccl.py"
    script.write_text(_NCCL_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert all("OK" in o for o in outs)


@pytest.mark.timeout(400)
def test_bench_two_gpus_rccl():
    """`python bench.py --gpus 2` on two real devices: ranks_seen == 2 over backend nccl, per-rank timings present."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "BIU_DDP_BACKEND", "BIU_SINGLE_DEVICE")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "3", "--workload", "cfg1"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=360)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert r["ddp"]["ranks_seen"] == 2 and r["ddp"]["backend"] == "nccl" and len(r["ddp"]["per_rank"]["rows"]) == 2
