#!/usr/bin/env python
"""Benchmark of the U-Net hot path on MI355X:  voxels/s of a full train step (forward + loss + backward + Adam).

    python bench.py --gpus N --steps K --warmup W [--workload cfg4|cfg2|cfg3|cfg5|cfg1]

N > 1: one rank per GPU over RCCL, either launched by the driver as  python -m torch.distributed.run --nproc-per-node N ...
bench.py --gpus N ...  (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), or -- when WORLD_SIZE is NOT set -- by this
script itself: `python bench.py --gpus N` starts N child ranks before anything touches a GPU (the parent never does), waits for
them and exits non-zero if any of them failed; rank 0's JSON line is the only line on stdout.  Per-GPU work is fixed (weak
scaling), `value` is the whole-job aggregate; `ddp.ranks_seen` (an all-reduce of 1 over the job) and the RCCL version are in the line.

Workloads (BASELINE.json `configs`) -- synthetic data x ~ U[0,1), y = (U > 0.5), seed 1234:
  cfg4 (default, the north-star target): UNet3D(1,1,32), (4,1,128,128,128) per GPU, bf16 storage / fp32 accumulate
  cfg2: Unet(1,2,64), (16,1,512,512), fp32          cfg3: Siam_UNet(32,'max'), 2 x (16,1,512,512), bf16
  cfg5: MultiOutputUnet3D(1, 3 heads, 64, interp), (1,1,128,256,256), bf16      cfg1: Unet(1,1,32), (2,1,256,256), fp32
One JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the dominant kernel launch, timed live with HIP events inside the timed region
  cpu_baseline -- the CPU oracle (a restatement of the reference's PyTorch-CPU path) on a bounded sample, host cores
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK = 8.0e12            # B/s   (MI355X_MICROARCH.md: HBM3E 8 TB/s spec)
MFMA_PEAK = {"bf16": 2.5e15, "f32": 157.3e12}
# --fp32-products (fp32 workloads only; profiles/HISTORY.md 3.4): how the 2-D 3x3 kernels multiply fp32 tensors.  bf16x6 (the library's default): six
# bf16 MFMA terms per product, fp32-grade; exact: the fp32 MFMA; bf16x3 (opt-in): three terms, <= 2^-15 per product.  The ceiling of
# the split launches is the bf16 dense peak over their number of terms.

HEADS5 = {"seg": {"channels": 1, "activation": "sigmoid"}, "flow": {"channels": 2, "activation": None},
          "dist": {"channels": 1, "activation": "sigmoid"}}

WORKLOADS = {
    "cfg1": dict(model="Unet", ctor=dict(in_channels=1, out_channels=1, n_filter=32), shape=(2, 1, 256, 256), dtype="f32", out=1),
    "cfg2": dict(model="Unet", ctor=dict(in_channels=1, out_channels=2, n_filter=64), shape=(16, 1, 512, 512), dtype="f32", out=2),
    "cfg3": dict(model="Siam_UNet", ctor=dict(n_filter=32, mode="max"), shape=(16, 1, 512, 512), dtype="bf16", out=1),
    "cfg4": dict(model="UNet3D", ctor=dict(in_channels=1, out_channels=1, n_filter=32), shape=(4, 1, 128, 128, 128), dtype="bf16", out=1),
    "cfg5": dict(model="MultiOutputUnet3D", ctor=dict(in_channels=1, output_heads=HEADS5, n_filter=64, use_interpolation=True),
                 shape=(1, 1, 128, 256, 256), dtype="bf16", out=4),
}


def make_step(wl, device, graph=False):
    """Returns (model, step_fn, fwd_fn, voxels_per_step). The loss expressions are the reference trainers' own.
    graph=True: the same step captured once in a hipGraph and replayed (bio_image_unet_amd/graph.py; single process)."""
    import bio_image_unet_amd as B
    from bio_image_unet_amd.losses import BCEDiceLoss
    from bio_image_unet_amd.optim import Adam

    torch.manual_seed(1234)
    cls = getattr(B, wl["model"])
    model = cls(**wl["ctor"]).to(device)
    if wl["model"] == "Unet":       # Trainer applies init_weights (Kaiming normal on nn.Conv2d only), unet/train.py:70
        model.apply(lambda m: torch.nn.init.kaiming_normal_(m.weight, nonlinearity="leaky_relu") if isinstance(m, torch.nn.Conv2d) else None)
    if wl["dtype"] == "bf16":
        model.set_compute_dtype(torch.bfloat16)
    model.train()
    shape = wl["shape"]
    x = torch.rand(shape, device=device)
    px = torch.rand(shape, device=device) if wl["model"] == "Siam_UNet" else None
    crit = BCEDiceLoss(0.5, 0.5)
    head_crit = BCEDiceLoss(1, 1)          # multi_output_unet3d/train.py: output_heads[name]['loss'] = 'BCEDiceLoss' -> BCEDiceLoss(1, 1)
    opt = Adam(model.parameters(), lr=1e-3)
    if wl["model"] == "MultiOutputUnet3D":
        tgt = {k: (torch.rand((shape[0], v["channels"]) + tuple(shape[2:]), device=device) > 0.5).float() for k, v in HEADS5.items()}
    else:
        y = (torch.rand((shape[0], wl["out"]) + tuple(shape[2:]), device=device) > 0.5).float()

    def loss_of(outs):
        if wl["model"] == "Unet":            # unet/train.py:133-134 (indexes the batch axis with the channel index)
            oc = wl["out"]
            return sum(crit(outs[1][ch], y[ch]) for ch in range(oc)) / oc
        if wl["model"] == "UNet3D":          # unet3d/train.py:140-145: criterion + SmoothL1(logits[1:], logits[:-1]) * 0.1, one fused pass
            return crit(outs[1], y, time_weight=0.1)
        if wl["model"] == "Siam_UNet":       # siam_unet/train.py:110
            return crit(outs[1], y)
        # multi_output_unet3d/train.py:183-195: per-head criterion on the (already activated) output, weighted sum (weights 1)
        return sum(head_crit(outs[k], tgt[k]) for k in outs)

    def fwd():
        return model(x, px) if px is not None else model(x)

    from bio_image_unet_amd.ddp import GradAverager
    avg = GradAverager(model)
    avg_ms = []                                    # host milliseconds every average() spent launching leftovers + waiting (per step)
    avg.wait_log = avg_ms

    def step():
        outs = fwd()
        loss = loss_of(outs)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        avg.average()                              # (a no-op at N = 1) -- BEFORE the clip: buckets that left during backward are not re-read
        avg_ms.append(avg.average_wait_ms)
        if wl["model"] == "MultiOutputUnet3D":     # multi_output_unet3d/train.py:201, on the global-batch gradient
            opt.clip_grad_norm_(1.0)
        opt.step()
        return loss

    if graph:
        from bio_image_unet_amd.graph import GraphedTrainStep
        clip = (lambda: opt.clip_grad_norm_(1.0)) if wl["model"] == "MultiOutputUnet3D" else None
        ins = [x, px] if px is not None else [x]
        gstep = GraphedTrainStep(model, lambda outs: loss_of(outs), opt, ins, [], after_backward=clip)
        eager_step = step

        def step():                                  # noqa: F811
            return gstep(ins, [])
        step.eager = eager_step

    nvox = 1
    for s in (shape[0],) + tuple(shape[2:]):
        nvox *= s
    return model, step, fwd, nvox, avg


def call_cost(eng, api, label, executed=False):
    """Algorithmic FLOPs and bytes (SURVEY.md 8d convention: the work of the reference ops the call stands for) of one C-ABI call, from the
    node's shapes.  ``executed=True``: the FLOPs the kernels actually issue -- fewer for the folded decoder ops, which compute
    the same function with 8 parity classes x 8 coarse taps instead of 27 fine taps on the up-sampled channels."""
    from bio_image_unet_amd import engine as E
    esz = 2 if eng.tdtype == torch.bfloat16 else 4
    node = next((n for n in eng.nodes if n.label == label.split(":")[0]), None)
    if node is None:
        return 0.0, 0.0
    if isinstance(node, E.ConvBlockNode) and api in ("biu_conv_fwd", "biu_conv_fwd_stats", "biu_conv_bwd_data", "biu_conv_bwd_data_bnred",
                                                     "biu_conv_bwd_weight", "biu_conv_bwd_weight_bn", "biu_conv_fwd_cat",
                                                     "biu_conv_bwd_data_cat", "biu_conv_bwd_weight_cat"):
        taps = node.kd * node.kh * node.kw
        v = node.y.nvox
        return 2.0 * v * taps * node.xin.c * node.y.c, float(v) * (node.xin.c + node.y.c) * esz
    if label.endswith("/chain"):               # the chain rule of a folded level on its tables (side stream): weight-space work only
        return 0.0, 0.0
    if isinstance(node, E.ConvBlockNode) and api in ("biu_foldt_fwd", "biu_foldt_bwd_data", "biu_foldt_bwd_weight_bn", "biu_foldt_bwd_weight_bn_phase") \
            and node.foldt is not None:
        # ConvTranspose + concat + conv as one op: 27 taps on the skip channels, 8 parity classes x 8 coarse taps on the ConvT's input channels
        # (the work the kernels do; the unfolded op would be 27 taps on all concat channels plus the ConvT); bytes: coarse input, skip, output
        v = node.y.nvox
        lo, skip, cup = node.foldt.xin, node.xin.parts[1], node.foldt.y.c
        by = float(v) * (lo.c / 8.0 + skip.c + node.y.c) * esz
        if executed:
            return 2.0 * v * node.y.c * (27 * skip.c + 8 * lo.c), by
        # the reference ops it replaces: Conv3d(k3) on all concat channels + ConvTranspose3d(k2, s2) (one tap per fine voxel)
        return 2.0 * v * node.y.c * 27 * (skip.c + cup) + 2.0 * v * lo.c * cup, by
    if isinstance(node, E.ConvBlockNode) and api.startswith("biu_upconv") and api != "biu_upconv_pack":
        # up-sampling folded into the conv: 8 parity classes x 8 coarse taps per FINE voxel instead of 27 fine taps (the work the kernel does);
        # bytes: the coarse input instead of the up-sampled one
        v = node.y.nvox
        by = float(v) * (node.xin.c / 8.0 + node.y.c) * esz
        return 2.0 * v * (8 if executed else 27) * node.xin.c * node.y.c, by          # reference: Conv3d(k3) on the up-sampled tensor
    if isinstance(node, E.ConvTNode) and api.startswith("biu_convt"):
        v = node.xin.nvox
        taps = node.kd * 4
        return 2.0 * v * taps * node.xin.c * node.y.c, float(v) * (node.xin.c + taps * node.y.c) * esz
    if isinstance(node, E.ConvBlockNode):      # BN / element-wise passes over y
        return 0.0, float(node.y.nvox) * node.y.c * esz * (3 if api == "biu_bn_bwd_apply" else (2 if api == "biu_bn_bwd_reduce" else 1))
    if isinstance(node, E.ResampleNode):
        bwd = api.endswith("_bwd") or "_bwd_" in api          # backward of a 2x pool: x, dx in, dx out (+ the pooled gradient)
        return 0.0, float(node.xin.nvox * node.xin.c * (3 if bwd else 1) + node.y.nvox * node.y.c) * esz
    return 0.0, 0.0


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown CPU"


def _time_oracle_steps(fwd, sd, shape, steps, budget_s, loss_of):
    """fwd + loss + bwd + Adam on the CPU oracle; returns (best seconds per step, timed steps)."""
    opt = torch.optim.Adam([v for v in sd.values() if v.requires_grad], lr=1e-3)
    x = torch.rand(shape)
    y = (torch.rand(shape) > 0.5).float()
    times, t_start = [], time.time()
    for i in range(steps + 1):
        t0 = time.time()
        _, logits = fwd(x)
        loss = loss_of(logits, y)
        opt.zero_grad()
        loss.backward()
        opt.step()
        if i > 0:
            times.append(time.time() - t0)
        if time.time() - t_start > budget_s and times:
            break
    return min(times), len(times)


def cpu_baseline(wl_name, budget_s=22.0):
    """The oracle (CPU restatement of the reference path, fp32, torch CPU threads = host cores): cfg1 verbatim (SURVEY 8d:
    Unet(1,1,32), 2 x 256^2, >= 5 timed steps) and a bounded sample of the benchmarked workload beside it."""
    from oracle import unet_oracle as O
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    ncores = min(ncores, 16)            # the GPU box's CPU share for one GPU is 16 cores
    torch.set_num_threads(ncores)
    torch.manual_seed(1234)
    cpu = _cpu_model()
    sd1 = O.clone_state(O.init_unet2d(1, 1, 32, seed=0), requires_grad=True)
    t1, n1 = _time_oracle_steps(lambda x: O.unet2d_forward(sd1, x, training=True), sd1, (2, 1, 256, 256), 5, 8.0,
                                lambda lg, y: O.trainer2d_loss(lg, y, 1))
    cfg1 = {"value": 2 * 256 * 256 / t1, "unit": "voxels/s", "ms_per_step": t1 * 1e3,
            "sample": f"cfg1 verbatim: Unet(1,1,32) oracle, (2,1,256,256) fp32, fwd+loss+bwd+Adam, best of {n1} timed steps"}
    if wl_name in ("cfg4", "cfg5"):
        sd = O.clone_state(O.init_unet3d(1, 1, 32, seed=0), requires_grad=True)
        shape = (1, 1, 64, 128, 128)
        t, n = _time_oracle_steps(lambda x: O.unet3d_forward(sd, x, training=True), sd, shape, 3, budget_s, O.bce_dice_loss)
        name = "UNet3D(1,1,32) oracle, (1,1,64,128,128) fp32 (1/8 of one cfg4 batch entry x 1), fwd+loss+bwd+Adam"
        nvox = 64 * 128 * 128
    else:
        t, n, name, nvox = t1, n1, cfg1["sample"], 2 * 256 * 256
    return {"value": nvox / t, "unit": "voxels/s", "cores": ncores, "cpu": cpu, "kind": "port",
            "sample": f"{name}; best of {n} timed steps after 1 warm-up ({t:.3f} s/step); torch {torch.__version__} CPU threads={ncores} on {cpu}",
            "cfg1_verbatim": cfg1}


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (the torchrun environment contract), one per
    GPU.  Runs BEFORE this process has imported the HIP library or made any GPU call, and it never makes one: the children are
    ordinary child processes (no exec of a GPU-initialised process), their stdout/stderr are inherited, so rank 0's JSON line is
    the only JSON on stdout.  Exit code: 0 only if every rank exited 0; a failed rank takes the others down (exact PIDs)."""
    import subprocess
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    env = dict(os.environ, MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=port, WORLD_SIZE=str(n),
               LOCAL_WORLD_SIZE=str(n), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)))
             for r in range(n)]
    rc, alive = 0, list(procs)
    while alive:
        for p in list(alive):
            try:
                code = p.wait(timeout=0.2)
            except subprocess.TimeoutExpired:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench: rank {procs.index(p)} exited with {code}; stopping the other ranks", file=sys.stderr)
                for q in alive:
                    q.terminate()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="cfg4", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured hipGraph (N = 1 only)")
    ap.add_argument("--fp32-products", default="bf16x6", choices=["exact", "bf16x3", "bf16x6"],
                    help="fp32 workloads: how the 2-D 3x3 kernels multiply (default: bf16x6, the library's default -- fp32-grade split products)")
    ap.add_argument("--breakdown", default=None, help="write the per-launch time table of one profiled step to this file")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    from bio_image_unet_amd import ddp
    from bio_image_unet_amd._lib import lib
    # rehearsal switches for a one-GPU box: BIU_DDP_BACKEND=gloo, BIU_SINGLE_DEVICE=1 (every rank on cuda:0)
    if os.environ.get("BIU_SINGLE_DEVICE") == "1":
        os.environ["LOCAL_RANK"] = "0"
    rank, local, world = ddp.init_from_env(os.environ.get("BIU_DDP_BACKEND", "nccl"))
    if world != args.gpus:
        sys.exit(f"bench: --gpus {args.gpus} but WORLD_SIZE={world} (launch one rank per GPU, or leave WORLD_SIZE unset)")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    wl = WORKLOADS[args.workload]
    if wl["dtype"] == "f32":
        import bio_image_unet_amd
        bio_image_unet_amd.set_fp32_products(args.fp32_products)
    model, step, fwd, nvox, avg = make_step(wl, device, graph=args.graph and world == 1)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # ---- warm-up; the last warm-up step is profiled per C-ABI call to find the dominant launch ------------------------
    for _ in range(max(args.warmup - 1, 0)):
        step()
    lib.prof = []
    getattr(step, "eager", step)()              # (a replayed graph makes no C-ABI calls: the profiled step is the eager one)
    torch.cuda.synchronize()
    prof, lib.prof = lib.prof, None
    eng = list(model._engines.values())[-1][-1]
    agg = {}
    for api, label, e0, e1 in prof:
        agg.setdefault((api, label), []).append(e0.elapsed_time(e1))
    rows = sorted(((sum(v), k) for k, v in agg.items()), reverse=True)
    total_kernel_ms = sum(r[0] for r in rows)
    dom_ms, dom_key = rows[0]
    step_exec_flop = sum(call_cost(eng, api, label, executed=True)[0] * len(agg[(api, label)]) for api, label in agg)
    if args.breakdown and rank == 0:
        with open(args.breakdown, "w") as f:
            f.write(f"# per-launch HIP-event times of one profiled step, workload {args.workload}; sum = {total_kernel_ms:.3f} ms\n")
            f.write("# TFLOP/s (executed) and GB/s: the FLOPs the kernels of the call ISSUE and its algorithmic bytes (SURVEY 8d) -- hardware utilisation; "
                    "'alg': the FLOPs of the reference ops the call stands for (larger for the folded decoder calls biu_foldt_*, biu_upconv_*: DESIGN.md 3.4); "
                    "'roof': max(executed FLOP / MFMA peak, bytes / 8 TB/s) / time\n")
            for ms, (api, label) in rows:
                fl, by = call_cost(eng, api, label)
                fx, _ = call_cost(eng, api, label, executed=True)
                pk = MFMA_PEAK["bf16" if eng.tdtype == torch.bfloat16 else "f32"]
                t_roof = max(fx / pk, by / HBM_PEAK) * 1e3
                f.write(f"{ms:9.4f} ms  {api:26s} {label:22s} {fx / ms / 1e9 if ms else 0:9.1f} TFLOP/s {by / ms / 1e6 if ms else 0:9.1f} GB/s   "
                        f"alg {fl / ms / 1e9 if ms else 0:7.1f}  roof {t_roof / ms if ms else 0:5.2f}\n")

    # ---- timed region --------------------------------------------------------------------------------------------------
    lib.watch, lib.watched = dom_key, []
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]     # per-step stamps on the compute stream (median)
    # The interpreter's cyclic garbage collector is paused inside the timed region, exactly as the package's Trainers pause it inside an
    # epoch's batch loop (workflow._EpochLoop._train_epoch): a generation-2 collection walks every object torch has imported and stalls
    # the host for 60-140 ms -- with steps of 5 ms (cfg1) the GPU queue runs dry and ONE such step took 60-115 ms (BENCH_GC=on shows it).
    # A step creates no reference cycles; reference counting frees everything it allocates.
    import gc
    gc_paused = os.environ.get("BENCH_GC") != "on"
    if gc_paused:
        gc.collect()
        gc.disable()
    barrier()
    t0 = time.perf_counter()
    use_marks = os.environ.get("BENCH_NO_MARKS") != "1"
    t_host = []
    if use_marks:
        marks[0].record()
    for i in range(args.steps):
        step()
        if use_marks:
            marks[i + 1].record()
        t_host.append(time.perf_counter())
    t_loop = time.perf_counter()
    barrier()
    dt = time.perf_counter() - t0
    if gc_paused:
        gc.enable()
    if os.environ.get("BENCH_DEBUG_STEPS") == "1":
        print(f"bench: host enqueue ms/step {(t_loop - t0) / args.steps * 1e3:.2f}, barrier wait {(t0 + dt - t_loop) * 1e3:.1f} ms; host per-step " +
              " ".join(f"{(b - a) * 1e3:.1f}" for a, b in zip([t0] + t_host[:-1], t_host)), file=sys.stderr)
    per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)] if use_marks else [dt / args.steps * 1e3] * args.steps
    if os.environ.get("BENCH_DEBUG_STEPS") == "1":
        print("bench: per-step ms " + " ".join(f"{v:.2f}" for v in per_step), file=sys.stderr)
    per_step.sort()
    median_ms = per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2])
    if hasattr(step, "eager"):                  # graph mode: the dominant launch is timed in eager steps after the timed region
        for _ in range(3):
            step.eager()
        torch.cuda.synchronize()
    watched, lib.watch = lib.watched, None
    ranks_seen = 1
    per_rank = None
    if world > 1:
        # every rank's own step time and the host time its average() calls took: a measured scaling curve can then be attributed
        # (stragglers vs. exposed all-reduce)
        waits = getattr(avg, "wait_log", [])[-args.steps:]
        mine = torch.tensor([dt / args.steps * 1e3, median_ms, sum(waits) / max(len(waits), 1), max(waits) if waits else 0.0], device=device, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(allr, mine)
        per_rank = [[round(float(v), 4) for v in r_.tolist()] for r_ in allr]
        t = torch.tensor([dt, median_ms], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt, median_ms = float(t[0].item()), float(t[1].item())
        one = torch.ones(1, device=device, dtype=torch.float64)
        torch.distributed.all_reduce(one)                      # every rank adds 1: the driver can see that N ranks really ran
        ranks_seen = int(one.item())
    ms_per_step = dt / args.steps * 1e3

    # forward-only (eval-style use, no_grad; BN in train mode exactly as the reference's validation loop)
    with torch.no_grad():
        fwd()
        barrier()
        t1 = time.perf_counter()
        nf = max(args.steps // 2, 1)
        for _ in range(nf):
            fwd()
        barrier()
        fwd_ms = (time.perf_counter() - t1) / nf * 1e3

    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    dom_launch_ms = sum(e0.elapsed_time(e1) for _, _, e0, e1 in watched) / max(len(watched), 1)
    # utilisation figures are taken on the FLOPs the kernels of the call issue (`executed`); for the folded decoder ops the FLOPs of the
    # reference ops they replace (SURVEY 8d's algorithmic convention, larger) are reported beside them as `algorithmic_equivalent`
    fl_alg, by = call_cost(eng, *dom_key)
    fl, _ = call_cost(eng, *dom_key, executed=True)
    dt_name = wl["dtype"]
    x3 = dt_name == "f32" and wl["model"] in ("Unet", "Siam_UNet") and args.fp32_products != "exact"
    nterms = {"bf16x3": 3, "bf16x6": 6}.get(args.fp32_products, 1)
    mfma_peak = MFMA_PEAK["bf16"] / nterms if x3 else MFMA_PEAK[dt_name]
    ai = fl / by if by else 0.0
    ridge = mfma_peak / HBM_PEAK
    if fl > 0 and ai >= ridge * 0.5:
        roof = {"bound": "mfma", "achieved": fl / (dom_launch_ms * 1e-3) / 1e12, "peak": mfma_peak / 1e12, "unit": "TFLOP/s"}
    else:
        roof = {"bound": "hbm", "achieved": by / (dom_launch_ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s"}
    roof["frac"] = roof["achieved"] / roof["peak"]
    if fl_alg != fl and fl > 0:
        roof["algorithmic_equivalent"] = {"TFLOP_per_call": fl_alg / 1e12, "TFLOPps": fl_alg / (dom_launch_ms * 1e-3) / 1e12,
                                          "note": "folded op (DESIGN.md 3.4): the kernels issue 8 parity classes x 2x2x2 coarse taps instead of 27 fine taps on the "
                                                  "up-sampled channels; `achieved` / `frac` count the issued FLOPs, this entry the FLOPs of the reference ops "
                                                  "(Conv3d on all concat channels + ConvTranspose3d) the call replaces -- not a utilisation figure"}
    roof["traffic"] = None          # HBM bytes per launch from PMC counters (tools/pmc_traffic.py), when measured for this call
    tr_file = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"pmc_traffic_{args.workload}.json")
    if os.path.exists(tr_file):
        tr = json.load(open(tr_file)).get(f"{dom_key[0]} @ {dom_key[1]}")
        if tr:
            roof["traffic"] = tr["traffic_bytes_per_call"]
            roof["traffic_source"] = f"profiles/pmc_traffic_{args.workload}.json (rocprofv3 PMC, separate run; algorithmic bytes {by:.4g})"
    roof["kernel"] = f"{dom_key[0]} @ {dom_key[1]}"
    roof["launch_ms"] = dom_launch_ms
    roof["launches_timed"] = len(watched)
    roof["share_of_step_kernel_time"] = dom_ms / total_kernel_ms if total_kernel_ms else None

    # whole-step roofline numbers of SURVEY.md 8d (cfg4: 1.316 MFLOP and 2915 B per voxel, fwd+bwd)
    per_vox = {"cfg4": (1.316e6, 2915.0), "cfg2": (3 * 1467.8e3, 3 * 5596.0), "cfg3": (3 * 469e3, 3 * 1940.0),
               "cfg5": (3 * 3243.7e3, 3 * 2179.0), "cfg1": (3 * 367.2e3, 3 * 2800.0)}[args.workload]
    vps_gpu = nvox / (ms_per_step * 1e-3)
    out = {
        "metric": "voxels/sec fwd+bwd", "value": vps_gpu * world, "unit": "voxels/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "median_ms_per_step": median_ms,
        "value_at_median": nvox / (median_ms * 1e-3) * world, "step_ms_min_max": [per_step[0], per_step[-1]],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dt_name, "data": "synthetic", "host": "cyclic GC paused inside the timed region (as in the package's Trainers)" if gc_paused else "cyclic GC on",
        "config": {"workload": f"{args.workload}: {wl['model']}({', '.join(f'{k}={v}' for k, v in wl['ctor'].items() if k != 'output_heads')}) input {wl['shape']} per GPU, "
                               "train step = forward + reference loss + backward + Adam", "parallelism": f"dp{world}"},
        "fwd_only": {"value": nvox / (fwd_ms * 1e-3) * world, "unit": "voxels/s", "ms": fwd_ms},
        "step_roofline": {"hbm_frac_algorithmic": vps_gpu * per_vox[1] / HBM_PEAK, "algorithmic_GBps": vps_gpu * per_vox[1] / 1e9,
                          "algorithmic_TFLOPps": vps_gpu * per_vox[0] / 1e12, "mfma_frac_algorithmic": vps_gpu * per_vox[0] / mfma_peak,
                          "executed_TFLOPps": step_exec_flop / (ms_per_step * 1e-3) / 1e12, "mfma_frac": step_exec_flop / (ms_per_step * 1e-3) / mfma_peak,
                          "note": "mfma_frac = FLOPs the step's kernels issue / time / peak (utilisation); *_algorithmic = SURVEY 8d's per-voxel figures of the "
                                  "reference ops (what `value` x 1.316 MFLOP is), larger where decoder levels run folded",
                          "kernel_time_ms_one_step": total_kernel_ms},
        "roofline": roof,
    }
    if hasattr(step, "eager"):
        out["graph"] = "step replayed from one captured hipGraph; roofline.launch_ms from eager steps after the timed region"
    if x3 and nterms == 3:
        out["arithmetic"] = ("fp32 tensors and accumulators; the products of the 3x3 convolutions (forward, data and weight gradient) are bf16x3: "
                             "operands split hi + lo in bf16, hi*hi + hi*lo + lo*hi on the bf16 MFMA, <= 2^-15 relative per product; "
                             "roofline peak = bf16 dense peak / 3")
    elif x3:
        out["arithmetic"] = ("fp32 tensors and accumulators; the products of the 3x3 convolutions (forward, data and weight gradient) are bf16x6: "
                             "operands split hi + mid + lo in bf16 (24 significant bits), the six terms of order >= 2^-16 on the bf16 MFMA, "
                             "<= 2^-23 relative per product (fp32-grade; every fp32 parity test runs in this mode); roofline peak = bf16 dense peak / 6")
    elif dt_name == "f32":
        out["arithmetic"] = "fp32 tensors, fp32 MFMA (v_mfma_f32_32x32x2_f32)"
    if world > 1:           # gradient all-reduce: decoder -> encoder buckets, issued from inside backward (bio_image_unet_amd/ddp.py)
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:                               # gloo rehearsal on a box without the collective library loaded
            rccl = None
        out["ddp"] = {"ranks_seen": ranks_seen, "backend": torch.distributed.get_backend(), "rccl_version": rccl,
                      "buckets": len(avg.buckets), "launched_in_backward": avg.launched_in_backward,
                      "bucket_mbytes": [round(b.flat.numel() * 4 / 2 ** 20, 2) for b in avg.buckets],
                      "grad_copies_in_backward": avg.copies_in_backward,
                      "per_rank": {"columns": ["ms_per_step", "median_ms_per_step", "average_wait_ms_mean", "average_wait_ms_max"], "rows": per_rank},
                      "step_ms_min_max_over_ranks": [min(r_[0] for r_ in per_rank), max(r_[0] for r_ in per_rank)] if per_rank else None}
    if not args.no_cpu_baseline and world == 1:      # reported at N = 1 only (rank 0's host cores, bounded sample)
        out["cpu_baseline"] = cpu_baseline(args.workload)
    print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
