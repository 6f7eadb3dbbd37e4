"""A training step replayed from a captured hipGraph (bio_image_unet_amd/graph.py) must train exactly like the eager step: same
parameters after K steps on changing batches, a learning-rate change in between included; constructing the graphed step must not train."""
import copy

import pytest
import torch

import bio_image_unet_amd as B
from bio_image_unet_amd.graph import GraphedTrainStep
from bio_image_unet_amd.losses import BCEDiceLoss
from bio_image_unet_amd.optim import Adam
from oracle import unet_oracle as O

pytestmark = pytest.mark.gpu


def _batches(k, shape):
    g = torch.Generator().manual_seed(11)
    return [(torch.rand(*shape, generator=g).cuda(), (torch.rand(*shape, generator=g) > 0.5).float().cuda()) for _ in range(k)]


def _rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_graphed_step_trains_like_the_eager_step(dtype):
    """Step by step (a trajectory comparison would only measure how fast two runs of a BatchNorm network drift apart): before every
    replay an eager twin takes the graphed model's state; after it, loss / gradients / BatchNorm buffers of the replay must equal the
    twin's eager forward + backward on the same batch, and the parameters must be exactly the Adam update of THOSE gradients with this
    step's learning rate and bias correction (learning-rate change in between, as ReduceLROnPlateau does)."""
    shape, K = (2, 1, 64, 64), 5
    sd = O.init_unet2d(1, 1, 16, seed=4)
    crit = BCEDiceLoss(0.5, 0.5)

    def make():
        m = B.Unet(1, 1, 16).cuda()
        m.load_state_dict(sd)
        if dtype == "bf16":
            m.set_compute_dtype(torch.bfloat16)
        return m.train()

    m, twin = make(), make()
    opt = Adam(m.parameters(), lr=1e-3)
    data = _batches(K, shape)
    before = copy.deepcopy(m.state_dict())
    gstep = GraphedTrainStep(m, lambda outs, y: crit(outs[1], y), opt, [data[0][0]], [data[0][1]])
    assert set(gstep.node_kinds) <= {"kernel"}, gstep.node_kinds      # this step is kernel nodes only
    for k, v in m.state_dict().items():                   # building it (warm-up steps + capture) left the model as it was
        assert torch.equal(v, before[k]), k
    assert not opt.state or all(int(st["step"]) == 0 for st in opt.state.values())
    names = [n for n, _ in m.named_parameters()]
    # fp32 pins the machinery (two runs differ by the weight gradient's atomics only).  bf16: two correct runs of this small network differ
    # by LeakyReLU / max-pool decision flips -- single tensors by up to ~10 %; a stale buffer or a lost ordering
    # in the graph gives O(1) or garbage (seen: 1e35 when a captured memset lost its order)
    # (bf16: the forward is bit-reproducible; backward sums vary in their last bits with the order of fp32 atomics, which single bf16
    # roundings of dy pass on -- tools/diag_repro.py: two eager runs agree to ~1e-6 over all gradients together)
    g_tol, b_tol = (1e-3, 1e-5) if dtype == "f32" else (5e-2, 1e-5)
    b1, b2, eps = 0.9, 0.999, 1e-8
    for i, (x, y) in enumerate(data):
        if i == 3:
            opt.param_groups[0]["lr"] = 3e-4
        lr = opt.param_groups[0]["lr"]
        twin.load_state_dict(m.state_dict())
        pre = {n: p.detach().clone() for n, p in m.named_parameters()}
        mom = {n: (opt.state[p]["exp_avg"].clone(), opt.state[p]["exp_avg_sq"].clone()) if p in opt.state and "exp_avg" in opt.state[p]
               else (torch.zeros_like(p), torch.zeros_like(p)) for n, p in m.named_parameters()}
        loss_g = float(gstep([x], [y]))
        loss_e = crit(twin(x)[1], y)
        twin.zero_grad(set_to_none=True)
        loss_e.backward()
        assert abs(loss_g - float(loss_e)) <= 1e-5 * max(1.0, abs(float(loss_e))), (i, loss_g, float(loss_e))
        pt = dict(twin.named_parameters())
        num = den = 0.0
        for n, p in m.named_parameters():
            if ".0.bias" in n and "final" not in n:       # conv bias in front of a train-mode BatchNorm: true gradient 0, rounding noise
                continue
            assert _rel(p.grad, pt[n].grad) <= g_tol, f"step {i} grad {n}: {_rel(p.grad, pt[n].grad)}"
            num += float((p.grad - pt[n].grad).double().pow(2).sum())
            den += float(pt[n].grad.double().pow(2).sum())
        assert (num / den) ** 0.5 <= (1e-3 if dtype == "f32" else 1e-2), f"step {i}: all gradients together differ by {(num / den) ** 0.5}"
        for (n, bg), (_, be) in zip(m.named_buffers(), twin.named_buffers()):
            if bg.dtype.is_floating_point:
                assert _rel(bg, be) <= b_tol, f"step {i} buffer {n}"
            else:
                assert torch.equal(bg, be), n
        step = i + 1
        for n, p in m.named_parameters():                 # torch.optim.Adam's update (no amsgrad, no decay) of the replay's own gradients
            g = p.grad
            m1 = b1 * mom[n][0] + (1 - b1) * g
            v1 = b2 * mom[n][1] + (1 - b2) * g * g
            want = pre[n] - (lr / (1 - b1 ** step)) * m1 / (v1.sqrt() / (1 - b2 ** step) ** 0.5 + eps)
            torch.testing.assert_close(p.detach(), want, rtol=1e-5, atol=1e-7, msg=lambda s_: f"step {i} Adam update of {n}: {s_}")
            assert int(opt.state[p]["step"]) == step
    assert names


def test_eager_forward_after_a_replay_sees_the_updated_weights():
    m = B.Unet(1, 1, 16).cuda()
    m.load_state_dict(O.init_unet2d(1, 1, 16, seed=4))
    m.train()
    opt = Adam(m.parameters(), lr=1e-2)
    crit = BCEDiceLoss(0.5, 0.5)
    (x, y), = _batches(1, (2, 1, 64, 64))
    gstep = GraphedTrainStep(m, lambda outs, t: crit(outs[1], t), opt, [x], [y])
    with torch.no_grad():
        before = m(x)[1].clone()
    gstep([x], [y])
    with torch.no_grad():
        after = m(x)[1].clone()
    ref = B.Unet(1, 1, 16).cuda()                      # a fresh module with the updated parameters packs from scratch
    ref.load_state_dict(m.state_dict())
    ref.train()
    with torch.no_grad():
        want = ref(x)[1]
    assert float((after - before).abs().max()) > 1e-4
    torch.testing.assert_close(after, want, rtol=1e-5, atol=1e-5)


def test_multi_head_networks_are_refused():
    heads = {"a": {"channels": 1, "activation": "sigmoid"}, "b": {"channels": 2, "activation": None}}
    m = B.MultiOutputUnet3D(1, heads, n_filter=4).cuda()
    opt = Adam(m.parameters(), lr=1e-3)
    x = torch.rand(1, 1, 8, 16, 16).cuda()
    with pytest.raises(NotImplementedError):
        GraphedTrainStep(m, lambda outs: sum(o.mean() for o in outs.values()), opt, [x], [])


def test_node_kinds_and_a_loss_with_copy_nodes():
    """The reference's 2-D loss indexes the logits (unet/train.py:133-134): autograd's select-backward puts device copy NODES into the
    captured step.  They must replay in order (a captured memset did not: GraphedTrainStep refuses those), and node_kinds reports them."""
    import warnings
    sd = O.init_unet2d(1, 1, 16, seed=4)
    crit = BCEDiceLoss(0.5, 0.5)
    lossf = lambda outs, y: crit(outs[1][0], y[0]) + crit(outs[1][1], y[1])      # noqa: E731
    m, twin = B.Unet(1, 1, 16).cuda().train(), B.Unet(1, 1, 16).cuda().train()
    m.load_state_dict(sd)
    opt = Adam(m.parameters(), lr=1e-3)
    data = _batches(4, (2, 1, 64, 64))
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        gstep = GraphedTrainStep(m, lossf, opt, [data[0][0]], [data[0][1]])
    kinds = gstep.node_kinds
    assert kinds.get("kernel", 0) > 100 and kinds.get("memset", 0) == 0, kinds
    assert kinds.get("memcpy", 0) > 0 and any("copy nodes" in str(w.message) for w in rec), kinds
    for x, y in data:
        twin.load_state_dict(m.state_dict())
        lg = float(gstep([x], [y]))
        le = lossf(twin(x), y)
        twin.zero_grad(set_to_none=True)
        le.backward()
        assert abs(lg - float(le)) <= 1e-5 * abs(float(le))
        pt = dict(twin.named_parameters())
        num = sum(float((p.grad - pt[n].grad).double().pow(2).sum()) for n, p in m.named_parameters())
        den = sum(float(pt[n].grad.double().pow(2).sum()) for n, p in m.named_parameters())
        assert (num / den) ** 0.5 <= 1e-3
