"""Race check of the engine's side stream: the same seeded training run (UNet3D(1,1,32), bf16, BCEDice + Adam) with the side
stream on, on again, and off (one child process each -- the switches are read once per process); the loss trajectories must agree to the
run-to-run noise of the fp32 atomics in the weight gradients.  A missing event would show as a step that reads a half-written gradient.

    python tools/side_stream_check.py [steps] [n,d,h,w]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "child" else 40
SHAPE = sys.argv[2] if len(sys.argv) > 2 and sys.argv[1] != "child" else "4,128,128,128"

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    import bio_image_unet_amd as B
    from bio_image_unet_amd.losses import BCEDiceLoss
    from bio_image_unet_amd.optim import Adam
    steps, shape = int(sys.argv[2]), tuple(int(v) for v in sys.argv[3].split(","))
    torch.manual_seed(0)
    m = B.UNet3D(1, 1, 32).cuda()
    m.set_compute_dtype(torch.bfloat16)
    m.train()
    opt = Adam(m.parameters(), lr=1e-3)
    crit = BCEDiceLoss(1.0, 1.0)
    g = torch.Generator(device="cuda").manual_seed(1)
    n, d, h, w = shape
    xs = [torch.rand(n, 1, d, h, w, device="cuda", generator=g) for _ in range(4)]
    ys = [(torch.rand(n, 1, d, h, w, device="cuda", generator=g) > 0.5).float() for _ in range(4)]
    out = []
    for i in range(steps):
        opt.zero_grad()
        p, l = m(xs[i % 4])
        loss = crit(l, ys[i % 4])
        loss.backward()
        opt.step()
        out.append(float(loss))
    torch.cuda.synchronize()
    fin = all(bool(torch.isfinite(q).all()) for q in m.parameters())
    print("RESULT " + json.dumps({"loss": out, "finite": fin}))
    sys.exit(0)


def run(env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(STEPS), SHAPE], env=env, capture_output=True, text=True, cwd=ROOT)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    assert r.returncode == 0 and line, r.stdout[-2000:] + r.stderr[-2000:]
    return json.loads(line[-1][7:])


a = run({})
b = run({})
c = run({"BIU_SIDE_WGRAD_VOX": "0", "BIU_DISABLE": "sidechain,prepack"})
assert a["finite"] and b["finite"] and c["finite"]
rel = lambda u, v: max(abs(p - q) / max(abs(q), 1e-6) for p, q in zip(u, v))
print(f"{STEPS} steps at {SHAPE}: loss {a['loss'][0]:.5f} -> {a['loss'][-1]:.5f}")
print(f"side stream on vs on again : max relative loss difference {rel(a['loss'], b['loss']):.2e}")
print(f"side stream on vs off      : max relative loss difference {rel(a['loss'], c['loss']):.2e}")
