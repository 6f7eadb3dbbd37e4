"""Practical HBM rates on this box with plain torch kernels (reference points for the streaming kernels): fill, copy, read-reduce."""
import torch
def t(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for gb in (0.5, 1.0, 2.0):
    n = int(gb * 2**30 // 2)
    a = torch.empty(n, dtype=torch.bfloat16, device="cuda"); b = torch.empty_like(a)
    a.normal_()
    ms_fill = t(lambda: b.zero_())
    ms_copy = t(lambda: b.copy_(a))
    ms_read = t(lambda: a.view(torch.int16).max())
    by = n * 2
    print(f"{gb:.1f} GiB: fill {by/ms_fill/1e9:.2f} TB/s, copy (r+w) {2*by/ms_copy/1e9:.2f} TB/s, read-reduce {by/ms_read/1e9:.2f} TB/s", flush=True)
