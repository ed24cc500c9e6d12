#!/usr/bin/env python
"""What the part sustains on the access patterns of the HBM-bound family: a pure write stream (torch fill), a copy, and a kernel that
writes eight bytes per byte read (the shape of gn_apply_up: one 16-bit tensor in, two tensors of four times its size out) - the
yardsticks for `roofline_hbm` besides the 8 TB/s data-sheet figure."""
import torch

dev = torch.device("cuda:0")
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


n = 1 << 30                                                     # 2 GiB of bf16
a = torch.empty(n, dtype=torch.bfloat16, device=dev)
b = torch.empty(n, dtype=torch.bfloat16, device=dev)
t = timed(lambda: a.fill_(1.0)); print(f"fill   2 GiB          : {2 * n / t / 1e12:5.2f} TB/s written")
t = timed(lambda: b.copy_(a)); print(f"copy   2 GiB -> 2 GiB : {4 * n / t / 1e12:5.2f} TB/s read + written")
s = torch.empty(n // 8, dtype=torch.bfloat16, device=dev)
t = timed(lambda: a.view(-1, 8).copy_(s.view(-1, 1).expand(-1, 8)))
print(f"1 -> 8 expanding write : {(2 * n + 2 * n // 8) / t / 1e12:5.2f} TB/s read + written (0.25 GiB in, 2 GiB out)")
