#!/usr/bin/env python
"""What a split of K across workgroups could buy on the few-tile 3x3 layers (ncsnpplarge's deep levels at configs[3], batch 8), MEASURED
before building it: S slices of the K loop on S x as many workgroups are emulated by the kernel that exists - the same convolution over
B*S images with Cin/S channels each and fp32 output slabs (same workgroup count, same chain of phases per workgroup, same bytes written
as the slices of a real split) - plus the reduction of the S slabs to the 16-bit output (one elementwise pass; timed as torch.sum over the
slab axis, which is what a fused combine kernel would read and a little less than it would do)."""
import sys

import torch

sys.path.insert(0, ".")
from storm_amd import ops  # noqa: E402

dev, dt = torch.device("cuda:0"), torch.bfloat16
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (B, H, W, cin) in [(16, 32, 64, 256), (16, 32, 64, 512), (8, 16, 64, 256), (8, 16, 64, 512), (8, 8, 32, 256), (8, 8, 32, 512), (8, 4, 16, 256), (8, 4, 16, 512)]:
    cout = 256
    x = rnd(B, H, W, cin).to(dt).to(dev)
    w = ops.pack_conv_weight((rnd(cout, cin, 3, 3) * 0.05).to(dev), dt)
    ss = ops.pack_gn_ss(1 + 0.1 * rnd(B, cin), 0.1 * rnd(B, cin)).to(dev)
    kw = dict(bias=rnd(cout).to(dev), tbias=rnd(B, cout).to(dev), gn_partials=True, scale=0.7)
    t_full = timed(lambda: ops.conv([ops.Seg(x, w, 9, gn_ss=ss, gn_silu=True)], cout, **kw))
    line = f"{B}x{H}x{W} {cin}->{cout}: {ops.conv_kernel_name([ops.Seg(x, w, 9, gn_ss=ss, gn_silu=True)], cout, bias=kw['bias'], tbias=kw['tbias'], scale=0.7).split('::')[-1]} {t_full:6.1f} us"
    for S in (2, 4, 8):
        if cin // S < 64:
            continue
        xs = rnd(B * S, H, W, cin // S).to(dt).to(dev)
        ws = ops.pack_conv_weight((rnd(cout, cin // S, 3, 3) * 0.05).to(dev), dt)
        sss = ops.pack_gn_ss(1 + 0.1 * rnd(B * S, cin // S), 0.1 * rnd(B * S, cin // S)).to(dev)
        t_slices = timed(lambda: ops.conv([ops.Seg(xs, ws, 9, gn_ss=sss, gn_silu=True)], cout, out_f32=True))
        slabs = ops.conv([ops.Seg(xs, ws, 9, gn_ss=sss, gn_silu=True)], cout, out_f32=True).view(S, B, H, W, -1)
        t_comb = timed(lambda: torch.sum(slabs, 0))
        line += f" | S={S}: slices {t_slices:5.1f} + combine {t_comb:4.1f} = {t_slices + t_comb:5.1f} us"
    print(line, flush=True)
