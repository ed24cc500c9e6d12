import sys, torch
sys.path.insert(0, "/root/repo")
from tests.backend import setup_backend
from storm_amd import ops
from storm_amd import _lib as L
dev = setup_backend(sys.argv[1] if len(sys.argv) > 1 else "sim")
dt = torch.bfloat16
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g)
for (B, H, W, cin, cout) in [(3, 32, 32, 256, 256), (3, 16, 16, 256, 256), (3, 8, 8, 256, 256), (3, 32, 32, 128, 256), (3, 16, 16, 512, 256)]:
    x = rnd(B, H, W, cin).to(dt).to(dev)
    w = ops.pack_conv_weight((rnd(cout, cin, 3, 3) * 0.05).to(dev), dt)
    ss = ops.pack_gn_ss(1 + 0.1 * rnd(B, cin), 0.1 * rnd(B, cin)).to(dev)
    bias, tb = rnd(cout).to(dev), rnd(B, cout).to(dev)
    y, part = ops.conv([ops.Seg(x, w, 9, gn_ss=ss, gn_silu=True)], cout, bias=bias, tbias=tb, gn_partials=True, scale=0.7)
    name = ops.conv_kernel_name([ops.Seg(x, w, 9, gn_ss=ss, gn_silu=True)], cout, bias=bias, tbias=tb, scale=0.7)
    for b in range(B):
        yb, pb = ops.conv([ops.Seg(x[b:b+1].contiguous(), w, 9, gn_ss=ss[b:b+1].contiguous(), gn_silu=True)], cout, bias=bias, tbias=tb[b:b+1].contiguous(), gn_partials=True, scale=0.7)
        nb = ops.conv_kernel_name([ops.Seg(x[b:b+1].contiguous(), w, 9, gn_ss=ss[b:b+1].contiguous(), gn_silu=True)], cout, bias=bias, tbias=tb[b:b+1].contiguous(), scale=0.7)
        print(B, H, W, cin, name.split("::")[-1], "| row", b, nb.split("::")[-1], "y equal", torch.equal(yb[0], y[b]), "part equal", torch.equal(pb[0], part[b]), float((yb[0].float()-y[b].float()).abs().max()))
