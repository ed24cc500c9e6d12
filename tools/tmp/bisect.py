import sys, torch
sys.path.insert(0, ".")
from tests.backend import setup_backend
from tests.test_net import build
from tests.util import rel_l2
from storm_amd import _lib as L
dev = setup_backend("hip")
m, sd, cfg = build(dict(input_channels=4), 11, dev)
g = torch.Generator().manual_seed(5)
x = (torch.randn(3, 2, 64, 64, dtype=torch.complex64, generator=g) * 0.5).to(dev)
t = torch.tensor([0.9, 0.4, 0.05], device=dev)
m.set_compute_dtype(torch.bfloat16)
for name, val in [("STORM_SPLITK", 0), ("STORM_SPLITK", 1), ("STORM_SPLITK", 2), ("STORM_CONV_VARIANT", 9), ("STORM_CONV_VARIANT", 7), ("STORM_CONV_VARIANT", 0)]:
    L.lib().storm_set_switch(b"STORM_SPLITK", 0); L.lib().storm_set_switch(b"STORM_CONV_VARIANT", -1)
    L.lib().storm_set_switch(name.encode(), val)
    yb = m(x, t)
    errs = [rel_l2(m(x[b:b + 1], t[b:b + 1]).cpu(), yb[b:b + 1].cpu()) for b in range(3)]
    print(name, val, ["%.2e" % e for e in errs])
    for B in (3, 1):
        ops_p, n_ops, _ = m.program(B, 64, 64)
        names = [L.lib().storm_program_kernel_name(ops_p, k, L.BF16).decode() for k in range(n_ops)]
        print("   B", B, [n.split("::")[-1][:28] for n in names if n][:60])
