#!/usr/bin/env python
"""Numerics gate for a Winograd F(2x2, 3x3) form of the wide 3x3 layers, on the CPU oracle.

The 3x3 convolutions with >= 128 input channels of the NCSN++ forward at the bench shape (fixture F8: the REFERENCE's
output for one [1,2,256,512] utterance) are replaced by emulations of what a 16-bit engine would compute:
  direct : operands rounded to bf16 (fp16), fp32 accumulation                 (what the HIP kernels do)
  wino   : V = B^T d B and U = G g G^T formed in fp32 and ROUNDED to bf16 (fp16) - they are the MFMA operands -, the 16
           element-wise products accumulated over the input channels in fp32, Y = A^T M A in fp32
every activation rounded to the 16-bit type between layers in both cases.  Prints the forward's rel-L2 against the reference.
Container-only experiment (runs the oracle, reads tests/golden): python tools/experiments/winograd_numerics.py [bf16|fp16]"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ncsnpp_ref as NR  # noqa: E402
from oracle.make_golden import seeded_input  # noqa: E402

dt = torch.float16 if (len(sys.argv) > 1 and sys.argv[1] == "fp16") else torch.bfloat16
rd = lambda t: t.to(dt).float()  # noqa: E731
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def wino_conv(x, w):
    """F(2x2, 3x3), stride 1, pad 1; x [B,C,H,W] (H, W even), w [K,C,3,3]; 16-bit operands in the transformed domain"""
    B, C, H, W = x.shape
    xp = F.pad(rd(x), (1, 1, 1, 1))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                      # [B,C,H/2,W/2,4,4]
    V = rd(torch.einsum("ij,bchwjk,lk->bchwil", BT, d, BT))      # input transform, rounded: the MFMA B operand
    U = rd(torch.einsum("ij,kcjl,ml->kcim", G, w, G))            # weight transform, rounded: the MFMA A operand
    M = torch.einsum("kcim,bchwim->bkhwim", U, V)                # 16 GEMMs, fp32 accumulation over c
    Y = torch.einsum("ij,bkhwjl,ml->bkhwim", AT, M, AT)          # [B,K,H/2,W/2,2,2]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B, w.shape[0], H, W)


mode = {"v": "direct"}
real_conv2d = F.conv2d


def conv2d(x, w, b=None, padding=0, **kw):
    if w.shape[-1] == 3 and w.shape[1] >= 128 and w.shape[0] >= 128 and mode["v"] == "wino":
        y = wino_conv(x, w)
    else:
        y = real_conv2d(rd(x), rd(w), None, padding=padding, **kw)
    if b is not None:
        y = y + b[None, :, None, None]
    return rd(y)                                                # activations are stored in the 16-bit type


g = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "f8_bench_shape.npz"))
cfg = NR.NCSNppConfig(input_channels=4)
sd = NR.seeded_state_dict(cfg, seed=11)
x = seeded_input((1, 2, 256, 512), 808, 0.5)
ref = torch.from_numpy(g["full4_y"])
torch.set_num_threads(8)
NR.F.conv2d = conv2d
for m in ("direct", "wino"):
    mode["v"] = m
    with torch.no_grad():
        y = NR.ncsnpp_forward(sd, cfg, x, torch.from_numpy(g["t"]))
    err = float((y - ref).abs().pow(2).sum().sqrt() / ref.abs().pow(2).sum().sqrt())
    print(f"{dt} {m:6s}: forward rel-L2 vs the reference {err:.3e}", flush=True)
