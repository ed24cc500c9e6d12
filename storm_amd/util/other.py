"""pad_spec (sgmse/util/other.py:102-109)."""
import torch


def pad_spec(Y):
    T = Y.size(3)
    num_pad = 64 - T % 64 if T % 64 != 0 else 0
    return torch.nn.functional.pad(Y, (0, num_pad, 0, 0))
