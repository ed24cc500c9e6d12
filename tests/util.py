import torch


def rel_l2(a, b):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    if a.is_complex() or b.is_complex():
        a, b = a.to(torch.complex128), b.to(torch.complex128)
    else:
        a, b = a.double(), b.double()
    return float((a - b).abs().pow(2).sum().sqrt() / (b.abs().pow(2).sum().sqrt() + 1e-30))
