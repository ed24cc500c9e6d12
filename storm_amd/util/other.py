"""Small utilities of sgmse/util/other.py that the hot path and the evaluation loop use:
pad_spec (:102-109), si_sdr / si_sdr_torch (:82-94)."""
import numpy as np
import torch


def pad_spec(Y):
    T = Y.size(3)
    num_pad = 64 - T % 64 if T % 64 != 0 else 0
    return torch.nn.functional.pad(Y, (0, num_pad, 0, 0))


def si_sdr(s, s_hat):
    """numpy, one pair (util/other.py:82-86) - the host-side definition the evaluation loop of the reference uses"""
    alpha = np.dot(s_hat, s) / np.linalg.norm(s) ** 2
    return 10 * np.log10(np.linalg.norm(alpha * s) ** 2 / np.linalg.norm(alpha * s - s_hat) ** 2)


def si_sdr_torch(s, s_hat):
    """one pair of 1-D tensors (util/other.py:88-94, the eps = 1e-10 variant): a 0-d tensor.  On a HIP device the sums run
    in one kernel (storm_si_sdr); CPU tensors (host-side metric code) use the same formula in torch."""
    min_len = min(s.size(-1), s_hat.size(-1))
    s, s_hat = s[..., :min_len], s_hat[..., :min_len]
    if s.is_cuda:
        from .. import ops
        return ops.si_sdr(s.reshape(1, -1).float().contiguous(), s_hat.reshape(1, -1).float().contiguous(), eps=1e-10)[0]
    alpha = torch.dot(s_hat, s) / torch.norm(s) ** 2
    return 10 * torch.log10(1e-10 + torch.norm(alpha * s) ** 2 / (1e-10 + torch.norm(alpha * s - s_hat) ** 2))


def si_sdr_batch(s, s_hat, eps=0.0):
    """[B, L] device tensors -> [B] dB (one launch for the whole evaluation batch)"""
    from .. import ops
    return ops.si_sdr(s.float(), s_hat.float(), eps=eps)
