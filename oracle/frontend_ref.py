"""Oracle: spectral front/back end, PyTorch CPU fp32.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates
  * get_window / SpecsDataModule.stft / istft   (sgmse/data_module.py:19-25, 195-223)
  * spec_fwd / spec_back                        (sgmse/data_module.py:182-193)
  * pad_spec                                    (sgmse/util/other.py:102-109)
  * the wav -> spectrogram -> wav wrapper of ScoreModel.enhance (sgmse/model.py:282-303)
The FFT itself lives in ATen (torch.stft / torch.istft), a third-party dependency of
the reference (requirements.txt pins torch==1.11.0; this image has 2.10.0); the call
sites and arguments are the reference's.
"""
import torch

N_FFT = 510
HOP = 128


def window(n_fft=N_FFT, kind="hann"):
    w = torch.hann_window(n_fft, periodic=True)                      # data_module.py:19-25
    if kind == "sqrthann":
        return torch.sqrt(w)
    if kind != "hann":
        raise NotImplementedError(f"Window type {kind} not implemented!")
    return w


def stft(sig, n_fft=N_FFT, hop=HOP, kind="hann"):
    return torch.stft(sig, n_fft=n_fft, hop_length=hop, window=window(n_fft, kind),
                      center=True, return_complex=True)               # data_module.py:217-219


def istft(spec, length=None, n_fft=N_FFT, hop=HOP, kind="hann"):
    return torch.istft(spec, n_fft=n_fft, hop_length=hop, window=window(n_fft, kind),
                       center=True, length=length)                    # data_module.py:221-223


def spec_fwd(spec, factor=0.15, e=0.5):                              # data_module.py:182-186
    if e != 1:
        spec = spec.abs() ** e * torch.exp(1j * spec.angle())
    return spec * factor


def spec_back(spec, factor=0.15, e=0.5):                             # data_module.py:188-193
    spec = spec / factor
    if e != 1:
        spec = spec.abs() ** (1 / e) * torch.exp(1j * spec.angle())
    return spec


def pad_spec(Y):                                                     # util/other.py:102-109
    T = Y.size(3)
    num_pad = 64 - T % 64 if T % 64 != 0 else 0
    return torch.nn.functional.pad(Y, (0, num_pad, 0, 0))


def wav_to_spec(y, factor=0.15, e=0.5):
    """model.py:282-286: y [1, L] -> (Y [1,1,F,Tpad] complex64, norm_factor, T_orig)."""
    T_orig = y.size(1)
    norm_factor = y.abs().max().item()
    y = y / norm_factor
    Y = torch.unsqueeze(spec_fwd(stft(y), factor, e), 0)
    return pad_spec(Y), norm_factor, T_orig


def spec_to_wav(sample, norm_factor, T_orig, factor=0.15, e=0.5):
    """model.py:301-303: sample [1,1,F,Tpad] complex -> x_hat [L]."""
    x_hat = istft(spec_back(sample.squeeze(), factor, e), T_orig)
    return (x_hat * norm_factor).squeeze()
