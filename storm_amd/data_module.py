"""Spectral front / back end — the inference subset of the reference SpecsDataModule
(sgmse/data_module.py:140-223: constructor knobs, window cache, spec_fwd/spec_back, stft/istft)
on the HIP kernels of csrc/spectral.hip.  Dataset / DataLoader code is out of scope (training)."""
import torch

from . import ops


def get_window(window_type, window_length):
    """data_module.py:19-25"""
    if window_type == "sqrthann":
        return torch.sqrt(torch.hann_window(window_length, periodic=True))
    elif window_type == "hann":
        return torch.hann_window(window_length, periodic=True)
    raise NotImplementedError(f"Window type {window_type} not implemented!")


class SpecsDataModule:
    def __init__(self, base_dir="", format="wsj0", spatial_channels=1, batch_size=8, n_fft=510, hop_length=128,
                 num_frames=256, window="hann", num_workers=8, dummy=False, spec_factor=0.15,
                 spec_abs_exponent=0.5, gpu=True, return_time=False, **kwargs):
        self.base_dir, self.format, self.spatial_channels, self.batch_size = base_dir, format, spatial_channels, batch_size
        self.n_fft, self.hop_length, self.num_frames = n_fft, hop_length, num_frames
        self.window_type = window
        self.window = get_window(window, n_fft)
        self.num_workers, self.dummy = num_workers, dummy
        self.spec_factor, self.spec_abs_exponent = spec_factor, spec_abs_exponent
        self.gpu, self.return_time, self.kwargs = gpu, return_time, kwargs

    # ---- transforms (data_module.py:182-193) ---------------------------------------------------
    def spec_fwd(self, spec):
        return ops.spec_transform(spec, self.spec_factor, self.spec_abs_exponent, inverse=False)

    def spec_back(self, spec):
        return ops.spec_transform(spec, self.spec_factor, self.spec_abs_exponent, inverse=True)

    @property
    def stft_kwargs(self):
        return {**self.istft_kwargs, "return_complex": True}

    @property
    def istft_kwargs(self):
        return dict(n_fft=self.n_fft, hop_length=self.hop_length, window=self.window, center=True)

    # ---- torch.stft / torch.istft replacements (data_module.py:217-223) -------------------------
    def stft(self, sig):
        """sig [C, L] (or [L]) float32 -> complex64 [C, F, frames]"""
        squeeze = sig.dim() == 1
        x = sig.unsqueeze(0) if squeeze else sig
        X = ops.stft(x.float().contiguous(), None, n_fft=self.n_fft, hop=self.hop_length, window=self.window_type)
        return X[0] if squeeze else X

    def istft(self, spec, length=None):
        """spec [C, F, T] (or [F, T]) complex64 -> [C, length]"""
        squeeze = spec.dim() == 2
        X = spec.unsqueeze(0) if squeeze else spec
        if length is None:
            length = self.hop_length * (X.shape[-1] - 1)
        w = ops.istft(X.contiguous(), int(length), None, n_fft=self.n_fft, hop=self.hop_length, window=self.window_type)
        return w[0] if squeeze else w

    # ---- fused batched forms used by enhance() ---------------------------------------------------
    def padded_frames(self, length, pad_to=64):
        """frames of one utterance after pad_spec (util/other.py:102-109): utterances with equal values can share a batch"""
        return ops.round_up(1 + int(length) // self.hop_length, pad_to)

    def wav_to_spec(self, wav, pad_to=64, lengths=None):
        """wav [B, L] -> (Y [B,1,F,Tpad] = pad_spec(spec_fwd(stft(wav / peak))), peak [B])  in two launches.
        lengths: per-row sample counts of a ragged batch (rows zero filled past them, one padded frame count for all)."""
        if lengths is not None:
            tp = {self.padded_frames(v, pad_to) for v in lengths}
            if len(tp) != 1 or max(int(v) for v in lengths) != wav.shape[1]:
                raise ValueError(f"a ragged batch must share one padded frame count and be as wide as its longest row: {sorted(tp)}")
        peak = ops.peak_abs(wav, lengths)
        Y = ops.stft(wav, peak, n_fft=self.n_fft, hop=self.hop_length, spec_factor=self.spec_factor,
                     spec_abs_exponent=self.spec_abs_exponent, pad_to=pad_to, window=self.window_type, lengths=lengths)
        return Y.unsqueeze(1), peak

    def spec_to_wav(self, spec, length, peak=None, lengths=None):
        """spec [B,1,F,T] -> wav [B, length] = istft(spec_back(spec), length) * peak."""
        return ops.istft(spec[:, 0].contiguous(), int(length), peak, n_fft=self.n_fft, hop=self.hop_length,
                         spec_factor=self.spec_factor, spec_abs_exponent=self.spec_abs_exponent, window=self.window_type,
                         lengths=lengths)

    @staticmethod
    def add_argparse_args(parser):
        parser.add_argument("--n_fft", type=int, default=510)
        parser.add_argument("--hop_length", type=int, default=128)
        parser.add_argument("--num_frames", type=int, default=256)
        parser.add_argument("--window", type=str, choices=("sqrthann", "hann"), default="hann")
        parser.add_argument("--spec_factor", type=float, default=0.33)
        parser.add_argument("--spec_abs_exponent", type=float, default=0.5)
        return parser
