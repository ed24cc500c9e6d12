"""Import the reference (sp-uhh/storm) on CPU in the build container.

Container-only helper for ``oracle/make_golden.py``: the reference lives at
/root/reference, which does not exist on the GPU box.  Nothing here is copied
from the reference; we only stub the third-party modules it imports that are
absent from this image (pytorch_lightning, torch_ema, torchaudio, ...), skip
its nvcc JIT build (the CPU path uses its own ``upfirdn2d_native``,
sgmse/backbones/ncsnpp_utils/op/upfirdn2d.py:145-150) and make ``.cuda()`` an
identity (sgmse/model.py:285 hard-codes it).
"""
import os
import sys
import types

REF_ROOT = os.environ.get("STORM_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "sgmse"))


def import_reference():
    """Returns the imported ``sgmse`` package namespace as a dict of modules."""
    import torch

    if not reference_available():
        raise RuntimeError(f"reference not found at {REF_ROOT}")
    sys.dont_write_bytecode = True
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)

    def stub(name, **attrs):
        if name in sys.modules:
            return
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m

    import torch.utils.cpp_extension as ce
    ce.load = lambda *a, **k: types.SimpleNamespace()

    class LightningModule(torch.nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

    class LightningDataModule:
        def __init__(self, *a, **k):
            pass

    stub("pytorch_lightning", LightningModule=LightningModule,
         LightningDataModule=LightningDataModule)

    class ExponentialMovingAverage:
        def __init__(self, params, decay):
            self.collected_params = None

        def to(self, *a, **k):
            pass

        def store(self, params):
            self.collected_params = [p.clone() for p in params]

        def copy_to(self, params):
            pass

        def restore(self, params):
            pass

        def update(self, params):
            pass

        def state_dict(self):
            return {}

        def load_state_dict(self, d):
            pass

    stub("torch_ema", ExponentialMovingAverage=ExponentialMovingAverage)
    for n in ("wandb", "h5py", "soundfile"):
        stub(n)
    stub("torchaudio", load=None, save=None)
    stub("pydub", AudioSegment=None)
    stub("pesq", pesq=None)
    stub("pystoi", stoi=None)
    torch.Tensor.cuda = lambda self, *a, **k: self

    import sgmse.model as model
    import sgmse.sdes as sdes
    import sgmse.sampling as sampling
    import sgmse.data_module as data_module
    import sgmse.backbones.ncsnpp as ncsnpp
    import sgmse.backbones.ncsnpp_utils.layerspp as layerspp
    import sgmse.backbones.ncsnpp_utils.up_or_down_sampling as updown
    import sgmse.util.other as other
    import sgmse.util.inference as inference   # (pesq / pystoi are stubs: callers set inference.pesq / .stoi to constants)
    torch.autograd.set_detect_anomaly(False)   # model.py:22 turns it on at import
    return dict(model=model, sdes=sdes, sampling=sampling, data_module=data_module,
                ncsnpp=ncsnpp, layerspp=layerspp, updown=updown, other=other, inference=inference)
