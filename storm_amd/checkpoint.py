"""Lightning-free reader for the reference's ``.ckpt`` files (pytorch-lightning 1.8.3 + torch-ema 0.3,
requirements.txt:9,15): a pickled dict with ``state_dict``, ``hyper_parameters`` and ``ema``.
Classes of packages that are not installed here (pytorch_lightning's AttributeDict, the reference's
``sgmse.data_module.SpecsDataModule`` stored in hyper_parameters['data_module_cls']) are mapped to
local equivalents / plain containers while unpickling."""
import pickle

import torch


class _AttributeDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


class _Opaque:
    """stand-in for any class we cannot import (kept so unpickling succeeds; never used)"""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.__dict__["state"] = state


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith("sgmse"):
            if name == "SpecsDataModule":
                from .data_module import SpecsDataModule
                return SpecsDataModule
            return _Opaque
        if module.startswith("pytorch_lightning") or module.startswith("lightning"):
            return _AttributeDict if name == "AttributeDict" else _Opaque
        try:
            return super().find_class(module, name)
        except (ImportError, AttributeError):
            return _Opaque


class _PickleModule:
    __name__ = "storm_amd_pickle"
    Unpickler = _Unpickler
    load = staticmethod(lambda f, **kw: _Unpickler(f, **kw).load())
    loads = staticmethod(pickle.loads)
    dump = staticmethod(pickle.dump)
    dumps = staticmethod(pickle.dumps)
    HIGHEST_PROTOCOL = pickle.HIGHEST_PROTOCOL


def load_checkpoint_file(path, map_location="cpu"):
    ckpt = torch.load(path, map_location=map_location, pickle_module=_PickleModule, weights_only=False)
    if "state_dict" not in ckpt:
        raise ValueError(f"{path}: not a Lightning checkpoint (no 'state_dict')")
    return ckpt
