"""Backend selection for tests: "sim" = the host simulation of the kernel sources (CPU, test
infrastructure), "hip" = the real libstorm_hip.so on cuda:0 (marked gpu)."""
import pytest
import torch

BACKENDS = [pytest.param("sim"), pytest.param("hip", marks=pytest.mark.gpu)]


def setup_backend(kind):
    from storm_amd import _lib
    if kind == "sim":
        from tests.sim.simenv import load_sim
        load_sim()
        return torch.device("cpu")
    assert torch.cuda.is_available(), "gpu test without a GPU"
    _lib._lib, _lib._sim = None, False
    _lib.lib()
    return torch.device("cuda:0")


@pytest.fixture(params=BACKENDS)
def dev(request):
    return setup_backend(request.param)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def tol(dtype, f32, bf16):
    return f32 if dtype == torch.float32 else bf16
