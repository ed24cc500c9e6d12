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


@pytest.fixture
def switch():
    """switch(name, value): set one of the library's A/B switches through its test hook (storm_set_switch; the names are the
    environment variables the table is initialised from) for the duration of the test."""
    from storm_amd import _lib
    saved = []

    def set_(name, value):
        lib = _lib.lib()
        saved.append((lib, name, lib.storm_get_switch(name.encode())))
        _lib.check(lib.storm_set_switch(name.encode(), int(value)), "storm_set_switch")
    yield set_
    for lib, name, v in reversed(saved):
        lib.storm_set_switch(name.encode(), v)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def tol(dtype, f32, bf16):
    return f32 if dtype == torch.float32 else bf16
