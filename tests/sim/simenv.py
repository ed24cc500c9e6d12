"""TEST INFRASTRUCTURE: bind storm_amd to the host simulation library (CPU tests only)."""
import pytest


def load_sim():
    from tests.sim.build_sim import build
    from storm_amd import _lib
    path = build()
    _lib._load_for_tests(path, sim=True)
    return _lib
