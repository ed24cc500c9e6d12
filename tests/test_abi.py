"""The C-ABI library builds for gfx950, loads, and exports every symbol include/storm_hip.h declares
(no compute calls: CPU suite)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "storm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(storm_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    from storm_amd.build import build
    path = build()
    lib = ctypes.CDLL(path)
    names = declared_symbols()
    assert len(names) >= 28
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    lib.storm_abi_version.restype = ctypes.c_int
    assert lib.storm_abi_version() == 1
    lib.storm_last_error.restype = ctypes.c_char_p
    assert lib.storm_last_error() is not None


def test_python_binding_covers_the_header():
    from storm_amd import _lib
    assert set(declared_symbols()) == set(_lib.EXPORTS)


def test_product_does_not_import_the_oracle():
    """the oracle is test infrastructure: nothing under storm_amd/ (or bench.py outside cpu_baseline) may use it"""
    import glob
    for f in glob.glob(os.path.join(ROOT, "storm_amd", "**", "*.py"), recursive=True):
        assert "oracle" not in open(f).read().replace("# oracle", ""), f
    bench = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"from oracle|import oracle", bench)]
    base = bench.index("def cpu_baseline")
    nxt = bench.index("\ndef ", base + 1)
    assert uses and all(base < u < nxt for u in uses)
