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
    assert lib.storm_abi_version() == 2
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


def _run_c_host(tmp_path, lib_path, compiler, extra, golden, ldflags=()):
    """compile tests/c/ncsnpp_host.c, hand it the reference's tiny4 golden input + the seeded state_dict as raw files, run it"""
    import subprocess

    import numpy as np
    import torch

    from oracle import ncsnpp_ref as NR
    g = golden["f2_tiny_nets"]
    cfg = NR.NCSNppConfig(nf=8, input_channels=4)
    sd = NR.seeded_state_dict(cfg, seed=7)
    d = str(tmp_path)
    for i, v in enumerate(sd.values()):
        v.detach().contiguous().numpy().astype(np.float32).tofile(os.path.join(d, f"w{i}.bin"))
    x = g["tiny4_x"]                                        # complex64 [2, 2, 32, 64]
    for j in range(2):
        np.ascontiguousarray(x[:, j]).tofile(os.path.join(d, f"x{j}.bin"))
    g["t"].astype(np.float32).tofile(os.path.join(d, "t.bin"))
    exe = os.path.join(d, "ncsnpp_host")
    libdir, libname = os.path.dirname(lib_path), os.path.basename(lib_path)[3:-3]
    cmd = [compiler] + extra + ["-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "ncsnpp_host.c"), "-o", exe,
                                "-L", libdir, "-l" + libname, "-Wl,-rpath," + libdir] + list(ldflags)
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe, d, "8", "4", "2", "32", "64", "0"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    out = np.fromfile(os.path.join(d, "out.bin"), dtype=np.complex64).reshape(2, 1, 32, 64)
    ref = g["tiny4_y"]
    err = float(np.linalg.norm(out - ref) / np.linalg.norm(ref))
    assert err < 1e-4, err
    return r.stdout


def test_c_host_runs_a_forward_through_the_whole_network_abi(tmp_path, golden):
    """A host written in C (no Python planning / packing / dispatch): storm_ncsnpp_tensor_info -> storm_ncsnpp_create ->
    storm_ncsnpp_workspace_bytes -> storm_ncsnpp_forward, linked against the host simulation of the kernels; its output
    equals the REFERENCE's forward of the tiny4 network (fixture F2, fp32 <= 1e-4)."""
    from tests.sim.build_sim import build
    out = _run_c_host(tmp_path, build(), "gcc", ["-std=c99", "-O1"], golden)
    assert "271 tensors" in out


import pytest  # noqa: E402


@pytest.mark.gpu
def test_c_host_on_the_gpu(tmp_path, golden):
    """the same C host against libstorm_hip.so on the MI355X: plain g++ (-DUSE_HIP: device memory through the HIP runtime
    API, hipMalloc / hipMemcpy), no device code in the host program"""
    from storm_amd.build import build
    _run_c_host(tmp_path, build(), "g++", ["-x", "c++", "-DUSE_HIP", "-D__HIP_PLATFORM_AMD__", "-O1", "-I", "/opt/rocm/include"], golden,
                ldflags=["-L", "/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])
