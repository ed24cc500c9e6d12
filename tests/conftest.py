import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # an xdist worker keeps to its share of the cores: four workers with torch's default of one OpenMP thread per core each spend their
    # time waiting for one another (single tests ran up to 3 x slower depending on what ran beside them)
    wi = getattr(config, "workerinput", None)
    if wi is not None:
        import torch
        torch.set_num_threads(max(1, (os.cpu_count() or 1) // max(1, int(wi.get("workercount", 1)))))


# ---- the CPU suite on several workers by default -------------------------------------------------------------------------------
# `python -m pytest tests -q -m "not gpu"` is ~270 tests, most of them kernels and whole sampler runs on the host simulator: half an hour
# on one core, a quarter of that on four.  When the run is the CPU suite (-m "not gpu"), pytest-xdist is installed, nobody chose -n and the
# host has the cores, four workers are started (STORM_TEST_WORKERS = N overrides, 0 / 1 = serial).  Never for -m gpu: tests of one GPU run
# one after the other (kernels of concurrent queues are not safe on this platform: profiles/r05_concurrent_streams_corruption.txt).
@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    if hasattr(config, "workerinput") or not config.pluginmanager.hasplugin("xdist"):
        return None
    opt = config.option
    if getattr(opt, "numprocesses", None) is not None or getattr(opt, "markexpr", "") != "not gpu":
        return None
    if getattr(opt, "usepdb", False) or getattr(opt, "collectonly", False):
        return None
    want = os.environ.get("STORM_TEST_WORKERS", "")
    n = int(want) if want.isdigit() else min(4, (os.cpu_count() or 1) // 2)
    if n >= 2:
        opt.numprocesses = n
        opt.dist = "load"
    return None


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    class G:
        def __getitem__(self, name):
            return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return G()


# ---- measured parity errors as an artefact -----------------------------------------------------------------------------------
# The golden tests PRINT what they measured ("... rel-L2 vs reference 1.08e-06 ..."); a quiet run keeps only "N passed".  Every
# such line of a passing or failing test is collected here and written to gpurun_out/parity.json (STORM_PARITY_JSON overrides
# the path) when the session ends: scripts/gpu_round.sh copies it to profiles/rNN_parity.json, so the measured numbers - not only
# the bounds the assertions hold them to - survive the GPU box.
_PARITY = []


def pytest_runtest_logreport(report):
    import re
    if report.when != "call":
        return
    for line in (getattr(report, "capstdout", "") or "").splitlines():
        if "rel-L2" in line or "nfev" in line:
            nums = [float(v) for v in re.findall(r"(?<![\w.])\d+\.\d+e[-+]\d+", line)]
            _PARITY.append({"test": report.nodeid, "outcome": report.outcome, "line": line.strip(), "values": nums})


def pytest_sessionfinish(session, exitstatus):
    import json
    if not _PARITY or hasattr(session.config, "workerinput"):          # (xdist workers forward their reports to the controller)
        return
    path = os.environ.get("STORM_PARITY_JSON") or os.path.join(ROOT, "gpurun_out", "parity.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        import torch
        dev = torch.cuda.get_device_name(0) if torch.cuda.is_available() else "cpu (host simulation of the kernel sources)"
        with open(path, "w") as f:
            json.dump({"device": dev, "exitstatus": int(exitstatus), "records": _PARITY}, f, indent=1)
    except OSError:
        pass
