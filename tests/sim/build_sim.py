"""TEST INFRASTRUCTURE: build tests/sim/libstorm_sim.so — the kernel sources of storm_amd/csrc
compiled for the HOST with the fiber shim (hip_host_shim.h), exporting the same C ABI as
libstorm_hip.so but operating on host memory.  Used only by CPU tests to exercise kernel index
math / launch code without a GPU."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "storm_amd", "csrc")
OUT = os.path.join(HERE, "libstorm_sim.so")
CXX = os.environ.get("STORM_SIM_CXX", "/opt/rocm/lib/llvm/bin/clang++")
SOURCES = ["abi", "conv_igemm", "conv_pipe", "conv_pipe128", "conv_duo", "conv_thin", "conv_narrow", "attention", "ncsnpp_graph", "norm_resample", "elementwise", "pyramid", "sde", "spectral", "program"]


def build(force=False):
    srcs = [os.path.join(CSRC, s + ".hip") for s in SOURCES] + [os.path.join(HERE, "simrt.cpp")]
    deps = srcs + [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "conv_index.h"), os.path.join(CSRC, "conv_params.h"), os.path.join(CSRC, "conv_pipe_common.h"), os.path.join(CSRC, "conv_epilogue.h"), os.path.join(CSRC, "conv_dispatch_table.h"), os.path.join(CSRC, "hw.h"),
                   os.path.join(HERE, "hip_host_shim.h"), os.path.join(ROOT, "include", "storm_hip.h")]
    fresh = lambda: os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps)  # noqa: E731
    if not force and fresh():
        return OUT
    # one builder at a time (the CPU suite runs on several pytest-xdist workers, each of which lands here on its first launch): the others
    # wait for the lock and then find the library fresh
    import fcntl
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    with open(os.path.join(HERE, "build", ".lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if not force and fresh():
            return OUT
        return _build_locked(srcs)


def _build_locked(srcs):
    objs = []
    bdir = os.path.join(HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    procs = []
    for s in srcs:
        o = os.path.join(bdir, os.path.basename(s) + ".o")
        objs.append(o)
        cmd = [CXX, "-x", "c++", "-std=c++17", "-O2", "-g", "-fPIC", "-DSTORM_HOST_SIM", "-DSTORM_WITH_DUO", "-I", HERE, "-I", CSRC,
               "-Wno-unknown-attributes", "-Wno-unknown-pragmas", "-Wno-pass-failed", "-c", s, "-o", o]
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f"sim build failed for {s}")
    subprocess.check_call([CXX, "-shared", "-fPIC", "-o", OUT + ".tmp"] + objs + ["-lpthread"])
    os.replace(OUT + ".tmp", OUT)                           # (atomic: a process that already mapped the old library keeps its copy)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
