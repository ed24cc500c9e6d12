"""Build libstorm_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m storm_amd.build [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(CSRC, "libstorm_hip.so")
SOURCES = ["abi", "conv_igemm", "conv_pipe", "conv_pipe128", "conv_thin", "conv_narrow", "attention", "ncsnpp_graph", "norm_resample", "elementwise", "pyramid", "sde", "spectral", "program"]
PROF_SOURCES = SOURCES + ["conv_duo"]   # conv_duo.hip (LAB_NOTES 2.3: built, measured, a tie - never dispatched) lives in the profiling library only
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "conv_index.h"), os.path.join(CSRC, "conv_params.h"), os.path.join(CSRC, "conv_pipe_common.h"), os.path.join(CSRC, "conv_epilogue.h"), os.path.join(CSRC, "conv_dispatch_table.h"), os.path.join(CSRC, "hw.h"),
           os.path.join(os.path.dirname(HERE), "include", "storm_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def up_to_date():
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, s + ".hip") for s in SOURCES] + HEADERS
    return all(os.path.getmtime(d) <= t for d in deps)


PROF_OUT = os.path.join(CSRC, "libstorm_hip_prof.so")


def build(force=False, verbose=False, profiling=False):
    """profiling=True builds libstorm_hip_prof.so with -DSTORM_PROFILING: the same library plus wave-timeline stamps and
    work-skipping kernel instantiations for tools/ (selected through STORM_LIB=<path>); never loaded by the product."""
    out = PROF_OUT if profiling else OUT
    if not profiling and not force and up_to_date():
        return OUT
    cc = hipcc()
    bdir = os.path.join(CSRC, "build_prof" if profiling else "build")
    os.makedirs(bdir, exist_ok=True)
    # one builder at a time (test workers / ranks that find the library stale at the same moment): the others wait, then find it fresh
    import fcntl
    with open(os.path.join(bdir, ".lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if not profiling and not force and up_to_date():
            return OUT
        return _build_locked(cc, bdir, out, force, verbose, profiling)


def _build_locked(cc, bdir, out, force, verbose, profiling):
    # (STORM_EXTRA_DEFS: extra -D switches of one-off experiments, profiling build only)
    flags = FLAGS + (["-DSTORM_PROFILING", "-DSTORM_WITH_DUO"] + os.environ.get("STORM_EXTRA_DEFS", "").split() if profiling else [])
    hdr_time = max(os.path.getmtime(h) for h in HEADERS + [os.path.abspath(__file__)])
    # every object is keyed on a hash of its flags (an object built with other -D switches can never be linked by mistake, whatever
    # failed or was interrupted in between) and appears under its name only when its compile succeeded (tmp file + atomic replace)
    import hashlib
    tag = hashlib.sha256(" ".join(flags).encode()).hexdigest()[:10]

    def compile_one(s):
        src, obj = os.path.join(CSRC, s + ".hip"), os.path.join(bdir, f"{s}.{tag}.o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(hdr_time, os.path.getmtime(src)):
            return obj                                     # this object is current: only edited sources recompile
        tmp = f"{obj}.{os.getpid()}.tmp"
        cmd = [cc] + flags + ["-c", src, "-o", tmp]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            if os.path.exists(tmp):
                os.remove(tmp)
            raise RuntimeError(f"hipcc failed on {s}.hip:\n{r.stdout.decode()}")
        os.replace(tmp, obj)
        if verbose and r.stdout:
            sys.stderr.write(r.stdout.decode())
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, PROF_SOURCES if profiling else SOURCES))
    for f in os.listdir(bdir):                             # objects of other flag sets / older layouts: not ours to link, not worth keeping
        if f.endswith(".o") and f".{tag}.o" not in f:
            os.remove(os.path.join(bdir, f))
    r = subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out + ".tmp"] + objs,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout.decode()}")
    os.replace(out + ".tmp", out)                          # (atomic: never a half-written library under the product's name)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, profiling="--profiling" in sys.argv))
