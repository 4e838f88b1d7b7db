"""Build recipe for csrc/libpygsd_hip.so (hipcc, gfx950 only).  Used by __graft_entry__.build().

Every .hip source is compiled to its own object (in parallel; only the ones older than their inputs) and the objects are
linked into the one shared library -- a change to one kernel file costs that file's compile, not all of them."""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libpygsd_hip.so")
OBJ = os.path.join(CSRC, "build")
SOURCES = ["runtime.hip", "spmm.hip", "dense.hip", "tall.hip", "gram.hip", "gemm.hip", "build.hip", "laplacian.hip", "magop.hip", "attention.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
HEADER = os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "pygsd_hip.h")


def _shared_deps():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")] + [HEADER, os.path.abspath(__file__)]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _stale():
    return _newer(LIB, [os.path.join(CSRC, f) for f in SOURCES] + _shared_deps())


def build_library(force=False, verbose=False):
    """Compile the HIP sources in-tree.  Returns the path of the shared library.

    Safe under N ranks started from a source checkout (`_cabi.lib()` builds on first use): the whole build runs under an
    exclusive file lock (`csrc/build/.lock`), staleness is re-checked once the lock is held (the rank that waited finds the
    library its peer just linked), and every object and the library itself are written to `*.tmp.<pid>` and moved into
    place with an atomic rename -- a reader can never `CDLL` or link a half-written file."""
    if not force and not _stale():
        return LIB
    import fcntl
    os.makedirs(OBJ, exist_ok=True)
    with open(os.path.join(OBJ, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():
                return LIB
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    shared = _shared_deps()
    pid = os.getpid()

    def run_into(cmd, target):
        tmp = f"{target}.tmp.{pid}"
        if verbose:
            print(" ".join(cmd + ["-o", target]), flush=True)
        try:
            subprocess.run(cmd + ["-o", tmp], cwd=CSRC, check=True)
            os.replace(tmp, target)
        finally:
            if os.path.exists(tmp):
                os.unlink(tmp)

    def compile_one(src):
        obj = os.path.join(OBJ, src[:-4] + ".o")
        if force or _newer(obj, [os.path.join(CSRC, src)] + shared):
            run_into([hipcc] + FLAGS + ["-c", src], obj)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    run_into([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs, LIB)
    return LIB
