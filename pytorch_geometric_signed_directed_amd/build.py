"""Build recipe for csrc/libpygsd_hip.so (hipcc, gfx950 only).  Used by __graft_entry__.build()."""
import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libpygsd_hip.so")
SOURCES = ["runtime.hip", "spmm.hip", "dense.hip", "tall.hip", "build.hip", "laplacian.hip", "magop.hip", "attention.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))]
    deps.append(os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "pygsd_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """Compile the HIP sources in-tree.  Returns the path of the shared library."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + FLAGS + ["-o", LIB] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, cwd=CSRC, check=True)
    return LIB
