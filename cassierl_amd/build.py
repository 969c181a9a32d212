"""Build the HIP extension (libcassie2d.so) in-tree for gfx950.

hipcc cross-compiles without a GPU, so this runs in the CPU-only container as well as on
the MI355X box.  The shared object is kept in cassierl_amd/lib/ (git-ignored, shipped by gpurun).
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libcassie2d.so")
SOURCES = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))  # cassie_cabi.hip includes the rest
HEADERS = [os.path.join(os.path.dirname(HERE), "include", f) for f in ("cassie2d.h", "cassie_vec.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile csrc/cassie_cabi.hip (which includes the kernels) into lib/libcassie2d.so."""
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [HIPCC] + FLAGS + ["-o", LIB, os.path.join(CSRC, "cassie_cabi.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
