"""The C-ABI shared library builds for gfx950 without a GPU, loads, and exports every symbol the public headers
declare.  No compute call is made here (no GPU in the CPU-test container)."""
import ctypes as ct
import os
import re

from conftest import ROOT


def declared_symbols():
    names = []
    for hdr in ("cassie2d.h", "cassie_vec.h", "cassie3d_vec.h", "cassie_trpo.h"):
        txt = open(os.path.join(ROOT, "include", hdr)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        for m in re.finditer(r"^[A-Za-z_][\w\s\*]*?\b(\w+)\s*\([^;{]*\)\s*;", txt, flags=re.M):
            names.append(m.group(1))
    return names


def test_library_builds_loads_and_exports():
    from cassierl_amd import build as B
    from cassierl_amd import _lib
    path = B.build()
    assert os.path.exists(path)
    L = ct.CDLL(path)
    syms = declared_symbols()
    assert len(syms) >= 31 and "Cassie2dInit" in syms and "CassieVecStep" in syms
    for s in syms:
        assert hasattr(L, s), "missing export " + s
    assert sorted(set(syms)) == sorted(set(_lib.EXPORTS))


def test_no_cpu_fallback_in_package():
    # the product must not import or link the oracle
    pkg = os.path.join(ROOT, "cassierl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle_py" not in txt and "liboracle" not in txt and "cassie_oracle" not in txt, f
