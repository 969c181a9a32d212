"""The packed physics kernel (and the Cassie3d kernel) solves mj_Euler's implicit-damping system (M + h B) x = g by the fixed-point iteration
x <- M^-1 g - M^-1 h B x (csrc/cassie_kernels_g16.hip, IMPLICIT_DAMPING_SWEEPS).  That is exact to rounding only if
E = M^-1 h B is a strong contraction for EVERY pose; this test pins the bound the kernel's comment states, with the oracle's
mass matrix (oracle/cassie_oracle.c, orc_mass_matrix) over random poses, and the three facts the code relies on."""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as O  # noqa: E402

H = 5e-4


def _table(name, n):
    txt = open(os.path.join(ROOT, "cassierl_amd", "csrc", "cassie2d_planar.h")).read()
    m = re.search(name + r"\[%d\] = \{([^}]*)\}" % n, txt)
    return np.array([float(x) for x in m.group(1).replace("\n", " ").split(",") if x.strip()])


def _sweeps():
    txt = open(os.path.join(ROOT, "cassierl_amd", "csrc", "cassie_kernels_g16.hip")).read()
    return int(re.search(r"constexpr int IMPLICIT_DAMPING_SWEEPS = (\d+);", txt).group(1))


def test_base_dofs_are_undamped():
    # the kernel's matvec only visits columns 3..12
    assert (_table("cp_dof_damping", 13)[:3] == 0).all()


def test_contraction_bound_and_truncation_error():
    damp = _table("cp_dof_damping", 13)
    o = O.Oracle()
    q0, _ = o.state()
    rng = np.random.default_rng(0)
    span = np.array([1, 0.3, 1.5, 1, 1.5, 1.5, 1.5, 1.5, 1, 1.5, 1.5, 1.5, 1.5])
    worst_rho, worst_err = 0.0, 0.0
    n = _sweeps()
    for _ in range(400):
        q = q0 + rng.uniform(-1, 1, 13) * span
        M = np.array(o.mass_matrix(q)).reshape(13, 13)
        Minv = np.linalg.inv(M)
        E = Minv * (H * damp)[None, :]
        worst_rho = max(worst_rho, np.abs(np.linalg.eigvals(E)).max())
        g = rng.normal(size=13) * np.array([100, 100, 50, 50, 50, 20, 20, 5, 50, 50, 20, 20, 5])
        qacc = Minv @ g
        x = qacc.copy()
        for _k in range(n):
            x = qacc - E @ x
        ref = np.linalg.solve(M + np.diag(H * damp), g)
        worst_err = max(worst_err, np.abs(x - ref).max() / np.abs(ref).max())
    assert worst_rho < 0.0395, worst_rho          # DESIGN.md / kernel comment: 0.0393
    assert worst_rho ** n < 1e-16, (worst_rho, n)  # truncation below double rounding
    assert worst_err < 5e-15, worst_err            # what remains is the rounding of the two direct solves being compared


def test_cassie3d_contraction_bound():
    """Same bound for model/cassie3d_stiff.xml (csrc/cassie3d_kernels.hip, IMPLICIT_DAMPING_SWEEPS3): floating base, 20 dofs."""
    txt = open(os.path.join(ROOT, "cassierl_amd", "csrc", "cassie3d_tables.h")).read()
    m = re.search(r"c3_dof_damping\[20\] = \{([^}]*)\}", txt)
    damp = np.array([float(x) for x in m.group(1).replace("\n", " ").split(",") if x.strip()])
    assert (damp[:6] == 0).all()   # the kernel's matvec visits columns 6..19 only
    src = open(os.path.join(ROOT, "cassierl_amd", "csrc", "cassie3d_kernels.hip")).read()
    n = int(re.search(r"constexpr int IMPLICIT_DAMPING_SWEEPS3 = (\d+);", src).group(1))
    o = O.Oracle3D()
    q0, _ = o.state()
    rng = np.random.default_rng(1)
    worst = 0.0
    for _ in range(300):
        q = q0.copy()
        q[:3] += rng.uniform(-1, 1, 3) * 0.3
        quat = q[3:7] + rng.normal(size=4) * 0.5
        q[3:7] = quat / np.linalg.norm(quat)
        q[7:] += rng.uniform(-1, 1, len(q) - 7)
        M = np.array(o.mass_matrix(q))
        worst = max(worst, np.abs(np.linalg.eigvals(np.linalg.solve(M, np.diag(H * damp)))).max())
    assert worst < 0.0395, worst
    assert worst ** n < 1e-16, (worst, n)
