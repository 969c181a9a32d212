"""Host-side mirror (structs, converters, gait table) against golden vectors generated from the
reference's importable Python modules (tests/golden/make_golden.py)."""
import ctypes as ct
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from cassierl_amd import structs as S
from cassierl_amd.trajectory import Cassie2dTraj


@pytest.fixture(scope="module")
def kat():
    with open(os.path.join(GOLDEN, "structs_kat.json")) as f:
        return json.load(f)


def test_struct_sizes_and_offsets(kat):
    # RobotInterface.h:14-50 == cassie2d_structs.py:5-51 : 48/48/56/48/208/144 bytes, no padding
    assert kat["sizes"] == {"ControllerTorque": 48, "ControllerForce": 48, "ControllerOsc": 56, "ControllerPd": 48,
                            "StateGeneral": 208, "StateOperationalSpace": 144}
    for name, size in kat["sizes"].items():
        cls = getattr(S, name)
        assert ct.sizeof(cls) == size
        for field, off in kat["offsets"][name].items():
            assert getattr(cls, field).offset == off


def test_converters_match_reference(kat):
    cv = S.InterfaceStructConverter()
    for c in kat["cases"]:
        g = cv.array_to_general_state(np.array(c["general_in"]))
        got = [list(g.base_pos), list(g.base_vel), list(g.left_pos), list(g.left_vel), list(g.right_pos), list(g.right_vel)]
        assert got == c["general_struct"]
        assert cv.general_state_to_array(g).tolist() == c["general_roundtrip"]
        x = S.StateOperationalSpace()
        for k, f in enumerate(("body_x", "body_xd", "left_x", "left_xd", "right_x", "right_xd")):
            for i in range(3):
                getattr(x, f)[i] = c["op_vals"][3 * k + i]
        arr = cv.operational_state_to_array(x)
        assert arr.tolist() == c["op_array"]
        assert cv.operational_state_array_to_pos_invariant_array(arr).tolist() == c["pos_invariant"]
        a = np.array(c["action_in"])
        osc = cv.array_to_operational_action(a)
        assert [list(osc.body_xdd), list(osc.left_xdd), list(osc.right_xdd), osc.pitch_add] == c["osc"]
        assert list(cv.array_to_pd_action(a).angles) == c["pd"]
        assert list(cv.array_to_torque_action(a).torques) == c["torque"]


def test_trajectory_lookup_semantics(kat, traj):
    assert kat["traj"]["shape_qpos"] == [1682, 13] and kat["traj"]["shape_qvel"] == [1682, 13] and kat["traj"]["shape_torque"] == [1682, 6]
    assert abs(kat["traj"]["tmax"] - 0.8405) < 1e-6
    tr = Cassie2dTraj.from_arrays(traj["time"], traj["qpos"], traj["qvel"], traj["torque"])
    np.testing.assert_allclose(tr.qpos[0], [0, 1.02381778, 0.03906015, 0.44288826, 0.95879775, 0.02641815, -0.28239858,
                                            -1.57079633, 0.35547572, 0.99434882, -0.00622221, -0.10587036, -1.57079633], atol=5e-9)
    for t, i, q in zip(traj["grid"], traj["grid_index"], traj["grid_qpos"]):
        assert tr.index(t) == i  # includes wrap-around at tmax
        assert np.array_equal(tr.state(t)[0], q)


@pytest.mark.skipif(not os.path.exists("/root/reference/rllab/trajectory/stepdata.bin"), reason="reference data only in the build container")
def test_trajectory_conversion_from_stepdata(traj):
    tr = Cassie2dTraj("/root/reference/rllab/trajectory/stepdata.bin")
    assert np.array_equal(tr.time, traj["time"])
    np.testing.assert_allclose(tr.qpos, traj["qpos"], rtol=0, atol=1e-15)
    assert np.array_equal(tr.qvel, traj["qvel"]) and np.array_equal(tr.torque, traj["torque"])


def test_pd_targets_match_step_traj_export_csv(traj):
    from cassierl_amd.trajectory import pd_targets
    tr = Cassie2dTraj.from_arrays(traj["time"], traj["qpos"], traj["qvel"], traj["torque"])
    for t, row in zip(traj["pd_t"], traj["pd_rows"]):
        ang, vel = pd_targets(tr, t)
        np.testing.assert_allclose(np.concatenate([ang, vel]), row, rtol=0, atol=1e-14)


def test_bench_tier_rule_and_cpu_limit_helpers():
    """bench.py's host-side helpers: the first-tier rule it mirrors from the library (whole rounds of one wavefront per SIMD), the thread
    ladder of the CPU-baseline leg and the cgroup probe (no GPU, no oracle)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert "g16" in b.dominant_kernel(4096) and "env_step_leg_kernel" in b.dominant_kernel(8192) and "env_step_leg_kernel" in b.dominant_kernel(32768)
    assert "env_step_duo_kernel" in b.dominant_kernel(40000) and "env_step_duo_kernel" in b.dominant_kernel(65536)
    assert "env_step_duo_kernel" in b.dominant_kernel(98304) and "env_step_duo_kernel" in b.dominant_kernel(131072) and "env_step_duo_kernel" in b.dominant_kernel(524288)   # 98 304: split (r06)
    assert b.thread_ladder(256) == [256, 128, 64, 32, 16, 8, 4, 2, 1] and b.thread_ladder(1) == [1] and b.thread_ladder(6) == [6, 3, 1]
    lim = b.cpu_limits()
    assert set(lim) == {"cpu_count", "affinity", "cpu_quota", "cpu_quota_source"} and (lim["cpu_quota"] is None or lim["cpu_quota"] > 0)
    assert b.preroll_count(65536) == 300 and b.preroll_count(4096) == 1000
