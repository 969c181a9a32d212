#!/usr/bin/env python3
"""Generate golden fixtures from the reference's importable Python modules.

Runs ONLY in the container that has /root/reference (the reference cannot travel).  Imports
rllab/envs/cassie2d_structs.py and rllab/envs/cassie2d_trajectory.py (numpy + ctypes only),
and writes data-only fixtures next to this script:
  traj2d.npz        Cassie2dTraj('stepdata.bin'): time[1682], qpos[1682,13], qvel[1682,13], torque[1682,6],
                    plus state(t)/action(t) lookups on a time grid (index behaviour incl. wrap at tmax)
  structs_kat.json  ctypes struct sizes/offsets and converter outputs on seeded random inputs
"""
import ctypes as ct
import json
import os
import sys

import numpy as np

REF = "/root/reference/rllab/envs"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
import cassie2d_structs as S  # noqa: E402
import cassie2d_trajectory as T  # noqa: E402


def main():
    tr = T.Cassie2dTraj(os.path.join(REF, "..", "trajectory", "stepdata.bin"))
    grid = np.concatenate([np.arange(0, 2.0, 0.0371), [0.0, 0.84049, 0.8405, 0.84051, 1.681, 5.0, 7.3]])
    st_q = np.array([tr.state(t)[0] for t in grid])
    st_v = np.array([tr.state(t)[1] for t in grid])
    ac_t = np.array([tr.action(t)[2] for t in grid])
    tmax = tr.time[-1]
    idx = np.array([int((t % tmax) / tmax * len(tr.time)) for t in grid])
    # PD targets of step_traj_export_csv (cassie2d.py:234-257), evaluated with the reference's own arithmetic
    pd_t = np.arange(0, 2.0, 0.0137)
    pd_rows = []
    for t in pd_t:
        qpos, qvel = tr.state(t)
        torques = tr.action(t)[2]
        joints, kp, kd = [3, 4, 6, 8, 9, 11], 10.0, 5.0
        angles = [(torques[i] - kd * (0.0 - qvel[joints[i]])) / kp + qpos[joints[i]] for i in range(len(joints))]
        pd_rows.append(angles + [qvel[j] for j in joints])
    np.savez_compressed(os.path.join(HERE, "traj2d.npz"), time=tr.time, qpos=tr.qpos, qvel=tr.qvel, torque=tr.torque,
                        grid=grid, grid_index=idx, grid_qpos=st_q, grid_qvel=st_v, grid_torque=ac_t,
                        pd_t=pd_t, pd_rows=np.array(pd_rows))
    # ---- structs
    kat = {"sizes": {}, "offsets": {}}
    for name in ("ControllerTorque", "ControllerForce", "ControllerOsc", "ControllerPd", "StateGeneral", "StateOperationalSpace"):
        cls = getattr(S, name)
        kat["sizes"][name] = ct.sizeof(cls)
        kat["offsets"][name] = {f[0]: getattr(cls, f[0]).offset for f in cls._fields_}
    rng = np.random.default_rng(123)
    cv = S.InterfaceStructConverter()
    cases = []
    for _ in range(5):
        s26 = rng.normal(size=26)
        g = cv.array_to_general_state(s26)
        back = cv.general_state_to_array(g)
        x = S.StateOperationalSpace()
        vals = rng.normal(size=18)
        for k, f in enumerate(("body_x", "body_xd", "left_x", "left_xd", "right_x", "right_xd")):
            for i in range(3):
                getattr(x, f)[i] = vals[3 * k + i]
        arr = cv.operational_state_to_array(x)
        inv = cv.operational_state_array_to_pos_invariant_array(arr)
        a7 = rng.normal(size=7)
        osc = cv.array_to_operational_action(a7)
        pd = cv.array_to_pd_action(a7)
        tq = cv.array_to_torque_action(a7)
        cases.append(dict(general_in=s26.tolist(), general_roundtrip=back.tolist(),
                          general_struct=[list(g.base_pos), list(g.base_vel), list(g.left_pos), list(g.left_vel),
                                          list(g.right_pos), list(g.right_vel)],
                          op_vals=vals.tolist(), op_array=arr.tolist(), pos_invariant=inv.tolist(), action_in=a7.tolist(),
                          osc=[list(osc.body_xdd), list(osc.left_xdd), list(osc.right_xdd), osc.pitch_add],
                          pd=list(pd.angles), torque=list(tq.torques)))
    kat["cases"] = cases
    kat["traj"] = dict(shape_qpos=list(tr.qpos.shape), shape_qvel=list(tr.qvel.shape), shape_torque=list(tr.torque.shape),
                       tmax=float(tmax), qpos0=tr.qpos[0].tolist())
    with open(os.path.join(HERE, "structs_kat.json"), "w") as f:
        json.dump(kat, f, indent=1)
    print("tmax", tmax, "qpos0", tr.qpos[0])


if __name__ == "__main__":
    main()
