"""ABI structs and array converters: host-side counterpart of rllab/envs/cassie2d_structs.py
(field order = src/Cassie2d/RobotInterface.h:14-50)."""
import ctypes as ct

import numpy as np


class ControllerTorque(ct.Structure):
    _fields_ = [("torques", ct.c_double * 6)]


class ControllerForce(ct.Structure):
    _fields_ = [("left_force", ct.c_double * 3), ("right_force", ct.c_double * 3)]


class ControllerOsc(ct.Structure):
    _fields_ = [("body_xdd", ct.c_double * 2), ("left_xdd", ct.c_double * 2), ("right_xdd", ct.c_double * 2),
                ("pitch_add", ct.c_double)]


class ControllerPd(ct.Structure):
    _fields_ = [("angles", ct.c_double * 6)]


class StateGeneral(ct.Structure):
    _fields_ = [("base_pos", ct.c_double * 3), ("base_vel", ct.c_double * 3), ("left_pos", ct.c_double * 5),
                ("left_vel", ct.c_double * 5), ("right_pos", ct.c_double * 5), ("right_vel", ct.c_double * 5)]


class StateOperationalSpace(ct.Structure):
    _fields_ = [("body_x", ct.c_double * 3), ("body_xd", ct.c_double * 3), ("left_x", ct.c_double * 3),
                ("left_xd", ct.c_double * 3), ("right_x", ct.c_double * 3), ("right_xd", ct.c_double * 3)]


_OP_FIELDS = ("body_x", "body_xd", "left_x", "left_xd", "right_x", "right_xd")


class InterfaceStructConverter:
    """Same method names and array layouts as the reference converter (cassie2d_structs.py:54-122)."""

    def operational_state_to_array(self, state):
        return np.concatenate([np.array(getattr(state, f)[:], dtype=np.double) for f in _OP_FIELDS])

    def operational_state_array_to_pos_invariant_array(self, state_array):
        s = np.zeros(26, dtype=np.double)
        s[:17] = np.asarray(state_array)[1:18]
        s[5] -= state_array[0]
        s[11] -= state_array[0]
        return s

    def general_state_to_array(self, state):
        return np.concatenate([np.array(state.base_pos[:]), np.array(state.base_vel[:]), np.array(state.left_pos[:]),
                               np.array(state.left_vel[:]), np.array(state.right_pos[:]), np.array(state.right_vel[:])])

    def array_to_general_state(self, s):
        st = StateGeneral()
        st.base_pos[:] = list(s[0:3]); st.base_vel[:] = list(s[3:6])
        st.left_pos[:] = list(s[6:11]); st.left_vel[:] = list(s[11:16])
        st.right_pos[:] = list(s[16:21]); st.right_vel[:] = list(s[21:26])
        return st

    def array_to_operational_action(self, action):
        a = ControllerOsc()
        a.body_xdd[:] = list(action[0:2]); a.left_xdd[:] = list(action[2:4]); a.right_xdd[:] = list(action[4:6])
        a.pitch_add = action[6]
        return a

    def array_to_torque_action(self, action):
        a = ControllerTorque()
        a.torques[:] = list(action[0:6])
        return a

    def array_to_pd_action(self, action):
        a = ControllerPd()
        a.angles[:] = list(action[0:6])
        return a


def general_array_to_qpos_qvel(s):
    """26-array (base_pos, base_vel, left_pos, left_vel, right_pos, right_vel) -> qpos[13], qvel[13]."""
    s = np.asarray(s, dtype=np.double)
    q = np.concatenate([s[0:3], s[6:11], s[16:21]])
    v = np.concatenate([s[3:6], s[11:16], s[21:26]])
    return q, v
