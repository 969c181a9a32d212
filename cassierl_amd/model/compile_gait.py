"""Offline: derive the product's 2-D reference-gait table from the reference's `rllab/trajectory/stepdata.bin`
(1682 x 98 float64, cassie2d_trajectory.py:6-14) with this package's own Cassie2dTraj, and store it as package data
(cassierl_amd/data/gait2d.npz: time[1682], qpos[1682,13], qvel[1682,13], torque[1682,6]).

Run in the build container only (needs /root/reference); the walk env loads the .npz through
cassierl_amd.trajectory.default_gait().  Same role as compile_model.py for the MJCF.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from cassierl_amd.trajectory import Cassie2dTraj  # noqa: E402

SRC = "/root/reference/rllab/trajectory/stepdata.bin"

if __name__ == "__main__":
    tr = Cassie2dTraj(SRC)
    out = os.path.join(os.path.dirname(HERE), "data", "gait2d.npz")
    np.savez_compressed(out, time=tr.time, qpos=tr.qpos, qvel=tr.qvel, torque=tr.torque)
    print(out, tr.qpos.shape, "tmax", tr.time[-1])
