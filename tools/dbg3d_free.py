import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import numpy as np, oracle_py as O
from cassierl_amd import vec_env3d as V3
CTRL = V3.CTRL_RANGE
rng = np.random.default_rng(2)
env = V3.Cassie3dVec(1)
o = O.Oracle3D()
q0, v0 = o.state()
env.set_state_host(V3.state_record(q0, v0, o.warmstart())[None])
for blk in range(100):
    u = rng.uniform(-0.3, 0.3, 10) * CTRL
    env.step_host(u[None], 10)
    for _ in range(10): o.step_torque(u)
    s = env.get_state_host()[0]
    q1, v1 = o.state()
    err = max(np.abs(s[:21] - q1).max(), np.abs(s[21:41] - v1).max() / (1 + np.abs(v1).max()))
    if err > 1e-6 or blk % 20 == 0: print(blk, err, env.counters(), "nefc", o.nefc, "rec nefc", s[73], "niter", s[72])
    if err > 1e-3: break
