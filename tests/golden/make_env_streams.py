#!/usr/bin/env python3
"""Record golden (action -> obs, reward, done) streams by running the REFERENCE's own env classes
(rllab/envs/cassie2d.py, cassie_stand2d.py -- imported from /root/reference, this container only)
against the oracle-backed drop-in library oracle/_dropin/bin/libcassie2d.so.

rllab / cached_property are absent, so inert stub modules are injected into sys.modules (the env
classes only use Env as a base class, Step as a tuple, Box for the spaces).  The reference resolves
'../../bin/libcassie2d.so' and '../trajectory/stepdata.bin' relative to the cwd, so a scratch tree
is laid out accordingly.  Output (data only): tests/golden/env_streams.npz
"""
import collections
import importlib
import os
import shutil
import subprocess
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))


def stub_modules():
    Step = collections.namedtuple("Step", ["observation", "reward", "done"])

    class Env:  # rllab.envs.base.Env
        pass

    class Box:
        def __init__(self, low, high):
            self.low, self.high = np.asarray(low), np.asarray(high)

    mods = {}
    for name in ("rllab", "rllab.envs", "rllab.envs.base", "rllab.misc", "rllab.misc.logger", "rllab.misc.overrides", "rllab.spaces",
                 "cached_property"):
        mods[name] = types.ModuleType(name)
    mods["rllab.envs.base"].Env = Env
    mods["rllab.envs.base"].Step = lambda observation, reward, done: Step(observation, reward, done)
    mods["rllab.misc"].logger = mods["rllab.misc.logger"]
    mods["rllab.misc.overrides"].overrides = lambda f: f
    mods["rllab.spaces"].Box = Box
    mods["cached_property"].cached_property = property
    sys.modules.update(mods)


def record(module_name, control_mode, actions, n_steps):
    """Import the reference env module with its module-level control_mode overridden."""
    src = open(os.path.join(REF, "rllab", "envs", module_name + ".py")).read()
    assert "control_mode = '" in src
    mod = types.ModuleType(module_name + "_" + control_mode)
    # the control mode is a module-level string literal in the reference (cassie2d.py:52): patch the VALUE at exec time
    code = src.replace("control_mode = 'PD'", "control_mode = '%s'" % control_mode).replace("control_mode = 'OSC'", "control_mode = '%s'" % control_mode)
    exec(compile(code, module_name + ".py", "exec"), mod.__dict__)
    env = mod.Cassie2dEnv()
    obs0 = env.reset()
    obs, rew, done, resets = [], [], [], []
    for t in range(n_steps):
        st = env.step(actions[t])
        obs.append(np.array(st.observation)); rew.append(float(st.reward)); done.append(bool(st.done))
        if st.done:
            resets.append(np.array(env.reset()))
        else:
            resets.append(np.zeros(26))
    return np.array(obs0), np.array(obs), np.array(rew), np.array(done), np.array(resets)


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle"), "legacy_shim"])
    tmp = tempfile.mkdtemp()
    try:
        os.makedirs(os.path.join(tmp, "bin")); os.makedirs(os.path.join(tmp, "rllab", "envs")); os.makedirs(os.path.join(tmp, "rllab", "trajectory"))
        shutil.copy(os.path.join(REPO, "oracle", "_dropin", "bin", "libcassie2d.so"), os.path.join(tmp, "bin"))
        os.symlink(os.path.join(REF, "rllab", "trajectory", "stepdata.bin"), os.path.join(tmp, "rllab", "trajectory", "stepdata.bin"))
        os.chdir(os.path.join(tmp, "rllab", "envs"))
        sys.path.insert(0, os.path.join(REF, "rllab", "envs"))
        stub_modules()
        rng = np.random.default_rng(2018)
        out = {}
        lo, hi = np.radians([-50, -164, -140] * 2), np.radians([80, -37, -30] * 2)
        T = 40
        a_pd = rng.uniform(lo, hi, size=(T, 6))
        a_tq = rng.uniform(-1, 1, size=(T, 6)) * np.array([12.0, 12.0, 0.9] * 2)
        a_osc = rng.uniform(-1, 1, size=(T, 7)) * np.array([3, 3, 1, 1, 1, 1, 3.0]); a_osc[:, 3] = np.abs(a_osc[:, 3]); a_osc[:, 5] = np.abs(a_osc[:, 5])
        for tag, module, mode, acts in (("walk_pd", "cassie2d", "PD", a_pd), ("walk_torque", "cassie2d", "Torque", a_tq),
                                        ("stand_torque", "cassie_stand2d", "Torque", a_tq), ("stand_pd", "cassie_stand2d", "PD", a_pd),
                                        ("stand_osc", "cassie_stand2d", "OSC", a_osc), ("walk_osc", "cassie2d", "OSC", a_osc)):
            o0, o, r, d, rs = record(module, mode, acts, T)
            out[tag + "_actions"], out[tag + "_obs0"], out[tag + "_obs"] = acts, o0, o
            out[tag + "_reward"], out[tag + "_done"], out[tag + "_reset_obs"] = r, d, rs
            print(tag, "reward[:4]", r[:4], "done", d.sum(), "/", T)
        np.savez_compressed(os.path.join(HERE, "env_streams.npz"), **out)
    finally:
        os.chdir(REPO)
        shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
