"""Batched Cassie2d environment: host-side mirror of the reference's env classes
(rllab/envs/cassie2d.py `Cassie2dEnv`, rllab/envs/cassie_stand2d.py) over the batched C-ABI
(include/cassie_vec.h).  Same vocabulary -- reset() / step(action) / observation_space /
action_space / control_mode -- but every array has a leading n_envs axis and lives in HBM
(torch tensors on the MI355X; torch is only the allocator / stream provider here).

No CPU fallback: constructing the env loads libcassie2d.so and needs a HIP device.
"""
import ctypes as ct

import numpy as np

from . import _lib

CONTROL_MODES = {"PD": 0, "Torque": 1, "OSC": 2, "Jacobian": 3}
ENV_KINDS = {"walk": 0, "stand": 1}
FIX_STALE_KIN, FIX_STALE_QSTATE, WAVE_PER_ENV, NO_PINV_SHORTCUT = 1, 2, 4, 8  # 8: tests only (literal SVD route)
LEG_TIER_OFF, LEG_TIER_ON = 16, 32  # first kernel tier: never / always the two-lanes-per-environment kernel (default: by batch size)
DUO_TIER_OFF, DUO_TIER_ON = 64, 128  # ... never / always in its 64-environments-per-wavefront form (default: above 32 768 envs where it saves whole rounds of the chip's SIMDs: CassieVecCreate; `tier_info()` says what was chosen)
STATE_STRIDE = 88


class Box:
    """Minimal stand-in for rllab.spaces.Box (low/high arrays + sample)."""

    def __init__(self, low, high):
        self.low, self.high = np.asarray(low, dtype=np.float64), np.asarray(high, dtype=np.float64)
        self.shape = self.low.shape

    def sample(self, rng=np.random):
        return rng.uniform(self.low, self.high)

    def contains(self, x):
        x = np.asarray(x, dtype=np.float64)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


def action_space(control_mode):
    """cassie2d.py:343-368."""
    if control_mode == "OSC":
        return Box(np.array([-2e1, -2e1, -2e1, 0, -2e1, 0, -2e1]), np.full(7, 2e1))
    if control_mode == "Torque":
        high = np.array([12.0, 12.0, 0.9, 12.0, 12.0, 0.9])
        return Box(-high, high)
    high = np.radians([80.0, -37.0, -30.0, 80.0, -37.0, -30.0])
    low = np.radians([-50.0, -164.0, -140.0, -50.0, -164.0, -140.0])
    return Box(low, high)


class CassieVecEnv:
    """N independent Cassie2d environments stepped by one kernel launch per Env.step."""

    def __init__(self, n_envs, kind="walk", control_mode="PD", n_substeps=10, flags=0, auto_reset=True, device=0,
                 trajectory=None):
        assert control_mode in CONTROL_MODES, "Invalid Control Mode"  # cassie2d.py:53
        self.L = _lib.load()
        self.n_envs, self.kind, self.control_mode, self.n_substeps = n_envs, kind, control_mode, n_substeps
        self.device = device
        cfg = _lib.CassieVecConfig(ENV_KINDS[kind], CONTROL_MODES[control_mode], n_substeps, flags, int(auto_reset))
        h = ct.c_void_p()
        rc = self.L.CassieVecCreate(ct.byref(h), n_envs, device, ct.byref(cfg))
        if rc != 0:
            raise RuntimeError("CassieVecCreate failed (%d): no HIP device / allocation failure; there is no CPU path" % rc)
        self.h = h
        self.adim = self.L.CassieVecActionDim(self.h)
        if trajectory is not None:
            self.set_trajectory(trajectory.time, trajectory.qpos)
        self._torch = None

    # ---------------------------------------------------------------- plumbing
    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError("libcassie2d error %d: %s" % (rc, self.L.CassieVecLastError(self.h).decode()))

    def close(self):
        if getattr(self, "h", None):
            self.L.CassieVecFree(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_trajectory(self, time, qpos):
        t = np.ascontiguousarray(time, dtype=np.float64)
        q = np.ascontiguousarray(qpos, dtype=np.float64)
        assert q.shape == (len(t), 13)
        self._chk(self.L.CassieVecSetTrajectory(self.h, t.ctypes.data, q.ctypes.data, len(t)))

    def set_heightfield(self, heights_m, size_x=10.0, size_y=10.0):
        """Terrain under every robot (counterpart of the <hfield>/<geom type=hfield> pair terrain_random.py writes into the
        MJCF): heights in metres, [nrow, ncol], spanning [-size_x, size_x] x [-size_y, size_y]; None = flat floor.
        See cassierl_amd.terrain for the PNG loader and the random terrain choice."""
        if heights_m is None:
            self._chk(self.L.CassieVecSetHeightField(self.h, None, 0, 0, 0.0, 0.0))
            return
        hm = np.ascontiguousarray(heights_m, dtype=np.float64)
        assert hm.ndim == 2
        self._chk(self.L.CassieVecSetHeightField(self.h, hm.ctypes.data, hm.shape[0], hm.shape[1], float(size_x), float(size_y)))

    def synchronize(self):
        self._chk(self.L.CassieVecSynchronize(self.h))

    def counters(self):
        """Event counters since create / reset_counters(): env-substeps requested, env-substeps that left the packed fast
        path, of those the ones done by the wave-per-environment kernel, environments stopped by the failure guard."""
        out = (ct.c_uint64 * 4)()
        self._chk(self.L.CassieVecGetCounters(self.h, out))
        req, cleanup, k1, bad = (int(x) for x in out)
        return dict(substeps=req, cleanup_substeps=cleanup, k1_substeps=k1, nonfinite_resets=bad,
                    cleanup_frac=(cleanup / req if req else 0.0), k1_frac=(k1 / req if req else 0.0))

    TIERS = ("wave_per_env", "g16", "leg", "duo")

    def tier_info(self):
        """First physics tier the library chose for this handle (by batch size, flags, environment overrides) and the state of the
        64-environments kernel's hand-over workspace: claim-table slots (0 = one slot per task), bytes, extra claim probes so far."""
        out = (ct.c_uint64 * 8)()
        self._chk(self.L.CassieVecTierInfo(self.h, out))
        return dict(first_tier=self.TIERS[int(out[0])], duo_table_slots=int(out[1]), duo_workspace_bytes=int(out[2]), ws_probes=int(out[3]),
                    handovers_per_launch=int(out[4]), duo_workspace_slots_per_wave=int(out[5]), duo_envs=int(out[6]))

    def qp_iterations(self):
        """Active-set iterations of the OSC QP since the previous call (the first call starts the counting and returns zeros):
        mean per StepOsc call and environment, maximum, calls counted, largest per-environment mean."""
        out = (ct.c_double * 4)()
        self._chk(self.L.CassieVecQpIterations(self.h, out))
        return dict(mean=out[0], max=int(out[1]), calls=int(out[2]), worst_env_mean=out[3])

    def debug_workspace_host(self):
        """The 64-environments kernel's hand-over workspace as the last launch left it: array [wavefront slot][W_N slots][64 lanes] (diagnosis)."""
        n = ct.c_uint64()
        self._chk(self.L.CassieVecDebugWorkspaceHost(self.h, None, 0, ct.byref(n)))
        buf = np.zeros(int(n.value))
        self._chk(self.L.CassieVecDebugWorkspaceHost(self.h, buf.ctypes.data, n.value, ct.byref(n)))
        wn = self.tier_info()["duo_workspace_slots_per_wave"]          # Duo::W_N
        per = wn * 64
        w = buf[:len(buf) // per * per].reshape(-1, wn // 2, 64, 2)       # [wave][slot pair][lane][2]
        return w.transpose(0, 1, 3, 2).reshape(w.shape[0], wn, 64)        # [wave][slot][lane]

    def reset_counters(self):
        self._chk(self.L.CassieVecResetCounters(self.h))

    @property
    def observation_space(self):
        """The space of the rows step() / reset() return: Box(26) for both env kinds, because the batched ABI always fills
        [n_envs, 26] (17 op-space values + 9 reference-gait joints, cassie2d.py:337-341; SURVEY Q5: "return 26; expose 17-view").
        A policy sized from this space fits the observations the env emits.  cassie_stand2d.py:241-244 declares Box(17) for the
        stand env: that is `reference_observation_space`, and `obs_view()` gives the matching 17-wide view (opt-in)."""
        high = np.full((26,), 1e20)
        return Box(-high, high)

    @property
    def reference_observation_space(self):
        """The observation space the reference's env class of this kind declares: Box(26) walk, Box(17) stand."""
        high = np.full((self.obs_dim,), 1e20)
        return Box(-high, high)

    @property
    def obs_dim(self):
        """Width of the reference env's own observation (17 for cassie_stand2d.py, 26 for cassie2d.py); the emitted rows are 26 wide."""
        return 17 if self.kind == "stand" else 26

    def obs_view(self, obs):
        """The observation as the reference's env of this kind returns it: [..., :17] for cassie_stand2d.py, all 26 for cassie2d.py
        (a view of the [n_envs, 26] tensor / array the step wrote; no copy)."""
        return obs[..., :self.obs_dim]

    @property
    def action_space(self):
        return action_space(self.control_mode)

    # ---------------------------------------------------------------- host (numpy) API: tests, small batches
    def reset_host(self):
        import torch
        obs = torch.empty((self.n_envs, 26), dtype=torch.float64, device="cuda:%d" % self.device)
        self._chk(self.L.CassieVecReset(self.h, None, obs.data_ptr()))
        self.synchronize()
        return obs.cpu().numpy()

    def step_host(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float64).reshape(self.n_envs, self.adim)
        obs, rew = np.empty((self.n_envs, 26)), np.empty(self.n_envs)
        done = np.empty(self.n_envs, dtype=np.uint8)
        self._chk(self.L.CassieVecStepHost(self.h, a.ctypes.data, obs.ctypes.data, rew.ctypes.data, done.ctypes.data))
        return obs, rew, done.astype(bool)

    def get_state_host(self):
        q, v = np.empty((self.n_envs, 13)), np.empty((self.n_envs, 13))
        self._chk(self.L.CassieVecGetStateHost(self.h, q.ctypes.data, v.ctypes.data))
        return q, v

    def get_full_state_host(self):
        s = np.empty((self.n_envs, STATE_STRIDE))
        self._chk(self.L.CassieVecGetFullStateHost(self.h, s.ctypes.data))
        return s

    def set_full_state_host(self, s):
        s = np.ascontiguousarray(s, dtype=np.float64).reshape(self.n_envs, STATE_STRIDE)
        self._chk(self.L.CassieVecSetStateHost(self.h, s.ctypes.data))

    def debug_substep_host(self, control_mode, actions):
        a = np.ascontiguousarray(actions, dtype=np.float64)
        dbg = np.zeros((self.n_envs, 512))
        self._chk(self.L.CassieVecDebugSubstepHost(self.h, CONTROL_MODES[control_mode], a.ctypes.data, dbg.ctypes.data))
        return dbg

    def substep_host(self, control_mode, actions, n_sub=1):
        import torch
        a = torch.as_tensor(np.ascontiguousarray(actions, dtype=np.float64), device="cuda:%d" % self.device)
        self._chk(self.L.CassieVecSubstep(self.h, CONTROL_MODES[control_mode], a.data_ptr(), n_sub))
        self.synchronize()

    def standing_step_host(self, control_mode, zpos, zvel, n_sub=1):
        """standing_controller_osc / standing_controller_jacobian (cassie2d.py:263-331) for every env."""
        import torch
        dev = "cuda:%d" % self.device
        zp = torch.as_tensor(np.broadcast_to(np.asarray(zpos, dtype=np.float64), (self.n_envs,)).copy(), device=dev)
        zv = torch.as_tensor(np.broadcast_to(np.asarray(zvel, dtype=np.float64), (self.n_envs,)).copy(), device=dev)
        self._chk(self.L.CassieVecStandingStep(self.h, CONTROL_MODES[control_mode], zp.data_ptr(), zv.data_ptr(), n_sub))
        self.synchronize()

    # ---------------------------------------------------------------- device (torch) API: rollouts
    def alloc(self):
        import torch
        dev = "cuda:%d" % self.device
        return dict(obs=torch.empty((self.n_envs, 26), dtype=torch.float64, device=dev),
                    reward=torch.empty(self.n_envs, dtype=torch.float64, device=dev),
                    done=torch.empty(self.n_envs, dtype=torch.uint8, device=dev))

    def use_torch_stream(self):
        import torch
        self._chk(self.L.CassieVecSetStream(self.h, ct.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def reset(self, out=None, mask=None):
        out = out or self.alloc()
        self._chk(self.L.CassieVecReset(self.h, None if mask is None else mask.data_ptr(), out["obs"].data_ptr()))
        return out["obs"]

    def reset_to(self, qpos, qvel, out=None, mask=None):
        """Cassie2d::Reset with caller-provided states (float64 CUDA tensors [n_envs, 13]); returns the observation tensor."""
        out = out or self.alloc()
        self._chk(self.L.CassieVecResetTo(self.h, None if mask is None else mask.data_ptr(), qpos.data_ptr(), qvel.data_ptr(),
                                          out["obs"].data_ptr()))
        return out["obs"]

    def get_state(self):
        """GetGeneralState for every env: (qpos, qvel) float64 CUDA tensors [n_envs, 13]."""
        import torch
        dev = "cuda:%d" % self.device
        q = torch.empty((self.n_envs, 13), dtype=torch.float64, device=dev)
        v = torch.empty((self.n_envs, 13), dtype=torch.float64, device=dev)
        self._chk(self.L.CassieVecGetState(self.h, q.data_ptr(), v.data_ptr()))
        return q, v

    def get_opstate(self):
        """GetOperationalSpaceState for every env: float64 CUDA tensor [n_envs, 18] in operational_state_to_array order
        (cassie2d_structs.py), computed like the reference from the kinematics of the last setState."""
        import torch
        x = torch.empty((self.n_envs, 18), dtype=torch.float64, device="cuda:%d" % self.device)
        self._chk(self.L.CassieVecGetOpState(self.h, x.data_ptr()))
        return x

    def step(self, actions, out=None, terminal_obs=None):
        """actions: float64 CUDA tensor [n_envs, adim].  Returns (obs, reward, done) tensors (views of `out`)."""
        out = out or self.alloc()
        assert actions.is_cuda and actions.dtype.is_floating_point and actions.element_size() == 8 and actions.is_contiguous()
        self._chk(self.L.CassieVecStep(self.h, actions.data_ptr(), out["obs"].data_ptr(), out["reward"].data_ptr(),
                                       out["done"].data_ptr(), None if terminal_obs is None else terminal_obs.data_ptr()))
        return out["obs"], out["reward"], out["done"]

    def accumulate(self, reward, done, returns=None, episodes=None):
        """Rollout bookkeeping of one Env.step in one launch on the env's stream: `returns += reward` (float64 CUDA tensors [n_envs]) and
        `episodes += done.count_nonzero()` (`episodes`: int64 CUDA tensor with one element).  Either accumulator may be None."""
        assert returns is None or (returns.is_cuda and returns.element_size() == 8 and returns.numel() == self.n_envs and returns.is_contiguous())
        assert episodes is None or (episodes.is_cuda and episodes.element_size() == 8 and episodes.numel() == 1)
        self._chk(self.L.CassieVecAccumulate(self.h, reward.data_ptr(), done.data_ptr(), None if returns is None else returns.data_ptr(),
                                             None if episodes is None else episodes.data_ptr()))

    def time_steps(self, actions, steps, out=None):
        """Average kernel time (ms) of `steps` back-to-back Env.steps, HIP events on the env's stream."""
        out = out or self.alloc()
        ms = ct.c_float()
        self._chk(self.L.CassieVecTimeSteps(self.h, actions.data_ptr(), steps, out["obs"].data_ptr(), out["reward"].data_ptr(),
                                            out["done"].data_ptr(), ct.byref(ms)))
        return ms.value
