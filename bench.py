#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the batched Cassie2d rollout (BASELINE.json metric).

Workload (every rank): 65 536 parallel Cassie2d envs -- the size BASELINE.json's target is quoted on ("64k parallel Cassie2d
envs at 1 GPU"; configs[3] = 8 x 64k) -- rllab/envs/cassie2d.py semantics: PD control mode, n=10 physics substeps per
Env.step, reward / termination / auto-reset, uniform random policy over the PD action box (configs[1]'s workload at the
target size; `--envs-per-gpu 4096` gives configs[1] itself).  One bench "step" = one vectorised Env.step over the whole
batch = 65 536 env-steps (655 360 physics substeps) per GPU.

Multi-GPU: weak scaling, one process per GPU.  Every rank owns `envs_per_gpu` envs (global ids rank*n ..), no collective
inside a step, ONE RCCL all_gather of the per-env returns at the end of the rollout batch (inside the timed region).
`python bench.py --gpus N` started WITHOUT torchrun spawns the N ranks itself -- before anything touches the GPU in the
parent -- and fails non-zero unless exactly N ranks join; under torchrun (WORLD_SIZE set) it is one of the ranks.

Inputs (actions, counter-based stream keyed by the global env id) are generated on the device before the timed region.
At N=1 the line also carries `extra`: the same kernels on workloads where robots move and fall (share of env-substeps
that leave the packed fast path included), configs[2] (OSC controller in the loop) and configs[4] (Cassie3d).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 65536
FIRST_TIER_KERNEL = {"duo": "cassie::leg::env_step_duo_kernel<0>", "leg": "cassie::leg::env_step_leg_kernel<0>",
                     "g16": "cassie::g16::env_step_g16_kernel<0, false>", "wave_per_env": "cassie::env_step_kernel<0, 1, 32, false>"}   # CassieVecEnv.tier_info()["first_tier"] -> kernel of one PD Env.step


def dominant_kernel(n_envs, simds=1024):
    """The kernel that steps most of a PD batch on the flat floor, by the library's rule (cassie_cabi.hip, CassieVecCreate: whole rounds of one wavefront per
    SIMD -- 64 environments per wavefront at ~1.0 ms a round against 32 at ~0.62 ms, the whole rounds of a batch in the former and a short remainder in the
    latter; LEG_MIN_ENVS below that).  The bench line itself asks the library (CassieVecTierInfo); this mirror is for tests and tools without a GPU."""
    rp, rj = -(-n_envs // (32 * simds)), -(-n_envs // (64 * simds))
    whole = n_envs // (64 * simds) * (64 * simds)
    rest = n_envs - whole
    split = 1.0 * (whole // (64 * simds)) + 0.62 * (-(-rest // (32 * simds)))
    if n_envs > 32768 and ((whole > 0 and rest > 0 and split < 0.62 * rp and split < 1.0 * rj) or 1.05 * 1.0 * rj < 0.62 * rp):
        return "cassie::leg::env_step_duo_kernel<0>"
    return "cassie::leg::env_step_leg_kernel<0>" if n_envs >= 6144 else "cassie::g16::env_step_g16_kernel<0, false>"


# Schema of the JSON line (the meaning of `roofline` and `cpu_baseline` changed between rounds; a reader comparing BENCH_rNN
# files must not mix them): r01-r03 roofline = HBM (GB/s); from r04 roofline.bound = fp64_valu (useful TFLOP/s; HBM under roofline.hbm)
# and cpu_baseline.value = the same-source AVX-512 leg (the oracle's leg under cpu_baseline.oracle).
BENCH_SCHEMA = "r06 (as r05 -- roofline: fp64_valu useful flops, hbm sub-object; cpu_baseline: same-source leg with per_threads / cpu_quota, oracle nested -- plus config.first_tier / duo_workspace_MB from CassieVecTierInfo and extra[configs[2]].qp_iterations_per_substep)"
PREROLL_STEPS_AT_64K = int(os.environ.get("CASSIE_BENCH_PREROLL_STEPS", "300"))     # untimed Env.steps before the timed region at 65 536 envs, on top of --warmup (see worker())
ALGO_BYTES_PER_ENV_STEP = 697 + 208  # state+action in, state+obs+reward+done out, + persisted warm-start vector
HBM_PEAK_GBPS = 8000.0               # MI355X_MICROARCH.md: 8 TB/s spec
USEFUL_FLOP_R04 = 300434.4270833333   # profiles/r04_n: counted useful FP64 per env-step of the bench workload as the source stood in rounds 3-4
FP64_VALU_PEAK_TFLOPS = 78.6         # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz


def preroll_count(n_envs):
    """Fixed pre-roll length: PREROLL_STEPS_AT_64K at 65 536 envs (~0.4 s), more steps for smaller batches (their steps are shorter,
    down to one wavefront's latency), fewer for larger ones; a function of the batch size only, so every rank and every run agree."""
    return int(min(1000, max(20, round(PREROLL_STEPS_AT_64K * min(4.0, 65536.0 / max(1, n_envs))))))


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--no-trpo", action="store_true", help="N > 1: skip the TRPO outer loop (configs[3]'s caller) after the random-policy rollout")
    ap.add_argument("--trpo-iters", type=int, default=6)
    ap.add_argument("--trpo-timeout", type=float, default=300.0, help="N > 1: seconds after which a TRPO stage that has not finished is taken for a lost rank (headline printed, exit 6)")
    ap.add_argument("--cpu-legs-only", action="store_true", help="internal: print the CPU legs of configs[0]/[2]/[4] as JSON (no GPU) and exit")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ launcher (no GPU call)
def visible_gpus():
    """Number of GPUs this process may use, WITHOUT touching torch.cuda / HIP in the launcher parent: the DRM render nodes of AMD
    devices (vendor 0x1002) that this process can actually open -- a container's /dev/dri may list nodes its cgroup denies, and
    /sys shows the whole host -- capped by every visibility list that is set (ROCR composes with HIP / CUDA: the effective set is
    no larger than the smallest)."""
    n = 0
    try:
        for f in sorted(os.listdir("/dev/dri")):
            if not f.startswith("renderD"):
                continue
            try:
                if open("/sys/class/drm/%s/device/vendor" % f).read().strip() != "0x1002":
                    continue
            except OSError:
                pass   # no sysfs entry to ask: count the node if it opens
            try:
                os.close(os.open(os.path.join("/dev/dri", f), os.O_RDWR))
                n += 1
            except OSError:
                pass
    except OSError:
        n = 0
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def spawn_ranks(args, poll_s=0.5, deadline_s=3600.0):
    """Start one child process per GPU with the torchrun environment; the parent never initialises the GPU.  All children are
    polled together: the first non-zero exit (or the deadline) ends the run -- the surviving ranks, which would otherwise sit in
    a collective waiting for the dead one, are killed by their exact PIDs -- and the launcher returns non-zero (128 + signal
    number for a rank that died from a signal, 124 for the deadline)."""
    have = visible_gpus()
    if have < args.gpus and not os.environ.get("CASSIE_DEVICE_MAP"):
        sys.stderr.write("bench.py: --gpus %d requested but only %d device(s) visible\n" % (args.gpus, have))
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc, deadline = 0, time.time() + deadline_s
    live = list(range(args.gpus))
    while live and rc == 0:
        time.sleep(poll_s)
        for r in list(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.remove(r)
            if code != 0:
                rc = 128 - code if code < 0 else code
                sys.stderr.write("bench.py: rank %d exited with %s; stopping the other ranks\n"
                                 % (r, "signal %d" % -code if code < 0 else "code %d" % code))
                break
        if rc == 0 and live and time.time() > deadline:
            rc = 124
            sys.stderr.write("bench.py: deadline of %.0f s passed with rank(s) %s still running\n" % (deadline_s, live))
    for p in procs:  # exact PIDs only.  SIGTERM first: rank 0 inside the TRPO stage prints the headline it has measured (trpo_outer_loop's watchdog)
        if p.poll() is None:
            p.terminate()
    t_end = time.time() + 8.0
    for p in procs:
        try:
            p.wait(timeout=max(0.1, t_end - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
    for p in procs:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            pass
    return rc


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_limits():
    """What this process may really use of the box's host cores: the affinity mask, the cgroup CPU quota (cgroup v2 `cpu.max`, v1
    `cpu.cfs_quota_us` / `cpu.cfs_period_us`; None = unlimited / not readable) and the cores the box shows."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else float(q) / float(per)
        src = "cgroup v2 cpu.max = %s %s" % (q, per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = None if q <= 0 else q / per
            src = "cgroup v1 cfs_quota_us / cfs_period_us = %g / %g" % (q, per)
        except Exception:
            src = "no readable cgroup CPU quota"
    return dict(cpu_count=os.cpu_count(), affinity=aff, cpu_quota=quota, cpu_quota_source=src)


def thread_ladder(cores):
    """Thread counts to try: the visible cores, then halves down to one (every figure is reported, the best is the value)."""
    tries, th = [], max(1, cores)
    while th >= 1:
        tries.append(th)
        th //= 2
    return tries


def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    return O


def cpu_baseline(traj, cores, budget_s=12.0):
    """Oracle (C restatement, -O3 -march=native build made on this box) on the host cores of this box, bounded sample of the
    same workload: (i) ONE thread, (ii) OpenMP over envs.  The visible CPU count of a container can exceed what its quota
    really schedules, so the OpenMP leg tries cores, cores/2, ... and reports the best, with the threads it used."""
    O = _oracle()
    from cassierl_amd import rollout as R
    import torch
    fast = O.use_fast_build() if hasattr(O, "use_fast_build") else False
    low, high = np.radians([-50, -164, -140] * 2), np.radians([80, -37, -30] * 2)

    def sample(n, threads, budget):
        envs = [O.OracleEnv("walk", "PD", traj=traj) for _ in range(n)]
        for e in envs:
            e.reset()
        ids = torch.arange(n)
        steps, t_used = 0, 0.0
        O.envs_step(envs, R.random_actions(1, ids, 0, low, high).numpy(), 10, True, threads)  # warm-up
        while t_used < budget and steps < 400:
            a = R.random_actions(1, ids, steps + 1, low, high).numpy()
            t0 = time.perf_counter()
            O.envs_step(envs, a, 10, True, threads)
            t_used += time.perf_counter() - t0
            steps += 1
        return n * steps / t_used, steps, t_used

    one, s1, t1 = sample(16, 1, 2.0)
    tries = [t for t in thread_ladder(cores) if t > 1][:6] or [1]
    per = max(1.0, (budget_s - 2.0) / len(tries))
    best, per_threads = None, {"1": one}
    for th in tries:
        v, st, tu = sample(16 * th, th, per)
        per_threads[str(th)] = v
        if best is None or v > best[0]:
            best = (v, th, st, tu)
    if one > best[0]:
        best = (one, 1, s1, t1)
    return dict(value=best[0], unit="env-steps/s", cores=best[1], kind="port", per_threads=per_threads,
                single_thread=dict(value=one, unit="env-steps/s", cores=1, sample="16 envs x %d Env.steps in %.1f s" % (s1, t1)),
                cores_visible=cores, threads_tried=tries,
                sample="%d envs x %d Env.steps (walk env, PD, random policy) in %.1f s, OpenMP over envs on %d threads (best of %s), "
                       "oracle built %s" % (16 * best[1], best[2], best[3], best[1], tries, "-O3 -march=native" if fast else "-O2"))


def cpu_same_source(traj, cores, budget_s=20.0):
    """The SAME SOURCE as the HIP kernel (cassierl_amd/csrc/cassie_leg_core.h) on the host cores: oracle/leg_host/leg_host.cpp
    instantiates it with an eight-lane backend (four environments per AVX-512 register), -O3 -march=native, built on this box;
    OpenMP over groups of environments.  Same workload as the headline (walk env, PD, random policy, auto-reset).  BASELINE.md
    section 3 / SURVEY.md 8(d) promise this leg beside the oracle's."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ctypes as ct
    import leg_host as LH
    import torch
    from cassierl_amd import rollout as R
    O = _oracle()
    L = LH.lib(fast=True)
    low, high = np.radians([-50, -164, -140] * 2), np.radians([80, -37, -30] * 2)
    oe = O.OracleEnv("walk", "PD", traj=traj)
    oe.reset()
    q, v = oe.oracle.state()
    ctor = O.Oracle()
    rec = np.zeros(88)
    rec[0:13], rec[13:26], rec[26:39] = q, v, oe.oracle.warmstart()
    rec[39:52], rec[52:65], rec[65:78] = ctor.state()[0], ctor.state()[1], q
    tq = np.ascontiguousarray(traj["qpos"], dtype=np.float64)
    dp, ip, bp = ct.POINTER(ct.c_double), ct.POINTER(ct.c_int), ct.POINTER(ct.c_ubyte)

    def sample(n, threads, budget):
        state = np.tile(rec, (n, 1)).copy()
        obs, rew, done, pend, bad = np.zeros((n, 26)), np.zeros(n), np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.int32), ct.c_int(0)
        ids = torch.arange(n)
        steps, t_used = 0, 0.0
        while t_used < budget and steps < 400:
            a = np.ascontiguousarray(R.random_actions(1, ids, steps, low, high).numpy())
            t0 = time.perf_counter()
            L.leg_host_step(state.ctypes.data_as(dp), a.ctypes.data_as(dp), n, 6, 0, 10, 0, 0, 1, tq.ctypes.data_as(dp),
                            ct.c_double(float(traj["time"][-1])), len(traj["time"]), obs.ctypes.data_as(dp), rew.ctypes.data_as(dp),
                            done.ctypes.data_as(bp), None, pend.ctypes.data_as(ip), ct.byref(bad), threads)
            if steps > 0:   # the first call warms the caches / the OpenMP team up
                t_used += time.perf_counter() - t0
            steps += 1
        assert np.isfinite(rew).all() and bad.value == 0 and pend.sum() == 0
        return n * (steps - 1) / t_used, steps - 1, t_used

    # Every thread count of the ladder is measured and REPORTED (`per_threads`): the sample is 256 environments per thread (64 register
    # groups each, so that a wide team has work for every member) stepped for at least `per` seconds after an untimed first call
    # (thread-team start-up, first touch of the state); the value is the best of them, and `limits` says what the box really grants --
    # a container can show 256 hardware threads and schedule a fraction of them (cgroup quota), in which case the wide teams
    # time-share cores and lose to the narrow ones.
    limits = cpu_limits()
    one, s1, t1 = sample(256, 1, 1.5)
    tries = [t for t in thread_ladder(cores) if t > 1] or [1]
    per = max(2.0, (budget_s - 1.5) / len(tries))
    best, per_threads = None, {"1": one}
    for th in tries:
        val, st, tu = sample(256 * th, th, per)
        per_threads[str(th)] = val
        if best is None or val > best[0]:
            best = (val, th, st, tu)
    if one > best[0]:
        best = (one, 1, s1, t1)
    q = limits.get("cpu_quota")
    why = ("the cgroup quota grants %.1f cores of the %d visible: wider teams time-share them" % (q, cores) if q and q < cores else
           "no CPU quota is set: the best thread count is where the same-source leg stops scaling on this box (shared L3 / memory, SMT pairs)")
    return dict(value=best[0], unit="env-steps/s", cores=best[1], kind="same-source", lanes_per_thread=int(L.leg_host_lanes()),
                single_thread=dict(value=one, unit="env-steps/s", cores=1, sample="256 envs x %d Env.steps in %.1f s" % (s1, t1)),
                per_threads=per_threads, cpu_quota=q, limits=limits, chosen_because=why,
                scaling_efficiency_at_best=best[0] / (one * best[1]) if one > 0 else None,
                cores_visible=cores, threads_tried=tries,
                sample="%d envs x %d Env.steps (walk env, PD, random policy) in %.1f s: cassie_leg_core.h through the host backend "
                       "(oracle/leg_host, 8 lanes = 4 envs per AVX-512 register), g++ -O3 -march=native, OpenMP over envs on %d threads "
                       "(best of %s; every figure in per_threads)" % (256 * best[1], best[2], best[3], best[1], [1] + tries))


def cpu_legs_other_configs(budget_s=4.0):
    """CPU legs beside the other BASELINE configs (oracle, ONE thread, bounded samples):
    configs[0] squatting.py (Jacobian standing controller, 2 kHz loop, 20 000 substeps = 10 s of robot time -- bounded by time,
    the substeps actually run are stated), configs[2] OSC standing controller per substep, configs[4] Cassie3d torque mode."""
    O = _oracle()
    rows = []
    qinit = np.array([0, 0.939, 0, 0.68111815, -1.40730357, 1.62972042, -1.77611107, -0.61968407,
                      0.68111815, -1.40730357, 1.62972042, -1.77611107, -0.61968407])

    def jacobian_force(s, zpos, zvel):  # standing_controller_jacobian, rllab/envs/cassie2d.py:297-331
        xt = (s[6] + s[12]) / 2.0
        fx = 200.0 * (xt - s[0]) + 50.0 * (0.0 - s[3])
        fz = max(0.5 * 9.806 * 31.0 + 200.0 * (zpos - s[1]) + 50.0 * (zvel - s[4]), 0.0)
        my = 100.0 * (0.0 - s[2]) + 10.0 * (0.0 - s[5])
        return np.array([fx, fz, my, fx, fz, my])

    o = O.Oracle()
    o.reset(qinit, np.zeros(13))
    n, t0 = 0, time.perf_counter()
    while n < 20000 and time.perf_counter() - t0 < 3 * budget_s:
        t = n * 0.0005
        s = o.opstate()
        o.step_jacobian(jacobian_force(s, 0.7 + 0.25 * np.sin(0.5 * 3.1415 * t), 0.25 * np.cos(0.5 * 3.1415 * t)))  # squatting.py:9-16
        n += 1
    dt = time.perf_counter() - t0
    q, _ = o.state()
    rows.append(dict(workload="configs[0]_squatting_cpu", cores=1, substeps=n, substeps_per_s=n / dt, seconds=dt, pelvis_z=float(q[1]),
                     note="squatting.py loop on the oracle: scripted Jacobian standing controller + StepJacobian per substep (2 kHz), one thread; "
                          "%d of the 20 000 substeps of configs[0] run within the time bound" % n))
    o = O.Oracle()
    o.reset(qinit, np.zeros(13))
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        s = o.opstate()
        a = np.zeros(7)   # standing_controller_osc(zpos=0.9, zvel=0), rllab/envs/cassie2d.py:263-295
        a[3] = 100.0 * (-5e-3 - s[7]); a[5] = 100.0 * (-5e-3 - s[13])
        a[0] = 100.0 * ((s[6] + s[12]) / 2.0 - s[0]) + 20.0 * (0.0 - s[3])
        a[1] = 100.0 * (0.9 - s[1]) + 20.0 * (0.0 - s[4])
        a[6] = 20.0 * (0.0 - s[2]) + 10.0 * (0.0 - s[5])
        o.step_osc(a)
        n += 1
    dt = time.perf_counter() - t0
    rows.append(dict(workload="configs[2]_osc_standing_cpu", cores=1, substeps=n, substeps_per_s=n / dt, env_steps_equiv_per_s=n / dt / 10, seconds=dt,
                     note="OSC QP (literal 39-variable form) + mj_step per substep on the oracle, one thread, one robot"))
    o3 = O.Oracle3D()
    rng = np.random.default_rng(5)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        o3.step_torque(rng.uniform(-1, 1, 10) * 4.5)
        n += 1
    dt = time.perf_counter() - t0
    rows.append(dict(workload="configs[4]_cassie3d_cpu", cores=1, substeps=n, substeps_per_s=n / dt, env_steps_per_s=n / dt / 10, seconds=dt,
                     note="cassie3d_stiff.xml physics on the oracle (-O2 parity build), random torques, one thread, one robot"))
    return rows


# ------------------------------------------------------------------------------------------------ extra workloads (N=1)
def run_env_workload(name, n, kind, mode, flags, traj, warmup, steps, action_fn, note, auto_reset=True, heightfield=None):
    import torch
    from cassierl_amd.vec_env import CassieVecEnv
    env = CassieVecEnv(n, kind=kind, control_mode=mode, n_substeps=10, flags=flags, auto_reset=auto_reset, device=0)
    if kind == "walk":
        env.set_trajectory(traj["time"], traj["qpos"])
    if heightfield is not None:
        env.set_heightfield(heightfield, 10.0, 10.0)
    env.use_torch_stream()
    out = env.alloc()
    env.reset(out)
    for t in range(warmup):
        env.step(action_fn(t), out)
    acts = [action_fn(warmup + t) for t in range(steps)]
    env.reset_counters()
    dones = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(steps):
        _, r, d = env.step(acts[t], out)
        env.accumulate(r, d, None, dones)   # episode count of the step (one launch; headline loop: returns too)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c = env.counters()
    row = dict(workload=name, note=note, envs=n, warmup_steps=warmup, steps=steps, env_steps_per_s=n * steps / dt, ms_per_step=dt / steps * 1e3,
               cleanup_frac=c["cleanup_frac"], k1_frac=c["k1_frac"], nonfinite_resets=c["nonfinite_resets"],
               episodes_terminated_per_env_step=float(dones.item()) / (n * steps))
    if mode == "OSC":
        # BASELINE.md C3 "QP iterations / step": active-set iterations of the OSC QP per substep, counted in a pass of its own BEHIND the timed region
        # (the controllers store nothing until the first qp_iterations() call), same action stream
        env.qp_iterations()
        for t in range(min(steps, 10)):
            env.step(acts[t], out)
        qi = env.qp_iterations()
        row["qp_iterations_per_substep"] = dict(mean=qi["mean"], max=qi["max"], worst_env_mean=qi["worst_env_mean"], stepOsc_calls=qi["calls"],
                                                note="primal active-set iterations to KKT convergence (cap 60), hot-started from the previous substep's working set; "
                                                     "the reference gives qpOASES 100 working-set changes / 500 us (OSC_RBDL.cpp:245-246)")
    q, v = env.get_state_host()
    env.close()
    row["finite"] = bool(np.isfinite(q).all() and np.isfinite(v).all())
    return row


def extra_workloads(traj, n):
    import torch
    from cassierl_amd import rollout as R
    from cassierl_amd import vec_env as VE
    ids = torch.arange(n, device="cuda:0")
    rows = []
    pd_box = VE.action_space("PD")
    tq_box = VE.action_space("Torque")
    # (0) configs[1] as BASELINE.json words it: 4096 envs, random-policy rollout, PD mode, walk env (the headline is the same
    #     workload at the north_star's 65 536 envs).  4096 envs are one wavefront per SIMD of the 4-envs-per-wave kernel: the
    #     step time is one wavefront's dependency chain, not throughput.
    n1 = 4096
    ids1 = torch.arange(n1, device="cuda:0")
    rows.append(run_env_workload("configs[1]_4096_envs_pd_random", n1, "walk", "PD", 0, traj, 300, 300,
                                 lambda t: R.random_actions(1, ids1, t, pd_box.low, pd_box.high),
                                 "configs[1] at its own size: walk env, StepPd with random joint targets, auto-reset"))
    # (a) stand env / PD mode, random joint targets over the PD box: robots thrash, hit joint limits and fall; reset at z < 0.5.
    #     (The walk env cannot serve here: its reward needs the pose to track the reference gait, so under a random policy every
    #     step ends the episode even with quirk Q3 fixed -- measured, episodes_terminated_per_env_step = 1.0.)
    rows.append(run_env_workload("stand_pd_random", n, "stand", "PD", 0, traj, 150, 40,
                                 lambda t: R.random_actions(2, ids, t, pd_box.low, pd_box.high),
                                 "cassie_stand2d.py reward/termination, StepPd with random joint targets"))
    # (b) stand env / torque mode, U(+-ctrlrange): free falls onto the ground, reset at z < 0.5 (random_agent.py-style)
    rows.append(run_env_workload("stand_torque_random", n, "stand", "Torque", 0, traj, 150, 40,
                                 lambda t: R.random_actions(3, ids, t, tq_box.low, tq_box.high),
                                 "cassie_stand2d.py reward/termination, random torques in ctrlrange"))
    # (b') the floor: no auto-reset, so after the warm-up every robot lies on the ground with many contacts and joint limits
    #      active -- the regime in which environments leave the 16-row packed kernel
    rows.append(run_env_workload("torque_random_no_reset_fallen", n, "stand", "Torque", 0, traj, 200, 20,
                                 lambda t: R.random_actions(3, ids, t, tq_box.low, tq_box.high),
                                 "random torques, auto_reset off: robots on the ground (throughput floor of the PD/torque path)", auto_reset=False))
    # counted useful FP64 of the two falling-robot rows (tools/count_flops.py: the first tier's own source through the op-counting CPU build, robots in the
    # same regime) against the FP64 vector peak, over the WHOLE step (first tier in segments + the lower tiers beside it)
    try:
        uf_all = json.load(open(os.path.join(ROOT, "profiles", "useful_flops.json")))
        for row in rows:
            uf = (uf_all.get(row["workload"]) or {}).get("flop_per_env_step")
            if uf:
                ach = uf * row["env_steps_per_s"] / 1e12
                row["roofline"] = dict(bound="fp64_valu", achieved=ach, peak=FP64_VALU_PEAK_TFLOPS, unit="TFLOP/s", frac=ach / FP64_VALU_PEAK_TFLOPS,
                                       useful_flop_per_env_step=uf,
                                       note="useful flops counted on the two-lanes-per-environment kernel's source (what the first tier executes here, in "
                                            "segments); an environment finished by a lower tier is priced at the same count")
    except Exception:
        pass
    # (b'') N4: the same random-PD rollout on a height field (terrain_random.py's <hfield>: here 3 cm rolling relief, 20 m x 20 m)
    xs = np.linspace(-10.0, 10.0, 2001)
    relief = np.tile(0.015 * (1.0 - np.cos(2.0 * np.pi * xs / 1.5)), (64, 1))
    rows.append(run_env_workload("terrain_stand_pd_random", n, "stand", "PD", 0, traj, 150, 40,
                                 lambda t: R.random_actions(2, ids, t, pd_box.low, pd_box.high),
                                 "stand env on a height field (N4), StepPd with random joint targets", heightfield=relief))
    # (c) configs[2]: OSC controller (QP) in every substep, cassie_stand2d Env.step with small random OSC targets
    osc_lo, osc_hi = np.array([-2.0, -2.0, -2.0, 0.0, -2.0, 0.0, -2.0]), np.full(7, 2.0)
    rows.append(run_env_workload("configs[2]_stand_osc_in_loop", n, "stand", "OSC", 0, traj, 20, 30,
                                 lambda t: R.random_actions(4, ids, t, osc_lo, osc_hi),
                                 "OSC_RBDL QP + mj_step per substep, 10 substeps per Env.step, random accelerations targets in +-2 m/s^2"))
    try:   # roofline of configs[2]: ISSUED FP64 lane-flops of the controller + physics kernels (PMC; an upper bound of the useful ones --
        # the controller's source has no op-counting build) and the HBM-side bytes of the 20+ launches of an Env.step, when the counters describe this tree
        o2 = pmc_for_this_tree().get("osc")
        r2 = rows[-1]
        if o2 and o2.get("envs") == n and "env_steps_per_s" in r2:
            ach = o2["valu_flop_issued_per_env_step"] * r2["env_steps_per_s"] / 1e12
            ea = o2.get("valu_flop_exec_active_per_env_step")
            r2["roofline"] = dict(bound="fp64_valu", achieved=ach, peak=FP64_VALU_PEAK_TFLOPS, unit="TFLOP/s", frac=ach / FP64_VALU_PEAK_TFLOPS, flops="issued (upper bound of useful)",
                                  exec_active=None if not ea else dict(achieved=ea * r2["env_steps_per_s"] / 1e12, frac=ea * r2["env_steps_per_s"] / 1e12 / FP64_VALU_PEAK_TFLOPS,
                                                                       active_lane_frac=o2.get("active_lane_frac"),
                                                                       note="issued x the EXEC-active share of each kernel's lanes (PMC SQ_THREAD_CYCLES_VALU / (64 SQ_INSTS_VALU), normalised by the "
                                                                            "lane-per-leg physics kernel): what the controller's 16-lane rows leave idle by masking is out; of an active row's 16 lanes 13 "
                                                                            "carry a dof (14 a QP variable), so the useful share of the controller's part is at most 13/16 of this"),
                                  traffic=o2["hbm_bytes_per_env_step_batch"], algorithmic_bytes=o2["algorithmic_bytes_per_env_step_batch"],
                                  kernel="cassie::g16::env_ctrl_g16_kernel<2,false> + cassie::leg::env_step_duo_kernel<2> per substep (+ hand-over passes)",
                                  note="two launches per substep, each re-staging the state records (and the physics kernel its hand-over workspace): latency- and "
                                       "launch-bound, not a throughput ceiling")
    except Exception:
        pass
    # (c') the scripted standing controller of squatting.py-style loops: substeps/s (one controller call per substep)
    env = VE.CassieVecEnv(n, kind="stand", control_mode="OSC", n_substeps=1, auto_reset=False, device=0)
    env.use_torch_stream()
    zp = torch.full((n,), 0.9, dtype=torch.float64, device="cuda:0")
    zv = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    env._chk(env.L.CassieVecStandingStep(env.h, VE.CONTROL_MODES["OSC"], zp.data_ptr(), zv.data_ptr(), 20))
    env.reset_counters()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        env._chk(env.L.CassieVecStandingStep(env.h, VE.CONTROL_MODES["OSC"], zp.data_ptr(), zv.data_ptr(), 20))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c = env.counters()
    env.close()
    rows.append(dict(workload="configs[2]_standing_controller_osc", note="standing_controller_osc(zpos=0.9) + StepOsc, 200 substeps",
                     envs=n, controller_substeps_per_s=n * 200 / dt, env_steps_equiv_per_s=n * 20 / dt, cleanup_frac=c["cleanup_frac"], k1_frac=c["k1_frac"]))
    # (c'') configs[0]'s controller at scale: standing_controller_jacobian + StepJacobian per substep (squatting.py's loop, 65 536 robots)
    env = VE.CassieVecEnv(n, kind="stand", control_mode="Jacobian", n_substeps=1, auto_reset=False, device=0)
    env.use_torch_stream()
    env._chk(env.L.CassieVecStandingStep(env.h, VE.CONTROL_MODES["Jacobian"], zp.data_ptr(), zv.data_ptr(), 20))
    env.reset_counters()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        env._chk(env.L.CassieVecStandingStep(env.h, VE.CONTROL_MODES["Jacobian"], zp.data_ptr(), zv.data_ptr(), 20))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c = env.counters()
    q, _ = env.get_state_host()
    env.close()
    rows.append(dict(workload="configs[0]_standing_controller_jacobian_at_scale", note="standing_controller_jacobian(zpos=0.9) + StepJacobian, 200 substeps (squatting.py's controller, 65 536 robots)",
                     envs=n, controller_substeps_per_s=n * 200 / dt, env_steps_equiv_per_s=n * 20 / dt, pelvis_z_mean=float(q[:, 1].mean()),
                     cleanup_frac=c["cleanup_frac"], k1_frac=c["k1_frac"]))
    # (d) configs[4]: Cassie3d, 16 384 envs, torque mode U(+-ctrlrange), 10 substeps per step
    from cassierl_amd.vec_env3d import Cassie3dVec, CTRL_RANGE
    n3 = 16384
    e3 = Cassie3dVec(n3)
    ids3 = torch.arange(n3, device="cuda:0")
    acts = [R.random_actions(5, ids3, t, -CTRL_RANGE, CTRL_RANGE) for t in range(60)]
    for t in range(30):
        e3.step(acts[t], 10)
    e3.synchronize()
    t0 = time.perf_counter()
    for t in range(30, 60):
        e3.step(acts[t], 10)
    e3.synchronize()
    dt = time.perf_counter() - t0
    row = dict(workload="configs[4]_cassie3d_torque_random", note="cassie3d_stiff.xml physics, random torques, robots fall during the run",
               envs=n3, warmup_steps=30, steps=30, env_steps_per_s=n3 * 30 / dt, ms_per_step=dt / 30 * 1e3)
    try:   # roofline of configs[4]: counted useful FP64 flops of exactly this workload (tools/count_flops.py, op-counting CPU build of
        # the lane-per-leg kernel's source) / the measured step time, against the FP64 vector peak
        uf = json.load(open(os.path.join(ROOT, "profiles", "useful_flops.json")))["cassie3d_torque_random"]["flop_per_env_step"]
        ach = uf * n3 * 30 / dt / 1e12
        row["roofline"] = dict(bound="fp64_valu", achieved=ach, peak=FP64_VALU_PEAK_TFLOPS, unit="TFLOP/s", frac=ach / FP64_VALU_PEAK_TFLOPS,
                               useful_flop_per_env_step=uf, kernel="cassie3d::leg::env_step3d_leg_kernel<32> (+ its hand-over tier)",
                               issued_flop_per_env_step=(pmc_for_this_tree().get("cassie3d") or {}).get("valu_flop_issued_per_env_step"),
                               traffic=(pmc_for_this_tree().get("cassie3d") or {}).get("hbm_bytes_per_launch"),
                               note="16 environments per wavefront, one wavefront per SIMD at 16 384 envs: the step time is ONE wavefront's dependent chain "
                                    "(500 Gauss-Seidel sweeps of ~11 us), not throughput; issued FP64 and HBM bytes: profiles/<tag>_pmc.json, section cassie3d")
    except Exception:
        pass
    if hasattr(e3, "counters"):
        row.update(e3.counters())
    e3.close()
    rows.append(row)
    # (e) configs[3]'s caller: the TRPO outer loop of trpo_cassie.py on the batched environment (walk env / PD, `n` envs x 8 steps per
    #     iteration; policy step, sampler, returns / baseline, Fisher-vector products and line search as HIP kernels)
    try:
        rows.append(trpo_outer_loop(n, 1, 0))
    except Exception as ex:
        rows.append(dict(workload="trpo_outer_loop_walk_pd", error=repr(ex)))
    return rows


def trpo_outer_loop(n, world, device, warm=4, iters=6, on_desync=None, timeout_s=300.0):
    """configs[3]'s caller (rllab/envs/trpo_cassie.py:21-48): TRPO iterations on `n` envs per rank x 8 Env.steps -- rollout, baseline,
    gradient, ten Fisher-vector products, line search -- with the data-parallel collectives of the update (all_reduce of the
    ~1.3 k-parameter gradient / Fisher-vector products / line-search scalars / baseline normal equations, all_gather of the per-env
    returns) timed one by one with HIP events.  Every rank calls this; the timed region is barrier + synchronize on both sides and
    the MAX over ranks.  Returns the row (meaningful on rank 0).

    The stage is made of collectives, so with more than one rank: (1) everything that can fail LOCALLY -- the env, its hand-over
    workspace, the policy -- is constructed first, and the ranks agree on success with one all_reduce(MIN); if any rank failed, every rank
    skips the stage and the row says so; (2) a failure AFTER that point leaves the ranks out of step for good: `on_desync(reason)` is called
    (the caller prints the headline it has already measured) and the process exits non-zero -- also from a watchdog when the stage does
    not finish within `timeout_s` because another rank died inside it; (3) T.COMM is reset and the env closed on every path."""
    import threading
    import torch
    from cassierl_amd import rollout as R
    from cassierl_amd import trpo as T
    from cassierl_amd.trajectory import default_gait
    algo, err = None, None
    try:
        if os.environ.get("CASSIE_TEST_HOOKS") == "1" and os.environ.get("CASSIE_TEST_TRPO_FAIL_RANK") == os.environ.get("RANK", "0"):
            raise RuntimeError("test hook: this rank fails to construct the TRPO stage")
        algo = T.make_cassie_trpo(n, kind="walk", control_mode="PD", device=device, trajectory=default_gait(), batch_size=n * world * 8, sync_policy=False)
    except Exception as ex:
        err = repr(ex)
    if R.min_over_ranks(0.0 if err else 1.0, device="cuda:%d" % device) < 1.0:
        if algo is not None:
            algo.env.close()
        return dict(workload="trpo_outer_loop_walk_pd", skipped=True, error=err or "another rank could not construct the stage; skipped on every rank")

    def desync(reason):
        try:
            if on_desync is not None:
                on_desync(reason)
        finally:
            os._exit(6)   # the ranks are out of step: nothing collective can follow, and a process that has touched the GPU is not re-executed

    # Watchdog thread (world > 1): fires when the stage has not finished within timeout_s, or at once when this process is told to terminate
    # (SIGTERM: torchrun's agent and bench.py's own launcher stop the surviving ranks that way when one rank dies).  The main thread may
    # be blocked inside a collective (C++, no Python signal handler can run there), so the signal is caught through the wake-up pipe:
    # the C-level handler writes a byte, the watchdog thread selects on the other end.
    dog = stop_dog = None
    if world > 1:
        import select
        import signal
        import socket as _socket
        r_sock, w_sock = _socket.socketpair()
        w_sock.setblocking(False)
        old_handler = signal.signal(signal.SIGTERM, lambda *_a: None)
        old_fd = signal.set_wakeup_fd(w_sock.fileno(), warn_on_full_buffer=False)
        done = threading.Event()

        def watch():
            ready, _, _ = select.select([r_sock], [], [], timeout_s)
            if done.is_set():
                return
            desync("terminated (another rank died?) inside the TRPO stage" if ready else
                   "the TRPO stage did not finish within %.0f s (a rank died inside a collective?)" % timeout_s)

        dog = threading.Thread(target=watch, daemon=True)
        dog.start()

        def stop_dog():
            done.set()
            signal.set_wakeup_fd(old_fd)
            signal.signal(signal.SIGTERM, old_handler)
            try:
                w_sock.send(b"x")   # wakes the watchdog, which sees `done`
            except OSError:
                pass
    try:
        T.broadcast_initial_policy(algo)
        if os.environ.get("CASSIE_TEST_HOOKS") == "1" and os.environ.get("CASSIE_TEST_TRPO_DIE_RANK") == os.environ.get("RANK", "0"):
            os._exit(9)   # test hook: this rank dies inside the stage, after the ranks agreed to run it
        for _ in range(warm):
            algo.train_iteration()
        T.COMM = T.CommTimer() if world > 1 else None
        torch.cuda.synchronize(); R.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        roll_s = upd_s = 0.0
        for _ in range(iters):
            st = algo.train_iteration()
            roll_s += st.get("seconds_rollout", 0.0); upd_s += st.get("seconds_update", 0.0)   # host clocks around the two halves (they include the queue's drain)
        torch.cuda.synchronize(); R.barrier(); torch.cuda.synchronize()
        dt = R.max_over_ranks(time.perf_counter() - t0, device="cuda:%d" % device)
        comm = T.COMM.summary() if T.COMM is not None else {}
    except Exception as ex:
        if world > 1:
            desync("the TRPO stage failed on this rank inside its collectives: %r" % (ex,))
        raise
    finally:
        if stop_dog is not None:
            stop_dog()
        T.COMM = None
        algo.env.close()
    per_iter = {k: dict(calls_per_iteration=v["calls"] / iters, ms_per_iteration=v["total_ms"] / iters, mean_ms=v["mean_ms"], max_ms=v["max_ms"]) for k, v in comm.items()}
    return dict(workload="trpo_outer_loop_walk_pd",
                note="TRPO iterations (rollout of 8 Env.steps + update) on %d envs per rank x %d rank(s); env-steps of the rollouts per second of the whole loop" % (n, world),
                envs=n, envs_total=n * world, ranks=world, iterations=iters, env_steps_per_s=iters * n * world * 8 / dt, ms_per_iteration=dt / iters * 1e3,
                samples_per_iteration=n * world * 8, kl=st.get("kl"), backtracks=st.get("backtracks"),
                rollout_ms_per_iteration=roll_s / iters * 1e3, update_ms_per_iteration=upd_s / iters * 1e3,
                collectives_ms=per_iter, collective_ms_per_iteration=sum(v["ms_per_iteration"] for v in per_iter.values()))


def pmc_for_this_tree():
    """profiles/pmc_traffic.json (profiles/summarize_pmc.py) if it describes THIS source tree (hash of cassierl_amd/csrc), else {}."""
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        from cassierl_amd.build import source_hash
        return pmc if pmc.get("csrc_sha16") == source_hash() else {}
    except Exception:
        return {}


def roofline_object(n_local, kernel_ms, dominant, pmc):
    """The ceiling that binds this path is the FP64 vector unit, not HBM: `achieved` = USEFUL FP64 flops of the
    algorithm per launch (counted by an op-counting build of the kernel's own source for this workload's row mix, tools/count_flops.py
    -> profiles/useful_flops.json; inside a Gauss-Seidel step only the owner lane counts) / the dominant kernel's launch time measured
    here with HIP events; `peak` = 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz.  `issued` (PMC: VALU instruction counters x 64 lanes,
    an upper bound of the useful work) and `traffic` (PMC: HBM bytes per launch) are only given when profiles/pmc_traffic.json
    describes THIS source tree and batch size.  The HBM figures BASELINE.json asks for are the `hbm` sub-object."""
    try:
        useful = json.load(open(os.path.join(ROOT, "profiles", "useful_flops.json")))["pd_bench"]["flop_per_env_step"]
    except Exception:
        useful = None
    sec = kernel_ms * 1e-3
    issued = pmc.get("valu_flop_per_env_step")
    ach = None if not useful else useful * n_local / sec / 1e12
    hbm_gbps = ALGO_BYTES_PER_ENV_STEP * n_local / sec / 1e9
    return {"bound": "fp64_valu", "achieved": ach, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": None if ach is None else ach / FP64_VALU_PEAK_TFLOPS, "traffic": pmc.get("hbm_bytes_per_launch"),
            "kernel": dominant + " (+ its hand-over passes)", "kernel_ms": kernel_ms,
            "useful_flop_per_env_step": useful,
            # the same launch time priced with the count of rounds 3-4 (0.3004 Mflop per env-step: before the structurally zero Jacobian terms, the
            # halved connect diagonals and the two empty row slots left the counted source) -- the figure comparable with earlier rounds' `frac`
            "frac_with_r04_count": USEFUL_FLOP_R04 * n_local / sec / 1e12 / FP64_VALU_PEAK_TFLOPS, "useful_flop_per_env_step_r04": USEFUL_FLOP_R04,
            "issued": None if not issued else {"achieved": issued * n_local / sec / 1e12, "frac": issued * n_local / sec / 1e12 / FP64_VALU_PEAK_TFLOPS,
                                               "flop_per_env_step": issued, "source": pmc.get("source")},
            "hbm": {"achieved": hbm_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": hbm_gbps / HBM_PEAK_GBPS,
                    "algo_bytes_per_env_step": ALGO_BYTES_PER_ENV_STEP, "algo_bytes_per_launch": ALGO_BYTES_PER_ENV_STEP * n_local,
                    "traffic_bytes_per_launch": pmc.get("hbm_bytes_per_launch")},
            "flop_model": "useful: +,-,* = 1 (a*b+c = 2); /, sqrt, 1/x, exp = 1; sincos = 2; compare/select/move = 0; reset pass of a "
                          "terminated environment included.  issued = 64 x (ADD + MUL + TRANS + 2 FMA) FP64 wave-instructions",
            "note": "FP64 vector unit: the joint Gauss-Seidel sweeps (a lane per environment, ~48 % of the launch) run at the 4-cycle issue rate of a lone "
                    "wavefront, the lane-per-leg set-up / finish (~30 %) at ~5 cycles per instruction, the phase hand-over through the per-wavefront "
                    "workspace ~12 %, per-step glue ~8 %; `traffic` is L2-to-fabric bytes (97 % of it that workspace, Infinity-Cache resident), not "
                    "HBM-bound (hbm.frac)"}


# ------------------------------------------------------------------------------------------------ one rank
def worker(args):
    import torch
    from cassierl_amd import rollout as R
    from cassierl_amd.trajectory import default_gait
    from cassierl_amd.vec_env import CassieVecEnv

    if os.environ.get("CASSIE_TEST_HOOKS") == "1" and os.environ.get("CASSIE_TEST_FAIL_RANK") == os.environ.get("RANK", "0"):
        return 7   # test hook (tests/test_gpu_bench.py, only with CASSIE_TEST_HOOKS=1): this rank dies before the rendezvous
    rank, local_rank, world = R.init_distributed()
    if world != max(1, args.gpus):
        sys.stderr.write("bench.py: --gpus %d but %d rank(s) joined (WORLD_SIZE); refusing to report a mislabelled number\n" % (args.gpus, world))
        return 3
    dev = R.local_device(local_rank) if world > 1 else 0
    torch.cuda.set_device(dev)
    device = "cuda:%d" % dev
    n_local = args.envs_per_gpu
    lo, hi = R.shard_bounds(n_local * world, rank, world)
    gait = default_gait()
    traj = dict(time=gait.time, qpos=gait.qpos)

    env = CassieVecEnv(n_local, kind="walk", control_mode="PD", n_substeps=10, auto_reset=True, device=dev)
    env.set_trajectory(traj["time"], traj["qpos"])
    env.use_torch_stream()
    out = env.alloc()
    ids = torch.arange(lo, hi, device=device)
    low, high = env.action_space.low, env.action_space.high
    total = args.warmup + args.steps
    actions = [R.random_actions(1, ids, t, low, high) for t in range(total)]  # resident in HBM before timing
    returns = torch.zeros(n_local, dtype=torch.float64, device=device)
    dones = torch.zeros(1, dtype=torch.int64, device=device)
    env.reset(out)
    # Rollout bookkeeping per Env.step: returns += reward (what the one collective gathers) and the episode count, ONE launch of the library
    # (CassieVecAccumulate; until r06 three torch expressions = four launches, ~30 us of every timed step: profiles/r06_accumulate.txt)
    for t in range(args.warmup):
        _, rew, dn = env.step(actions[t], out)
        env.accumulate(rew, dn, returns, dones)
    # Fixed pre-roll, separate from the caller's --warmup: a fresh process on a fresh box ramps its clocks over the first few
    # hundred milliseconds (r02: the driver's `--warmup 5` left the 20 timed steps 19 % slower than steady state), so Env.steps
    # of the same workload run for at least PREROLL_SECONDS before the timed region, whatever --warmup says.  Their actions come
    # from a different stream (seed 9), so the timed steps consume exactly actions[warmup:total] as before.
    # Everything with a one-time cost (RCCL communicator set-up in the first gather, the counters' memset, event objects) happens
    # BEFORE the pre-roll, so that nothing but the mandated synchronize / barrier / synchronize sits between the last pre-roll step
    # and the first timed one: r03 kernel trace of the driver's invocation (--steps 20 --warmup 5) showed a 30 ms idle gap there,
    # after which the dominant kernel ran 12 % slow and took ~10 launches to recover (clocks drop when the GPU idles) -- with 20
    # timed steps that was 1.6-2.6 ms per step against 1.43 ms in steady state.
    R.gather_returns(returns)
    env.reset_counters()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    kev0, kev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    kev0.record(); kev1.record()   # (first use of a timing event outside the timed region too)
    # The number of pre-roll steps is a function of the batch size only (a wall-clock bound made the state at the start of
    # the timed region differ from run to run and from rank to rank): ~0.4 s of Env.steps at the measured 1.4 ms per 65 536 envs.
    pre_actions = [R.random_actions(9, ids, t, low, high) for t in range(10)]
    preroll_steps = preroll_count(n_local)
    torch.cuda.synchronize()
    for k in range(preroll_steps):   # the timed loop's body, kernel for kernel: the first use of a torch kernel costs its module load
        _, rew, dn = env.step(pre_actions[k % 10], out)   # (r03: a torch kernel first ran inside the timed region -- 8-25 ms of host time in
        env.accumulate(rew, dn, returns, dones)             # its first step; r02's 15 % driver gap was the same thing)
    torch.cuda.synchronize()
    returns.zero_(); dones.zero_()
    env.reset_counters()
    torch.cuda.synchronize()
    R.barrier()
    torch.cuda.synchronize()
    # Two HIP events bracket the timed region on the stream its kernels are launched on (use_torch_stream above): GPU time per
    # Env.step = the dominant kernel + its two hand-over passes (~12 us) + the return accumulation (one launch, ~3 us).  (Events around every single step would agree with rocprofv3's per-kernel average even more directly, but their
    # barrier packets cost 0.15 ms per step -- they would change the number being reported.)
    t0 = time.perf_counter()
    kev0.record()
    for t in range(args.warmup, total):
        _, rew, dn = env.step(actions[t], out)
        env.accumulate(rew, dn, returns, dones)
    kev1.record()
    ev0.record()
    all_returns = R.gather_returns(returns)  # the single collective of the rollout batch (RCCL over xGMI)
    ev1.record()
    torch.cuda.synchronize()
    R.barrier()
    torch.cuda.synchronize()
    elapsed = R.max_over_ranks(time.perf_counter() - t0, device=device)
    gather_ms = ev0.elapsed_time(ev1)
    ranks_joined = int(R.max_over_ranks(world, device=device))
    counters = env.counters()

    kernel_ms = kev0.elapsed_time(kev1) / args.steps
    q, v = env.get_state_host()
    finite = bool(np.isfinite(q).all() and np.isfinite(v).all())
    tier = env.tier_info()   # what the library chose for this handle (batch size, flags, CASSIE2D_* overrides), not a re-derivation of its rule
    env.close()
    backend = R.dist.get_backend() if R.dist.is_initialized() else None

    rc = 0
    line = None
    if rank == 0:
        n_total = n_local * world
        value = n_total * args.steps / elapsed
        achieved_gbps = ALGO_BYTES_PER_ENV_STEP * n_local / (kernel_ms * 1e-3) / 1e9
        # PMC-derived figures (profiles/pmc_traffic.json, written by profiles/summarize_pmc.py from a rocprofv3 --pmc session) are
        # only reported when they describe THIS source tree (hash of cassierl_amd/csrc) and this batch size; otherwise null.
        pmc = {}
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        except Exception:
            pmc = {}
        from cassierl_amd.build import source_hash
        pmc_ok = pmc.get("envs") == n_local and pmc.get("csrc_sha16") == source_hash()
        dominant = FIRST_TIER_KERNEL[tier["first_tier"]]
        line = {
            "metric": "env-steps/sec (whole node) for Cassie2d batched rollout", "value": value, "unit": "env-steps/s",
            "n_gpus": ranks_joined, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic", "schema": BENCH_SCHEMA,
            "config": {"workload": "%d parallel Cassie2d envs per GPU (%s), random-policy rollout, PD mode, 10 substeps/step, walk env "
                                   "reward/done/auto-reset, reference semantics (flags=0)"
                                   % (n_local, "BASELINE north_star target size; configs[1] workload" if n_local == ENVS_PER_GPU else
                                      ("configs[1] as written" if n_local == 4096 else "configs[1] workload at a non-default size")),
                       "envs_per_gpu": n_local, "envs_total": n_total, "substeps_per_env_step": 10, "parallelism": "env-shards x%d" % world,
                       "collective": "one all_gather of per-env returns per rollout batch", "gather_ms": gather_ms,
                       "backend": backend, "ranks_joined": ranks_joined, "preroll_steps": preroll_steps,
                       "first_tier": tier["first_tier"], "duo_workspace_MB": tier["duo_workspace_bytes"] / 1e6, "duo_claim_table_slots": tier["duo_table_slots"]},
            "roofline": roofline_object(n_local, kernel_ms, dominant, pmc if pmc_ok else {}),
            "physics_substeps_per_s": value * 10, "returns_checksum": float(all_returns.sum().item()), "finite": finite,
            "episodes_terminated_per_env_step": float(dones.item()) / (n_local * args.steps),
            "cleanup_frac": counters["cleanup_frac"], "k1_frac": counters["k1_frac"], "nonfinite_resets": counters["nonfinite_resets"],
            "workload_note": "reference quirk Q3 (stale qstate) makes reward < 0.6 on every step, so every env terminates and auto-resets "
                             "each step: the headline is the reference-faithful but degenerate regime; `extra` holds the regimes where robots move",
            "cpu_baseline": None,
        }

    def print_line():
        # RCCL writes its banner ("Librccl path : ...") through C stdio, which is flushed at exit when stdout is a pipe: flush
        # it now so that the JSON line is the LAST line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line), flush=True)

    # configs[3] as written ("512k envs sharded 8 x MI355X, TRPO outer loop, RCCL return gather"): with more than one rank every rank
    # also runs the TRPO loop on its shard (gradient / Fisher-vector-product all-reduces, return gather); the random-policy rollout
    # above stays the headline metric and is ALREADY in `line`: if the ranks fall out of step inside this stage, rank 0 prints the
    # headline with the reason and every rank exits non-zero (trpo_outer_loop: on_desync).
    trpo_row = None
    if world > 1 and not args.no_trpo:
        def on_desync(reason):
            if rank == 0:
                line["config"]["trpo_outer_loop"] = dict(workload="trpo_outer_loop_walk_pd", error=reason)
                print_line()
            sys.stderr.write("bench.py rank %d: %s\n" % (rank, reason))
        trpo_row = trpo_outer_loop(n_local, world, dev, warm=3, iters=args.trpo_iters, on_desync=on_desync, timeout_s=args.trpo_timeout)

    if rank == 0:
        if not finite or ranks_joined != max(1, args.gpus):
            rc = 4
        if trpo_row is not None:
            line["config"]["trpo_outer_loop"] = trpo_row
            line["config"]["trpo_outer_loop_env_steps_per_s"] = trpo_row.get("env_steps_per_s")
        if world == 1 and not args.no_extra:
            try:
                line["extra"] = extra_workloads(traj, n_local)
                # the regimes where robots move, fall and lie on the ground, where the driver's record keeps them (env-steps/s)
                line["config"]["env_steps_per_s_other_workloads"] = {
                    r["workload"]: r.get("env_steps_per_s", r.get("env_steps_equiv_per_s")) for r in line["extra"] if "workload" in r}
                line["config"]["trpo_outer_loop_env_steps_per_s"] = line["config"]["env_steps_per_s_other_workloads"].get("trpo_outer_loop_walk_pd")
            except Exception as ex:  # the headline stays valid; say what failed
                line["extra"] = [{"error": repr(ex)}]
        if world == 1 and not args.no_cpu_baseline:
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            # two CPU legs on this box's host cores: the SAME SOURCE as the HIP kernel through its host backend (the fair one), and
            # the oracle (dense 3-D O(n^3) restatement, the parity checker)
            try:
                orc = cpu_baseline(traj, cores)   # first: it binds the oracle's -O3 -march=native build before anything loads the library
            except Exception as ex:
                orc = {"error": repr(ex)}
            try:
                line["cpu_baseline"] = cpu_same_source(traj, cores)
            except Exception as ex:
                line["cpu_baseline"] = {"error": "same-source leg: " + repr(ex)}
            line["cpu_baseline"]["oracle"] = orc
            if "extra" in line:
                try:
                    line["extra"] += cpu_legs_other_configs()
                except Exception as ex:
                    line["extra"].append({"error": "cpu legs: " + repr(ex)})
    if R.dist.is_initialized():
        R.dist.barrier()
        R.dist.destroy_process_group()
    if rank == 0:
        print_line()
    return rc


def main():
    args = parse_args()
    if args.cpu_legs_only:
        print(json.dumps(cpu_legs_other_configs()))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    sys.exit(worker(args))


if __name__ == "__main__":
    main()
