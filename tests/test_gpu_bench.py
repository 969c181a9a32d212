"""bench.py as the driver runs it: the JSON contract, the self-spawning `--gpus N` launcher, and the RCCL gather on hardware
(a process group is forced for the single rank so that a 1-GPU box exercises the communicator).  -m gpu only."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(args, env_extra=None, timeout=900, attempts=2):
    """bench.py in its own session: a run that outlives `timeout` is stopped together with the ranks it spawned (their process group --
    r05: a two-rank run that hung on one box was killed by the launcher's pid alone, and its ranks went on sharing the GPU with the tests
    behind it), and tried once more (a rendezvous that hangs does so at start-up; the second attempt gets a fresh port)."""
    import signal
    env = dict(os.environ)
    env.update(env_extra or {})
    for attempt in range(attempts):
        p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                             start_new_session=True)
        try:
            out, err = p.communicate(timeout=timeout if attempt == attempts - 1 else min(timeout, 300))
            return p.returncode, out, err
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p.pid, signal.SIGKILL)   # the session we started: launcher + its ranks, nothing else
            except ProcessLookupError:
                pass
            out, err = p.communicate()
            if attempt == attempts - 1:
                return 124, out, err + "\n[test] bench.py timed out"


def _line(out):
    lines = [l for l in out.strip().splitlines() if l.strip()]
    assert lines and lines[-1].startswith("{"), "the JSON line must be the last line of stdout: %r" % lines[-3:]
    return json.loads(lines[-1])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_bench_line_contract_small():
    rc, out, err = _run(["--steps", "6", "--warmup", "2", "--envs-per-gpu", "2048", "--no-cpu-baseline", "--no-extra"])
    assert rc == 0, err[-2000:]
    line = _line(out)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == 6 and line["config"]["envs_per_gpu"] == 2048 and line["finite"]
    assert line["unit"] == "env-steps/s" and line["dtype"] == "f64" and line["scaling"] == "weak"
    r = line["roofline"]
    assert r["bound"] == "fp64_valu" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["hbm"]["unit"] == "GB/s" and abs(r["hbm"]["frac"] - r["hbm"]["achieved"] / r["hbm"]["peak"]) < 1e-12
    assert abs(r["hbm"]["achieved"] - 905 * 2048 / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-9 * r["hbm"]["achieved"]
    assert abs(line["value"] - 2048 * 6 / (line["ms_per_step"] * 6e-3)) < 1e-6 * line["value"]
    assert line["episodes_terminated_per_env_step"] == 1.0  # quirk Q3: the reference-faithful walk env ends every step


def test_rccl_gather_with_a_forced_single_rank_process_group():
    env = dict(CASSIE_FORCE_PROCESS_GROUP="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", LOCAL_RANK="0",
               WORLD_SIZE="1")
    rc, out, err = _run(["--steps", "4", "--warmup", "1", "--envs-per-gpu", "1024", "--no-cpu-baseline", "--no-extra"], env)
    assert rc == 0, err[-2000:]
    line = _line(out)
    assert line["n_gpus"] == 1 and line["config"]["gather_ms"] >= 0.0 and line["finite"]


def test_gpus_n_spawns_n_ranks_or_fails_loudly():
    """`python bench.py --gpus 2` without torchrun: on a box with >= 2 GPUs two NCCL ranks must join (n_gpus == 2 in the line);
    on a 1-GPU box the launcher must refuse with a non-zero exit code instead of silently reporting one rank."""
    import torch
    rc, out, err = _run(["--gpus", "2", "--steps", "4", "--warmup", "1", "--envs-per-gpu", "1024", "--no-cpu-baseline"])
    if torch.cuda.device_count() >= 2:
        assert rc == 0, err[-2000:]
        line = _line(out)
        assert line["n_gpus"] == 2 and line["config"]["envs_total"] == 2048
    else:
        assert rc != 0 and "only 1 device" in err


def test_two_ranks_sharing_one_gpu_exercise_the_n_rank_path():
    """A 1-GPU box cannot host two RCCL ranks (RCCL refuses duplicate devices), but everything else of the N > 1 path can run
    there: the self-spawning launcher, the rendezvous, global env ids per shard, the gather and the max-over-ranks timing --
    with both ranks mapped to device 0 and gloo carrying the two collectives (test hooks CASSIE_DEVICE_MAP / CASSIE_BACKEND)."""
    env = dict(CASSIE_DEVICE_MAP="0,0", CASSIE_BACKEND="gloo")
    rc, out, err = _run(["--gpus", "2", "--steps", "4", "--warmup", "1", "--envs-per-gpu", "2048", "--no-cpu-baseline", "--trpo-iters", "2"], env)
    assert rc == 0, err[-2000:]
    line = _line(out)
    assert line["n_gpus"] == 2 and line["config"]["envs_total"] == 4096 and line["finite"] and "extra" not in line
    # configs[3] as written: with N > 1 the line also carries the TRPO outer loop on the sharded envs, the backend and the collectives
    cfg = line["config"]
    assert cfg["backend"] == "gloo" and cfg["ranks_joined"] == 2
    tr = cfg["trpo_outer_loop"]
    assert "error" not in tr, tr
    assert tr["ranks"] == 2 and tr["envs_total"] == 4096 and tr["samples_per_iteration"] == 4096 * 8 and tr["iterations"] == 2
    assert cfg["trpo_outer_loop_env_steps_per_s"] == tr["env_steps_per_s"] > 0
    for name in ("gradient_all_reduce", "fvp_all_reduce", "line_search_all_reduce", "returns_all_gather", "baseline_all_reduce"):
        c = tr["collectives_ms"][name]
        assert c["calls_per_iteration"] >= 1 and c["ms_per_iteration"] >= 0.0, (name, c)
    assert tr["collectives_ms"]["fvp_all_reduce"]["calls_per_iteration"] == 11   # one per conjugate-gradient iteration (trpo_cassie.py: cg_iters 10) + s'Hs of the step length
    # weak scaling bookkeeping: value counts the envs of BOTH ranks
    assert abs(line["value"] - 4096 * 4 / (line["ms_per_step"] * 4e-3)) < 1e-6 * line["value"]
    # the gathered returns cover both shards: same checksum as one rank stepping all 4096 envs (actions are keyed by global id)
    rc1, out1, err1 = _run(["--steps", "4", "--warmup", "1", "--envs-per-gpu", "4096", "--no-cpu-baseline", "--no-extra"])
    assert rc1 == 0, err1[-2000:]
    assert abs(_line(out1)["returns_checksum"] - line["returns_checksum"]) < 1e-6 * abs(line["returns_checksum"])


def test_bench_under_torchrun_as_the_driver_launches_it():
    """The driver's own command for N > 1 -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py --gpus N --steps K --warmup W` -- with the two ranks mapped onto the one GPU of this box and gloo carrying the collectives (RCCL refuses
    two ranks on one device): ranks take RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment, rank 0 prints the JSON line last."""
    import signal
    env = dict(os.environ, CASSIE_DEVICE_MAP="0,0", CASSIE_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--envs-per-gpu", "2048", "--no-cpu-baseline", "--trpo-iters", "2"]
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=600)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        out, err = p.communicate()
        raise AssertionError("torchrun bench timed out: " + err[-1500:])
    assert p.returncode == 0, err[-2000:]
    line = _line(out)
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["warmup"] == 1 and line["scaling"] == "weak" and line["finite"]
    assert line["config"]["ranks_joined"] == 2 and line["config"]["envs_total"] == 4096 and line["config"]["backend"] == "gloo"
    assert abs(line["value"] - 4096 * 4 / (line["ms_per_step"] * 4e-3)) < 1e-6 * line["value"]
    assert sum(1 for l in out.splitlines() if l.startswith("{")) == 1   # ONE JSON line (rank 0's)


def test_mismatched_world_size_is_refused():
    env = dict(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    rc, out, err = _run(["--gpus", "4", "--steps", "2", "--warmup", "1", "--envs-per-gpu", "256", "--no-cpu-baseline", "--no-extra"], env)
    assert rc != 0 and "refusing" in err and out.strip() == ""


def test_configs3_eight_ranks_of_65536_envs_on_one_gpu():
    """configs[3] = 8 x 65 536 envs, one rank per GPU.  A 1-GPU box cannot give every rank its own device, but it can run the
    whole configuration: the self-spawning launcher starts 8 ranks of 65 536 envs each (all mapped to device 0, gloo carrying the
    collectives -- RCCL refuses duplicate devices), all 8 join, `envs_total` is 524 288, and the gathered returns cover all eight
    shards: their checksum equals ONE rank stepping the same 524 288 global env ids.  What remains untested afterwards is RCCL
    with N > 1 alone."""
    env = dict(CASSIE_DEVICE_MAP="0,0,0,0,0,0,0,0", CASSIE_BACKEND="gloo")
    rc, out, err = _run(["--gpus", "8", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], env, timeout=2400)
    assert rc == 0, err[-2000:]
    line = _line(out)
    assert line["n_gpus"] == 8 and line["config"]["envs_per_gpu"] == 65536 and line["config"]["envs_total"] == 524288
    assert line["finite"] and line["nonfinite_resets"] == 0 and "extra" not in line
    assert abs(line["value"] - 524288 * 3 / (line["ms_per_step"] * 3e-3)) < 1e-6 * line["value"]
    rc1, out1, err1 = _run(["--steps", "3", "--warmup", "1", "--envs-per-gpu", "524288", "--no-cpu-baseline", "--no-extra"], timeout=1200)
    assert rc1 == 0, err1[-2000:]
    one = _line(out1)
    assert one["config"]["envs_total"] == 524288 and one["finite"]
    assert abs(one["returns_checksum"] - line["returns_checksum"]) < 1e-9 * abs(one["returns_checksum"])


def test_launcher_stops_all_ranks_when_one_dies():
    """a rank that dies must take the run down at once (the survivors would otherwise wait in a collective until a
    watchdog fires).  Rank 1 is made to exit before the rendezvous; the launcher must return non-zero within seconds."""
    import time
    env = dict(CASSIE_DEVICE_MAP="0,0", CASSIE_BACKEND="gloo", CASSIE_TEST_HOOKS="1", CASSIE_TEST_FAIL_RANK="1")
    t0 = time.time()
    rc, out, err = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--envs-per-gpu", "256", "--no-cpu-baseline"], env, timeout=600)
    assert rc != 0 and "rank 1 exited" in err and time.time() - t0 < 300
    assert not [l for l in out.strip().splitlines() if l.startswith("{")]


def test_trpo_stage_is_skipped_on_every_rank_when_one_cannot_construct_it():
    """the TRPO stage behind the headline is made of collectives; a rank that fails to build its env / workspace / policy must not leave
    the others waiting in a broadcast.  Rank 1 is made to fail its construction: the ranks agree (all_reduce MIN) to skip, the run ends with rc 0,
    ONE JSON line, the headline measured before the stage intact and the row saying why it was skipped."""
    env = dict(CASSIE_DEVICE_MAP="0,0", CASSIE_BACKEND="gloo", CASSIE_TEST_HOOKS="1", CASSIE_TEST_TRPO_FAIL_RANK="1")
    rc, out, err = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--envs-per-gpu", "2048", "--no-cpu-baseline", "--trpo-iters", "2", "--trpo-timeout", "60"], env, timeout=150, attempts=1)
    assert rc == 0, err[-2000:]
    line = _line(out)
    assert sum(1 for l in out.splitlines() if l.startswith("{")) == 1
    assert line["n_gpus"] == 2 and line["finite"] and line["value"] > 0 and line["config"]["first_tier"] in ("g16", "leg", "duo")   # (g16 by the size rule; the forced-tier suites -- CASSIE2D_LEG / CASSIE2D_DUO -- run this test too)
    row = line["config"]["trpo_outer_loop"]
    assert row["skipped"] and "another rank" in row["error"] and line["config"]["trpo_outer_loop_env_steps_per_s"] is None


def test_headline_survives_a_rank_that_dies_inside_the_trpo_stage():
    """... and a rank that dies INSIDE the stage (after the agreement) takes the run down with a non-zero exit -- but rank 0, told to terminate by the
    launcher while it waits in a collective, still prints the headline it measured before the stage, with the reason in the TRPO row."""
    import time
    env = dict(CASSIE_DEVICE_MAP="0,0", CASSIE_BACKEND="gloo", CASSIE_TEST_HOOKS="1", CASSIE_TEST_TRPO_DIE_RANK="1")
    t0 = time.time()
    rc, out, err = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--envs-per-gpu", "2048", "--no-cpu-baseline", "--trpo-iters", "2", "--trpo-timeout", "60"], env,
                        timeout=150, attempts=1)
    assert rc != 0 and rc != 124 and time.time() - t0 < 120, (rc, err[-1500:])
    line = _line(out)
    assert line["n_gpus"] == 2 and line["finite"] and line["value"] > 0
    assert "error" in line["config"]["trpo_outer_loop"] and "TRPO stage" in line["config"]["trpo_outer_loop"]["error"]
