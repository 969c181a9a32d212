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


def _run(args, env_extra=None, timeout=900):
    env = dict(os.environ)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    return p.returncode, p.stdout, p.stderr


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
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
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
    rc, out, err = _run(["--gpus", "2", "--steps", "4", "--warmup", "1", "--envs-per-gpu", "2048", "--no-cpu-baseline"], env)
    assert rc == 0, err[-2000:]
    line = _line(out)
    assert line["n_gpus"] == 2 and line["config"]["envs_total"] == 4096 and line["finite"] and "extra" not in line
    # weak scaling bookkeeping: value counts the envs of BOTH ranks
    assert abs(line["value"] - 4096 * 4 / (line["ms_per_step"] * 4e-3)) < 1e-6 * line["value"]
    # the gathered returns cover both shards: same checksum as one rank stepping all 4096 envs (actions are keyed by global id)
    rc1, out1, err1 = _run(["--steps", "4", "--warmup", "1", "--envs-per-gpu", "4096", "--no-cpu-baseline", "--no-extra"])
    assert rc1 == 0, err1[-2000:]
    assert abs(_line(out1)["returns_checksum"] - line["returns_checksum"]) < 1e-6 * abs(line["returns_checksum"])


def test_mismatched_world_size_is_refused():
    env = dict(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    rc, out, err = _run(["--gpus", "4", "--steps", "2", "--warmup", "1", "--envs-per-gpu", "256", "--no-cpu-baseline", "--no-extra"], env)
    assert rc != 0 and "refusing" in err and out.strip() == ""
