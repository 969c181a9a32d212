#!/usr/bin/env python3
"""Turn gpurun_out/<tag>/ (written by profiles/collect.sh on the GPU box) into the committed summaries:
  profiles/<tag>_bench_4096.json, _bench_65536.json, _other_configs.jsonl, _kernel_stats.csv, _pmc.json
and refresh profiles/pmc_traffic.json (read by bench.py for roofline.traffic).

HBM bytes per launch follow MI355X_MICROARCH.md's rocprofv3 section: FETCH_SIZE and WRITE_SIZE are collected in
separate passes and are in KiB... the guide's gfx950 correction doubles FETCH_SIZE (wide coalesced reads are
reported at half size)."""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def counter_per_kernel(path):
    """mean counter value per dispatch for each kernel name, skipping the first (cold) dispatch of each"""
    vals = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            vals[r["Kernel_Name"]].append((float(r["Counter_Value"]), int(r["Scratch_Size"]), int(r["LDS_Block_Size"]), int(r["Grid_Size"])))
    out = {}
    for k, v in vals.items():
        body = v[1:] if len(v) > 1 else v
        out[k] = dict(mean=sum(x[0] for x in body) / len(body), n=len(v), scratch=v[0][1], lds=v[0][2], grid=v[0][3])
    return out


def traffic(src, prefix, step_kernels):
    fe = counter_per_kernel(os.path.join(src, prefix + "FETCH_SIZE", "pmc_counter_collection.csv"))
    wr = counter_per_kernel(os.path.join(src, prefix + "WRITE_SIZE", "pmc_counter_collection.csv"))
    rows, total = [], 0.0
    for k in fe:
        if not any(s in k for s in step_kernels):
            continue
        fetch_b, write_b = 2.0 * fe[k]["mean"] * 1024.0, wr[k]["mean"] * 1024.0
        rows.append(dict(kernel=k, dispatches=fe[k]["n"], grid=fe[k]["grid"], scratch_bytes_per_lane=fe[k]["scratch"], lds_bytes_per_block=fe[k]["lds"],
                         fetch_size_kb_raw=fe[k]["mean"], write_size_kb_raw=wr[k]["mean"], hbm_read_bytes=fetch_b, hbm_write_bytes=write_b))
        total += fetch_b + write_b
    return rows, total


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", tag)
    for name in ("bench_4096.json", "bench_65536.json", "other_configs.jsonl", "trpo_65536.jsonl"):
        shutil.copy(os.path.join(src, name), os.path.join(HERE, "%s_%s" % (tag, name)))
    shutil.copy(os.path.join(src, "stats", "bench_kernel_stats.csv"), os.path.join(HERE, tag + "_kernel_stats.csv"))
    pd_rows, pd_total = traffic(src, "pmc_", ("env_step_g16_kernel", "env_step_kernel"))
    osc_rows, osc_total = traffic(src, "pmc_osc_", ("env_ctrl_step_g16_kernel", "env_ctrl_step_kernel"))
    c3_rows, c3_total = traffic(src, "pmc_3d_", ("env_step3d_kernel",))
    summary = dict(source="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, no tracing), tests/prof_step.py 4096 6 [PD|OSC]; "
                          "mean over dispatches after the first; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950)",
                   n_envs=4096, pd=dict(kernels=pd_rows, hbm_bytes_per_env_step_launch=pd_total),
                   osc=dict(kernels=osc_rows, hbm_bytes_per_env_step_launch=osc_total),
                   cassie3d=dict(kernels=c3_rows, hbm_bytes_per_env_step_launch=c3_total))
    with open(os.path.join(HERE, tag + "_pmc.json"), "w") as f:
        json.dump(summary, f, indent=1)
    with open(os.path.join(HERE, "pmc_traffic.json"), "w") as f:
        json.dump(dict(source="profiles/%s_pmc.json (profiles/collect.sh + profiles/summarize.py)" % tag, hbm_bytes_per_launch=int(pd_total),
                       note="per Env.step of the bench workload (4096 envs): fast-path kernel + clean-up kernel; algorithmic bytes are 905 B x 4096 = 3.7 MB, "
                            "the rest is register-spill scratch written at the kernel prologue and flushed at the kernel boundary"), f, indent=1)
    print(json.dumps(summary, indent=1)[:3000])


if __name__ == "__main__":
    main()
