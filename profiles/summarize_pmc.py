#!/usr/bin/env python3
"""Condense gpurun_out/<tag>/pmc_* (profiles/collect_pmc.sh) into profiles/<tag>_pmc.json and refresh
profiles/pmc_traffic.json (read by bench.py for roofline.traffic and the counted FP64 figure).

Conventions (MI355X_MICROARCH.md, HBM / rocprofv3 sections):
  * FETCH_SIZE and WRITE_SIZE come from separate passes, in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide
    coalesced reads, so it is doubled; WRITE_SIZE is taken as is;
  * SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles (x4 = shader cycles);
  * SQ_INSTS_* count wave-instructions; the FP64 flop figure is 64 lanes x (ADD + MUL + TRANS + 2 FMA), i.e. ISSUED lane-flops
    whatever the EXEC mask or the usefulness of a lane -- an upper bound of the useful work, stated as such.
Every value is the mean per dispatch over the dispatches after the first (cold) one of each kernel.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
ALGO_BYTES_PER_ENV_STEP = 905


def read_pass(d):
    """{kernel: {counter: mean per dispatch, '_meta': {...}}}"""
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        return {}
    per = defaultdict(lambda: defaultdict(list))
    meta = {}
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"]
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta.setdefault(k, dict(scratch_bytes_per_lane=int(r.get("Scratch_Size", 0) or 0), lds_bytes_per_block=int(r.get("LDS_Block_Size", 0) or 0),
                                    grid=int(r.get("Grid_Size", 0) or 0), vgpr=int(r.get("VGPR_Count", 0) or 0), sgpr=int(r.get("SGPR_Count", 0) or 0),
                                    accum_vgpr=int(r.get("Accum_VGPR_Count", 0) or 0)))
    out = {}
    for k, cs in per.items():
        out[k] = {c: (sum(v[1:]) / len(v[1:]) if len(v) > 1 else v[0]) for c, v in cs.items()}
        out[k]["_dispatches"] = max(len(v) for v in cs.values())
        out[k]["_meta"] = meta[k]
    return out


def workload(src, wl, want, envs, units_per_launch_name):
    merged = defaultdict(dict)
    for grp in ("sq1", "sq2", "sq3", "fetch", "write"):
        for k, v in read_pass(os.path.join(src, "pmc_%s_%s" % (wl, grp))).items():
            if any(w in k for w in want):
                merged[k].update(v)
    rows = []
    for k, v in merged.items():
        m = v.get("_meta", {})
        waves = v.get("SQ_WAVES", 0.0)
        valu = v.get("SQ_INSTS_VALU", 0.0)
        f64 = 64.0 * (v.get("SQ_INSTS_VALU_ADD_F64", 0.0) + v.get("SQ_INSTS_VALU_MUL_F64", 0.0) + v.get("SQ_INSTS_VALU_TRANS_F64", 0.0) +
                      2.0 * v.get("SQ_INSTS_VALU_FMA_F64", 0.0))
        row = dict(kernel=k, dispatches=v.get("_dispatches"), grid=m.get("grid"), scratch_bytes_per_lane=m.get("scratch_bytes_per_lane"),
                   lds_bytes_per_block=m.get("lds_bytes_per_block"), vgpr=m.get("vgpr"), accum_vgpr=m.get("accum_vgpr"), sgpr=m.get("sgpr"),
                   counters={c: x for c, x in v.items() if not c.startswith("_")},
                   valu_insts_per_wave=(valu / waves if waves else None),
                   lds_insts_per_wave=(v.get("SQ_INSTS_LDS", 0.0) / waves if waves else None),
                   wave_cycles_per_wave=(4.0 * v.get("SQ_WAVE_CYCLES", 0.0) / waves if waves else None),
                   cycles_per_valu_inst=(4.0 * v.get("SQ_WAVE_CYCLES", 0.0) / valu if valu else None),
                   valu_busy_frac_of_wave_cycles=(v.get("SQ_ACTIVE_INST_VALU", 0.0) / v["SQ_WAVE_CYCLES"] if v.get("SQ_WAVE_CYCLES") else None),
                   fp64_lane_flops_issued=f64, fp64_valu_insts=sum(v.get(c, 0.0) for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64",
                                                                                           "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_FMA_F64")),
                   hbm_read_bytes=2.0 * 1024.0 * v.get("FETCH_SIZE", 0.0), hbm_write_bytes=1024.0 * v.get("WRITE_SIZE", 0.0))
        row["hbm_bytes"] = row["hbm_read_bytes"] + row["hbm_write_bytes"]
        # r06: SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU) = EXEC-active lanes per VALU instruction (a kernel whose instructions all run with 64
        # active lanes reads 1.015: the calibration is K1d; see active_lane_frac below)
        thr = v.get("SQ_THREAD_CYCLES_VALU", 0.0)
        row["valu_thread_cycles_per_lane_inst"] = thr / (64.0 * valu) if (thr and valu) else None
        # Per SIMD (1024 of them): GRBM_GUI_ACTIVE is summed over the 8 XCDs.  cycles_per_valu_inst above is per WAVEFRONT (how long a
        # wavefront lives per instruction it issues); with W wavefronts resident on a SIMD the SIMD issues W times as often.
        gui = v.get("GRBM_GUI_ACTIVE", 0.0)
        row["simd_cycles_per_valu_inst"] = (gui / 8.0) * 1024.0 / valu if (gui and valu) else None
        row["simd_valu_busy_frac"] = 4.0 * v.get("SQ_ACTIVE_INST_VALU", 0.0) / ((gui / 8.0) * 1024.0) if gui else None
        rows.append(row)
    full = max([r["valu_thread_cycles_per_lane_inst"] or 0.0 for r in rows] + [0.0])
    for r in rows:   # normalised by the fullest kernel of the workload when that one is a lane-per-leg kernel (all lanes active): ~1.015
        t = r["valu_thread_cycles_per_lane_inst"]
        r["active_lane_frac"] = min(1.0, t / max(full, 1.0)) if t else None
    rows.sort(key=lambda r: -(r["counters"].get("SQ_INSTS_VALU", 0.0)))
    tot_flop = sum(r["fp64_lane_flops_issued"] for r in rows)
    tot_active = sum(r["fp64_lane_flops_issued"] * (r["active_lane_frac"] if r["active_lane_frac"] is not None else 1.0) for r in rows)
    tot_bytes = sum(r["hbm_bytes"] for r in rows)
    return dict(envs=envs, launch=units_per_launch_name, kernels=rows, hbm_bytes_per_launch=tot_bytes,
                fp64_lane_flops_issued_per_launch=tot_flop, fp64_lane_flops_issued_per_env_step=tot_flop / envs,
                fp64_lane_flops_exec_active_per_env_step=tot_active / envs)


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", tag)
    pd = workload(src, "pd", ("env_step_duo_kernel", "env_step_leg_kernel", "env_step_g16_kernel", "env_step_kernel"), 65536, "one Env.step of 65 536 envs (10 substeps), walk env / PD, reference semantics")
    osc = workload(src, "osc", ("env_ctrl_g16_kernel", "env_ctrl_kernel", "env_step_duo_kernel<2", "env_step_leg_kernel<2", "env_step_g16_kernel<2", "env_step_kernel<2", "env_osc_fused"), 65536, "one Env.step of 65 536 envs = 10 x (controller kernel + physics kernel + hand-over pass), stand env / OSC QP in every substep; per-dispatch means, i.e. per SUBSTEP for these kernels")
    c3 = workload(src, "c3", ("env_step3d_leg_kernel", "env_step3d_kernel", "env_step3d_pair_kernel"), 16384, "one 10-substep step of 16 384 Cassie3d envs, torque mode")
    summary = dict(source="profiles/collect_pmc.sh %s: rocprofv3 --pmc, one run per counter group, no tracing; see the docstring of profiles/summarize_pmc.py "
                          "for units and the gfx950 FETCH_SIZE correction" % tag, pd=pd, osc=osc, cassie3d=c3)
    with open(os.path.join(HERE, tag + "_pmc.json"), "w") as f:
        json.dump(summary, f, indent=1)
    with open(os.path.join(HERE, "pmc_traffic.json"), "w") as f:
        sys.path.insert(0, ROOT)
        from cassierl_amd.build import source_hash
        useful = None
        try:  # counted useful flops of the same workload (tools/count_flops.py writes it)
            useful = json.load(open(os.path.join(HERE, "useful_flops.json")))["pd_bench"]["flop_per_env_step"]
        except Exception:
            pass
        dom = pd["kernels"][0]["kernel"] if pd["kernels"] else None
        json.dump(dict(source="profiles/%s_pmc.json" % tag, envs=65536, csrc_sha16=source_hash(), dominant_kernel=dom,
                       useful_flop_per_env_step=useful, hbm_bytes_per_launch=int(pd["hbm_bytes_per_launch"]),
                       algorithmic_bytes_per_launch=ALGO_BYTES_PER_ENV_STEP * 65536,
                       valu_flop_per_env_step=pd["fp64_lane_flops_issued_per_env_step"],
                       # the other configs the bench line attaches a roofline object to (per Env.step of 10 substeps: the OSC kernels are
                       # profiled per substep, hence x 10)
                       osc=dict(envs=65536, valu_flop_issued_per_env_step=10.0 * osc["fp64_lane_flops_issued_per_env_step"],
                                valu_flop_exec_active_per_env_step=10.0 * osc["fp64_lane_flops_exec_active_per_env_step"],
                                active_lane_frac={r["kernel"][:48]: r["active_lane_frac"] for r in osc["kernels"][:2]},
                                hbm_bytes_per_env_step_batch=int(10.0 * osc["hbm_bytes_per_launch"]), algorithmic_bytes_per_env_step_batch=(ALGO_BYTES_PER_ENV_STEP + 8) * 65536),
                       cassie3d=dict(envs=16384, valu_flop_issued_per_env_step=c3["fp64_lane_flops_issued_per_env_step"], hbm_bytes_per_launch=int(c3["hbm_bytes_per_launch"])),
                       note="bench workload at 65 536 envs: packed kernel + its hand-over pass per Env.step.  valu_flop_per_env_step = ISSUED FP64 lane-flops "
                            "(64 x (ADD + MUL + TRANS + 2 FMA) wave-instructions) / envs: an upper bound of the useful flops"), f, indent=1)
    for name, w in (("pd", pd), ("osc", osc), ("cassie3d", c3)):
        print(name, "hbm MB/launch %.2f" % (w["hbm_bytes_per_launch"] / 1e6), "fp64 Mflop issued/env-step %.3f" % (w["fp64_lane_flops_issued_per_env_step"] / 1e6))
        for r in w["kernels"]:
            print("   %-70s grid %s scratch %s lds %s vgpr %s  valu/wave %s  cyc/valu %s" % (r["kernel"][:70], r["grid"], r["scratch_bytes_per_lane"],
                  r["lds_bytes_per_block"], r["vgpr"], r["valu_insts_per_wave"] and round(r["valu_insts_per_wave"]), r["cycles_per_valu_inst"] and round(r["cycles_per_valu_inst"], 2)))


if __name__ == "__main__":
    main()
