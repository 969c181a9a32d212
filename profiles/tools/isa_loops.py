#!/usr/bin/env python3
"""Loops of a function in an AMDGPU .s file (hipcc --cuda-device-only -S): for each backward branch, the span and its instruction mix.
usage: python profiles/tools/isa_loops.py file.s <first line> <last line>   (the line range of the function of interest)"""
import re, sys, collections
lines = open(sys.argv[1]).read().split("\n"); lo, hi = int(sys.argv[2]), int(sys.argv[3])
labels = {}
for i in range(lo, hi):
    m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
    if m: labels[m.group(1)] = i
def mix(a, b):
    c = collections.Counter(); n = 0
    for i in range(a, b):
        s = lines[i].strip()
        if not s or s.startswith((".", ";")) or s.endswith(":"): continue
        op = s.split()[0]; n += 1
        if op.startswith("v_fmac_f64") or op.startswith("v_max_f64") or op.startswith("v_fma_f64") or op.startswith("v_mul_f64") or op.startswith("v_add_f64") or op.startswith("v_pk_") : c["f64"] += 1
        elif op.startswith(("v_rcp_f64", "v_sqrt_f64", "v_rsq_f64", "v_div")): c["trans"] += 1
        elif op.startswith("v_cndmask"): c["cndmask"] += 1
        elif op.startswith("v_cmp"): c["vcmp"] += 1
        elif op.startswith(("v_accvgpr", "v_mov", "v_pk_mov")): c["mov/acc"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith(("scratch_", "buffer_", "global_", "flat_")): c["mem"] += 1
        elif op.startswith("s_waitcnt"): c["wait"] += 1
        elif op.startswith("s_nop"): c["nop"] += 1
        elif op.startswith("s_"): c["salu"] += 1
        elif op.startswith("v_"): c["valu_other"] += 1
        else: c["other"] += 1
    return n, c
for i in range(lo, hi):
    m = re.match(r"^\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)", lines[i]) or re.match(r"^\s*s_branch\s+(\.LBB\d+_\d+)", lines[i])
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        a = labels[m.group(1)]; n, c = mix(a, i + 1)
        print("loop %s lines %d-%d: %d instr  %s" % (m.group(1), a, i, n, dict(c)))
