#!/usr/bin/env python3
"""Compiler-reported resources of every kernel of a translation unit (not a test): registers, scratch, LDS, occupancy.
usage: python tools/kernel_resources.py tu_g16 [tu_3d ...]   (extra hipcc flags through CASSIE_HIPCC_FLAGS)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cassierl_amd", "csrc")
for unit in sys.argv[1:] or ["tu_g16"]:
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "--cuda-device-only",
           "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", unit + ".hip"] + os.environ.get("CASSIE_HIPCC_FLAGS", "").split()
    err = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
    cur = {}
    for line in err.splitlines():
        m = re.search(r"remark:\s+(.+?): (\S+) \[-Rpass", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k == "Function Name":
            cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()[:90]}
        else:
            cur[k] = v
        if k == "LDS Size [bytes/block]":
            print("%-92s vgpr %-4s agpr %-3s sgpr %-4s scratch %-5s lds %-6s occ %s sgpr-spill %s vgpr-spill %s" % (cur["name"], cur.get("VGPRs"), cur.get("AGPRs"), cur.get("TotalSGPRs"),
                  cur.get("ScratchSize [bytes/lane]"), cur.get("LDS Size [bytes/block]"), cur.get("Occupancy [waves/SIMD]"), cur.get("SGPRs Spill"), cur.get("VGPRs Spill")))
