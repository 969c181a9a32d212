"""Build-time guard against a register-allocator defect of the hipcc this tree is built with (ROCm 7.2.0, AMD clang 22.0.0git roc-7.2.0): a VECTOR
instruction placed in an EXEC HOLE -- between the label a divergent region's skip branch jumps to and the `s_or_b64 exec, exec, s[A:B]` that
re-enables the lanes which skipped the region:

        s_and_saveexec_b64 s[4:5], s[12:13]        ; if (act & (ncon < 3)) { st_pair(...) }   -- the last contact candidate of sub_setup
        s_cbranch_execz .LBB1_83
        ... region body (partial EXEC) ...
    .LBB1_83:
        v_mov_b32_e32 v56, v241                    ; <-- live-range split copy of the LANE NUMBER, executed with the region's EXEC
        s_mov_b32 s90, s39
        s_mov_b32 s89, s38
        s_or_b64 exec, exec, s[4:5]
        ...
        v_mov_b32_e32 v19, v56
        v_lshl_add_u32 v54, v19, 3, s0             ; LDS address of the lane's contact-pair descriptor (ld_pair)
        ds_read2st64_b64 v[72:75], v54 offset0:42 offset1:43

This is the -ftrivial-auto-var-init=pattern build of env_step_duo_kernel<1> (r06): the allocator split the live range of the lane number and put
the copy at the head of the join block, AHEAD of the exec restore (scalar copies may sit there; a vector copy only reaches the lanes that were active
inside the region -- none at all if the region was skipped).  Every robot whose last contact candidate (rear toe sphere) was out of contact then read
its contact descriptors through a stale lane number: wrong rows, wrong forces, results off by 1e-5 .. 1e+3 after one substep, on the GPU only, in a
build whose LLVM IR is the shipped build's (the pattern stores all die; what survives are phis on infeasible paths).  It is the r05 "device build
depends on dead code" phenomenon: any change to the code around the set-up moves the allocator's split points.

`scan()` finds the pattern in assembly (`hipcc -S`) or in `llvm-objdump -d` output of a code object; `check_object()` extracts the gfx950 code
object from a compiled translation unit and scans it.  cassierl_amd.build runs it on every unit it compiles and REFUSES to link a library that
contains a hit (first retrying the unit with flags that move the split points); tests/test_isa_guard.py holds the known-bad excerpt.
usage: python -m cassierl_amd.isa_guard <file.s | objdump.txt | unit.o> ..."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

VECTOR = ("v_", "ds_", "buffer_", "global_", "scratch_", "flat_")
LANE_OPS = ("v_readlane", "v_writelane", "v_readfirstlane")   # ignore EXEC by definition
OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
_SAVEEXEC = re.compile(r"s_(?:and|or|xor|andn2|orn2|nand|nor|xnor)_saveexec_b64 (s\[\d+:\d+\]|vcc),")


def _parse(text):
    """-> list of (position key, instruction text, kernel name); positions are labels' line indices (assembly) or addresses (objdump)."""
    insts, labels, syms, kernel = [], {}, {}, None
    objdump = "// 0000" in text
    for raw in text.split("\n"):
        m = re.match(r"^([0-9a-fA-F]{8,16}) <(\S+)>:", raw)
        if m:
            kernel = m.group(2)
            syms[kernel] = int(m.group(1), 16)
            continue
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", raw)
        if m and not raw.startswith(".L"):
            kernel = m.group(1)
            continue
        m = re.match(r"^(\.L\w+):", raw)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        t = raw.strip()
        if not t or t.startswith((";", ".", "//")):
            continue
        addr = None
        if objdump:
            m = re.match(r"^(.*?)\s*//\s*([0-9A-Fa-f]+):", t)
            if not m:
                continue
            t, addr = m.group(1).strip(), int(m.group(2), 16)
            rest = raw.split("//", 1)[1]
            tm = re.search(r"<(\S+?)\+0x([0-9a-fA-F]+)>", rest)
            if tm and t.startswith(("s_cbranch", "s_branch")):
                t = t.split()[0] + " @%s+%d" % (tm.group(1), int(tm.group(2), 16))
        else:
            t = t.split(";")[0].strip()
        insts.append((addr, t, kernel))
    by_addr = {a: i for i, (a, _, _) in enumerate(insts) if a is not None}
    return insts, labels, syms, by_addr


def scan(text, want=""):
    """-> [(kernel, instruction index, vector instruction, skip-target description)] for every vector instruction in an exec hole."""
    insts, labels, syms, by_addr = _parse(text)
    hits = []
    for i, (_, t, kernel) in enumerate(insts):
        m = _SAVEEXEC.match(t)
        if not m or i + 1 >= len(insts) or (want and not (kernel and want in kernel)):
            continue
        saved = m.group(1)
        b = re.match(r"s_cbranch_execz (\S+)", insts[i + 1][1])
        if not b:
            continue
        tgt = b.group(1)
        if tgt.startswith("@"):
            sym, off = tgt[1:].rsplit("+", 1)
            k = by_addr.get(syms.get(sym, -1 << 60) + int(off))
        else:
            k = labels.get(tgt)
        if k is None or k <= i:
            continue
        pending = []
        for j in range(k, min(k + 40, len(insts))):
            x = insts[j][1]
            if re.match(r"s_or_b64 exec, exec, " + re.escape(saved) + r"\s*$", x):
                hits += [(kernel, n, txt, "%s (saveexec %s at instruction %d, exec restored at %d)" % (tgt, saved, i, j)) for n, txt in pending]
                break
            if x.startswith(VECTOR) and not x.startswith(LANE_OPS):
                pending.append((j, x))
            elif x.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")) or re.search(r"\bexec\b", x):
                break
    return hits


def disassemble_object(obj):
    """llvm-objdump -d of the gfx950 code object bundled in a hipcc-compiled translation unit."""
    d = tempfile.mkdtemp(prefix="isa_guard_")
    try:
        o = os.path.join(d, "unit.o")
        shutil.copy(obj, o)
        subprocess.run([OBJDUMP, "--offloading", o], check=True, capture_output=True, cwd=d)
        cos = [f for f in os.listdir(d) if "amdgcn" in f]
        if not cos:
            return ""   # a unit without device code (the C-ABI host file)
        return "".join(subprocess.run([OBJDUMP, "-d", os.path.join(d, f)], check=True, capture_output=True, text=True).stdout for f in sorted(cos))
    finally:
        shutil.rmtree(d, ignore_errors=True)


def check_object(obj):
    return scan(disassemble_object(obj))


def describe(hits):
    return "\n".join("  %s: `%s` ahead of the exec restore of %s" % (k, txt, where) for k, _, txt, where in hits)


if __name__ == "__main__":
    bad = 0
    for f in sys.argv[1:]:
        hits = check_object(f) if f.endswith(".o") else scan(open(f).read())
        print("%s: %d vector instruction(s) in an exec hole" % (f, len(hits)))
        if hits:
            print(describe(hits))
        bad += len(hits)
    sys.exit(1 if bad else 0)
