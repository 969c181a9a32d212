"""Build the HIP extension (libcassie2d.so) in-tree for gfx950.

hipcc cross-compiles without a GPU, so this runs in the CPU-only container as well as on
the MI355X box.  The shared object is kept in cassierl_amd/lib/ (git-ignored, shipped by gpurun).
One translation unit per kernel family (csrc/tu_*.hip + the C-ABI in cassie_cabi.hip), compiled
in parallel to objects under lib/obj/ and linked into one shared library.
"""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libcassie2d.so")
UNITS = ["cassie_cabi", "tu_base", "tu_g16", "tu_leg", "tu_leg_seg", "tu_duo", "tu_duo_hf", "tu_hf", "tu_ctrl", "tu_ctrl_g16", "tu_3d", "tu_trpo", "tu_trpo_baseline"]
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"] + os.environ.get("CASSIE_HIPCC_FLAGS", "").split()
# Units that instantiate cassie_leg_core.h (two lanes per environment: one launch, segments / reset, 64 environments per wavefront) are
# compiled with -ffp-contract=on: a multiply-add is fused where the SOURCE writes `a * b + c` in one expression (a front-end decision),
# not where the back end's combiner finds it profitable in the code around it (hipcc's default, `fast`).  The three units compile the same
# source into different kernels whose results are compared bit for bit (tests/test_gpu_duo.py, the segment-order tests): with `fast` the
# back end fused one multiply-add in one kernel and not in the other (r05: state records apart by an ulp in a handful of fields).
UNIT_FLAGS = {"tu_leg": ["-ffp-contract=on"], "tu_leg_seg": ["-ffp-contract=on"], "tu_duo": ["-ffp-contract=on"], "tu_duo_hf": ["-ffp-contract=on"],
              "tu_hf": ["-ffp-contract=on"]}   # (holds the height-field form of the two-lanes kernel, which env_step_duo_hf_kernel is compared with bit for bit)


def _deps():
    d = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".inc"))]
    d += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    return d


def source_hash():
    """sha256 (first 16 hex digits) over the kernel / C-ABI sources: ties PMC-derived figures (profiles/pmc_traffic.json) and
    build stamps to the source tree they were measured on."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h", ".inc")):
            h.update(f.encode())
            with open(os.path.join(CSRC, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def _units():
    return [u for u in UNITS if os.path.exists(os.path.join(CSRC, u + ".hip"))]


STAMP = os.path.join(OBJDIR, "FLAGS.stamp")
LIBSTAMP = LIB + ".stamp"   # travels with the library (gpurun ships built files): "<flags>\n<hash of sources + public headers>"


def _tree_hash():
    import hashlib
    h = hashlib.sha256()
    for d in sorted(_deps()):
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _flags_id():
    return " ".join(FLAGS) + " | " + " ".join("%s: %s" % (u, " ".join(f)) for u, f in sorted(UNIT_FLAGS.items()))


def _stamp_matches():
    """The objects under lib/obj were compiled with exactly the current FLAGS (an A/B or profiling build -- CASSIE_HIPCC_FLAGS --
    leaves a different stamp, and the next plain build() then recompiles everything instead of mixing objects)."""
    try:
        return open(STAMP).read() == _flags_id()
    except OSError:
        return False


def needs_build():
    """Decided by CONTENT, not by file times: a copy of the tree to another box (gpurun, the driver's push) keeps no promise about
    mtimes, and a spurious rebuild there costs minutes of a GPU lease (the library is ~2.5 CPU-minutes of hipcc)."""
    if not os.path.exists(LIB):
        return True
    try:
        return open(LIBSTAMP).read() != _flags_id() + "\n" + _tree_hash()
    except OSError:
        return True


def build(force=False, verbose=False, only=None):
    """Compile every translation unit (or those named in `only`, re-using the other objects) and link lib/libcassie2d.so."""
    if not force and not only and not needs_build():
        return LIB
    os.makedirs(OBJDIR, exist_ok=True)
    import fcntl
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:  # ranks of one job may call build() at the same time
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not only and not needs_build():  # another process built it while this one waited
            return LIB
        return _build_locked(force, verbose, only)


# ISA guard (cassierl_amd/isa_guard.py): the hipcc of this image (ROCm 7.2.0) can place a live-range split copy of a VECTOR register ahead of the
# `s_or_b64 exec` of a control-flow join -- the copy then only reaches the lanes that were active inside the divergent region (r06: the
# auto-var-init=pattern build of env_step_duo_kernel<1>, wrong on the GPU for every robot whose last contact candidate was out of contact).  Every
# unit is scanned after it is compiled; a unit with a hit is recompiled with flags that move the allocator's split points (each of them gave a
# clean AND correct build of the affected kernel, profiles/r06_exec_hole.txt) and the library is not linked while any unit has one.
ISA_RETRY_FLAGS = [["-mllvm", "-amdgpu-opt-exec-mask-pre-ra=0"], ["-mllvm", "-disable-machine-sink"], ["-mllvm", "-disable-machine-licm"]]


def _compile_guarded(cmd, obj, verbose, log=None):
    """Run the compile command; scan the object; retry with ISA_RETRY_FLAGS while the scan finds an exec hole.  Returns the extra flags used."""
    from . import isa_guard
    tried = []
    for extra in [[]] + ISA_RETRY_FLAGS:
        c = cmd[:1] + extra + cmd[1:]
        if verbose:
            print(" ".join(c), flush=True)
        subprocess.check_call(c, cwd=CSRC)
        hits = isa_guard.check_object(obj)
        if not hits:
            if extra:
                msg = "isa_guard: %s compiled clean with %s after: %s" % (os.path.basename(obj), " ".join(extra), "; ".join(tried))
                print(msg, flush=True)
                if log is not None:
                    log.append(msg)
            return extra
        tried.append("%s -> %d exec hole(s): %s" % (" ".join(extra) or "(regular flags)", len(hits), hits[0][2]))
    raise RuntimeError("isa_guard: %s still has a vector instruction ahead of an exec restore with every retry flag set:\n%s" %
                       (obj, "\n".join(tried)))


def _build_locked(force, verbose, only):
    if not _stamp_matches():
        force = True   # objects of another flag set: recompile every unit
        only = None
    newest = max(os.path.getmtime(d) for d in _deps())

    def compile_one(u):
        obj = os.path.join(OBJDIR, u + ".o")
        stale = force or not os.path.exists(obj) or os.path.getmtime(obj) < newest
        if only is not None:
            stale = u in only or not os.path.exists(obj)
        if stale:
            cmd = [HIPCC] + FLAGS + UNIT_FLAGS.get(u, []) + ["-c", "-o", obj, os.path.join(CSRC, u + ".hip")]
            _compile_guarded(cmd, obj, verbose)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, _units()))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(_flags_id())
    with open(LIBSTAMP, "w") as f:
        f.write(_flags_id() + "\n" + _tree_hash())
    return LIB


VARDIR = os.path.join(LIBDIR, "variants")


def variant_path(name):
    return os.path.join(VARDIR, "libcassie2d_%s.so" % name)


def build_variant(name, units, extra_flags, verbose=False, guarded=True):
    """A second build of the library in which the translation units `units` are compiled with `extra_flags` ON TOP of the regular flags
    (FLAGS + UNIT_FLAGS: the contraction mode the bit-identity between kernels depends on stays what the shipped build uses) and every other
    unit is the regular build's object: lib/variants/libcassie2d_<name>.so, loaded through CASSIE2D_LIB.  Used by the A/B scripts under
    profiles/tools and by the auto-var-init guard builds (tests/test_gpu_build_guard.py).  Content-stamped like the main library.
    guarded: the units go through the ISA guard like the shipped build's (scan, retry flags); False keeps whatever the compiler produced
    (diagnosis: the known-wrong builds of profiles/r06_exec_hole.txt)."""
    build()
    os.makedirs(VARDIR, exist_ok=True)
    out = variant_path(name)
    stamp = out + ".stamp"
    ident = _flags_id() + "\n" + _tree_hash() + "\n" + " ".join(units) + " | " + " ".join(extra_flags) + (" | guarded" if guarded else " | unguarded")
    notes = []
    try:
        if os.path.exists(out) and open(stamp).read() == ident:
            return out
    except OSError:
        pass

    def compile_one(u):
        obj = os.path.join(VARDIR, "%s_%s.o" % (u, name))
        cmd = [HIPCC] + FLAGS + UNIT_FLAGS.get(u, []) + list(extra_flags) + ["-c", "-o", obj, os.path.join(CSRC, u + ".hip")]
        if guarded:
            _compile_guarded(cmd, obj, verbose, notes)
        else:
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd, cwd=CSRC)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        mine = list(ex.map(compile_one, units))
    objs = [os.path.join(OBJDIR, u + ".o") for u in _units() if u not in units] + mine
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    with open(stamp, "w") as f:
        f.write(ident)
    with open(out + ".notes", "w") as f:   # which units needed a retry flag set (tests print it)
        f.write("\n".join(notes))
    return out


# Guard builds: the units of the 64-environments-per-wavefront kernel with every automatic variable pre-set to a byte pattern /
# to zero.  A read of an undefined value in device code -- or a register-allocation accident of this 512-register kernel -- shows up as a
# difference between these builds and the shipped one (tests/test_gpu_build_guard.py compares them bit for bit on the GPU).
# "view": the r05 experiment whose device build depended on dead code, recreated (-DDUO_VIEW_EXPERIMENT, cassie_duo_core.h: joint_solve_view): a second
# joint-solve path compiled into env_step_duo_kernel behind a flag no caller sets -- 1.2 KB of scratch instead of 264 B, another register allocation of
# the whole kernel.  Must give the shipped build's results with the branch never taken AND with it taken (CASSIE_DUO_VIEW_FLAG).
GUARD_VARIANTS = {"avi_pattern": (["tu_duo", "tu_duo_hf"], ["-ftrivial-auto-var-init=pattern"]),
                  "avi_zero": (["tu_duo", "tu_duo_hf"], ["-ftrivial-auto-var-init=zero"]),
                  "view": (["tu_duo"], ["-DDUO_VIEW_EXPERIMENT"])}
DUO_VIEW_FLAG = 0x20000000   # CASSIE_DUO_VIEW_FLAG (cassie_duo_core.h), honoured by the "view" build only


def build_guards(verbose=False):
    libs = {n: build_variant(n, u, f, verbose) for n, (u, f) in sorted(GUARD_VARIANTS.items())}
    # the pattern build of tu_duo exactly as the compiler emits it, NOT passed through the ISA guard: with hipcc 7.2.0 its env_step_duo_kernel<1> holds
    # the exec-hole copy and is wrong on the GPU (the guard's known-bad reference: tests/test_isa_guard.py, tests/test_gpu_build_guard.py)
    libs["avi_pattern_raw"] = build_variant("avi_pattern_raw", ["tu_duo"], ["-ftrivial-auto-var-init=pattern"], verbose, guarded=False)
    return libs


if __name__ == "__main__":
    import sys
    if sys.argv[1:2] == ["--variant"]:   # python -m cassierl_amd.build --variant <name> "<unit> <unit>" <flags...>
        print(build_variant(sys.argv[2], sys.argv[3].split(), sys.argv[4:], verbose=True))
    elif sys.argv[1:2] == ["--guards"]:
        print("\n".join(build_guards(verbose=True).values()))
    else:
        only = sys.argv[1:] or None
        print(build(force=only is None, verbose=True, only=only))
