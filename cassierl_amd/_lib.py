"""ctypes binding of libcassie2d.so (the HIP extension).  No fallback: if the library is
missing or no HIP device is present the import / create call raises."""
import ctypes as ct
import os

from . import build as _build

_LIB = None


class CassieVecConfig(ct.Structure):
    _fields_ = [("env_kind", ct.c_int), ("control_mode", ct.c_int), ("n_substeps", ct.c_int),
                ("flags", ct.c_int), ("auto_reset", ct.c_int)]


EXPORTS = [
    # legacy ABI (include/cassie2d.h)
    "Cassie2dInit", "Reset", "StepOsc", "StepTorque", "StepJacobian", "StepPd", "GetGeneralState",
    "GetOperationalSpaceState", "Display", "Render",
    # batched ABI (include/cassie_vec.h)
    "CassieVecCreate", "CassieVecFree", "CassieVecLastError", "CassieVecNumEnvs", "CassieVecActionDim",
    "CassieVecSetStream", "CassieVecSynchronize", "CassieVecGetCounters", "CassieVecResetCounters", "CassieVecTierInfo", "CassieVecQpIterations", "CassieVecSetTrajectory", "CassieVecSetHeightField", "CassieVecReset", "CassieVecResetTo",
    "CassieVecStep", "CassieVecSubstep", "CassieVecStandingStep", "CassieVecAccumulate", "CassieVecGetState", "CassieVecGetOpState", "CassieVecStatePtr",
    "CassieVecStepHost", "CassieVecGetStateHost", "CassieVecSetStateHost", "CassieVecGetFullStateHost",
    "CassieVecDebugSubstepHost", "CassieVecDebugWorkspaceHost", "CassieVecTimeSteps",
    # batched Cassie3d physics (include/cassie3d_vec.h)
    "Cassie3dVecCreate", "Cassie3dVecFree", "Cassie3dVecLastError", "Cassie3dVecSetStream", "Cassie3dVecSynchronize",
    "Cassie3dVecReset", "Cassie3dVecStep", "Cassie3dVecStatePtr", "Cassie3dVecGetCounters", "Cassie3dVecResetCounters", "Cassie3dVecStepHost", "Cassie3dVecGetStateHost",
    "Cassie3dVecSetStateHost", "Cassie3dVecDebugForwardHost", "Cassie3dVecTimeSteps",
    # fused policy kernels of the TRPO outer loop (include/cassie_trpo.h)
    "CassieTrpoParamCount", "CassieTrpoPartialRows", "CassieTrpoFvp", "CassieTrpoVjp", "CassieTrpoSurrogate", "CassieTrpoCgUpdate", "CassieTrpoPolicyStep", "CassieTrpoSamplerRows", "CassieTrpoSamplerStep",
    "CassieTrpoBaselineFeatures", "CassieTrpoBaselinePredict", "CassieTrpoReturnsAdvantages", "CassieTrpoGramRows", "CassieTrpoGramRowSize", "CassieTrpoBaselineGram", "CassieTrpoRidgeSolve",
]


def lib_path():
    # CASSIE2D_LIB lets experiments (A/B kernel builds) point at another build of the same HIP extension
    return os.environ.get("CASSIE2D_LIB", _build.LIB)


def load():
    """dlopen the HIP extension; raises OSError if it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise OSError("HIP extension %s is missing: run `python -m cassierl_amd.build` (needs hipcc); "
                      "cassierl_amd has no CPU fallback" % path)
    # torch bundles its own libamdhip64.so.7; two HIP runtimes in one process cannot both own the GPU.
    # Importing torch first makes the dynamic loader bind libcassie2d.so to the runtime torch already loaded
    # (same SONAME), so tensors, streams and our kernels share one runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ct.CDLL(path)
    vp, dp, u8p = ct.c_void_p, ct.c_void_p, ct.c_void_p
    L.CassieVecCreate.argtypes = [ct.POINTER(ct.c_void_p), ct.c_int, ct.c_int, ct.POINTER(CassieVecConfig)]
    L.CassieVecFree.argtypes = [vp]
    L.CassieVecFree.restype = None
    L.CassieVecLastError.argtypes = [vp]
    L.CassieVecLastError.restype = ct.c_char_p
    L.CassieVecNumEnvs.argtypes = [vp]
    L.CassieVecActionDim.argtypes = [vp]
    L.CassieVecSetStream.argtypes = [vp, vp]
    L.CassieVecSynchronize.argtypes = [vp]
    L.CassieVecGetCounters.argtypes = [vp, ct.POINTER(ct.c_uint64)]
    L.CassieVecResetCounters.argtypes = [vp]
    if hasattr(L, "CassieVecQpIterations"):
        L.CassieVecQpIterations.argtypes = [vp, ct.POINTER(ct.c_double)]
    if hasattr(L, "CassieVecTierInfo"):   # (absent from A/B libraries built from earlier rounds' sources: CASSIE2D_LIB)
        L.CassieVecTierInfo.argtypes = [vp, ct.POINTER(ct.c_uint64)]
    L.CassieVecSetTrajectory.argtypes = [vp, dp, dp, ct.c_int]
    L.CassieVecSetHeightField.argtypes = [vp, dp, ct.c_int, ct.c_int, ct.c_double, ct.c_double]
    L.CassieVecReset.argtypes = [vp, u8p, dp]
    L.CassieVecResetTo.argtypes = [vp, u8p, dp, dp, dp]
    L.CassieVecStep.argtypes = [vp, dp, dp, dp, u8p, dp]
    L.CassieVecSubstep.argtypes = [vp, ct.c_int, dp, ct.c_int]
    L.CassieVecStandingStep.argtypes = [vp, ct.c_int, dp, dp, ct.c_int]
    if hasattr(L, "CassieVecAccumulate"):   # (absent from A/B libraries built from earlier sources: CASSIE2D_LIB)
        L.CassieVecAccumulate.argtypes = [vp, dp, u8p, dp, vp]
    L.CassieVecGetState.argtypes = [vp, dp, dp]
    L.CassieVecGetOpState.argtypes = [vp, dp]
    L.CassieVecStatePtr.argtypes = [vp]
    L.CassieVecStatePtr.restype = ct.c_void_p
    L.CassieVecStepHost.argtypes = [vp, dp, dp, dp, u8p]
    L.CassieVecGetStateHost.argtypes = [vp, dp, dp]
    L.CassieVecSetStateHost.argtypes = [vp, dp]
    L.CassieVecGetFullStateHost.argtypes = [vp, dp]
    L.CassieVecDebugSubstepHost.argtypes = [vp, ct.c_int, dp, dp]
    if hasattr(L, "CassieVecDebugWorkspaceHost"):
        L.CassieVecDebugWorkspaceHost.argtypes = [vp, dp, ct.c_uint64, ct.POINTER(ct.c_uint64)]
    L.CassieVecTimeSteps.argtypes = [vp, dp, ct.c_int, dp, dp, u8p, ct.POINTER(ct.c_float)]
    L.Cassie3dVecCreate.argtypes = [ct.POINTER(ct.c_void_p), ct.c_int, ct.c_int]
    L.Cassie3dVecFree.argtypes = [vp]
    L.Cassie3dVecFree.restype = None
    L.Cassie3dVecLastError.argtypes = [vp]
    L.Cassie3dVecLastError.restype = ct.c_char_p
    L.Cassie3dVecSetStream.argtypes = [vp, vp]
    L.Cassie3dVecSynchronize.argtypes = [vp]
    L.Cassie3dVecReset.argtypes = [vp, dp, dp]
    L.Cassie3dVecStep.argtypes = [vp, dp, ct.c_int]
    L.Cassie3dVecStatePtr.argtypes = [vp]
    L.Cassie3dVecStatePtr.restype = ct.c_void_p
    L.Cassie3dVecGetCounters.argtypes = [vp, ct.POINTER(ct.c_uint64)]
    L.Cassie3dVecResetCounters.argtypes = [vp]
    L.Cassie3dVecStepHost.argtypes = [vp, dp, ct.c_int]
    L.Cassie3dVecGetStateHost.argtypes = [vp, dp]
    L.Cassie3dVecSetStateHost.argtypes = [vp, dp]
    L.Cassie3dVecDebugForwardHost.argtypes = [vp, dp, dp]
    L.Cassie3dVecTimeSteps.argtypes = [vp, dp, ct.c_int, ct.c_int, ct.POINTER(ct.c_float)]
    L.Cassie2dInit.restype = ct.c_void_p
    _LIB = L
    return L
