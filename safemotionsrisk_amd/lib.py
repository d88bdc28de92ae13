"""ctypes binding of libsmr_hip.so (include/safemotions_hip.h) and its in-tree build.

This is the reference-side stub a maintainer adds under ``SafeMotionsEnv``.  There is
no CPU fallback: if the shared library is missing, or no GPU is present, every constructor raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

from .scene import Scene

_PKG = os.path.dirname(os.path.abspath(__file__))
#: SMR_LIB: another build of the same library (the profiling build of scripts/profile_rows.py, A/B variants); never a different backend
LIB_PATH = os.environ.get("SMR_LIB") or os.path.join(_PKG, "libsmr_hip.so")
CSRC = os.path.join(_PKG, "csrc")
INCLUDE = os.path.join(os.path.dirname(_PKG), "include")

ABI_VERSION = 10
INFO_N = 15
K_COUNT = 10
KERNEL_NAMES = ("plan", "collide", "finish", "reset", "spawn", "brake", "risk", "keys", "enumerate", "jobs")

#: every symbol include/safemotions_hip.h declares
EXPORTS = ("smr_abi_version", "smr_scene_sizeof", "smr_create", "smr_destroy", "smr_last_error", "smr_num_envs",
           "smr_obs_size", "smr_reset", "smr_step", "smr_observe", "smr_get_state", "smr_set_state",
           "smr_synchronize", "smr_set_profiling", "smr_get_profile", "smr_test_fk", "smr_test_safe_acc_range",
           "smr_test_distances", "smr_test_ball_distances", "smr_test_self_bounds", "smr_test_brake_plan", "smr_mlp_create", "smr_mlp_destroy", "smr_mlp_forward",
           "smr_set_risk_networks", "smr_set_risk_state_config", "smr_set_risk_rollout", "smr_episode_count", "smr_link_distances")


class HostState(C.Structure):
    _fields_ = [("q", C.c_void_p), ("v", C.c_void_p), ("a", C.c_void_p), ("planet_idx", C.c_void_p),
                ("step_count", C.c_void_p), ("moving_hit", C.c_void_p), ("reset_count", C.c_void_p),
                ("ball", C.c_void_p), ("ball_i", C.c_void_p), ("target", C.c_void_p), ("target_count", C.c_void_p)]


def _sources():
    return [os.path.join(CSRC, "smr_hip.hip"), os.path.join(CSRC, "smr_device.h"), os.path.join(CSRC, "smr_tables.h"),
            os.path.join(CSRC, "smr_mlp.h"), os.path.join(INCLUDE, "safemotions_hip.h"), os.path.join(INCLUDE, "smr_scene.h")]


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> safemotionsrisk_amd/libsmr_hip.so (cross-compiles without a GPU)."""
    srcs = _sources()
    if not force and os.path.exists(LIB_PATH) and all(not os.path.exists(s) or os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", LIB_PATH, srcs[0]]
    cmd += os.environ.get("SMR_HIPCC_FLAGS", "").split()       # profiling builds (scripts/profile_rows.py: -DSMR_PROFILE_ROWS)
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def load():
    """Load the HIP library; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch bundles its own HIP runtime; it has to be in the process before libsmr_hip.so is bound, or the library
    # binds the system copy and the two runtimes do not see each other's device, streams and memory
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the product has no CPU path)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.smr_abi_version.restype = C.c_int
    L.smr_scene_sizeof.restype = C.c_size_t
    L.smr_last_error.restype = C.c_char_p
    L.smr_last_error.argtypes = [C.c_void_p]
    L.smr_create.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_int32, C.c_uint64, C.c_void_p, C.POINTER(C.c_void_p)]
    L.smr_destroy.argtypes = [C.c_void_p]
    L.smr_num_envs.argtypes = [C.c_void_p]
    L.smr_obs_size.argtypes = [C.c_void_p]
    L.smr_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.smr_step.argtypes = [C.c_void_p] + [C.c_void_p] * 6
    L.smr_observe.argtypes = [C.c_void_p, C.c_void_p]
    L.smr_get_state.argtypes = [C.c_void_p, C.POINTER(HostState)]
    L.smr_set_state.argtypes = [C.c_void_p, C.POINTER(HostState)]
    L.smr_synchronize.argtypes = [C.c_void_p]
    L.smr_set_profiling.argtypes = [C.c_void_p, C.c_int32]
    L.smr_get_profile.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.smr_test_fk.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    L.smr_test_safe_acc_range.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                          C.c_void_p, C.c_void_p]
    L.smr_test_distances.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    L.smr_test_ball_distances.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    L.smr_test_self_bounds.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    L.smr_test_brake_plan.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.smr_mlp_create.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
    L.smr_mlp_destroy.argtypes = [C.c_void_p]
    L.smr_mlp_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    L.smr_set_risk_networks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]
    L.smr_set_risk_state_config.argtypes = [C.c_void_p, C.c_int32]
    L.smr_episode_count.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    L.smr_link_distances.argtypes = [C.c_void_p, C.c_void_p]
    L.smr_set_risk_rollout.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]
    if L.smr_abi_version() != ABI_VERSION:
        raise RuntimeError("libsmr_hip.so ABI %d != binding ABI %d" % (L.smr_abi_version(), ABI_VERSION))
    if L.smr_scene_sizeof() != C.sizeof(Scene):
        raise RuntimeError("smr_scene_t layout mismatch: library %d bytes, binding %d bytes"
                           % (L.smr_scene_sizeof(), C.sizeof(Scene)))
    _lib = L
    return L


class SmrError(RuntimeError):
    pass


class SamplingError(SmrError, ValueError):
    """SMR_E_SAMPLING: a rejection-sampling loop ran out of attempts; the reference raises ValueError there (ctlp.py:2127-2136)."""


E_SAMPLING = -4


def check(L, handle, rc, what):
    if rc != 0:
        msg = L.smr_last_error(handle)
        raise (SamplingError if rc == E_SAMPLING else SmrError)("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
