"""ctypes wrapper around oracle/libsmr_oracle.so (CPU ORACLE - test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_DIR, "libsmr_oracle.so")
INFO_N = 15
NJ = 7


def build(force=False):
    src = os.path.join(_DIR, "smr_oracle.cpp")
    hdr = os.path.join(_DIR, "..", "include", "smr_scene.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    subprocess.check_call(["g++", "-O2", "-fopenmp", "-shared", "-fPIC", "-std=c++17", "-o", _LIB_PATH, src])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.smo_create.restype = C.c_void_p
        L.smo_create.argtypes = [C.c_void_p, C.c_int32, C.c_uint64, C.c_int32]
        L.smo_destroy.argtypes = [C.c_void_p]
        L.smo_scene_sizeof.restype = C.c_size_t
        L.smo_gjk.restype = C.c_double
        L.smo_link_planet_piece_distance.restype = C.c_double
        L.smo_self_pair_distance.restype = C.c_double
        L.smo_support_vertex_visits.restype = C.c_int64
        L.smo_vel_upper_a1max.restype = C.c_double
        L.smo_vel_upper_a1max.argtypes = [C.c_double] * 6
        L.smo_vel_peak_bruteforce.restype = C.c_double
        L.smo_vel_peak_bruteforce.argtypes = [C.c_double] * 7
        L.smo_braking_trajectory_unsafe.restype = C.c_int32
        L.smo_philox_word.restype = C.c_uint32
        L.smo_philox_word.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _d(x):
    return np.ascontiguousarray(x, dtype=np.float64)


class OracleEnv:
    """Batched float64 oracle env with the same call shape as the product's VecSafeMotionsEnv."""

    def __init__(self, scene_bundle, num_envs, seed=0, num_threads=1, env_index_base=0):
        self.L = lib()
        assert self.L.smo_scene_sizeof() == C.sizeof(type(scene_bundle.struct)), "smr_scene_t layout mismatch"
        self.bundle = scene_bundle
        self.B = int(num_envs)
        self.obs_size = scene_bundle.obs_size
        self.h = self.L.smo_create(C.cast(scene_bundle.pointer(), C.c_void_p), self.B, int(seed), int(num_threads))
        if not self.h:
            raise ValueError("smo_create rejected the scene")
        self.h = C.c_void_p(self.h)
        if env_index_base:
            self.L.smo_set_env_index_base(self.h, C.c_uint32(int(env_index_base)))

    def close(self):
        if self.h:
            self.L.smo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, mask=None):
        obs = np.zeros((self.B, self.obs_size), dtype=np.float32)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self.L.smo_reset(self.h, _p(m), _p(obs))
        if mask is not None:
            self.L.smo_observe(self.h, _p(obs))
        return obs

    def observe(self):
        obs = np.zeros((self.B, self.obs_size), dtype=np.float32)
        self.L.smo_observe(self.h, _p(obs))
        return obs

    def step(self, actions):
        act = _d(actions).reshape(self.B, NJ)
        obs = np.zeros((self.B, self.obs_size), dtype=np.float32)
        reward = np.zeros(self.B, dtype=np.float64)
        done = np.zeros(self.B, dtype=np.uint8)
        reason = np.zeros(self.B, dtype=np.int8)
        info = np.zeros((self.B, INFO_N), dtype=np.float64)
        self.L.smo_step(self.h, _p(act), _p(obs), _p(reward), _p(done), _p(reason), _p(info))
        return obs, reward, done, reason, info

    def get_state(self):
        B = self.B
        st = dict(q=np.zeros((B, NJ)), v=np.zeros((B, NJ)), a=np.zeros((B, NJ)),
                  planet_idx=np.zeros((B, 2), dtype=np.int32), step_count=np.zeros(B, dtype=np.int32),
                  moving_hit=np.zeros(B, dtype=np.int32), reset_count=np.zeros(B, dtype=np.uint32),
                  ball=np.zeros((B, 10)), ball_i=np.zeros((B, 5), dtype=np.int32),
                  target=np.zeros((B, 5)), target_count=np.zeros(B, dtype=np.uint32))
        self.L.smo_get_state(self.h, _p(st["q"]), _p(st["v"]), _p(st["a"]), _p(st["planet_idx"]),
                             _p(st["step_count"]), _p(st["moving_hit"]), _p(st["reset_count"]), _p(st["ball"]), _p(st["ball_i"]),
                             _p(st["target"]), _p(st["target_count"]))
        return st

    def set_state(self, q=None, v=None, a=None, planet_idx=None, step_count=None, moving_hit=None, reset_count=None,
                  ball=None, ball_i=None, target=None, target_count=None):
        def f(x, dt, shape):
            return None if x is None else np.ascontiguousarray(np.broadcast_to(np.asarray(x, dtype=dt), shape))
        B = self.B
        arrs = [f(q, np.float64, (B, NJ)), f(v, np.float64, (B, NJ)), f(a, np.float64, (B, NJ)),
                f(planet_idx, np.int32, (B, 2)), f(step_count, np.int32, (B,)), f(moving_hit, np.int32, (B,)),
                f(reset_count, np.uint32, (B,)), f(ball, np.float64, (B, 10)), f(ball_i, np.int32, (B, 5)),
                f(target, np.float64, (B, 5)), f(target_count, np.uint32, (B,))]
        self.L.smo_set_state(self.h, *[_p(x) for x in arrs])

    # ---- component entry points ----
    def fk(self, q):
        pos = np.zeros((8, 3)); rot = np.zeros((8, 3, 3))
        self.L.smo_fk(self.h, _p(_d(q)), _p(pos), _p(rot))
        return pos, rot

    def safe_acc_range(self, p, v, a):
        lo = np.zeros(NJ); hi = np.zeros(NJ); code = np.zeros(NJ, dtype=np.int32)
        self.L.smo_safe_acc_range(self.h, _p(_d(p)), _p(_d(v)), _p(_d(a)), _p(lo), _p(hi), _p(code))
        return lo, hi, code

    def distances(self, q, planet_idx):
        out = np.zeros(4)
        idx = np.ascontiguousarray(planet_idx, dtype=np.int32)
        self.L.smo_distances(self.h, _p(_d(q)), _p(idx), _p(out))
        return out

    def ball_distances(self, q, ball, k):
        out = np.zeros(4)
        self.L.smo_ball_distances(self.h, _p(_d(q)), _p(_d(ball)), C.c_int32(int(k)), _p(out))
        return out

    def link_planet_piece_distance(self, q, link, planet, piece, idx):
        return self.L.smo_link_planet_piece_distance(self.h, _p(_d(q)), C.c_int32(link), C.c_int32(planet),
                                                     C.c_int32(piece), C.c_int32(idx))

    def body_frames(self, q):
        n = int(self.bundle.struct.n_bodies)
        pos, rot = np.zeros((n, 3)), np.zeros((n, 9))
        self.L.smo_body_frames(self.h, _p(_d(q)), _p(pos), _p(rot))
        return pos, rot.reshape(n, 3, 3)

    def self_pair_distance(self, q, i):
        return self.L.smo_self_pair_distance(self.h, _p(_d(q)), C.c_int32(i))

    def braking_acceleration(self, q, v, a):
        """(braking acceleration clipped into the safe range, robot_stopped): ctlp.py:3494-3506, btg.py:44-93"""
        b = np.zeros(NJ)
        stopped = self.L.smo_braking_acceleration(self.h, _p(_d(q)), _p(_d(v)), _p(_d(a)), _p(b))
        return b, bool(stopped)

    def braking_trajectory_unsafe(self, q, v, a, a1):
        """check_braking_trajectory_method, ctlp.py:3026-3053"""
        return bool(self.L.smo_braking_trajectory_unsafe(self.h, _p(_d(q)), _p(_d(v)), _p(_d(a)), _p(_d(a1))))

    def position_is_valid(self, q):
        return bool(self.L.smo_position_is_valid(self.h, _p(_d(q))))


def gjk(va, Ra, ta, vb, Rb, tb, cutoff=1e30):
    va, vb = _d(va), _d(vb)
    return lib().smo_gjk(_p(va), C.c_int32(len(va)), _p(_d(Ra)), _p(_d(ta)), _p(vb), C.c_int32(len(vb)),
                         _p(_d(Rb)), _p(_d(tb)), C.c_double(cutoff))


def safe_acc_range_joint(plo, phi, V, A, J, dt, p, v, a):
    lo, hi, code = C.c_double(), C.c_double(), C.c_int32()
    lib().smo_safe_acc_range_joint(*[C.c_double(x) for x in (plo, phi, V, A, J, dt, p, v, a)],
                                   C.byref(lo), C.byref(hi), C.byref(code))
    return lo.value, hi.value, code.value


def philox_word(seed, env, stream, tag, draw):
    return lib().smo_philox_word(seed, env, stream, tag, draw)


def vel_upper_a1max(V, A, J, dt, v0, a0):
    return lib().smo_vel_upper_a1max(V, A, J, dt, v0, a0)


def vel_peak_bruteforce(V, A, J, dt, v0, a0, a1):
    return lib().smo_vel_peak_bruteforce(V, A, J, dt, v0, a0, a1)
