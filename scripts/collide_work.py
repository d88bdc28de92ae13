"""GJK calls / iterations per (env, link) lane of the collide kernel's end-of-step sample (CPU, host emulation)."""
import sys, ctypes as C, subprocess
sys.path.insert(0, '.')
import numpy as np
from oracle import oracle as O
from safemotionsrisk_amd.scene import build_scene, SPACE_ENV_CONFIG
lib = '/tmp/libemul_cnt.so'
subprocess.check_call(['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-ffp-contract=fast', '-mfma', '-o', lib, 'tests/host_emul/emul.cpp'])
E = C.CDLL(lib); E.emul_scene_create.restype = C.c_void_p
cnt = (C.c_long * 8).in_dll(E, 'emul_counters')
sc = build_scene(SPACE_ENV_CONFIG, auto_reset=True)
H = C.c_void_p(E.emul_scene_create(C.cast(sc.pointer(), C.c_void_p)))
B = 512
ora = O.OracleEnv(sc, B, seed=0); ora.reset()
rng = np.random.default_rng(0)
for _ in range(7): ora.step(rng.uniform(-1, 1, (B, 7)))
st = ora.get_state()
calls = np.zeros((B, 8)); its = np.zeros((B, 8))
for e in range(B):
    q = st['q'][e].astype(np.float32); idx = st['planet_idx'][e].astype(np.int32)
    for l in range(8):
        cnt[2] = cnt[3] = 0
        out = np.zeros(3, np.float32)
        E.emul_link_query(H, q.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p), l, 1, 1, 1, out.ctypes.data_as(C.c_void_p))
        calls[e, l] = cnt[2]; its[e, l] = cnt[3]
print('per link mean GJK calls', calls.mean(0).round(2)); print('per link mean GJK iterations', its.mean(0).round(2))
print('per link p99 iterations', np.percentile(its, 99, axis=0)); print('wave-level (max over 64 envs) iterations', its.reshape(B // 64, 64, 8).max(1).mean(0))
# wave cost of the lockstep scheme: sum over pieces of the max over the 64 lanes of a wave of that piece's iterations
slots = (C.c_long * 64).in_dll(E, 'emul_slot_counters')
per = np.zeros((B, 8, 64))
for e in range(B):
    q = st['q'][e].astype(np.float32); idx = st['planet_idx'][e].astype(np.int32)
    for l in range(3, 8):
        for i in range(64): slots[i] = 0
        out = np.zeros(3, np.float32)
        E.emul_link_query(H, q.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p), l, 0, 1, 1, out.ctypes.data_as(C.c_void_p))
        per[e, l] = np.array(slots[:])
w = per.reshape(B // 64, 64, 8, 64)
print('lockstep wave iterations per link (sum over pieces of max over lanes)', w.max(1).sum(-1).mean(0).round(1))
print('lane-autonomous wave iterations (max over lanes of sum over pieces) ', w.sum(-1).max(1).mean(0).round(1))
print('mean per lane', w.sum(-1).mean((0, 1)).round(1))
