"""How much work does the safe-acceleration-range root finding do on states of a random-action rollout? (CPU, host emulation)"""
import sys, ctypes as C, subprocess, os
sys.path.insert(0, '.')
import numpy as np
from oracle import oracle as O
from safemotionsrisk_amd.scene import build_scene, SPACE_ENV_CONFIG
lib = '/tmp/libemul_cnt.so'
subprocess.check_call(['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-ffp-contract=fast', '-mfma', '-o', lib, 'tests/host_emul/emul.cpp'])
E = C.CDLL(lib)
E.emul_safe_acc_range.argtypes = [C.c_double] * 6 + [C.c_float] * 3 + [C.c_void_p] * 3
cnt = (C.c_long * 4).in_dll(E, 'emul_counters')
sc = build_scene(SPACE_ENV_CONFIG, auto_reset=True)
s = sc.struct
B = 256
ora = O.OracleEnv(sc, B, seed=0)
ora.reset()
rng = np.random.default_rng(0)
evals, brakes = [], []
for step in range(30):
    st = ora.get_state()
    for e in range(B):
        for j in range(7):
            L = (s.pos_lo[j], s.pos_hi[j], s.vel_max[j], s.acc_max[j], s.jerk_max[j], s.dt)
            cnt[0] = cnt[1] = 0
            lo, hi, code = C.c_float(), C.c_float(), C.c_int()
            E.emul_safe_acc_range(*[float(x) for x in L], float(st['q'][e, j]), float(st['v'][e, j]), float(st['a'][e, j]), C.byref(lo), C.byref(hi), C.byref(code))
            evals.append(cnt[0]); brakes.append(cnt[1])
    ora.step(rng.uniform(-1, 1, (B, 7)))
evals = np.array(evals); brakes = np.array(brakes)
print('calls', len(evals), 'margin evals mean', evals.mean(), 'pcts', np.percentile(evals, [50, 90, 99, 100]))
print('brake knots per call mean', brakes.mean(), 'pcts', np.percentile(brakes, [50, 90, 99, 100]), 'knots per eval', brakes.sum() / evals.sum())
print('frac calls with >2 evals', (evals > 2).mean())
# wave-level: max over 64 consecutive calls (same joint ordering as plan kernel: lanes = envs of one joint)
ev = evals.reshape(30, B, 7)
w = ev.transpose(0, 2, 1).reshape(30, 7, B // 64, 64).max(-1)
print('wave max evals mean', w.mean())
