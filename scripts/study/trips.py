"""CPU study: GJK iterations per (env, link) lane of the collide kernel's planet loop under different cutoff policies
(current: own running minimum; ideal: the env's true minimum known in advance; first: the link with the smallest key bound first)."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from safemotionsrisk_amd.scene import build_scene, SPACE_ENV_CONFIG
HERE = os.path.dirname(os.path.abspath(__file__))
lib = os.path.join(HERE, "libtrips.so")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=fast", "-mfma", "-o", lib, os.path.join(HERE, "trips.cpp")])
E = C.CDLL(lib)
E.emul_scene_create.restype = C.c_void_p
bm = len(sys.argv) > 1 and sys.argv[1] == "bm"
sc = build_scene(dict(SPACE_ENV_CONFIG, ball_machine_mode=bm))
s = sc.struct
H = C.c_void_p(E.emul_scene_create(C.cast(sc.pointer(), C.c_void_p)))
n = 4096
rng = np.random.default_rng(0)
q = rng.uniform(np.array(s.pos_lo), np.array(s.pos_hi), (n, 7)).astype(np.float32)
i0 = rng.integers(1200, size=n)
idx = np.stack([i0, (i0 - 480) % 1200], axis=1).astype(np.int32)
links = np.array([5, 6, 4, 7, 3] + ([8, 9] if bm else []), dtype=np.int32)
L = len(links)
t0, ti, tf = (np.zeros((n, L), np.int32) for _ in range(3))
r0, ri = np.zeros((n, L), np.int32), np.zeros((n, L), np.int32)
dmin, lb = np.zeros(n, np.float32), np.zeros((n, L), np.float32)
p = lambda a: a.ctypes.data_as(C.c_void_p)
E.study_trips(H, n, p(q), p(idx), p(links), L, p(t0), p(ti), p(tf), p(dmin), p(r0), p(ri), p(lb))
def wave_stats(t, name):
    w = t.reshape(n // 64, 64, L)            # a wavefront = 64 consecutive envs of one link
    mx = w.max(axis=1)
    print("%-8s iterations per lane: mean %.2f  | per wavefront: mean of max %.1f, p95 of max %d, max %d | per link mean-of-max %s"
          % (name, t.mean(), mx.mean(), np.percentile(mx, 95), mx.max(), np.round(mx.mean(axis=0), 1)))
wave_stats(t0, "current"); wave_stats(ti, "ideal"); wave_stats(tf, "first")
print("runs per lane: current %.2f ideal %.2f; envs with a planet within 0.6 m: %.2f" % (r0.mean(), ri.mean(), (dmin < 0.6).mean()))
print("lanes with any iteration: current %.2f ideal %.2f first %.2f" % ((t0 > 0).mean(), (ti > 0).mean(), (tf > 0).mean()))
nj, itot, imax, ifirst = (np.zeros(n, np.int32) for _ in range(4))
dj = np.zeros(n, np.float32); hist = np.zeros(33, np.int32)
E.study_jobs(H, n, p(q), p(idx), p(links), L, p(nj), p(itot), p(imax), p(ifirst), p(dj), p(hist))
print("job form: jobs per env mean %.2f p95 %d max %d | iterations per env: first %.2f + jobs %.2f (current: %.2f) | per job mean %.2f, max %d"
      % (nj.mean(), np.percentile(nj, 95), nj.max(), ifirst.mean(), itot.mean(), t0.sum(axis=1).mean(), itot.sum() / max(nj.sum(), 1), imax.max()))
print("  distance agrees with the lane form: max |diff| %.2e" % np.abs(np.minimum(dj, 0.602) - np.minimum(dmin, 0.602)).max())
print("  iterations per run histogram:", hist.tolist())
# a wave of 64 consecutive jobs runs as long as its longest job
its = []
cl = lambda x: np.where(x < 0.001, 0.0, np.minimum(x, 0.602))
print("  after the 1 mm clamp: max |diff| %.2e" % np.abs(cl(dj) - cl(dmin)).max())
je, jb = np.zeros(7), np.zeros(7)
E.study_ub(H, n, p(q), p(idx), p(links), L, p(je), p(jb))
print("pairs per env with key bound <= the best pair's bound after k iterations (k = 0: exact):", np.round(je, 2))
print("   ... when every body only knows the bound of ITS best piece:", np.round(jb, 2))
hr = np.zeros((10, 16), np.int32)
E.study_ratio(H, n, p(q), p(idx), p(links), L, p(hr))
print("jobs by (key bound / cutoff): count, mean iterations, share with > 2 iterations")
for b in range(10):
    c = hr[b].sum()
    if c: print("  ratio %.1f-%.1f: %6d jobs, mean %.2f iterations, %.0f%% > 2, max %d" % (b / 10, (b + 1) / 10, c, (hr[b] * np.arange(16)).sum() / c, 100 * hr[b, 3:].sum() / c, np.nonzero(hr[b])[0].max()))
