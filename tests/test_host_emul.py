"""The DEVICE functions of safemotionsrisk_amd/csrc/smr_device.h compiled for the host (float32, same source,
tests/host_emul/emul.cpp) against the float64 oracle.  Runs without a GPU; it is how numerical problems of the
kernels' GJK / range code are reproduced and debugged on a CPU.  Not a product path: nothing outside tests/
loads libemul.so."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_emul", "emul.cpp")
LIB = os.path.join(HERE, "host_emul", "libemul.so")


@pytest.fixture(scope="module", params=["off", "fast"])
def emul(request, built):
    """Two builds: without and with FMA contraction (hipcc contracts by default)."""
    lib = LIB.replace(".so", "_%s.so" % request.param)
    flags = ["-ffp-contract=off"] if request.param == "off" else ["-ffp-contract=fast", "-mfma"]
    csrc = os.path.join(HERE, "..", "safemotionsrisk_amd", "csrc")
    deps = [SRC, os.path.join(csrc, "smr_device.h"), os.path.join(csrc, "smr_tables.h")]
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", lib, SRC] + flags)
    E = C.CDLL(lib)
    E.emul_scene_create.restype = C.c_void_p
    return E


def test_device_distances_on_host(emul, space_scene):
    s = space_scene.struct
    ora = O.OracleEnv(space_scene, 1)
    H = C.c_void_p(emul.emul_scene_create(C.cast(space_scene.pointer(), C.c_void_p)))
    rng = np.random.default_rng(0)
    clamp = lambda x: 0.0 if x < 0.001 else x
    worst = 0.0
    for _ in range(1500):
        q = rng.uniform(np.array(s.pos_lo), np.array(s.pos_hi)).astype(np.float32)
        i0 = int(rng.integers(1200))
        idx = np.array([i0, (i0 - 480) % 1200], dtype=np.int32)
        out = np.zeros(4, np.float32)
        emul.emul_distances(H, q.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p), 1, 0, out.ctypes.data_as(C.c_void_p))
        d = ora.distances(q.astype(np.float64), idx)
        if min(abs(d[0] - 0.001), abs(d[2] - 0.001)) < 2e-5:
            continue
        worst = max(worst, abs(clamp(out[0]) - clamp(d[0])), abs(clamp(out[1]) - clamp(d[2])))
        c = np.zeros(4, np.float32)
        emul.emul_distances(H, q.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p), 0, 1, c.ctypes.data_as(C.c_void_p))
        if abs(d[2] - 0.02) > 2e-5:
            assert c[2] == d[3]
    assert worst < 2e-6


def test_lane_autonomous_planet_queries_equal_lockstep(emul, space_scene):
    """planet_queries_auto (every lane takes its own candidate pieces best first, keys from capsule bounds, GJK in the
    link frame) against the wave-lockstep loop of collide_link and against the oracle: per-link moving distance and the
    contact flag, for the end-of-step sample (distance + contact) and for a contact-only sample."""
    s = space_scene.struct
    ora = O.OracleEnv(space_scene, 1)
    H = C.c_void_p(emul.emul_scene_create(C.cast(space_scene.pointer(), C.c_void_p)))
    rng = np.random.default_rng(3)
    worst = 0.0
    hits = 0
    for _ in range(400):
        q = rng.uniform(np.array(s.pos_lo), np.array(s.pos_hi)).astype(np.float32)
        i0 = int(rng.integers(1200))
        idx = np.array([i0, (i0 - 480) % 1200], dtype=np.int32)
        d = ora.distances(q.astype(np.float64), idx)
        per_link = []
        for l in range(8):
            lock, auto, con_l, con_a, dist_a = (np.zeros(3, np.float32) for _ in range(5))
            args = (H, q.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p), l, 0)
            emul.emul_link_query(*args, 1, 1, lock.ctypes.data_as(C.c_void_p))
            emul.emul_link_query(*args, 3, 3, auto.ctypes.data_as(C.c_void_p))
            emul.emul_link_query(*args, 0, 1, con_l.ctypes.data_as(C.c_void_p))
            emul.emul_link_query(*args, 0, 3, con_a.ctypes.data_as(C.c_void_p))
            emul.emul_link_query(*args, 3, 0, dist_a.ctypes.data_as(C.c_void_p))      # distance only: refined below 0.02 too
            near = lambda x: abs(x - 0.02) < 2e-5 or abs(x - 0.001) < 2e-5
            if not near(lock[1]):
                assert abs(min(lock[1], 0.6) - min(auto[1], 0.6)) < 1e-6 or (lock[1] < 0.001 and auto[1] < 0.001)
                assert lock[2] == auto[2] and con_l[2] == con_a[2]
            per_link.append(dist_a[1])
            hits += int(auto[2])
        if d[2] < 0.6 and abs(d[2] - 0.001) > 2e-5:
            got = min(per_link)
            worst = max(worst, abs((0.0 if got < 0.001 else got) - (0.0 if d[2] < 0.001 else d[2])))
    assert worst < 2e-6 and hits > 10


def test_links_dropped_from_the_contact_queries_never_reach_a_planet(emul, space_scene, ball_scene):
    """reachable_contact_mask (smr_tables.h) drops iiwa_link_0..2 in the space scene - the shoulder never gets within the
    contact distance of a planet - and nothing in a ball scene; brute force over random poses and orbit phases agrees."""
    H = C.c_void_p(emul.emul_scene_create(C.cast(space_scene.pointer(), C.c_void_p)))
    mask = emul.emul_contact_mask(H)
    assert mask == 0b11111000
    assert emul.emul_static_mask(H) == 0b11111000 and space_scene.struct.static_link_mask == 0b11111100    # iiwa_link_2 stays 0.16 m up
    Hb = C.c_void_p(emul.emul_scene_create(C.cast(ball_scene.pointer(), C.c_void_p)))
    assert emul.emul_contact_mask(Hb) == ball_scene.struct.contact_link_mask
    s = space_scene.struct
    ora = O.OracleEnv(space_scene, 1)
    rng = np.random.default_rng(5)
    closest = np.inf
    for _ in range(25):
        q = rng.uniform(np.array(s.pos_lo), np.array(s.pos_hi))
        i0 = int(rng.integers(1200))
        for link in range(3):
            for p, n in ((0, s.planet[0].piece_count), (1, s.planet[1].piece_count)):
                for c in range(n):
                    closest = min(closest, ora.link_planet_piece_distance(q, link, p, c, i0 if p == 0 else (i0 - 480) % 1200))
    assert closest > 0.05


def test_device_distances_with_ball_machine_on_host(emul, bm_scene):
    """robot_ball_machine.urdf: the two convex pieces mounted on iiwa_link_7 take part in the table / planet queries and
    pair with the base adapter's cylinder and iiwa_link_0..5 for the self-collision distance (A7 self)."""
    s = bm_scene.struct
    assert s.n_bodies == 10 and s.n_self_pairs == 14
    ora = O.OracleEnv(bm_scene, 1)
    H = C.c_void_p(emul.emul_scene_create(C.cast(bm_scene.pointer(), C.c_void_p)))
    assert H.value
    rng = np.random.default_rng(3)
    clamp = lambda x: 0.0 if x < 0.001 else x
    worst, self_hits, self_seen = 0.0, 0, 0
    for _ in range(1200):
        q = rng.uniform(np.array(s.pos_lo), np.array(s.pos_hi)).astype(np.float32)
        i0 = int(rng.integers(1200))
        idx = np.array([i0, (i0 - 480) % 1200], dtype=np.int32)
        out = np.zeros(4, np.float32)
        emul.emul_distances(H, q.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p), 1, 0, out.ctypes.data_as(C.c_void_p))
        d = ora.distances(q.astype(np.float64), idx)                      # static, self, moving, contact
        if min(abs(d[0] - 0.001), abs(d[1] - 0.001), abs(d[2] - 0.001)) < 2e-5:
            continue
        worst = max(worst, abs(clamp(out[0]) - clamp(d[0])), abs(clamp(out[3]) - clamp(d[1])), abs(clamp(out[1]) - clamp(d[2])))
        self_seen += d[1] < 0.1
        self_hits += d[1] < 0.001
    assert worst < 3e-6 and self_seen > 100 and self_hits > 10


def test_hill_climbing_support_equals_scan(emul, space_scene, ball_scene):
    """Climbing along the hull's edge graph (the link-hull support of the kernels) reaches the same maximum as the
    exhaustive scan, for every piece, from the cube-map start and from arbitrary warm starts."""
    rng = np.random.default_rng(5)
    emul.emul_support_check.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    for scene in (space_scene, ball_scene):
        H = C.c_void_p(emul.emul_scene_create(C.cast(scene.pointer(), C.c_void_p)))
        assert H.value
        for pi in range(scene.struct.n_pieces):
            n = emul.emul_piece_count(H, pi)
            dirs = rng.normal(size=(60, 3)).astype(np.float32)
            dirs[:6] = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float32)
            dirs[6] *= 1e-6                               # tiny and huge directions
            dirs[7] *= 1e6
            for d in dirs:
                for warm in (-1, int(rng.integers(n))):
                    out = np.zeros(3, np.float32)
                    emul.emul_support_check(H, pi, d.ctypes.data_as(C.c_void_p), warm, out.ctypes.data_as(C.c_void_p))
                    scale = float(np.linalg.norm(d))
                    assert out[0] >= out[1] - 2e-7 * scale, (pi, d, warm, out)


def test_gjk_continues_through_a_rounding_level_stall(emul, ball_scene):
    """Regression (found by the ball rollout on the GPU): the triangle built with a far new vertex improved |v|^2 by
    4e-9 - less than float32 resolves - and the "no progress" exit returned a distance 1.5e-5 m too large."""
    q = np.array([-2.9045886993408203, -0.8068791031837463, -2.3876144886016846, -1.90622878074646, 2.388410806655884,
                  0.18951769173145294, 2.6782050132751465], np.float32)
    ball = np.array([-1.1466519832611084, 2.185420513153076, 0.10109367221593857, 1.195295810699463, -1.5054339170455933,
                     5.683743000030518, 0.25892138481140137, 0.20055784285068512, 0.026196885854005814, 6.032756805419922], np.float32)
    k = 240
    H = C.c_void_p(emul.emul_scene_create(C.cast(ball_scene.pointer(), C.c_void_p)))
    ora = O.OracleEnv(ball_scene, 1)
    want = ora.ball_distances(q.astype(np.float64), ball.astype(np.float64), k)[2]
    for use_scan in (0, 1):
        out = np.zeros(8, np.float32)
        emul.emul_ball_link_distances(H, q.ctypes.data_as(C.c_void_p), ball.ctypes.data_as(C.c_void_p), k,
                                      out.ctypes.data_as(C.c_void_p), use_scan)
        assert abs(out.min() - want) < 1e-6


def test_device_acc_range_on_host(emul, space_scene):
    s = space_scene.struct
    rng = np.random.default_rng(1)
    emul.emul_safe_acc_range.argtypes = [C.c_double] * 6 + [C.c_float] * 3 + [C.c_void_p] * 3
    worst = 0.0
    for _ in range(3000):
        j = int(rng.integers(7))
        L = (s.pos_lo[j], s.pos_hi[j], s.vel_max[j], s.acc_max[j], s.jerk_max[j] * float(rng.choice([1.0, 0.5, 0.2])), 0.1)
        p = np.float32(rng.uniform(L[0], L[1])); v = np.float32(rng.uniform(-L[2], L[2])); a = np.float32(rng.uniform(-L[3], L[3]))
        lo, hi, code = C.c_float(), C.c_float(), C.c_int()
        emul.emul_safe_acc_range(*[float(x) for x in L], float(p), float(v), float(a), C.byref(lo), C.byref(hi), C.byref(code))
        olo, ohi, ocode = O.safe_acc_range_joint(*[float(x) for x in L], float(p), float(v), float(a))
        assert code.value == ocode
        worst = max(worst, abs(lo.value - olo) / L[3], abs(hi.value - ohi) / L[3])
    assert worst < 1e-5


def test_reset_kernels_lds_pool_is_a_copy_of_the_global_tables(emul, space_scene, bm_scene):
    """body_pool_tables (smr_tables.h): the image the reset-side kernels stage in LDS holds every body's vertices, id groups, start map
    and piece record at the offsets of its header - climbs over it return the vertices the climbs over the global tables return -
    and fits the kernels' static LDS array, with and without the ball machine."""
    for scene in (space_scene, bm_scene):
        H = C.c_void_p(emul.emul_scene_create(C.cast(scene.pointer(), C.c_void_p)))
        mism = C.c_int(-1)
        size = emul.emul_body_pool_check(H, C.cast(scene.pointer(), C.c_void_p), 200, C.byref(mism))
        assert 0 < size <= 15360 and mism.value == 0, (size, mism.value)


def test_state_code_alone_equals_the_code_of_the_range(emul, space_scene):
    """safe_state_code (kinematic start-state sampling: only "is this (q, v, a) feasible") returns the violation code of
    safe_acc_range_joint without finding the range ends - over random states, including the ones next to a position limit where
    the brackets open, both must agree bit for bit."""
    s = space_scene.struct
    rng = np.random.default_rng(5)
    emul.emul_safe_acc_range.argtypes = [C.c_double] * 6 + [C.c_float] * 3 + [C.c_void_p] * 3
    emul.emul_state_code.argtypes = [C.c_double] * 6 + [C.c_float] * 3
    seen = set()
    for i in range(6000):
        j = int(rng.integers(7))
        L = (s.pos_lo[j], s.pos_hi[j], s.vel_max[j], s.acc_max[j], s.jerk_max[j] * float(rng.choice([1.0, 0.5, 0.2])), 0.1)
        if i % 3 == 0:      # next to a limit, moving towards it: position brackets open / infeasible states
            side = rng.integers(2)
            p = np.float32(L[side] + (1 - 2 * side) * rng.uniform(0, 0.08) * (L[1] - L[0]))
            v = np.float32((2 * side - 1) * rng.uniform(0, 1) * L[2])
        else:
            p = np.float32(rng.uniform(L[0], L[1])); v = np.float32(rng.uniform(-L[2], L[2]))
        a = np.float32(rng.uniform(-L[3], L[3]))
        lo, hi, code = C.c_float(), C.c_float(), C.c_int()
        emul.emul_safe_acc_range(*[float(x) for x in L], float(p), float(v), float(a), C.byref(lo), C.byref(hi), C.byref(code))
        assert emul.emul_state_code(*[float(x) for x in L], float(p), float(v), float(a)) == code.value
        seen.add(code.value)
    assert {0, 2}.issubset(seen) and len(seen) >= 3


def _cubic(rng, s, n):
    """random start states and end accelerations inside the limits (the sweep only needs kinematically plausible cubics)"""
    lo, hi, V, A = (np.array(x) for x in (s.pos_lo, s.pos_hi, s.vel_max, s.acc_max))
    q = rng.uniform(lo, hi, (n, 7)).astype(np.float32)
    v = (rng.uniform(-1, 1, (n, 7)) * V).astype(np.float32)
    a = (rng.uniform(-1, 1, (n, 7)) * A).astype(np.float32)
    a1 = (rng.uniform(-1, 1, (n, 7)) * A).astype(np.float32)
    return q, v, a, a1


def _fp(x):
    return x.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("N", [24, 3])
def test_contact_sweep_equals_testing_every_sample(emul, space_scene, N):
    """The end-of-step lane of a body (collide_lane) rules out the intermediate contact samples N-1, N-2, ... that its lower bound
    of the planet distance at t = dt covers (motion bounds of the body and of the planets) and asks for the rest as work items:
    items + end-of-step flag must equal testing every sample with the same device functions - no tolerance - and the float64
    oracle; and most lanes need no item.  Half of the cases are drawn near a planet (oracle distance below 0.15 m in mid-step)
    so that contacts occur."""
    s = space_scene.struct
    ora = O.OracleEnv(space_scene, 1)
    H = C.c_void_p(emul.emul_scene_create(C.cast(space_scene.pointer(), C.c_void_p)))
    emul.emul_sweep_R.restype = C.c_float
    emul.emul_planet_step.restype = C.c_float
    assert 0.002 < emul.emul_planet_step(H, 0) < 0.01 and 0.002 < emul.emul_planet_step(H, 1) < 0.01     # ~ 0.9 m/s / 240 Hz + rotation
    p7 = s.pieces[s.link_piece[7]]
    rho7 = np.linalg.norm(space_scene.verts[p7.vert_begin:p7.vert_begin + p7.vert_count], axis=1).max() + p7.margin
    assert abs(emul.emul_sweep_R(H, 7, 6) - rho7) < 1e-4 and emul.emul_sweep_R(H, 7, 0) > 1.0     # link 7 about joint 1: the whole arm
    rng = np.random.default_rng(11 + N)
    dt, upd = float(s.dt), 24 // N
    cubic = lambda q, v, a, a1, t: q.astype(np.float64) + v * t + 0.5 * a * t * t + (a1.astype(np.float64) - a) / dt * t ** 3 / 6
    hits = items = lanes = lanes_far = items_far = oracle_checked = 0
    for i in range(240):
        while True:
            (q,), (v,), (a,), (a1,) = _cubic(rng, s, 1)
            v *= 0.5                                          # typical rollouts stay well below the velocity limits
            i0 = int(rng.integers(1200))
            idx0 = np.array([i0, (i0 - 480) % 1200], dtype=np.int32)
            if i % 2 or ora.distances(cubic(q, v, a, a1, 0.5 * dt), (idx0 + 12) % 1200)[2] < 0.15:
                break
        any_sweep = False
        # the step that follows (same planets 24 updates on, a new end acceleration): its lanes also get the bound this step's lanes
        # leave behind for t = dt, as lb_start, and rule out their first samples with it
        f32 = np.float32
        q2 = (q + v * f32(dt) + f32(0.5) * a * f32(dt) * f32(dt) + f32(1 / 6) * ((a1 - a) / f32(dt)) * f32(dt) ** 3).astype(np.float32)
        v2 = (v + a * f32(dt) + f32(0.5) * ((a1 - a) / f32(dt)) * f32(dt) ** 2).astype(np.float32)
        a2 = (rng.uniform(-1, 1, 7) * np.array(s.acc_max)).astype(np.float32)
        idx2 = ((idx0 + 24) % 1200).astype(np.int32)
        for l in range(3, 8):
            out, lit, nxt, nxt_lit, nxt_plain = (np.zeros(6, np.float32) for _ in range(5))
            emul.emul_collide_lane(H, l, _fp(q), _fp(v), _fp(a), _fp(a1), _fp(idx0), N, 0, None, None, C.c_float(-1.0), _fp(out))
            emul.emul_collide_lane(H, l, _fp(q), _fp(v), _fp(a), _fp(a1), _fp(idx0), N, 1, None, None, C.c_float(-1.0), _fp(lit))
            # a contact at the end-of-step sample latches the hit anyway: no items are asked for then
            assert out[0] or out[4] == lit[4], (i, l)
            any_sweep |= bool(lit[4])
            if not (out[0] or out[4]):                            # no hit latched: the env steps on
                emul.emul_collide_lane(H, l, _fp(q2), _fp(v2), _fp(a1), _fp(a2), _fp(idx2), N, 0, None, None, C.c_float(out[5]), _fp(nxt))
                emul.emul_collide_lane(H, l, _fp(q2), _fp(v2), _fp(a1), _fp(a2), _fp(idx2), N, 0, None, None, C.c_float(-1.0), _fp(nxt_plain))
                emul.emul_collide_lane(H, l, _fp(q2), _fp(v2), _fp(a1), _fp(a2), _fp(idx2), N, 1, None, None, C.c_float(-1.0), _fp(nxt_lit))
                assert nxt[0] or nxt[4] == nxt_lit[4], (i, l)
                assert nxt[3] <= nxt_plain[3]
                items += nxt[3]; lanes += 1
                if i % 2:
                    items_far += nxt[3]; lanes_far += 1
        hits += any_sweep
        if i % 4 == 0:                                            # the oracle on every sample (float64, all links at once)
            flag, tie = False, False
            for k in range(1, N):
                d = ora.distances(cubic(q, v, a, a1, dt * k / N), (idx0 + k * upd) % 1200)
                flag |= bool(d[3]); tie |= abs(d[2] - 0.02) < 2e-5
            if not tie:
                assert flag == any_sweep, i
                oracle_checked += 1
    print("contact items N=%d: %.2f per (env, body) lane; %.2f for uniformly drawn states; %d envs with a contact" % (N, items / lanes, items_far / lanes_far, hits))
    assert hits > 15 and oracle_checked > 40
    assert items_far / lanes_far < 0.2 * (N - 1), items_far / lanes_far
