"""Shared helpers of the parity tests: which done / reason mismatches between the float32 kernels and the float64 oracle are
explained by a distance sitting on a decision threshold.

Decisions of a step: d_static / d_self / d_moving < 0.001 m at the end of the step (rewards.py:115-121,153-155) and the contact
latch `distance <= contact_distance` (0.02 m) at each of the N contact samples of the step (ctlp.py:2631-2637,2849-2861).  An env
is a TIE when the oracle's float64 distance of one of these decisions lies within TIE of its threshold: float32 may then fall on
the other side.  Anything else that differs is a finding, not a budget: the tests assert zero unexplained mismatches.
"""
import numpy as np

TIE = 2e-5


def self_cut(scene):
    """The range of self-collision distances the reference reads: the reward's min(1, d / max_reward_distance) and the flag
    d < 0.001 (rewards.py:119-126).  The product reports the exact distance below it and the cut-off value ("no closest point")
    at or beyond it (include/safemotions_hip.h, SMR_INFO_D_SELF); the oracle makes the literal queries up to the cut-off."""
    s = scene.struct
    return min(float(s.static_max_distance), max(float(s.self_max_reward_distance) if s.self_max_reward != 0 else 0.0, float(s.collision_distance)))


def relevant(d, scene):
    """distance columns (d_static, d_self, d_moving, ...) with the self distance clipped to the range the reference reads"""
    d = np.array(d, dtype=np.float64, copy=True)
    d[..., 1] = np.minimum(d[..., 1], self_cut(scene))
    return d


def cubic(q0, v0, a0, a1, t, dt):
    q0, v0, a0, a1 = (np.asarray(x, dtype=np.float64) for x in (q0, v0, a0, a1))
    return q0 + v0 * t + 0.5 * a0 * t * t + (a1 - a0) / dt * t ** 3 / 6


def ball_retire_update(k, kmax, khit, updates_per_step=24):
    """first update u in 1..24 at which the ball is retired (ctlp.py:2836-2848), 25 if it survives the step"""
    u = max(1, min(kmax - k + 1, khit - k))
    return updates_per_step + 1 if u > updates_per_step else u


def explain_mismatches(ora, scene, st0, st1, bad, tie=TIE):
    """For the envs flagged in `bad` (done / reason / counters differ): True where a threshold tie explains it.
    st0 / st1: the product's get_state() before / after the step (st1["a"] is the end acceleration of the step's cubic).
    Uses the oracle's float64 distances at the end of the step and at every intermediate contact sample."""
    s = scene.struct
    N, dt, upd = int(s.collision_checks_per_step), float(s.dt), int(s.updates_per_step) // int(s.collision_checks_per_step)
    idx_bad = np.nonzero(bad)[0]
    out = np.zeros(len(idx_bad), dtype=bool)
    for n, e in enumerate(idx_bad):
        q0, v0, a0, a1 = st0["q"][e], st0["v"][e], st0["a"][e], st1["a"][e]
        samples = range(1, N + 1)
        for k in samples:
            qk = cubic(q0, v0, a0, a1, dt * k / N, dt)
            if s.use_ball:
                k0, kmax, khit, active = (int(x) for x in st0["ball_i"][e][:4])
                if not active:
                    break
                if k * upd >= ball_retire_update(k0, kmax, khit):
                    break
                d = ora.ball_distances(qk, st0["ball"][e].astype(np.float64), k0 + k * upd)
            else:
                n_orbit = [int(s.planet[p].n_orbit) if p < s.n_planets else 1 for p in range(2)]
                idx = np.array([(int(st0["planet_idx"][e][p]) + k * upd) % n_orbit[p] for p in range(2)], dtype=np.int32)
                d = ora.distances(qk, idx)
            if abs(d[2] - float(s.contact_distance)) < tie:
                out[n] = True
            if k == N and (np.abs(np.asarray(d[:3]) - float(s.collision_distance)) < tie).any():
                out[n] = True
        if not out[n]:                                  # static / self distances do not depend on the moving obstacle (a retired ball skips the loop end)
            qn = cubic(q0, v0, a0, a1, dt, dt)
            d = ora.ball_distances(qn, st0["ball"][e].astype(np.float64), 0) if s.use_ball else ora.distances(qn, np.zeros(2, np.int32))
            out[n] = bool((np.abs(np.asarray(d[:2]) - float(s.collision_distance)) < tie).any())
    return out


def braking_decision_is_tie(ora, scene, q, v, a, action, tie=TIE):
    """The braking-trajectory check of one env (ctlp.py:3155-3207) replayed with the oracle's component functions: True when one of
    its decisions sits on a threshold - a static / self distance of a sample within `tie` of the safety distance, a joint
    within 1e-6 of the stop band that ends the braking trajectory (btg.py:45), or a joint that rides ON a position limit at a knot
    of the braking trajectory (within 3e-4 rad: the clip of its braking acceleration into the safe range, btg.py:80-93, then hangs
    on a margin of micrometres - seen: 19.02 against 20.0 m/s^2 with the joint 1.0e-4 rad from its limit and a dip of 1.1e-4 rad
    ahead of it - and the two trajectories part by millimetres) - so that float32 may decide the other way."""
    s = scene.struct
    dt, Nb, safety = float(s.dt), int(s.brake_checks_per_step), float(s.brake_safety_distance)
    q, v, a = (np.asarray(x, dtype=np.float64) for x in (q, v, a))
    lo, hi, _ = ora.safe_acc_range(q, v, a)
    an = lo + 0.5 * (np.asarray(action, dtype=np.float64) + 1.0) * (hi - lo)
    p, vv, ap = q.copy(), v.copy(), a.copy()
    for m in range(int(s.brake_timeout_steps) + 1):
        for k in range(1, Nb + 1):
            d = ora.distances(cubic(p, vv, ap, an, dt if k == Nb else dt * k / Nb, dt), np.zeros(2, np.int32))
            if (np.abs(np.asarray(d[:2]) - safety) < tie).any():
                return True
            if min(d[0], d[1]) < safety:
                return False                                # a clear collision ends the check
        p1 = cubic(p, vv, ap, an, dt, dt)
        v1 = vv + 0.5 * (ap + an) * dt
        if (np.minimum(p1 - np.array(s.pos_lo), np.array(s.pos_hi) - p1) < 3e-4).any():
            return True
        if (np.abs(np.abs(v1) - float(s.brake_stop_velocity)) < 1e-6).any() or (np.abs(np.abs(an) - float(s.brake_stop_acceleration)) < 1e-6).any():
            return True
        b, stopped = ora.braking_acceleration(p1, v1, an)
        if stopped:
            return False
        p, vv, ap, an = p1, v1, an, b
    return False
