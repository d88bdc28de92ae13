"""Pins the oracle's kinematic pieces: A3 cubic identities and the A4 safe acceleration range
(invariants of Kiemel & Kroeger's scheme + closed forms derived in DESIGN.md).  CPU only."""
import numpy as np
import pytest

from oracle import oracle as O

LIMS = dict(plo=-2.93206, phi=2.93206, V=1.7104, A=15.0, J=300.0, dt=0.1)


def rng_state(rng, L, n):
    return (rng.uniform(L["plo"], L["phi"], n), rng.uniform(-L["V"], L["V"], n), rng.uniform(-L["A"], L["A"], n))


def step(p, v, a, a1, dt):
    return p + v * dt + (a / 3 + a1 / 6) * dt * dt, v + 0.5 * (a + a1) * dt, a1


def test_range_inactive_far_from_limits(built):
    lo, hi, code = O.safe_acc_range_joint(**LIMS, p=0.0, v=0.0, a=0.0)
    assert (lo, hi, code) == (-15.0, 15.0, 0)
    # jerk limit binds when jerk*dt < 2 A
    L = dict(LIMS, J=50.0)
    lo, hi, code = O.safe_acc_range_joint(**L, p=0.0, v=0.0, a=3.0)
    assert (lo, hi, code) == (pytest.approx(-2.0), pytest.approx(8.0), 0)


def test_velocity_bound_closed_form(built):
    """J dt = 2A: the brake phase after a1 > 0 is one segment to -A, so the peak velocity is
    v0 + (a0 + a1) dt/2 + a1^2 dt / (2 (a1 + A)) = V  ('safe acceleration range')."""
    L = LIMS
    for v0, a0 in [(1.6, 0.0), (1.5, 2.0), (1.0, 5.0), (1.69, -1.0)]:
        lo, hi, code = O.safe_acc_range_joint(**L, p=0.0, v=v0, a=a0)
        V = L["V"] * (1 + 1e-5)
        K = (V - v0 - a0 * L["dt"] / 2) * 2 / L["dt"]
        a1 = ((K - L["A"]) + np.sqrt((L["A"] - K) ** 2 + 8 * K * L["A"])) / 4
        if a1 > 0:
            assert hi == pytest.approx(min(a1, L["A"]), rel=1e-9)
        assert code == 0 and lo == -L["A"]


def test_position_bound_stops_exactly_at_limit(built):
    """Taking the upper end of the range and then always the lower end must touch, not cross, the limit."""
    L = LIMS
    p, v, a = 2.6, 1.2, 0.0
    lo, hi, code = O.safe_acc_range_joint(**L, p=p, v=v, a=a)
    assert code == 0 and hi < L["A"]
    p, v, a = step(p, v, a, hi, L["dt"])
    peak = p
    for _ in range(30):
        lo, hi, code = O.safe_acc_range_joint(**L, p=p, v=v, a=a)
        # fine sampling of the segment for the true maximum
        ts = np.linspace(0, L["dt"], 200)
        seg = p + v * ts + 0.5 * a * ts ** 2 + (lo - a) / L["dt"] * ts ** 3 / 6
        peak = max(peak, seg.max())
        p, v, a = step(p, v, a, lo, L["dt"])
        if v < 0:
            break
    tol = 1e-6 * (L["phi"] - L["plo"])
    assert peak <= L["phi"] + tol + 1e-9
    assert peak >= L["phi"] - 1e-4            # the bound is tight, not conservative


def test_velocity_closed_form_against_bruteforce(built):
    """The closed-form velocity bound reproduces the simulated peak of the braking profile exactly,
    for jerk limits from 2A/dt down to 5 % of it (several full-jerk knots)."""
    rng = np.random.default_rng(0)
    checked = 0
    for _ in range(4000):
        A, V, dt = rng.uniform(5, 20), rng.uniform(1, 3), 0.1
        J = 2 * A / dt * rng.choice([1.0, 0.7, 0.5, 0.3, 0.15, 0.05])
        v0, a0 = rng.uniform(-V, V), rng.uniform(-A, A)
        u = O.vel_upper_a1max(V, A, J, dt, v0, a0)
        D = J * dt
        blo, bhi = max(-A, a0 - D), min(A, a0 + D)
        if u < -1e200:
            assert O.vel_peak_bruteforce(V, A, J, dt, v0, a0, blo) > V - 1e-9
        elif blo <= u <= bhi:
            assert O.vel_peak_bruteforce(V, A, J, dt, v0, a0, u) == pytest.approx(V, abs=1e-9)
            checked += 1
        elif u > bhi:
            assert O.vel_peak_bruteforce(V, A, J, dt, v0, a0, bhi) <= V + 1e-9
        else:
            assert O.vel_peak_bruteforce(V, A, J, dt, v0, a0, blo) >= V - 1e-9
    assert checked > 300


@pytest.mark.parametrize("jerk_factor", [1.0, 0.5, 0.15])
def test_rollouts_never_violate_limits(built, space_scene, jerk_factor):
    """The reference's own detectors - |pos|,|vel|,|acc| rel > 1.001 (obs.py:379-403), jerk > 1.002
    (rewards.py:195-201) - must never fire, the range must never be empty and no violation code may be
    raised, for random, bang-bang, always-max, always-min, square-wave and biased action sequences on
    all seven joints' limits."""
    rng = np.random.default_rng(0)
    s = space_scene.struct
    for j in range(7):
        L = dict(plo=s.pos_lo[j], phi=s.pos_hi[j], V=s.vel_max[j], A=s.acc_max[j], J=s.jerk_max[j] * jerk_factor, dt=0.1)
        for trial in range(18):
            p, v, a = rng.uniform(L["plo"], L["phi"]) * 0.9, 0.0, 0.0
            mode = trial % 6
            for k in range(100):
                lo, hi, code = O.safe_acc_range_joint(**L, p=p, v=v, a=a)
                assert code == 0 and lo <= hi
                u = [rng.uniform(-1, 1), rng.choice([-1.0, 1.0]), 1.0, -1.0, 1.0 if (k // 7) % 2 else -1.0,
                     1.0 if rng.uniform() < 0.9 else -1.0][mode]
                a1 = lo + 0.5 * (u + 1) * (hi - lo)
                assert abs(a1 - a) / L["dt"] <= L["J"] * 1.002
                ts = np.linspace(0, L["dt"], 25)
                pp = p + v * ts + 0.5 * a * ts ** 2 + (a1 - a) / L["dt"] * ts ** 3 / 6
                vv = v + a * ts + 0.5 * (a1 - a) / L["dt"] * ts ** 2
                assert np.abs(-1 + 2 * (pp - L["plo"]) / (L["phi"] - L["plo"])).max() <= 1.00001
                assert np.abs(vv).max() / L["V"] <= 1.0001
                assert abs(a1) / L["A"] <= 1.0000001
                p, v, a = step(p, v, a, a1, L["dt"])


def test_violation_codes(built):
    L = LIMS
    # moving up at full speed right below the upper limit: nothing can stop it -> position violation
    assert O.safe_acc_range_joint(**L, p=2.93, v=1.7, a=0.0)[2] & 2
    # accelerating hard while already at the velocity limit -> velocity violation
    assert O.safe_acc_range_joint(**L, p=0.0, v=1.71, a=15.0)[2] & 1
    # mirrored
    assert O.safe_acc_range_joint(**L, p=-2.93, v=-1.7, a=0.0)[2] & 2
    assert O.safe_acc_range_joint(**L, p=0.0, v=-1.71, a=-15.0)[2] & 1
    # symmetric problem -> symmetric range
    lo, hi, _ = O.safe_acc_range_joint(**L, p=1.0, v=0.7, a=3.0)
    lo2, hi2, _ = O.safe_acc_range_joint(**L, p=-1.0, v=-0.7, a=-3.0)
    assert (lo, hi) == (pytest.approx(-hi2), pytest.approx(-lo2))


def test_philox_known_answer(built):
    """Random123 known-answer test for philox4x32-10 (counter 0, key 0)."""
    assert [O.philox_word(0, 0, 0, 0, d) for d in range(4)] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
