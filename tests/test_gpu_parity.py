"""Parity tests proper: the HIP path (through the C-ABI) against the float64 oracle and the committed golden
vectors.  Run on the GPU box: python -m pytest tests -m gpu.

Tolerances (float32 kernels vs float64 oracle, one step from an injected float32 state):
  FK positions 2e-6 m; distances 5e-6 m; acceleration range 2e-5 of acc_max (also while a joint rides on a
  limit: the kernels carry the float64 residual of every limit and work on displacements relative to it);
  obs: position columns 2e-6, velocity 1e-5, acceleration 2e-5, planet columns 1e-6; reward 5e-5;
  done / termination reason / collision flags: exact, except envs whose oracle distance lies within
  2e-5 m of a decision threshold (0.001 m collision, 0.02 m contact), which are excluded and counted.
"""
import os

import numpy as np
import pytest

from parity_util import TIE, braking_decision_is_tie, explain_mismatches, relevant, self_cut

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "space_step.npz")
STATE_KEYS = ("q", "v", "a", "planet_idx", "step_count", "moving_hit", "reset_count")
OBS_TOL = np.concatenate([np.full(7, 2e-6), np.full(7, 1e-5), np.full(7, 2e-5), np.full(2, 1e-6)])


def _env(B, **kw):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from safemotionsrisk_amd import VecSafeMotionsEnv, SPACE_ENV_CONFIG
    kw.setdefault("auto_reset", False)
    return VecSafeMotionsEnv(B, device=0, **kw, **SPACE_ENV_CONFIG)


def test_native_library_is_loaded(built):
    from safemotionsrisk_amd import lib
    L = lib.load()
    maps = open("/proc/self/maps").read()
    assert "libsmr_hip.so" in maps and L.smr_abi_version() == lib.ABI_VERSION


def test_components_against_golden(built):
    g = np.load(GOLD)
    env = _env(256, seed=0)
    pos, rot = env.test_fk(g["comp_q"])
    assert np.abs(pos - g["comp_fk"]).max() < 2e-6
    d = env.test_distances(g["comp_q"], g["comp_idx"])
    gd = g["comp_dist"]
    # below the 1 mm collision threshold only "d < 0.001" is defined (rewards.py:115-121,153-155): the kernel stops
    # refining a distance as soon as the collision is certain, so values are compared after that clamp
    clamp = lambda x: np.where(x < 0.001, 0.0, x)
    near = (np.abs(gd[:, 0] - 0.001) < TIE) | (np.abs(gd[:, 2] - 0.001) < TIE)
    assert np.abs(clamp(d[:, 0]) - clamp(gd[:, 0]))[~near].max() < 5e-6     # static (table)
    assert np.abs(clamp(d[:, 2]) - clamp(gd[:, 2]))[~near].max() < 5e-6     # moving (planets)
    safe = np.abs(gd[:, 2] - 0.02) > TIE
    assert np.array_equal(d[safe, 3], gd[safe, 3])                      # contact flags
    lo, hi, code = env.test_safe_acc_range(g["comp_p"], g["comp_v"], g["comp_a"])
    A = np.array(env.scene.struct.acc_max)
    assert np.abs((lo - g["comp_lo"]) / A).max() < 2e-5 and np.abs((hi - g["comp_hi"]) / A).max() < 2e-5
    assert np.array_equal(code, g["comp_code"])
    env.close()


@pytest.mark.parametrize("name", ["space_step.npz", "space_step_n3.npz", "space_step_n1.npz", "space_bm_step.npz", "reach_step.npz",
                                  "reach_brake_step.npz"])
def test_step_against_golden(built, name):
    """space_step.npz: 24 contact samples per step (the reference's cadence = the default); _n3 / _n1: the opt-in variants;
    reach_brake: the reaching task with the braking-trajectory safety layer (executed in ~12 % of the fixture's env-steps)"""
    import torch
    g = np.load(os.path.join(os.path.dirname(GOLD), name))
    B, T = int(g["B"]), int(g["T"])
    reach, brake = "reach" in name, "brake" in name
    env = (_brake_env if brake else _reach_env if reach else _bm_env if "bm" in name else _env)(B, seed=int(g["seed"]), collision_checks_per_step=int(g["checks"]))
    assert int(env.scene.struct.collision_checks_per_step) == int(g["checks"]) == (int(name.split("_n")[1].split(".")[0]) if "_n" in name else 24)
    env.set_state(q=g["q0"], v=g["v0"], a=g["a0"], planet_idx=g["planet_idx0"], step_count=g["step_count0"],
                  moving_hit=g["moving_hit0"], **(dict(target=g["target0"], target_count=g["target_count0"]) if reach else {}))
    from oracle.oracle import OracleEnv
    ora = OracleEnv(env.scene, 1)
    excluded = 0
    for t in range(T):
        st0 = env.get_state()
        obs, rew, done, info = env.step(torch.from_numpy(g["actions"][t]).cuda())
        obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        reason = info["termination_reason"].cpu().numpy()
        vals = info["values"].cpu().numpy()
        st = env.get_state()
        bad = (done != g["done"][t]) | (reason != g["reason"][t])
        if bad.any():                               # every mismatch must be a distance on a decision threshold (parity_util)
            why = explain_mismatches(ora, env.scene, st0, st, bad)
            assert why.all(), "%d unexplained done / reason mismatches at step %d" % (int((~why).sum()), t)
            excluded += int(bad.sum())
        if brake:                                   # the braking decision itself: exact, threshold ties aside (they change the whole step)
            diff = vals[:, 8] != g["info"][t][:, 8]
            for e in np.nonzero(diff)[0]:
                assert braking_decision_is_tie(ora, env.scene, st0["q"][e], st0["v"][e], st0["a"][e], g["actions"][t][e]), (t, e)
            excluded += int(diff.sum())
            bad = bad | diff
            assert g["info"][t][:, 8].sum() > 20
        ok = ~bad
        tol = np.broadcast_to(REACH_OBS_TOL if reach else OBS_TOL, obs.shape)
        if reach:                                   # a freshly drawn target comes from float32 FK of a random pose
            prev = g["target_count"][t - 1] if t else g["target_count0"]
            tol = np.where((g["target_count"][t] != prev)[:, None], 5e-5, tol)
        assert (np.abs(obs[ok] - g["obs"][t][ok]) < tol[ok]).all()
        assert np.abs(rew[ok] - g["reward"][t][ok]).max() < (2e-4 if reach else 5e-5)
        if reach:
            assert np.array_equal(st["target_count"][ok], g["target_count"][t][ok]) and np.abs(st["target"] - g["target"][t])[ok].max() < 2e-5
        gi = g["info"][t]
        near = (np.abs(gi[:, :3] - 0.001) < TIE).any(axis=1)      # below 1 mm only "d < 0.001" is defined: the clamp to 0 may differ on a tie
        assert np.abs(relevant(vals[:, :3], env.scene)[ok & ~near] - relevant(gi[:, :3], env.scene)[ok & ~near]).max() < 5e-6
        gs = g["state"][t]
        A = np.array(env.scene.struct.acc_max)
        assert np.abs(st["q"] - gs[:, :7])[ok].max() < 2e-6 and np.abs(st["v"] - gs[:, 7:14])[ok].max() < 2e-5
        assert np.abs((st["a"] - gs[:, 14:]) / A)[ok].max() < 2e-5
        # golden protocol: continue from the float32 image of the ORACLE state
        env.set_state(q=gs[:, :7].astype(np.float32), v=gs[:, 7:14].astype(np.float32), a=gs[:, 14:].astype(np.float32),
                      step_count=np.where(g["done"][t], 0, st["step_count"]), moving_hit=np.where(g["done"][t], 0, st["moving_hit"]),
                      **(dict(target=g["target"][t].astype(np.float32), target_count=g["target_count"][t]) if reach else {}))
    assert excluded <= 0.01 * B * T
    env.close()


@pytest.mark.parametrize("checks,lockstep", [(24, False), (3, False), (1, False), (24, True)])
def test_rollout_against_oracle_with_state_injection(built, checks, lockstep):
    """4096 envs x 12 steps of uniform random actions; after every step the oracle is restarted from the GPU's
    float32 state, so each step is a one-step parity check on states the GPU path itself produced.  24 contact samples per
    step (the reference's cadence, the default: the kernels sweep them by conservative advancement, the oracle tests every one),
    3 and 1; lockstep: SMR_LOCKSTEP_PLANETS=1, the wave-lockstep planet queries (also the fallback for planets beyond the LDS
    budget of the lane-autonomous ones)."""
    import torch
    from oracle.oracle import OracleEnv
    B, T = 4096, 12
    if lockstep:
        os.environ["SMR_LOCKSTEP_PLANETS"] = "1"
    try:
        env = _env(B, seed=5, collision_checks_per_step=checks)
    finally:
        os.environ.pop("SMR_LOCKSTEP_PLANETS", None)
    assert int(env.scene.struct.collision_checks_per_step) == checks
    env.reset()
    ora = OracleEnv(env.scene, B, seed=5, num_threads=16)
    rng = np.random.default_rng(9)
    mism, excluded, dones, hits = 0, 0, 0, 0
    for t in range(T):
        st = env.get_state()
        ora.set_state(**{k: st[k] for k in STATE_KEYS})
        act = rng.uniform(-1, 1, (B, 7)).astype(np.float32)
        obs, rew, done, info = env.step(torch.from_numpy(act).cuda())
        o_obs, o_rew, o_done, o_reason, o_info = ora.step(act.astype(np.float64))
        obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        reason = info["termination_reason"].cpu().numpy()
        bad = (done != o_done) | (reason != o_reason)
        if bad.any():
            why = explain_mismatches(ora, env.scene, st, env.get_state(), bad)
            excluded += int(why.sum())
            mism += int((~why).sum())
        good = ~bad
        assert (np.abs(obs[good] - o_obs[good]) < OBS_TOL).all()
        assert np.abs(rew[good] - o_rew[good]).max() < 5e-5
        dones += int(o_done.sum()); hits += int((o_reason == 5).sum())
        # envs that finished are re-sampled on the GPU so the rollout keeps all lanes busy
        if done.any():
            env.reset(mask=torch.from_numpy(done).cuda())
    assert mism == 0, "%d unexplained done/reason mismatches" % mism
    assert excluded <= 8 and dones > 0.2 * B and hits > 100
    env.close()


def test_reset_properties_and_oracle_agreement(built):
    from oracle.oracle import OracleEnv
    B = 8192
    env = _env(B, seed=21)
    obs = env.reset().cpu().numpy()
    st = env.get_state()
    s = env.scene.struct
    lo, hi = np.array(s.pos_lo), np.array(s.pos_hi)
    assert np.abs(obs).max() <= 1.0 and obs.shape == (B, 23)
    assert (st["q"] >= lo - 1e-6).all() and (st["q"] <= hi + 1e-6).all()
    assert (np.abs(st["v"]) <= np.array(s.vel_max) * 1.0001).all() and (np.abs(st["a"]) <= np.array(s.acc_max) * 1.0001).all()
    assert (st["step_count"] == 0).all() and ((st["planet_idx"][:, 1] - st["planet_idx"][:, 0]) % 1200 == 720).all()
    ora = OracleEnv(env.scene, 1)
    for e in range(0, B, 64):                               # oracle-checked properties on a sample
        q = st["q"][e].astype(np.float64)
        assert ora.position_is_valid(q) or abs(ora.distances(q, st["planet_idx"][e])[0] - 0.001) < TIE
        lo_a, hi_a, code = ora.safe_acc_range(q, st["v"][e].astype(np.float64), st["a"][e].astype(np.float64))
        assert (code == 0).all()
        d = ora.distances(q, st["planet_idx"][e])
        assert d[3] == st["moving_hit"][e] or abs(d[2] - 0.02) < TIE
    # same seed -> bit-identical states; the float64 oracle with the same Philox streams agrees on almost all envs
    env2 = _env(B, seed=21)
    env2.reset()
    st2 = env2.get_state()
    assert all(np.array_equal(st[k], st2[k]) for k in STATE_KEYS)
    ora_b = OracleEnv(env.scene, B, seed=21, num_threads=16)
    ora_b.reset()
    ost = ora_b.get_state()
    same = (ost["planet_idx"] == st["planet_idx"]).all(axis=1) & (np.abs(ost["q"] - st["q"]).max(axis=1) < 1e-3)
    assert (ost["reset_count"] == 1).all() and (st["reset_count"] == 1).all()
    assert same.mean() > 0.97
    # kinematic-sampling envs agree to float32 rounding; random-walk envs accumulate it over up to 64 walk steps
    assert np.abs(st["q"][same] - ost["q"][same]).max() < 2e-4
    assert np.median(np.abs(st["q"][same] - ost["q"][same]).max(axis=1)) < 1e-6
    A = np.array(s.acc_max)
    assert np.abs((st["a"][same] - ost["a"][same]) / A).max() < 2e-3
    # planet phases are uniform over the orbit (chi-square on 12 bins, very loose)
    hist = np.bincount(st["planet_idx"][:, 0] // 100, minlength=12)
    assert hist.min() > 0.6 * B / 12
    env.close(); env2.close()


def test_full_size_invariants_and_determinism(built):
    """BASELINE size (65 536 envs, 24 contact samples per step = the default, auto-reset): size-independent properties over 45 steps."""
    import torch
    B, T = 65536, 45
    outs = []
    for rep in range(2):
        env = _env(B, seed=3, auto_reset=True)
        env.reset()
        s = env.scene.struct
        gen = torch.Generator(device="cuda"); gen.manual_seed(77)
        lens = np.zeros(B, dtype=np.int64)
        n_done, reasons = 0, set()
        acc = torch.zeros(B, device="cuda")
        for t in range(T):
            act = torch.rand((B, 7), device="cuda", generator=gen) * 2 - 1
            if t % 5 == 0:
                act = torch.sign(act)                        # bang-bang steps ride the limits
            obs, rew, done, info = env.step(act)
            assert float(obs.abs().max()) <= 1.0
            acc += rew
            d = done.cpu().numpy().astype(bool)
            r = info["termination_reason"].cpu().numpy()
            vals = info["values"].cpu().numpy()
            lens += 1
            assert (r[~d] == -1).all() and np.isin(r[d], [2, 4, 5]).all()
            assert (lens[d] <= 20).all() and (lens[d & (r == 2)] == 20).all()
            assert (vals[:, 7] == lens).all()
            # reward bounds: 0.4 + 1 + 1 + 3 (+15) / (-15)
            rw = rew.cpu().numpy()
            assert rw.max() <= 5.4 + 15 + 1e-4 and rw.min() >= -15 - 1e-4
            assert (rw[d & (r == 2)] >= 15 - 1e-4).all() and (rw[d & (r >= 4)] <= 5.4 - 15 + 1e-4).all()
            lens[d] = 0
            n_done += int(d.sum()); reasons |= set(r[d].tolist())
            st = env.get_state() if t % 15 == 14 else None
            if st is not None:
                lo, hi = np.array(s.pos_lo), np.array(s.pos_hi)
                assert np.abs(-1 + 2 * (st["q"] - lo) / (hi - lo)).max() <= 1.001          # obs.py:379
                assert (np.abs(st["v"]) / np.array(s.vel_max)).max() <= 1.001               # obs.py:389
                assert (np.abs(st["a"]) / np.array(s.acc_max)).max() <= 1.001               # obs.py:399
                assert (st["step_count"] == lens).all()
        assert n_done > B and {2, 5} <= reasons
        outs.append((acc.cpu().numpy(), env.get_state()))
        env.close()
    assert np.array_equal(outs[0][0], outs[1][0])
    assert all(np.array_equal(outs[0][1][k], outs[1][1][k]) for k in STATE_KEYS)


def test_profiling_counters(built):
    import torch
    env = _env(4096, seed=1, auto_reset=True)
    env.reset()
    env.set_profiling(True)
    act = torch.zeros((4096, 7), device="cuda")
    for _ in range(5):
        env.step(act)
    prof = env.get_profile()
    assert all(prof[k]["launches"] == 5 for k in ("plan", "collide", "finish", "reset"))
    assert prof["collide"]["ms"] > 0
    # a period: the launches of every third step are timed (bench.py: the events around every launch of every step cost the step 12 %)
    env.set_profiling(3)
    for _ in range(7):
        env.step(act)
    prof = env.get_profile()
    assert all(prof[k]["launches"] in (2, 3) for k in ("plan", "collide", "finish")) and prof["plan"]["launches"] == prof["finish"]["launches"]
    env.set_profiling(False)
    env.step(act)
    assert env.get_profile()["plan"]["launches"] == 0
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,switch", [("space_step.npz", "SMR_COLLIDE_LANES"), ("space_bm_step.npz", "SMR_COLLIDE_LANES"),
                                         ("reach_step.npz", "SMR_SELF_ROWS"), ("space_bm_step.npz", "SMR_SELF_ROWS")])
def test_step_against_golden_under_the_ab_switches(built, monkeypatch, name, switch):
    """The A/B switches that keep an older form of the distance stage alive stay correct: SMR_COLLIDE_LANES=1 (planet scenes in the
    lane-per-(env, body) collide kernel, with the next step's ranges as rows of its grid), SMR_SELF_ROWS=1 (the self-collision pairs' bound
    tests as rows of the keys grid)."""
    monkeypatch.setenv(switch, "1")
    test_step_against_golden(built, name)


# ------------------------------------------------------------------ ball scene (A10, BASELINE configs[3])
def _ball_env(B, ball_machine=False, **kw):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from safemotionsrisk_amd import VecSafeMotionsEnv, BALL_ENV_CONFIG
    kw.setdefault("auto_reset", False)
    return VecSafeMotionsEnv(B, device=0, **kw, **dict(BALL_ENV_CONFIG, ball_machine_mode=ball_machine))


BALL_OBS_TOL = np.concatenate([np.full(7, 2e-6), np.full(7, 1e-5), np.full(7, 2e-5), np.full(6, 5e-6)])


def test_ball_components_against_oracle(built):
    from oracle.oracle import OracleEnv
    env = _ball_env(512, seed=2)
    ora = OracleEnv(env.scene, 1)
    rng = np.random.default_rng(0)
    q = rng.uniform(-1.8, 1.8, (512, 7)).astype(np.float32)
    pos, _ = env.test_fk(q)
    ball = np.zeros((512, 10), np.float32)
    ball[:, :3] = pos[np.arange(512), rng.integers(3, 8, 512)] + rng.normal(size=(512, 3)) * rng.choice([0.02, 0.1, 0.3], (512, 1))
    ball[:, 3:6] = rng.normal(size=(512, 3)); ball[:, 6:9] = rng.uniform(-3, 3, (512, 3)); ball[:, 9] = rng.uniform(0, 6.28, 512)
    k = rng.integers(0, 12, 512).astype(np.int32)
    d = env.test_ball_distances(q, ball, k)
    od = np.array([ora.ball_distances(q[i].astype(np.float64), ball[i].astype(np.float64), k[i]) for i in range(512)])
    clamp = lambda x: np.where(x < 0.001, 0.0, x)
    near = (np.abs(od[:, 0] - 0.001) < TIE) | (np.abs(od[:, 2] - 0.001) < TIE)
    assert np.abs(clamp(d[:, 0]) - clamp(od[:, 0]))[~near].max() < 5e-6
    assert np.abs(clamp(d[:, 2]) - clamp(od[:, 2]))[~near].max() < 5e-6
    safe = np.abs(od[:, 2] - 0.02) > TIE
    assert np.array_equal(d[safe, 3], od[safe, 3]) and od[:, 3].sum() > 20
    env.close()


@pytest.mark.parametrize("name", ["ball_step.npz", "ball_bm_step.npz"])
def test_ball_step_against_golden(built, name):
    """ball_bm_step.npz: the same scene on robot_ball_machine.urdf (the shipped backup_networks/ball/params.json) - self-collision pairs,
    and the runs against the ball as jobs of the jobs kernel"""
    import torch
    g = np.load(os.path.join(os.path.dirname(GOLD), name))
    B, T = int(g["B"]), int(g["T"])
    env = _ball_env(B, name == "ball_bm_step.npz", seed=int(g["seed"]))
    env.set_state(q=g["q0"], v=g["v0"], a=g["a0"], step_count=g["step_count0"], moving_hit=0, ball=g["ball0"], ball_i=g["ball_i0"])
    from oracle.oracle import OracleEnv
    ora = OracleEnv(env.scene, 1)
    for t in range(T):
        st0 = env.get_state()
        obs, rew, done, info = env.step(torch.from_numpy(g["actions"][t]).cuda())
        obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        reason = info["termination_reason"].cpu().numpy()
        vals = info["values"].cpu().numpy()
        st = env.get_state()
        gi = g["info"][t]
        bad = (done != g["done"][t]) | (reason != g["reason"][t]) | (st["ball_i"] != g["ball_i"][t]).any(axis=1)   # counters, active flag, spawn count: exact
        if bad.any():
            why = explain_mismatches(ora, env.scene, st0, st, bad)
            assert why.all(), "%d unexplained mismatches at step %d" % (int((~why).sum()), t)
        ok = ~bad & ~((np.abs(gi[:, 0] - 0.001) < TIE) | (np.abs(gi[:, 2] - 0.001) < TIE))
        assert (np.abs(obs[ok] - g["obs"][t][ok]) < BALL_OBS_TOL).all()
        assert np.abs(rew[ok] - g["reward"][t][ok]).max() < 5e-5
        assert np.abs(relevant(vals[:, :3], env.scene)[ok] - relevant(gi[:, :3], env.scene)[ok]).max() < 5e-6
        assert np.abs(st["ball"] - g["ball"][t])[ok].max() < 2e-5                  # new balls: same launch state
        gs = g["state"][t]
        env.set_state(q=gs[:, :7].astype(np.float32), v=gs[:, 7:14].astype(np.float32), a=gs[:, 14:].astype(np.float32),
                      ball=g["ball"][t].astype(np.float32), ball_i=g["ball_i"][t], step_count=np.where(g["done"][t], 0, st["step_count"]))
    env.close()


@pytest.mark.parametrize("ball_machine,switches", [(False, {}), (False, {"SMR_BALL_JOBS": "1"}), (True, {}), (True, {"SMR_BALL_JOBS": "0"}),
                                                   (True, {"SMR_JOB_CAP": "200"})])
def test_ball_rollout_and_reset_against_oracle(built, monkeypatch, ball_machine, switches):
    """Both forms of the runs against the ball - in the lanes of the collide kernel (the default without self-collision pairs) and as jobs
    of the jobs kernel (the default with the ball machine mounted, where the self pairs' jobs run anyway), SMR_BALL_JOBS forcing the
    other one - and a job list that is far too short (overflow_kernel's ball branch): every one against the float64 oracle."""
    import torch
    from oracle.oracle import OracleEnv
    B, T = 4096, 16
    for k, v in switches.items():
        monkeypatch.setenv(k, v)
    env = _ball_env(B, ball_machine, seed=5)
    for k in switches:
        monkeypatch.delenv(k)
    obs0 = env.reset().cpu().numpy()
    ora = OracleEnv(env.scene, B, seed=5, num_threads=16)
    ora1 = OracleEnv(env.scene, 1)
    o_obs0 = ora.reset()
    st, ost = env.get_state(), ora.get_state()
    same = (np.abs(ost["q"] - st["q"]).max(axis=1) < 1e-3) & (ost["ball_i"] == st["ball_i"]).all(axis=1)
    assert same.mean() > 0.97                                                      # same Philox streams -> same start states and balls
    assert np.abs(ost["ball"] - st["ball"])[same].max() < 2e-5
    keys = env.STATE_KEYS
    rng = np.random.default_rng(9)
    mism, spawned, hits = 0, 0, 0
    for t in range(T):
        st = env.get_state()
        ora.set_state(**{k: st[k] for k in keys})
        act = rng.uniform(-1, 1, (B, 7)).astype(np.float32)
        obs, rew, done, info = env.step(torch.from_numpy(act).cuda())
        o_obs, o_rew, o_done, o_reason, o_info = ora.step(act.astype(np.float64))
        obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        reason = info["termination_reason"].cpu().numpy()
        st1, ost1 = env.get_state(), ora.get_state()
        good = (done == o_done) & (reason == o_reason) & (st1["ball_i"] == ost1["ball_i"]).all(axis=1)
        if (~good).any():
            mism += int((~explain_mismatches(ora1, env.scene, st, st1, ~good)).sum())
        assert (np.abs(obs[good] - o_obs[good]) < BALL_OBS_TOL).all()
        assert np.abs(rew[good] - o_rew[good]).max() < 5e-5
        spawned += int((st1["ball_i"][:, 4] != st["ball_i"][:, 4]).sum())
        hits += int((o_reason == 5).sum())
        if done.any():
            env.reset(mask=torch.from_numpy(done).cuda())
    assert mism == 0, "%d unexplained mismatches" % mism
    assert spawned > B and hits > 50
    env.close()


def test_ball_full_size_invariants(built):
    import torch
    B, T = 65536, 30
    env = _ball_env(B, seed=3, auto_reset=True)
    env.reset()
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    reasons = set()
    for t in range(T):
        obs, rew, done, info = env.step(torch.rand((B, 7), device="cuda", generator=gen) * 2 - 1)
        assert float(obs.abs().max()) <= 1.0
        r = info["termination_reason"].cpu().numpy()
        d = done.cpu().numpy().astype(bool)
        assert (r[~d] == -1).all() and np.isin(r[d], [2, 4, 5]).all()
        reasons |= set(r[d].tolist())
        if t % 10 == 9:
            st = env.get_state()
            k, kmax, khit, active, spawns = st["ball_i"].T
            assert (active == 1).all() and (k >= 0).all() and (k <= np.minimum(kmax, khit) + 24).all() and (spawns >= 1).all()
            assert np.allclose(np.linalg.norm(st["ball"][:, 3:6], axis=1), 6.0, atol=1e-4)
    assert {2, 4, 5} <= reasons
    env.close()


# ---- robot_ball_machine.urdf (ball_machine_mode=True: what the shipped backup / task network env_configs ask for) ----------

def _bm_env(B, **kw):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from safemotionsrisk_amd import VecSafeMotionsEnv, SPACE_ENV_CONFIG
    kw.setdefault("auto_reset", False)
    return VecSafeMotionsEnv(B, device=0, **kw, **dict(SPACE_ENV_CONFIG, ball_machine_mode=True))


@pytest.mark.gpu
def test_ball_machine_components_against_oracle(built):
    """table / self / planet distances and contact flags with the ball machine mounted (10 bodies, 14 self pairs)."""
    from oracle.oracle import OracleEnv
    env = _bm_env(64, seed=0)
    s = env.scene.struct
    assert s.n_bodies == 10 and s.n_self_pairs == 14
    ora = OracleEnv(env.scene, 1)
    rng = np.random.default_rng(17)
    n = 512
    q = rng.uniform(np.array(s.pos_lo), np.array(s.pos_hi), (n, 7)).astype(np.float32)
    i0 = rng.integers(0, 1200, n)
    idx = np.stack([i0, (i0 - 480) % 1200], axis=1).astype(np.int32)
    d = env.test_distances(q, idx)
    clamp = lambda x: np.where(x < 0.001, 0.0, x)
    od = np.array([ora.distances(q[i].astype(np.float64), idx[i]) for i in range(n)])
    near = (np.abs(od[:, :3] - 0.001) < TIE).any(axis=1)
    for c in range(3):                                       # static, self, moving
        assert np.abs(clamp(d[:, c]) - clamp(od[:, c]))[~near].max() < 5e-6
    safe = np.abs(od[:, 2] - 0.02) > TIE
    assert np.array_equal(d[safe, 3], od[safe, 3])
    assert (od[:, 1] < 0.001).sum() > 5 and (od[:, 1] < 0.1).sum() > 50     # the self pairs are exercised
    env.close()


@pytest.mark.gpu
def test_start_position_sampling_attempts_column(built):
    """_starting_point_sampling_attempts (ctlp.py:1480-1488 -> obstacles_starting_point_sampling_attempts, ctlp.py:1154): the info column
    of an episode carries the number of candidates its start position took - the oracle's count on the same Philox streams where the
    two accepted the same candidate, >= 1 everywhere, with the rejection rate of the README scene (about one candidate in ten)."""
    import torch
    from oracle.oracle import OracleEnv
    B = 4096
    env = _bm_env(B, seed=21)
    env.reset()
    ora = OracleEnv(env.scene, B, seed=21, num_threads=16)
    ora.reset()
    st, ost = env.get_state(), ora.get_state()
    same = np.abs(ost["q"] - st["q"]).max(axis=1) < 1e-3
    act = np.zeros((B, 7), np.float32)
    _, _, _, info = env.step(torch.from_numpy(act).cuda())
    _, _, _, _, o_info = ora.step(act.astype(np.float64))
    att = info["values"][:, 14].cpu().numpy()
    assert same.mean() > 0.97 and np.array_equal(att[same], o_info[same, 14]) and (att >= 1).all()
    assert 1.02 < att.mean() < 1.6 and att.max() >= 3
    env.close()


@pytest.mark.gpu
def test_self_pair_bound_tables_are_lower_bounds(built):
    """The bound tables of the near self-collision pairs (csrc/smr_device.h "Self pairs: bound tables"): for random and for
    near-collision configurations the bound never exceeds the oracle's float64 distance of the literal query (ctlp.py:3345-3372),
    and it clears the range the reward reads for most configurations - that is what it is for."""
    from oracle.oracle import OracleEnv
    env = _bm_env(64, seed=0)
    s = env.scene.struct
    ora = OracleEnv(env.scene, 1)
    rng = np.random.default_rng(3)
    lo, hi = np.array(s.pos_lo), np.array(s.pos_hi)
    q = rng.uniform(lo, hi, (1500, 7))
    q[1000:, 4:] = q[1000:1001, 4:] + rng.normal(size=(500, 3)) * 0.05           # a cluster inside a few cells
    q = np.clip(q, lo, hi).astype(np.float32)
    lb = env.test_self_bounds(q)
    tabled = [i for i in range(s.n_self_pairs) if s.self_pair[i][1] in (4, 5)]
    assert len(tabled) == 4 and (lb[:, tabled] > -1e30).all() and (lb[:, [i for i in range(s.n_self_pairs) if i not in tabled]] < -1e30).all()
    cut = self_cut(env.scene)
    for i in tabled:
        d = np.array([ora.self_pair_distance(q[n].astype(np.float64), i) for n in range(len(q))])
        informative = lb[:, i] > 0
        assert (lb[informative, i] <= d[informative] + 1e-6).all(), (i, float((lb[:, i] - d)[informative].max()))
        cleared = lb[:1000, i] > cut
        assert cleared.mean() > (0.75 if s.self_pair[i][1] == 5 else 0.85), (i, float(cleared.mean()))
        assert (d[:1000][cleared] > cut).all()
        # tightness where it matters: the two-angle tables lose a few millimetres, the three-angle tables a few centimetres
        loss = (np.minimum(d, 0.25) - lb[:, i])[d < 0.25]
        assert np.median(loss) < (0.006 if s.self_pair[i][1] == 5 else 0.05), (i, float(np.median(loss)))
    env.close()


@pytest.mark.gpu
def test_job_list_overflow_is_worked_by_the_overflow_kernel(built, monkeypatch):
    """A job list that is far too short (SMR_JOB_CAP): the envs whose jobs did not fit get all their queries from overflow_kernel -
    same distances, flags and rewards as the run with the full list."""
    import torch
    B = 1024
    full = _bm_env(B, seed=11)
    full.reset()
    st = full.get_state()
    monkeypatch.setenv("SMR_JOB_CAP", "300")
    short = _bm_env(B, seed=11)
    monkeypatch.delenv("SMR_JOB_CAP")
    rng = np.random.default_rng(2)
    for t in range(3):
        short.set_state(**{k: st[k] for k in STATE_KEYS})
        full.set_state(**{k: st[k] for k in STATE_KEYS})
        act = torch.from_numpy(rng.uniform(-1, 1, (B, 7)).astype(np.float32)).cuda()
        o1, r1, d1, i1 = full.step(act)
        o2, r2, d2, i2 = short.step(act)
        v1, v2 = i1["values"].cpu().numpy(), i2["values"].cpu().numpy()
        tie = (np.abs(v1[:, :3] - 0.001) < TIE).any(axis=1) | (np.abs(v2[:, :3] - 0.001) < TIE).any(axis=1)
        assert np.abs(v1[~tie, :3] - v2[~tie, :3]).max() < 2e-6
        flags = (d1 != d2).cpu().numpy() | (i1["termination_reason"] != i2["termination_reason"]).cpu().numpy()
        assert int((flags & ~tie).sum()) <= 2                                  # (a contact distance on its 0.02 m threshold)
        assert np.abs((r1 - r2).cpu().numpy()[~tie & ~flags]).max() < 2e-5
        st = full.get_state()
        if d1.any():
            full.reset(mask=d1)
            st = full.get_state()
    full.close(); short.close()


@pytest.mark.gpu
def test_ball_machine_rollout_and_reset_against_oracle(built):
    import torch
    from oracle.oracle import OracleEnv
    B, T = 4096, 12
    env = _bm_env(B, seed=5)
    env.reset()
    ora = OracleEnv(env.scene, B, seed=5, num_threads=16)
    ora.reset()
    st, ost = env.get_state(), ora.get_state()
    same = (ost["planet_idx"] == st["planet_idx"]).all(axis=1) & (np.abs(ost["q"] - st["q"]).max(axis=1) < 1e-3)
    assert same.mean() > 0.97                                 # same Philox streams -> same start states (self pairs included)
    rng = np.random.default_rng(9)
    mism, reasons = 0, []
    for t in range(T):
        st = env.get_state()
        ora.set_state(**{k: st[k] for k in STATE_KEYS})
        act = rng.uniform(-1, 1, (B, 7)).astype(np.float32)
        obs, rew, done, info = env.step(torch.from_numpy(act).cuda())
        o_obs, o_rew, o_done, o_reason, o_info = ora.step(act.astype(np.float64))
        obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        reason = info["termination_reason"].cpu().numpy()
        bad = (done != o_done) | (reason != o_reason)
        if bad.any():
            mism += int((~explain_mismatches(ora, env.scene, st, env.get_state(), bad)).sum())
        good = ~bad
        assert (np.abs(obs[good] - o_obs[good]) < OBS_TOL).all()
        assert np.abs(rew[good] - o_rew[good]).max() < 5e-5
        assert np.abs(relevant(info["values"].cpu().numpy()[:, :3], env.scene)[good, 1] - relevant(o_info[:, :3], env.scene)[good, 1]).max() < 5e-6      # self distance
        reasons += o_reason[o_done.astype(bool)].tolist()
        if done.any():
            env.reset(mask=torch.from_numpy(done).cuda())
    assert mism == 0, "%d unexplained mismatches" % mism
    assert reasons.count(3) > 20 and reasons.count(4) > 20 and reasons.count(5) > 20     # self / static / moving terminations
    env.close()


# ---- target-point reaching task (A14): SafeMotionsEnv with TargetPointReachingReward, shipped task config minus its risk network ----

def _brake_env(B, **kw):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from safemotionsrisk_amd import VecSafeMotionsEnv
    from safemotionsrisk_amd.scene import reach_brake_env_config
    kw.setdefault("auto_reset", False)
    return VecSafeMotionsEnv(B, device=0, **kw, **reach_brake_env_config())


REACH_OBS_TOL = np.concatenate([np.full(7, 2e-6), np.full(7, 1e-5), np.full(7, 2e-5), np.full(6, 2e-6), np.full(2, 1e-6)])


def _reach_env(B, **kw):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from safemotionsrisk_amd import VecSafeMotionsEnv, REACH_ENV_CONFIG
    kw.setdefault("auto_reset", False)
    return VecSafeMotionsEnv(B, device=0, **kw, **REACH_ENV_CONFIG)


@pytest.mark.gpu
def test_reaching_task_rollout_and_reset_against_oracle(built):
    """target points: sampled at reset and when reached, observed (position + relative position), rewarded by the decrease of
    the distance normalised to the initial one + bonus; episodes start at rest; 4096 envs x 14 steps with state injection.
    To see targets reached within 14 steps, a quarter of the envs get their target put 8 cm from the target link point."""
    import torch
    from oracle.oracle import OracleEnv
    B, T = 4096, 14
    env = _reach_env(B, seed=5)
    obs0 = env.reset().cpu().numpy()
    assert obs0.shape == (B, 29)
    ora = OracleEnv(env.scene, B, seed=5, num_threads=16)
    ora.reset()
    st, ost = env.get_state(), ora.get_state()
    same = np.abs(ost["q"] - st["q"]).max(axis=1) < 1e-3
    assert same.mean() > 0.97 and (st["v"] == 0).all() and (st["a"] == 0).all()          # episodes start at rest
    assert np.abs(ost["target"] - st["target"])[same].max() < 1e-4 and (st["target_count"] == 1).all()
    pos, _ = zip(*[ora.body_frames(st["q"][e].astype(np.float64)) for e in range(0, B, 4)])
    near = st["target"].copy()
    near[::4, :3] = np.array([p[8] for p in pos]) + np.array([0.0, 0.0, 0.18])            # ee = ball_machine frame + (0, 0, 0.1)
    near[::4, 3] = near[::4, 4] = 0.08
    env.set_state(target=near)
    rng = np.random.default_rng(9)
    mism, reached, reasons = 0, 0, []
    for t in range(T):
        st = env.get_state()
        ora.set_state(**{k: st[k] for k in env.STATE_KEYS})
        act = rng.uniform(-1, 1, (B, 7)).astype(np.float32)
        obs, rew, done, info = env.step(torch.from_numpy(act).cuda())
        o_obs, o_rew, o_done, o_reason, o_info = ora.step(act.astype(np.float64))
        obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        reason = info["termination_reason"].cpu().numpy()
        st1, ost1 = env.get_state(), ora.get_state()
        bad = (done != o_done) | (reason != o_reason) | (st1["target_count"] != ost1["target_count"])
        if bad.any():
            why = explain_mismatches(ora, env.scene, st, st1, bad)
            # ... or the target link point passes the 0.065 m target sphere within rounding at one of the 24 updates
            for n, e in enumerate(np.nonzero(bad)[0]):
                if not why[n] and st1["target_count"][e] != ost1["target_count"][e]:
                    from parity_util import cubic
                    dmin = min(abs(np.linalg.norm(_ee_exact(ora, env, cubic(st["q"][e], st["v"][e], st["a"][e], st1["a"][e], 0.1 * u / 24, 0.1))
                                                  - st["target"][e, :3]) - 0.065) for u in range(1, 25))
                    why[n] = dmin < TIE
            mism += int((~why).sum())
        good = ~bad
        new = good & (st1["target_count"] != st["target_count"])
        reached += int(new.sum())
        assert (np.abs(obs[good] - o_obs[good]) < 5e-5).all()                             # a fresh target: float32 FK of a random pose
        assert (np.abs(obs[good & ~new] - o_obs[good & ~new]) < REACH_OBS_TOL).all()
        assert np.abs(rew[good] - o_rew[good]).max() < 2e-4                               # distance differences / (0.1 s x initial distance)
        assert np.abs(st1["target"][good] - ost1["target"][good]).max() < 2e-5
        reasons += o_reason[o_done.astype(bool)].tolist()
        if done.any():
            env.reset(mask=torch.from_numpy(done).cuda())
    assert mism == 0, "%d unexplained mismatches" % mism
    assert reached > 30 and len(reasons) > 100
    env.close()


@pytest.mark.gpu
def test_episode_metrics_on_the_gpu_env(built):
    """EpisodeMetrics over the real env (auto-reset): every env finishes at least one episode in 25 steps, episode lengths
    are those of the info column, the termination rates partition the episodes, distances feed the min metrics."""
    import torch
    from safemotionsrisk_amd import EpisodeMetrics
    B = 2048
    env = _env(B, seed=2, auto_reset=True)
    env.reset()
    m = EpisodeMetrics(B, env.INFO_COLUMNS, device=env.device, auto_reset=True, trajectory_length=21)
    gen = torch.Generator(device="cuda").manual_seed(0)
    n, agg = 0, {}
    for _ in range(25):
        act = torch.rand((B, 7), device="cuda", generator=gen) * 2 - 1
        obs, rew, done, info = env.step(act)
        fin = m.update(obs, rew, done, info)
        if len(fin):
            n += len(fin)
            for k, v in fin.metrics.items():
                agg[k] = agg.get(k, 0.0) + float(v.sum())
            assert (fin.episode_length >= 1).all() and (fin.episode_length <= 20).all()
            assert (fin.metrics["joint_0_pos_abs_max"] <= 1.0001).all()
            assert (fin.metrics["minimum_distance_to_moving_obstacles_min"] <= fin.metrics["minimum_distance_to_moving_obstacles_average"] + 1e-6).all()
    assert n >= B
    rates = sum(agg[k] for k in agg if k.endswith("_termination_rate")) / n
    assert abs(rates - 1.0) < 1e-6 and agg["success_rate"] / n == 1.0 and 1.0 < agg["episode_length"] / n <= 20.0
    moving = agg["collision_moving_obstacles_termination_rate"] / n
    assert 0.0 < moving < 1.0 and agg["collision_rate_moving_obstacles_max"] / n >= moving - 1e-6
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["space", "ball", "reach"])
def test_ranges_prepared_one_step_ahead_equal_the_plan_kernels_own(built, monkeypatch, scene):
    """The safe acceleration ranges of step t + 1 are computed during step t (extra blocks of the jobs launch / extra rows of the collide
    grid; restarting envs bring the range of their staged start state) - SMR_RANGE_AHEAD=0 lets the plan kernel compute them as it did
    before.  Same function of the same state: the two runs agree as two batch compositions do (test_ragged_batches), episodes end in the same
    steps but for distances on a threshold, and a state injected in between (set_state
    invalidates the prepared ranges) changes nothing."""
    import torch
    import safemotionsrisk_amd as smr
    cfg = {"space": smr.SPACE_ENV_CONFIG, "ball": smr.BALL_ENV_CONFIG, "reach": smr.REACH_ENV_CONFIG}[scene]
    B, T = 2048, 30
    acts = np.random.default_rng(5).uniform(-1, 1, (T, B, 7)).astype(np.float32)
    runs = []
    for ahead in ("1", "0"):
        monkeypatch.setenv("SMR_RANGE_AHEAD", ahead)
        env = smr.VecSafeMotionsEnv(B, device=0, seed=3, auto_reset=True, **cfg)
        monkeypatch.delenv("SMR_RANGE_AHEAD")
        rec = [env.reset().cpu().numpy().copy()]
        for t in range(T):
            if t == 12:                                   # an injected state: the same one, so the runs stay comparable
                st = env.get_state()
                env.set_state(**{k: st[k] for k in env.STATE_KEYS})
            obs, rew, done, info = env.step(torch.from_numpy(acts[t]).cuda())
            rec.append(np.concatenate([obs.cpu().numpy(), rew.cpu().numpy()[:, None], done.cpu().numpy()[:, None].astype(np.float32)], axis=1))
        runs.append(rec)
        env.close()
    assert np.array_equal(runs[0][0], runs[1][0])
    # (the tolerance of test_ragged_batches: a range end is the largest feasible point the wavefront's shared k-section evaluated, and which
    # points it evaluates depends on how many lanes of the wavefront have an open bracket - the two runs differ in exactly that)
    alive = np.ones(B, bool)                              # envs whose episodes ended in the same steps so far
    close, tight = [], []
    for a, b in zip(runs[0][1:], runs[1][1:]):
        alive &= a[:, -1] == b[:, -1]
        d = np.abs(a[alive, :-2] - b[alive, :-2]).max(axis=1)
        close.append((d < 1e-4).mean()); tight.append((d < 2e-6).mean())
    assert alive.mean() > 0.99 and min(close) > 0.97 and np.mean(tight) > 0.8, (alive.mean(), min(close), np.mean(tight))


@pytest.mark.gpu
@pytest.mark.parametrize("auto_reset", [False, True])
def test_ragged_batches_give_the_same_envs(built, auto_reset):
    """Batch sizes that fill neither a wavefront nor a 256-lane block (1, 63, 257, 1000): env i is the same env in every batch
    (Philox streams are keyed by the env index) through resets - bit for bit where its wavefront holds the same 64 envs,
    to rounding where the last wavefront is partly empty (the cooperative root finder of the plan kernel places its
    evaluation points by the number of open brackets in the wavefront); and the oracle agrees on the small one."""
    import torch
    from oracle.oracle import OracleEnv
    T = 24
    rng = np.random.default_rng(4)
    acts = rng.uniform(-1, 1, (T, 1000, 7)).astype(np.float32)
    outs = {}
    for B in (1000, 257, 63, 1):
        env = _env(B, seed=11, auto_reset=auto_reset)
        rec = [env.reset().cpu().numpy().copy()]
        for t in range(T):
            obs, rew, done, info = env.step(torch.from_numpy(acts[t, :B]).cuda())
            rec.append(np.concatenate([obs.cpu().numpy(), rew.cpu().numpy()[:, None], done.cpu().numpy()[:, None].astype(np.float32),
                                       info["termination_reason"].cpu().numpy()[:, None].astype(np.float32), info["values"].cpu().numpy()], axis=1))
            if not auto_reset and done.any():
                env.reset(mask=done)
        outs[B] = rec
        env.close()
    for B in (257, 63, 1):
        full = (B // 64) * 64
        assert np.array_equal(outs[B][0], outs[1000][0][:B])
        close = []
        for t in range(1, T + 1):
            assert np.array_equal(outs[B][t][:full], outs[1000][t][:full]), (B, t)
            close.append(np.abs(outs[B][t][full:] - outs[1000][t][full:B]).max(axis=1) < 1e-4)
        assert np.mean(close) > 0.97, (B, np.mean(close))
    assert sum(r[:, 24].sum() for r in outs[1000][1:]) > 900            # every env finished about once
    ora = OracleEnv(_scene_of(auto_reset), 63, seed=11, num_threads=8)
    o0 = ora.reset()
    same = np.abs(o0 - outs[63][0]).max(axis=1) < 1e-4                  # start states: accept/reject ties aside
    assert same.mean() > 0.95
    o1, r1, d1, rs1, _ = ora.step(acts[0, :63].astype(np.float64))
    assert (np.abs(o1[same] - outs[63][1][same, :23]) < 1e-4).mean() > 0.99


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["space", "ball_bm", "reach"])
def test_shards_are_a_partition_of_the_global_batch(built, scene):
    """env_index_base (smr_create): a handle over the envs [base, base + B) of a global batch draws what ONE handle over the whole
    batch draws for them - start states, planet phases, ball launches, target points, through resets, auto-resets and a masked
    reset - so the concatenation of the shards' (obs, reward, done, reason, info) is bit-identical to the single handle's, for an
    even and an uneven split (block-aligned bases: the wavefronts then hold the same envs; see test_ragged_batches for the rest).
    North star: "envs shard trivially ... only to concatenate the rollout batch"."""
    import torch
    import safemotionsrisk_amd as smr
    from safemotionsrisk_amd import VecSafeMotionsEnv
    cfg = {"space": smr.SPACE_ENV_CONFIG, "ball_bm": dict(smr.BALL_ENV_CONFIG, ball_machine_mode=True), "reach": smr.REACH_ENV_CONFIG}[scene]
    G, T = 1024, 25
    acts = np.random.default_rng(8).uniform(-1, 1, (T, G, 7)).astype(np.float32)
    acts[::4] = np.sign(acts[::4])                              # bang-bang steps: collisions and early terminations

    def run(splits):
        envs = [VecSafeMotionsEnv(hi - lo, device=0, seed=5, auto_reset=True, env_index_base=lo, **cfg) for lo, hi in splits]
        rec = [np.concatenate([e.reset().cpu().numpy() for e in envs])]
        for t in range(T):
            rows = []
            for e, (lo, hi) in zip(envs, splits):
                obs, rew, done, info = e.step(torch.from_numpy(acts[t, lo:hi]).cuda())
                rows.append(np.concatenate([obs.cpu().numpy(), rew.cpu().numpy()[:, None], done.cpu().numpy()[:, None].astype(np.float32),
                                            info["termination_reason"].cpu().numpy()[:, None].astype(np.float32), info["values"].cpu().numpy()], axis=1))
            rec.append(np.concatenate(rows))
            if t == 12:                                         # a masked reset in the middle: every third env of the GLOBAL batch
                obs_rows = []
                for e, (lo, hi) in zip(envs, splits):
                    mask = torch.from_numpy((np.arange(lo, hi) % 3 == 0)).cuda()
                    obs_rows.append(e.reset(mask=mask).cpu().numpy())
                rec.append(np.concatenate(obs_rows))
        states = [e.get_state() for e in envs]
        for e in envs:
            e.close()
        return rec, {k: np.concatenate([st[k] for st in states]) for k in states[0]}

    one, st_one = run([(0, G)])
    for splits in ([(0, 512), (512, 1024)], [(0, 256), (256, 768), (768, 1024)]):
        rec, st = run(splits)
        for t, (a, b) in enumerate(zip(one, rec)):
            assert np.array_equal(a, b), (scene, splits, t, int((a != b).any(axis=1).sum()))
        assert all(np.array_equal(st_one[k], st[k]) for k in st_one)
    dones = sum(r[:, one[0].shape[1] + 1].sum() for r in one[1:] if r.shape[1] > one[0].shape[1])
    assert dones > 100                                          # auto-resets happened inside the compared window


@pytest.mark.gpu
@pytest.mark.parametrize("ball_machine", [False, True])
def test_ragged_batches_in_the_ball_scenes(built, ball_machine):
    """The same for the ball scenes, whose collide grid has 128-lane blocks and rows behind the bodies (the next step's acceleration ranges
    on the plain robot; with the ball machine mounted the runs against the ball are jobs and the ranges ride in the jobs launch): batch
    sizes that fill neither a wavefront nor a block give every env the values it has in a batch of 1000 - bit for bit in complete
    wavefronts - through auto-resets, ball launches and respawns."""
    import torch
    T = 24
    acts = np.random.default_rng(6).uniform(-1, 1, (T, 1000, 7)).astype(np.float32)
    outs = {}
    for B in (1000, 321, 129, 5):
        env = _ball_env(B, ball_machine, seed=23, auto_reset=True)
        rec = [env.reset().cpu().numpy().copy()]
        for t in range(T):
            obs, rew, done, info = env.step(torch.from_numpy(acts[t, :B]).cuda())
            rec.append(np.concatenate([obs.cpu().numpy(), rew.cpu().numpy()[:, None], done.cpu().numpy()[:, None].astype(np.float32),
                                       info["termination_reason"].cpu().numpy()[:, None].astype(np.float32), info["values"].cpu().numpy()], axis=1))
        st = env.get_state()
        rec.append(np.concatenate([st["ball"], st["ball_i"].astype(np.float32)], axis=1))
        outs[B] = rec
        env.close()
    for B in (321, 129, 5):
        full = (B // 64) * 64
        close = []
        for t in range(T + 2):
            assert np.array_equal(outs[B][t][:full], outs[1000][t][:full]), (B, t)
            if full < B:
                close.append(np.abs(outs[B][t][full:] - outs[1000][t][full:B]).max(axis=1) < 1e-4)
        assert np.mean(np.concatenate(close)) > 0.95, (B, np.mean(np.concatenate(close)))
    assert sum(r[:, -17].sum() for r in outs[1000][1:-1]) > 500            # (done column) episodes ended


def _scene_of(auto_reset):
    from safemotionsrisk_amd import SPACE_ENV_CONFIG, build_scene
    return build_scene(SPACE_ENV_CONFIG, auto_reset=auto_reset)


PLANET_VARIANTS = {
    "iss_only": dict(planet_two_center=None, planet_two_time_shift=None),
    "asteroid_only": dict(planet_one_center=None),
    "uncoupled": dict(planet_two_time_shift=None, planet_two_period=4.0),
    "no_termination": dict(terminate_on_collision_with_moving_obstacle=False),
    "pos_obs": dict(obs_planet_size_per_planet=3),
    # 48 obstacle updates per step: N = 48 exceeds what the job words / item queues carry (rejected, tests/test_scene_config.py), 24 samples of
    # two updates each is the explicit choice - the sample index of end-of-step jobs (k = N) must survive the job word's packing
    "time_step_0p2": dict(trajectory_time_step=0.2, trajectory_duration=4.0),
}
PLANET_VARIANT_KW = {"time_step_0p2": dict(collision_checks_per_step=24)}


@pytest.mark.gpu
@pytest.mark.parametrize("variant", sorted(PLANET_VARIANTS))
def test_planet_scene_variants_against_oracle(built, variant):
    """one planet (either), two uncoupled planets with different periods, planets that do not end the episode, planet
    positions in the observation: 1024 envs x 8 steps with state injection, as the rollout test above."""
    import torch
    from oracle.oracle import OracleEnv
    from safemotionsrisk_amd import VecSafeMotionsEnv, SPACE_ENV_CONFIG
    B, T = 1024, 8
    cfg = dict(SPACE_ENV_CONFIG, **PLANET_VARIANTS[variant])
    env = VecSafeMotionsEnv(B, device=0, seed=8, auto_reset=False, **PLANET_VARIANT_KW.get(variant, {}), **cfg)
    o0 = env.reset().cpu().numpy()
    ora = OracleEnv(env.scene, B, seed=8, num_threads=16)
    ora1 = OracleEnv(env.scene, 1)
    same = np.abs(ora.reset() - o0).max(axis=1) < 1e-4
    assert same.mean() > 0.95
    rng = np.random.default_rng(1)
    mism, n_done = 0, 0
    for t in range(T):
        st = env.get_state()
        ora.set_state(**{k: st[k] for k in STATE_KEYS})
        act = rng.uniform(-1, 1, (B, 7)).astype(np.float32)
        obs, rew, done, info = env.step(torch.from_numpy(act).cuda())
        o_obs, o_rew, o_done, o_reason, o_info = ora.step(act.astype(np.float64))
        obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        vals = info["values"].cpu().numpy()
        bad = (done != o_done) | (info["termination_reason"].cpu().numpy() != o_reason)
        tie = (np.abs(o_info[:, :3] - 0.001) < TIE).any(axis=1) | (np.abs(o_info[:, 2] - 0.02) < 10 * TIE)
        if bad.any():
            mism += int((~explain_mismatches(ora1, env.scene, st, env.get_state(), bad)).sum())
        good = ~bad
        assert np.abs(obs[good] - o_obs[good]).max() < 2e-5 and np.abs(rew[good] - o_rew[good]).max() < 5e-5
        ok = good & ~tie
        assert np.abs(relevant(vals[:, :3], env.scene)[ok] - relevant(o_info[:, :3], env.scene)[ok]).max() < 5e-6
        n_done += int(o_done.sum())
        if done.any():
            env.reset(mask=torch.from_numpy(done).cuda())
    assert mism == 0, "%d unexplained mismatches" % mism
    assert n_done > (0 if variant == "no_termination" else 20)
    env.close()


@pytest.mark.gpu
def test_staged_target_points_equal_the_synchronous_path(built):
    """reaching task with auto-reset: start states AND target points prepared ahead on the side stream (default) against
    SMR_SYNC_RESET=1 (everything drawn inside step()): bit-identical observations, rewards, dones and target state over 60
    steps, with targets reached and episodes restarting on the way."""
    import torch
    B, T = 2048, 60
    rng = np.random.default_rng(2)
    acts = rng.uniform(-1, 1, (T, B, 7)).astype(np.float32)
    runs = []
    for sync in ("0", "1"):
        os.environ["SMR_SYNC_RESET"] = sync
        try:
            env = _reach_env(B, seed=13, auto_reset=True)
        finally:
            del os.environ["SMR_SYNC_RESET"]
        rec = [env.reset().cpu().numpy().copy()]
        st = env.get_state()
        near = st["target"].copy()                            # put a third of the targets close by, so that they are reached
        near[::3, :3] = np.array([0.0, 0.0, 0.0]) + st["target"][::3, :3] * 0 + _ee_of(env, st["q"][::3]) + np.array([0.0, 0.0, 0.07])
        near[::3, 3] = near[::3, 4] = 0.07
        env.set_state(target=near)
        counts = 0
        for t in range(T):
            obs, rew, done, info = env.step(torch.from_numpy(acts[t]).cuda())
            rec.append(np.concatenate([obs.cpu().numpy(), rew.cpu().numpy()[:, None], done.cpu().numpy()[:, None].astype(np.float32)], axis=1))
        st = env.get_state()
        rec.append(st["target"]); rec.append(st["target_count"].astype(np.float32)[:, None])
        runs.append(rec)
        env.close()
    # the installed target's distances come from the end-of-step target link point finish has at hand, the synchronous kernel
    # recomputes it from the stored joint positions: equal to rounding, so are the rewards that divide by them
    for i, (a, b) in enumerate(zip(*runs)):
        assert a.shape == b.shape and np.abs(a - b).max() < (2e-4 if 0 < i <= T else 2e-6), (i, np.abs(a - b).max())
        if 0 < i <= T:
            assert np.array_equal(a[:, -1], b[:, -1]) and np.abs(a[:, :-2] - b[:, :-2]).max() < 2e-6      # dones exact, observations to rounding
    assert (runs[0][-1] > 2).sum() > 50                                   # targets were reached and re-drawn (beyond the resets)


def _ee_exact(ora, env, q):
    """target link point (scene.py:185-208) of joint vector q, float64"""
    s = env.scene.struct
    pos, rot = ora.body_frames(np.asarray(q, dtype=np.float64))
    return pos[s.ee_body] + rot[s.ee_body] @ np.array(s.ee_offset)


def _ee_of(env, q):
    from oracle.oracle import OracleEnv
    ora = OracleEnv(env.scene, 1)
    return np.array([ora.body_frames(x.astype(np.float64))[0][8] for x in q]) + np.array([0.0, 0.0, 0.1])


@pytest.mark.gpu
@pytest.mark.parametrize("ball_machine", [False, True])
def test_staged_balls_equal_the_synchronous_path(built, ball_machine):
    """ball scene with auto-reset: the launch of every env's next ball prepared ahead on the side stream (default) against
    SMR_SYNC_RESET=1 (drawn inside step()): identical observations, rewards, dones and ball state over 60 steps."""
    import torch
    from safemotionsrisk_amd import VecSafeMotionsEnv, BALL_ENV_CONFIG
    B, T = 1024, 60
    rng = np.random.default_rng(3)
    acts = rng.uniform(-1, 1, (T, B, 7)).astype(np.float32)
    runs = []
    for sync in ("0", "1"):
        os.environ["SMR_SYNC_RESET"] = sync
        try:
            env = VecSafeMotionsEnv(B, device=0, seed=17, auto_reset=True, **dict(BALL_ENV_CONFIG, ball_machine_mode=ball_machine))
        finally:
            del os.environ["SMR_SYNC_RESET"]
        rec = [env.reset().cpu().numpy().copy()]
        for t in range(T):
            obs, rew, done, info = env.step(torch.from_numpy(acts[t]).cuda())
            rec.append(np.concatenate([obs.cpu().numpy(), rew.cpu().numpy()[:, None], done.cpu().numpy()[:, None].astype(np.float32)], axis=1))
        st = env.get_state()
        rec.append(st["ball"]); rec.append(st["ball_i"].astype(np.float32))
        runs.append(rec)
        env.close()
    for i, (a, b) in enumerate(zip(*runs)):
        assert np.array_equal(a, b), i
    assert (runs[0][-1].reshape(B, -1)[:, 4] > 3).sum() > B // 4            # several balls per env were spawned


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["space", "ball", "reach"])
def test_staging_survives_manual_resets_and_state_injection(built, scene):
    """auto-reset stepping mixed with env.reset(mask), env.reset() and set_state of the counters the staged entries are keyed by:
    the staged pipeline (side streams, alternating lists) against SMR_SYNC_RESET=1."""
    import torch
    from safemotionsrisk_amd import VecSafeMotionsEnv, SPACE_ENV_CONFIG, BALL_ENV_CONFIG, REACH_ENV_CONFIG
    cfg = {"space": SPACE_ENV_CONFIG, "ball": BALL_ENV_CONFIG, "reach": REACH_ENV_CONFIG}[scene]
    B, T = 777, 45
    rng = np.random.default_rng(6)
    acts = rng.uniform(-1, 1, (T, B, 7)).astype(np.float32)
    masks = rng.random((T, B)) < 0.2
    runs = []
    for sync in ("0", "1"):
        os.environ["SMR_SYNC_RESET"] = sync
        try:
            env = VecSafeMotionsEnv(B, device=0, seed=23, auto_reset=True, **cfg)
        finally:
            del os.environ["SMR_SYNC_RESET"]
        rec = [env.reset().cpu().numpy().copy()]
        for t in range(T):
            obs, rew, done, info = env.step(torch.from_numpy(acts[t]).cuda())
            rec.append(np.concatenate([obs.cpu().numpy(), rew.cpu().numpy()[:, None], done.cpu().numpy()[:, None].astype(np.float32)], axis=1))
            if t % 7 == 3:
                rec.append(env.reset(mask=torch.from_numpy(masks[t])).cpu().numpy().copy())
            if t == 20:
                rec.append(env.reset().cpu().numpy().copy())
            if t == 30:                                        # jump the counters: every staged entry is stale afterwards
                st = env.get_state()
                kw = dict(reset_count=st["reset_count"] + 5)
                if scene == "ball":
                    bi = st["ball_i"].copy(); bi[:, 4] += 3; kw["ball_i"] = bi
                if scene == "reach":
                    kw["target_count"] = st["target_count"] + 4
                env.set_state(**kw)
        env.synchronize()
        runs.append(rec)
        env.close()
    tol = 2e-4 if scene == "reach" else 0.0
    for i, (a, b) in enumerate(zip(*runs)):
        assert a.shape == b.shape and np.abs(a - b).max() <= tol, (i, np.abs(a - b).max())


@pytest.mark.gpu
def test_limits_hold_under_adversarial_actions_on_the_gpu(built):
    """The invariants the reference asserts (obs.py:379-403: |pos|, |vel|, |acc| <= 1.001) over 8 s episodes of actions that
    push every joint into a limit and hold it there (constant signs per env), flip (every 7 steps) and chatter (every step):
    the range code runs on 1-ulp reciprocals / square roots on the GPU, the CPU tests of the same loops on exact ones."""
    import torch
    from safemotionsrisk_amd import VecSafeMotionsEnv, SPACE_ENV_CONFIG
    B, T = 8192, 80
    cfg = dict(SPACE_ENV_CONFIG, trajectory_duration=8.0, terminate_on_collision_with_moving_obstacle=False,
               terminate_on_collision_with_static_obstacle=False)
    for jerk in (1.0, 0.15):
        env = VecSafeMotionsEnv(B, device=0, seed=31, auto_reset=True, **dict(cfg, jerk_limit_factor=jerk))
        env.reset()
        s = env.scene.struct
        lo, hi, V, A = (np.array(x) for x in (s.pos_lo, s.pos_hi, s.vel_max, s.acc_max))
        gen = torch.Generator(device="cuda"); gen.manual_seed(5)
        base = torch.sign(torch.rand((B, 7), device="cuda", generator=gen) - 0.5)
        worst = np.zeros(3)
        for t in range(T):
            mode = torch.arange(B, device="cuda") % 3
            flip = -1.0 if (t // 7) % 2 else 1.0
            chat = -1.0 if t % 2 else 1.0
            act = base * torch.where(mode == 0, 1.0, torch.where(mode == 1, flip, chat)).unsqueeze(1)
            obs, rew, done, info = env.step(act)
            if t % 8 == 7:
                st = env.get_state()
                worst = np.maximum(worst, [np.abs(-1 + 2 * (st["q"] - lo) / (hi - lo)).max(), (np.abs(st["v"]) / V).max(), (np.abs(st["a"]) / A).max()])
        assert (worst <= 1.001).all(), worst
        assert worst[0] > 0.999 and worst[2] > 0.999                         # the limits were actually reached
        env.close()


@pytest.mark.gpu
def test_braking_layer_reward_terms_against_oracle(built):
    """punish_adaptation, punish_end_min_distance and punish_braking_trajectory_min_distance (rewards.py:315-334,357-375; weights and
    thresholds of trained_networks/human_network/params.json): the two punishment columns and the reward against the oracle, which
    makes the literal get_minimum_distance walk over every sample of the braking trajectory (ctlp.py:3155-3207,3282-3374)."""
    import torch
    from safemotionsrisk_amd import VecSafeMotionsEnv
    from safemotionsrisk_amd.scene import reach_brake_punish_env_config
    from oracle.oracle import OracleEnv
    B, T = 1024, 6
    env = VecSafeMotionsEnv(B, device=0, seed=5, auto_reset=False, **reach_brake_punish_env_config())
    env.reset()
    ora = OracleEnv(env.scene, B, seed=5, num_threads=16)
    ora1 = OracleEnv(env.scene, 1)
    rng = np.random.default_rng(9)
    st = env.get_state()
    st["q"][::2, 1] = rng.uniform(1.2, 1.9, B // 2) * rng.choice([-1, 1], B // 2)
    st["q"][::2, 3] = rng.uniform(-1.0, 1.0, B // 2)
    st["v"][::2, 1] = np.sign(st["q"][::2, 1]) * rng.uniform(0.2, 1.2, B // 2)
    env.set_state(q=st["q"], v=st["v"])
    n_end = n_bt = n_braked = 0
    for t in range(T):
        st = env.get_state()
        ora.set_state(**{k: st[k] for k in env.STATE_KEYS})
        act = rng.uniform(-1, 1, (B, 7)).astype(np.float32)
        obs, rew, done, info = env.step(torch.from_numpy(act).cuda())
        o_obs, o_rew, o_done, o_reason, o_info = ora.step(act.astype(np.float64))
        vals, rew, done = info["values"].cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        diff = vals[:, 8] != o_info[:, 8]
        for e in np.nonzero(diff)[0]:
            assert braking_decision_is_tie(ora1, env.scene, st["q"][e], st["v"][e], st["a"][e], act[e]), (t, e)
        tie = diff | (done != o_done) | (np.abs(o_info[:, :2] - 0.001) < TIE).any(axis=1)
        ok = ~tie
        assert np.abs(vals[ok, 12] - o_info[ok, 12]).max() < 2e-4                      # quadratic of a distance within 5e-6 over 4 cm
        # the trajectory's minimum: float32 rounding may move a distance across the safety distance (-> 1) or the sample set of a tie
        close = np.abs(vals[:, 13] - o_info[:, 13]) < 5e-4
        # the minimum over a whole braking trajectory: the float32 braking law (tolerances of 1e-5 of the limits per knot) moves its
        # samples by up to 1e-4 m (quadratic: 5e-3); beyond that two trajectories have parted at a joint riding on a position limit
        odd = np.nonzero(ok & ~close)[0]
        assert len(odd) <= 0.005 * B
        for e in odd:
            assert abs(vals[e, 13] - o_info[e, 13]) < 5e-3 or braking_decision_is_tie(ora1, env.scene, st["q"][e], st["v"][e], st["a"][e], act[e]), \
                (t, e, vals[e, 13], o_info[e, 13])
        good = ok & close
        assert np.abs(rew[good] - o_rew[good]).max() < 5e-4
        n_end += int((o_info[:, 12] > 0).sum()); n_bt += int(((o_info[:, 13] > 0) & (o_info[:, 13] < 1)).sum()); n_braked += int(o_info[:, 8].sum())
        if done.any():
            env.reset(mask=torch.from_numpy(done).cuda())
    assert n_end > 50 and n_bt > 10 and n_braked > 50
    env.close()


@pytest.mark.gpu
def test_braking_layer_rollout_against_oracle(built):
    """the reaching task with check_braking_trajectory_collisions (README.md:95): 2048 envs x 16 steps with state injection; a third of the
    envs are pushed towards the table so that braking trajectories are executed.  The braking decision (info column 8) and
    done / reason are exact, threshold ties aside; where the decisions agree, observations and rewards agree to rounding; and with the
    layer on no episode of the auto-reset run that follows ends by a collision with the table or the robot itself."""
    import torch
    from oracle.oracle import OracleEnv
    B, T = 2048, 16
    env = _brake_env(B, seed=5)
    env.reset()
    ora = OracleEnv(env.scene, B, seed=5, num_threads=16)
    ora1 = OracleEnv(env.scene, 1)
    rng = np.random.default_rng(9)
    st = env.get_state()
    st["q"][::3, 1] = rng.uniform(1.2, 1.9, len(st["q"][::3])) * rng.choice([-1, 1], len(st["q"][::3]))
    st["q"][::3, 3] = rng.uniform(-1.0, 1.0, len(st["q"][::3]))
    st["v"][::3, 1] = np.sign(st["q"][::3, 1]) * rng.uniform(0.2, 1.2, len(st["q"][::3]))
    env.set_state(q=st["q"], v=st["v"])
    mism = braked = ties = 0
    for t in range(T):
        st = env.get_state()
        ora.set_state(**{k: st[k] for k in env.STATE_KEYS})
        act = rng.uniform(-1, 1, (B, 7)).astype(np.float32)
        obs, rew, done, info = env.step(torch.from_numpy(act).cuda())
        o_obs, o_rew, o_done, o_reason, o_info = ora.step(act.astype(np.float64))
        obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        reason, vals = info["termination_reason"].cpu().numpy(), info["values"].cpu().numpy()
        st1, ost1 = env.get_state(), ora.get_state()
        diff = vals[:, 8] != o_info[:, 8]
        for e in np.nonzero(diff)[0]:
            if braking_decision_is_tie(ora1, env.scene, st["q"][e], st["v"][e], st["a"][e], act[e]):
                ties += 1
            else:
                mism += 1
        bad = (done != o_done) | (reason != o_reason) | (st1["target_count"] != ost1["target_count"])
        if (bad & ~diff).any():
            mism += int((~explain_mismatches(ora1, env.scene, st, st1, bad & ~diff)).sum())
        good = ~bad & ~diff
        braked += int(o_info[:, 8].sum())
        A = np.array(env.scene.struct.acc_max)
        assert np.abs((st1["a"][good] - ost1["a"][good]) / A).max() < 2e-5              # the executed (braking) accelerations
        new = good & (st1["target_count"] != st["target_count"])
        assert (np.abs(obs[good & ~new] - o_obs[good & ~new]) < REACH_OBS_TOL).all()
        assert np.abs(rew[good] - o_rew[good]).max() < 2e-4
        if done.any():
            env.reset(mask=torch.from_numpy(done).cuda())
    assert mism == 0, "%d unexplained mismatches" % mism
    assert braked > 500 and ties <= 10
    env.close()
    env = _brake_env(B, seed=6, auto_reset=True)
    env.reset()
    gen = torch.Generator(device="cuda"); gen.manual_seed(3)
    reasons, n_braked = [], 0
    for t in range(100):
        obs, rew, done, info = env.step(torch.rand((B, 7), device="cuda", generator=gen) * 2 - 1)
        d = done.cpu().numpy().astype(bool)
        reasons += info["termination_reason"].cpu().numpy()[d].tolist()
        n_braked += int(info["values"][:, 8].sum().item())
    env.synchronize()
    assert not set(reasons) & {3, 4} and 5 in reasons and n_braked > 1000
    env.close()


@pytest.mark.gpu
def test_braking_timeout_is_punished_by_the_walked_minimum(built):
    """A braking trajectory that does not come to a standstill within the braking timeout counts as a collision (ctlp.py:3201-3203), but its
    punish_braking_trajectory_min_distance term is the quadratic of the minimum distance over the samples that were walked (ctlp.py:2393-2415,
    3185-3203), not 1.  The timeout is never reached with the shipped limits (every joint stops within 12 steps), so the scene's
    brake_timeout_steps is cut to 2 here - for the kernels and the oracle alike."""
    import torch
    from safemotionsrisk_amd import VecSafeMotionsEnv
    from safemotionsrisk_amd.scene import reach_brake_punish_env_config
    from oracle.oracle import OracleEnv
    B = 1024
    env = VecSafeMotionsEnv(B, device=0, seed=5, auto_reset=False, **reach_brake_punish_env_config())
    env.scene.struct.brake_timeout_steps = 2
    env.set_seed(5)                                      # a fresh handle from the edited scene
    env.reset()
    ora = OracleEnv(env.scene, B, seed=5, num_threads=16)
    rng = np.random.default_rng(3)
    n_timeout = n_partial = 0
    for t in range(4):
        st = env.get_state()
        st["v"] = (rng.uniform(-0.8, 0.8, (B, 7)) * np.array(env.scene.struct.vel_max)).astype(np.float32)      # moving joints do not stop in 3 segments
        env.set_state(v=st["v"])
        ora.set_state(**{k: st[k] for k in env.STATE_KEYS})
        act = rng.uniform(-1, 1, (B, 7)).astype(np.float32)
        obs, rew, done, info = env.step(torch.from_numpy(act).cuda())
        o_obs, o_rew, o_done, o_reason, o_info = ora.step(act.astype(np.float64))
        vals = info["values"].cpu().numpy()
        same = (vals[:, 8] == o_info[:, 8]) & (done.cpu().numpy() == o_done)
        assert same.mean() > 0.98
        close = np.abs(vals[:, 13] - o_info[:, 13]) < 5e-3
        assert (close | ~same).mean() > 0.99
        braked = o_info[:, 8] > 0
        n_timeout += int(braked.sum()); n_partial += int((braked & (o_info[:, 13] < 1.0)).sum())
        if done.any():
            env.reset(mask=done)
    assert n_timeout > 0.5 * 4 * B and n_partial > 100        # most envs time out, and many of them with a punishment below 1
    env.close()


@pytest.mark.gpu
def test_episode_counter_counts_auto_resets_and_link_distances_fold_to_the_static_distance(built):
    """episode_counter (base.py:913-916: one count per reset()) includes the restarts inside step(); env.link_distances() - the per-body
    distances to the static obstacles behind log_obstacle_data (ctlp.py:2639-2660) - have the step's static distance as their minimum,
    and EpisodeMetrics reports their episode minima as obstacles_link_data_*."""
    import torch
    from safemotionsrisk_amd.metrics import EpisodeMetrics
    B, T = 4096, 30
    env = _env(B, seed=2, auto_reset=True)
    env.reset()
    assert env.episode_counter == B
    met = EpisodeMetrics(B, env.INFO_COLUMNS, device=env.device, auto_reset=True)
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    n_done, seen = 0, False
    for t in range(T):
        act = torch.rand((B, 7), device="cuda", generator=gen) * 2 - 1
        act[:, 1] = 1.0; act[:, 3] = -1.0                    # towards the table: static distances inside the query range
        obs, rew, done, info = env.step(act)
        ld = env.link_distances()
        fin = met.update(obs, rew, done, info, link_distances=ld)
        d_static = info["values"][:, 0]
        fold = torch.where(ld.min(dim=0).values < 0.001, torch.zeros_like(d_static), ld.min(dim=0).values)      # rewards.py:115-117
        assert float((fold - d_static).abs().max()) < 1e-6
        n_done += int(done.sum())
        if len(fin):
            key = "obstacles_link_data_table_body_7_closest_point_distance_set_min"
            assert key in fin.metrics and fin.metrics[key].shape[0] == len(fin)
            seen = seen or bool((fin.metrics[key] < 0.102).any())
    assert env.episode_counter == B + n_done and n_done > B and seen
    env.close()
