"""oracle/risk_rollout.py (the CPU restatement of safe_motions_base.py:1498-1659 the GPU tests check against) by itself: properties the
reference's text implies, on a small batch.  No GPU."""
import numpy as np

from oracle.oracle import OracleEnv
from oracle.risk_rollout import backup_action, is_action_risky
from safemotionsrisk_amd import SPACE_ENV_CONFIG, build_scene


def _nets(rng, R):
    w1, w2 = rng.normal(size=(R, 14)).astype(np.float32), rng.normal(size=(R, 1)).astype(np.float32)
    return (lambda x: 1.0 / (1.0 + np.exp(-(x @ w2)))), (lambda x: np.tanh(x @ w1))


def test_background_copy_rollout_properties():
    B, K = 96, 4
    scene = build_scene(SPACE_ENV_CONFIG, auto_reset=False)
    sh_scene = build_scene(dict(SPACE_ENV_CONFIG, trajectory_duration=(K + 1) * 0.1), auto_reset=False)
    assert int(sh_scene.struct.episode_steps) == K + 1                                   # trajectory_length = K + 2, base.py:1819
    main, shadow = OracleEnv(scene, B, seed=3, num_threads=8), OracleEnv(sh_scene, B, seed=3, num_threads=8)
    main.reset()
    rng = np.random.default_rng(0)
    R = scene.obs_size
    cols = np.arange(R)
    risk_fn, backup_fn = _nets(rng, R)
    # a few steps towards the table so that some copies collide
    for _ in range(6):
        a = rng.uniform(-1, 1, (B, 7)); a[:, 1] = 1.0; a[:, 3] = -1.0
        _, _, done, _, _ = main.step(a)
        if done.any():
            main.reset(mask=done)
    before = main.get_state()
    act = rng.uniform(-1, 1, (B, 7)).astype(np.float32); act[:, 1] = 1.0; act[:, 3] = -1.0
    m4 = is_action_risky(main, shadow, act, None, backup_fn, cols, 4, K, 0.5, store_ground_truth=True)
    after = main.get_state()
    assert all(np.array_equal(before[k], after[k]) for k in before)                      # the env itself is not stepped
    assert np.array_equal(m4["risky"], m4["collided"]) and set(np.unique(m4["rating"])) <= {0.0, 1.0}      # base.py:1560-1566
    assert m4["collided"].sum() > 3
    m4_short = is_action_risky(main, shadow, act, None, backup_fn, cols, 4, 1, 0.5)
    assert not (m4_short["collided"] & ~m4["collided"]).any()                          # a longer rollout only adds collisions
    m3 = is_action_risky(main, shadow, act, risk_fn, backup_fn, cols, 3, K, 0.5)
    first_col = m4["trace"][0][3] & (m4["trace"][0][4] != 2)
    assert (m3["rating"][first_col] == 1.0).all()                                        # base.py:1548-1551
    net = risk_fn(m4["state"][:, :R])[:, 0]
    assert np.allclose(m3["rating"][~first_col], net[~first_col], atol=1e-6)            # base.py:1553-1556: the network on the copy's next state
    # ground-truth rows (base.py:1567-1586): state | action | risk | prediction | valid, and the state table of the first simulated step
    sa, s2 = m4["state_action"], m4["state"]
    assert np.array_equal(sa[:, :R], main.observe()) and np.array_equal(sa[:, R:R + 7], act) and np.array_equal(sa[:, R + 7], m4["collided"].astype(np.float32))
    assert np.array_equal(s2[:, R + 1], (~m4["first_done"]).astype(np.float32)) and (sa[:, R + 9] == 1.0).all()
    # the copy's episode ends by its length at step K + 1, not before (base.py:1819-1823): such an end is not a collision
    last = m4["trace"][-1]
    assert (last[3] | ~last[2]).all() and not (last[2] & (last[4] == 2) & m4["collided"] & ~np.any([tr[2] & tr[3] & (tr[4] != 2) for tr in m4["trace"]], axis=0)).any()
    # risk_ignore_estimation_probability (base.py:1652-1655)
    none = is_action_risky(main, shadow, act, None, backup_fn, cols, 4, K, 0.5, ignore_probability=1.0, seed=3)
    assert not none["risky"].any()
    some = is_action_risky(main, shadow, np.tile(act, (1, 1)), None, backup_fn, cols, 4, K, 0.5, ignore_probability=0.5, seed=3)
    assert not (some["risky"] & ~m4["risky"]).any()
    assert np.array_equal(backup_action(backup_fn, sa[:, :R]), np.clip(backup_fn(sa[:, :R])[:, :7], -1, 1).astype(np.float32))
