"""CPU restatement of SafeMotionsBase._is_action_risky (safe_motions_base.py:1498-1659) for a batch - TEST INFRASTRUCTURE ONLY.

Only tests/ import this module (like oracle.py).  It drives two OracleEnv objects the way the reference drives its main and its
backup physics client: `main` is the env, `shadow` the background copy (same scene, episodes of backup_steps + 1 steps -
trajectory_length = backup_steps + 2, base.py:1819 -, no auto-reset).  The networks are plain callables on float32 numpy arrays.
PARITY UNPINNED like the rest of the oracle: the reference ships no vectors for this path; the restatement follows its text.
"""
import numpy as np

from . import oracle as _o

TERMINATION_TRAJECTORY_LENGTH = 2       # base.py:66
S_RISK_IGNORE = 15                      # csrc/smr_hip.hip: the Philox stream of risk_ignore_estimation_probability


def backup_action(backup_fn, risk_obs):
    """_get_action_from_backup_action of the deterministic backup agent: its first 7 outputs, clipped (actions.py:327-335)"""
    return np.clip(backup_fn(risk_obs.astype(np.float32))[:, :7], -1.0, 1.0).astype(np.float32)


def rollout(main_state, shadow, first_actions, backup_fn, cols, backup_steps, robs_now):
    """switch_to_backup_client ... switch_back_to_main_client (base.py:1527-1562 / 1608-1643): step the copy with `first_actions`
    (None: backup actions only, base.py:1616-1618) and then `backup_steps` actions of the backup policy.  Returns
    (collided [B] bool - the copy ended in a termination other than the trajectory length; first_done [B] - its first step ended
    it; first_collided [B]; first_state [B, R] float32 - the copy's risk observation after its first step; per-step states for
    the tie analysis of the tests)."""
    B = shadow.B
    st = dict(main_state)
    st["step_count"] = np.zeros(B, np.int32)                # _episode_length = 0, base.py:1823
    shadow.set_state(**st)
    alive = np.ones(B, bool)
    collided, first_done, first_collided = np.zeros(B, bool), np.zeros(B, bool), np.zeros(B, bool)
    first_state = np.zeros((B, len(cols)), np.float32)
    src = robs_now
    n_steps = (0 if first_actions is None else 1) + backup_steps
    trace = []
    for i in range(n_steps):
        act = first_actions if (i == 0 and first_actions is not None) else backup_action(backup_fn, src)
        s0 = shadow.get_state()
        obs, _, done, reason, info = shadow.step(act.astype(np.float64))
        trace.append((s0, shadow.get_state(), alive.copy(), done.astype(bool), reason.copy()))
        ended = alive & done.astype(bool)
        col = ended & (reason != TERMINATION_TRAJECTORY_LENGTH)           # base.py:1548-1551 / 1639-1641
        collided |= col
        if i == 0:
            first_done, first_collided = ended.copy(), col.copy()
            first_state = obs[:, cols].astype(np.float32)
        alive &= ~done.astype(bool)
        src = obs[:, cols].astype(np.float32)
    return collided, first_done, first_collided, first_state, trace


def is_action_risky(main, shadow, actions, risk_fn, backup_fn, cols, mode, backup_steps, threshold, store_ground_truth=False,
                    ignore_probability=0.0, seed=0, env_index_base=0):
    """The decision of one step for every env of `main` (which is not stepped).  mode: -1 state-action network, 0 .. 4
    risk_state_config.  Returns dict(risky, rating, state_action rows, state rows) with the row layout of smr_set_risk_rollout."""
    B, R = main.B, len(cols)
    actions = np.asarray(actions, np.float32)
    st = main.get_state()
    robs_now = main.observe()[:, cols].astype(np.float32)
    simulate = mode in (3, 4)
    rating = np.zeros(B, np.float32)
    collided = first_done = first_collided = None
    first_state = np.zeros((B, R), np.float32)
    trace = []
    if simulate or store_ground_truth:
        first = None if (store_ground_truth and not simulate and mode == 0) else actions          # base.py:1608-1618
        K = backup_steps if (store_ground_truth or mode == 4) else 0
        collided, first_done, first_collided, first_state, trace = rollout(st, shadow, first, backup_fn, cols, K, robs_now)
    if mode == 4:                                            # base.py:1560-1566: 1.0 / 0.0, no network
        rating = collided.astype(np.float32)
    elif mode == 3:                                          # base.py:1548-1556: a collision of the simulated step, else the network on the next state
        net = risk_fn(first_state)[:, 0].astype(np.float32)
        rating = np.where(first_collided, 1.0, np.where(first_done, 0.0, net)).astype(np.float32)
    elif mode == -1:
        rating = risk_fn(np.concatenate([robs_now, actions], axis=1))[:, 0].astype(np.float32)
    elif mode == 0:
        rating = risk_fn(robs_now)[:, 0].astype(np.float32)
    else:
        raise NotImplementedError("forecasting modes 1, 2: tests/test_gpu_policy.py::test_state_risk_network_modes builds their input")
    risky = rating >= np.float32(threshold)                  # base.py:1602-1603
    if ignore_probability > 0.0:                             # base.py:1652-1655 (the draw: csrc/smr_hip.hip risk_apply_kernel)
        for e in np.nonzero(risky)[0]:
            w = _o.philox_word(seed, env_index_base + int(e), S_RISK_IGNORE, int(st["reset_count"][e]), int(st["step_count"][e]))
            if np.float32(w >> 8) * np.float32(1.0 / 16777216.0) < np.float32(ignore_probability):
                risky[e] = False
    out = dict(risky=risky, rating=rating, collided=collided, first_done=first_done, trace=trace)
    if store_ground_truth:
        label = collided.astype(np.float32)
        next_state = mode in (1, 2, 3)
        sa = np.zeros((B, R + 7 + 3), np.float32)
        sa[:, :R] = first_state if next_state else robs_now
        sa[:, R:R + 7] = actions
        sa[:, R + 7] = label
        sa[:, R + 8] = rating
        sa[:, R + 9] = (~first_done).astype(np.float32) if next_state else 1.0
        out["state_action"] = sa
        if mode == 4:
            s2 = np.zeros((B, R + 2), np.float32)
            s2[:, :R] = first_state
            s2[:, R] = label
            s2[:, R + 1] = (~first_done).astype(np.float32)
            out["state"] = s2
    return out
