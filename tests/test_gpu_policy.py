"""The MFMA network kernel against a plain float32 numpy reference of the same layers, and risk-aware stepping (SURVEY section 8f row 4:
actions.py:303-340, safe_motions_base.py:1498-1502,1598-1603) against the same decision made on the host and the float64 oracle's step.
Weights are synthetic (seeded): the reference's checkpoints are pickle files / SavedModels and are not loaded."""
import numpy as np
import pytest

from parity_util import explain_mismatches

pytestmark = pytest.mark.gpu


def _weights(rng, sizes, scale=1.0):
    ws = [(rng.normal(size=(a, b)) * scale / np.sqrt(a)).astype(np.float32) for a, b in zip(sizes[:-1], sizes[1:])]
    bs = [(rng.normal(size=b) * 0.1).astype(np.float32) for b in sizes[1:]]
    return ws, bs


@pytest.mark.parametrize("sizes,acts", [([23, 256, 128, 14], ["swish", "swish", "tanh"]),          # backup policy (shipped model config)
                                         ([30, 256, 128, 1], ["swish", "swish", "sigmoid"]),       # state-action risk network
                                         ([29, 64, 7], ["selu", "linear"]), ([5, 3], ["relu"]),
                                         ([256, 256, 256, 200, 17], ["tanh", "relu", "swish", "linear"])])
def test_mlp_kernel_against_numpy(built, sizes, acts):
    import torch
    from safemotionsrisk_amd import Mlp
    rng = np.random.default_rng(len(sizes) * 100 + sizes[0])
    ws, bs = _weights(rng, sizes)
    net = Mlp(ws, bs, acts)
    for rows, pad in ((1, 0), (16, 0), (1000, 3), (65536, 0)):
        x = rng.uniform(-1, 1, (rows, sizes[0] + pad)).astype(np.float32)
        y = net(torch.from_numpy(x).cuda()).cpu().numpy()
        ref = net.reference(x)
        assert y.shape == ref.shape == (rows, sizes[-1])
        assert np.abs(y - ref).max() < 2e-5 * max(1.0, np.abs(ref).max()), (rows, np.abs(y - ref).max())
    # exact integer data: every product and sum is exact in float32, so any operand-layout error shows as a wrong integer
    wi = [rng.integers(-2, 3, w.shape).astype(np.float32) for w in ws[:1]]
    net_i = Mlp(wi, [np.zeros(sizes[1], np.float32)], ["linear"])
    xi = rng.integers(-3, 4, (77, sizes[0])).astype(np.float32)
    assert np.array_equal(net_i(torch.from_numpy(xi).cuda()).cpu().numpy(), xi @ wi[0])
    net.close(); net_i.close()


def test_mlp_rejects_bad_shapes(built):
    from safemotionsrisk_amd import Mlp
    from safemotionsrisk_amd.lib import SmrError
    with pytest.raises(SmrError):
        Mlp([np.zeros((23, 300), np.float32)], [np.zeros(300, np.float32)], ["tanh"])             # wider than 256
    with pytest.raises(ValueError):
        Mlp([np.zeros((23, 64), np.float32), np.zeros((32, 7), np.float32)], [np.zeros(64, np.float32), np.zeros(7, np.float32)], ["tanh", "tanh"])


@pytest.mark.parametrize("config", ["reach", "space"])
def test_risk_aware_stepping(built, config):
    """the task policy's action is replaced by the backup policy's where the risk network rates (risk observation, action) at or above the
    threshold: the env's decision and the executed action against the same networks evaluated on the host (numpy float32) from the
    env's observation without its target-point columns, and the step that follows against the oracle fed with that executed action."""
    import torch
    from oracle.oracle import OracleEnv
    from safemotionsrisk_amd import REACH_ENV_CONFIG, SPACE_ENV_CONFIG, VecSafeMotionsEnv, backup_policy, risk_network
    cfg = REACH_ENV_CONFIG if config == "reach" else SPACE_ENV_CONFIG
    B, T, thr = 2048, 10, 0.5
    env = VecSafeMotionsEnv(B, device=0, seed=4, auto_reset=False, **cfg)
    robs = env.risk_observation_size
    assert robs == 23 and env.observation_size == (29 if config == "reach" else 23)
    rng = np.random.default_rng(7)
    risk = risk_network(*_weights(rng, [robs + 7, 256, 128, 1], scale=2.0))
    backup = backup_policy(*_weights(rng, [robs, 256, 128, 14]))
    env.set_risk_networks(risk, backup, risk_threshold=thr)
    obs = env.reset().cpu().numpy()
    ora = OracleEnv(env.scene, B, seed=4, num_threads=16)
    keys = env.STATE_KEYS
    cols = np.r_[0:21, (27 if config == "reach" else 21):env.observation_size]       # obs.py:419-431: no target-point columns
    n_risky = mism = 0
    for t in range(T):
        st = env.get_state()
        ora.set_state(**{k: st[k] for k in keys})
        robs_host = env.observe().cpu().numpy()[:, cols]
        act = rng.uniform(-1, 1, (B, 7)).astype(np.float32)
        r = risk.reference(np.concatenate([robs_host, act], axis=1))[:, 0]
        risky = r >= thr
        executed = np.where(risky[:, None], np.clip(backup.reference(robs_host)[:, :7], -1, 1), act)
        o, rew, done, info = env.step(torch.from_numpy(act).cuda())
        vals = info["values"].cpu().numpy()
        tie = np.abs(r - thr) < 1e-4
        assert np.array_equal(vals[~tie, 11], risky[~tie].astype(np.float32))
        n_risky += int(risky.sum())
        o_obs, o_rew, o_done, o_reason, o_info = ora.step(executed.astype(np.float64))
        st1, ost1 = env.get_state(), ora.get_state()
        A = np.array(env.scene.struct.acc_max)
        ok = ~tie
        assert np.abs((st1["a"][ok] - ost1["a"][ok]) / A).max() < 1e-4                  # the executed action, through the safe range
        bad = ((done.cpu().numpy() != o_done) | (info["termination_reason"].cpu().numpy() != o_reason)) & ok
        if bad.any():
            mism += int((~explain_mismatches(ora, env.scene, st, st1, bad, tie=1e-4)).sum())
        # the action punishment is the ORIGINAL action's (safe_motions_base.py:1066, rewards.py:164-169)
        pun = np.clip((np.abs(act).max(axis=1) - 0.95) / 0.05, 0, 1) ** 2
        assert np.abs(vals[:, 3] - pun).max() < 1e-5
        if done.any():
            env.reset(mask=done)
    assert mism == 0 and 0.05 * B * T < n_risky < 0.95 * B * T
    env.set_risk_networks(None, None)
    o, rew, done, info = env.step(torch.zeros((B, 7), device="cuda"))
    assert float(info["values"][:, 11].sum()) == 0.0
    env.close(); risk.close(); backup.close()


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_state_risk_network_modes(built, mode):
    """A risk network over the risk observation alone (risk_config.json "action_size": null) rates a STATE (safe_motions_base.py:1503-1517):
    0 the current one, 1 the next one with its joint columns forecast for the requested action (obs.py:192-231), 2 with the planets one
    step on as well.  The env's decision against the same network on the host, fed with the observation the env itself reports - for
    the forecasts: the joint columns of the NEXT observation the un-replaced action leads to (an env stepped in parallel without the
    risk check), the planet columns of now (1) / of then (2)."""
    import torch
    from safemotionsrisk_amd import SPACE_ENV_CONFIG, VecSafeMotionsEnv, backup_policy, risk_network
    B, T, thr = 1024, 6, 0.5
    env = VecSafeMotionsEnv(B, device=0, seed=4, auto_reset=False, **SPACE_ENV_CONFIG)
    shadow = VecSafeMotionsEnv(B, device=0, seed=4, auto_reset=False, **SPACE_ENV_CONFIG)      # the same step without the risk check
    robs = env.risk_observation_size
    rng = np.random.default_rng(11 + mode)
    risk = risk_network(*_weights(rng, [robs, 256, 128, 1], scale=3.0))
    backup = backup_policy(*_weights(rng, [robs, 256, 128, 14]))
    env.set_risk_networks(risk, backup, risk_threshold=thr, risk_state_config=mode)
    env.reset()
    n_risky = 0
    for t in range(T):
        st = env.get_state()
        shadow.set_state(**{k: st[k] for k in env.STATE_KEYS})
        now = env.observe().cpu().numpy().copy()
        act = rng.uniform(-1, 1, (B, 7)).astype(np.float32)
        nxt = shadow.step(torch.from_numpy(act).cuda())[0].cpu().numpy().copy()
        x = now.copy()
        if mode >= 1:
            x[:, :21] = nxt[:, :21]
        if mode == 2:
            x[:, 21:] = nxt[:, 21:]
        r = risk.reference(x)[:, 0]
        risky = r >= thr
        executed = np.where(risky[:, None], np.clip(backup.reference(now)[:, :7], -1, 1), act)
        _, _, done, info = env.step(torch.from_numpy(act).cuda())
        tie = np.abs(r - thr) < 2e-4
        assert np.array_equal(info["values"][:, 11].cpu().numpy()[~tie], risky[~tie].astype(np.float32))
        # the executed action: the end acceleration the env took equals the one the shadow env takes for `executed`
        shadow.set_state(**{k: st[k] for k in env.STATE_KEYS})
        shadow.step(torch.from_numpy(executed.astype(np.float32)).cuda())
        A = np.array(env.scene.struct.acc_max)
        assert np.abs((env.get_state()["a"] - shadow.get_state()["a"])[~tie] / A).max() < 1e-4
        n_risky += int(risky.sum())
        if done.any():
            env.reset(mask=done)
    assert 0.05 * B * T < n_risky < 0.95 * B * T
    env.close(); shadow.close()


def _explained(ora1, scene, trace, e, tie):
    """a label of env e that differs between the copy on the GPU and the oracle's copy: is one of the decisions of the oracle's
    rollout of that env (while it was alive) a threshold tie?  The later steps of a rollout start from float32 / float64 states
    that have drifted apart, so their ties are judged with a wider band."""
    for i, (s0, s1, alive, done, reason) in enumerate(trace):
        if not alive[e]:
            break
        one = np.zeros(len(alive), bool); one[e] = True
        if explain_mismatches(ora1, scene, s0, s1, one, tie=tie if i == 0 else 20 * tie)[0]:
            return True
    return False


@pytest.mark.parametrize("mode,config", [(3, "space"), (4, "space"), (4, "ball"), (4, "reach")])
def test_risk_modes_that_step_a_background_copy(built, mode, config):
    """RISK_CHECK_NEXT_STATE_SIMULATE_NEXT_STEP (3) and ..._AND_BACKUP_TRAJECTORY (4), safe_motions_base.py:1518-1577: the env copies its
    state into a second handle, steps that with the requested action (3: the network then rates the copy's next risk observation unless the
    step collided) and, mode 4, with `backup_steps` more actions of the backup policy - risky if the copy ends in a collision.  Against
    oracle/risk_rollout.py, which drives two OracleEnv objects the same way with the same networks in numpy: the decision (info column
    risky_action), the executed action and the ground-truth rows; a label that differs must sit on a threshold tie of the oracle's rollout.
    The env itself is not touched by the copy: its step equals the step of an env without the rollout fed with the executed action."""
    import torch
    from oracle.oracle import OracleEnv
    from oracle.risk_rollout import is_action_risky
    from parity_util import TIE
    from safemotionsrisk_amd import BALL_ENV_CONFIG, REACH_ENV_CONFIG, SPACE_ENV_CONFIG, VecSafeMotionsEnv, backup_policy, build_scene, risk_network
    cfg = dict(space=SPACE_ENV_CONFIG, ball=BALL_ENV_CONFIG, reach=REACH_ENV_CONFIG)[config]
    B, T, K, thr = 1024, 5, 3, 0.5
    env = VecSafeMotionsEnv(B, device=0, seed=6, auto_reset=False, **cfg)
    plain = VecSafeMotionsEnv(B, device=0, seed=6, auto_reset=False, **cfg)            # the same env without risk-aware stepping
    R = env.risk_observation_size
    cols = np.r_[0:21, (env.observation_size - (R - 21)):env.observation_size]          # obs.py:419-431: no target-point columns
    rng = np.random.default_rng(20 + mode)
    risk = risk_network(*_weights(rng, [R, 256, 128, 1], scale=3.0)) if mode == 3 else None
    backup = backup_policy(*_weights(rng, [R, 256, 128, 14], scale=2.0))
    env.set_risk_networks(risk, backup, risk_threshold=thr, risk_state_config=mode)
    env.set_risk_rollout(backup_steps=K, store_ground_truth=True)
    assert env.risk_ground_truth["state_action"].shape == (B, R + 10) and env.risk_ground_truth["state"].shape == (B, R + 2)
    env.reset()
    ora = OracleEnv(env.scene, B, seed=6, num_threads=16)
    ora1 = OracleEnv(env.scene, 1)
    sh_scene = build_scene(dict(env.scene.env_config, trajectory_duration=(K + 1) * float(env.scene.struct.dt)), auto_reset=False)
    ora_sh = OracleEnv(sh_scene, B, seed=6, num_threads=16)
    keys = env.STATE_KEYS
    n_risky = n_label = unexplained = 0
    for t in range(T):
        st = env.get_state()
        ora.set_state(**{k: st[k] for k in keys})
        act = rng.uniform(-1, 1, (B, 7)).astype(np.float32)
        if t % 2:                                             # half of the steps push towards the table so that copies collide
            act[: B // 2, 1] = 1.0; act[: B // 2, 3] = -1.0
        want = is_action_risky(ora, ora_sh, act, None if risk is None else risk.reference, backup.reference, cols, mode, K, thr, store_ground_truth=True, seed=6)
        plain.set_state(**{k: st[k] for k in keys})
        _, _, done, info = env.step(torch.from_numpy(act).cuda())
        got = info["values"][:, 11].cpu().numpy() > 0.5
        gt_sa, gt_s = env.risk_ground_truth["state_action"].cpu().numpy(), env.risk_ground_truth["state"].cpu().numpy()
        # labels: exact, threshold ties of the oracle's rollout aside
        diff = gt_sa[:, R + 7] != want["state_action"][:, R + 7]
        for e in np.nonzero(diff)[0]:
            unexplained += 0 if _explained(ora1, sh_scene, want["trace"], e, TIE) else 1
        same = ~diff
        assert diff.sum() <= 0.01 * B
        near = np.abs(want["rating"] - thr) < 2e-4           # (mode 3: the network's rating on the threshold)
        first_same = gt_sa[:, R + 9] == want["state_action"][:, R + 9]
        ok = same & ~near & first_same
        assert np.array_equal(got[ok], want["risky"][ok])
        assert np.abs(gt_sa[ok][:, :R + 7] - want["state_action"][ok][:, :R + 7]).max() < 3e-5           # state (current, or mode 3: the copy's next one) | action
        assert np.array_equal(gt_sa[ok][:, R + 9], want["state_action"][ok][:, R + 9])
        assert np.abs(gt_sa[ok][:, R + 8] - want["state_action"][ok][:, R + 8]).max() < 2e-4              # the rating
        if mode == 4:
            v = ok & (gt_s[:, R + 1] > 0.5)
            assert np.array_equal(gt_s[ok][:, R + 1], want["state"][ok][:, R + 1]) and np.array_equal(gt_s[ok][:, R], want["state"][ok][:, R])
            assert np.abs(gt_s[v][:, :R] - want["state"][v][:, :R]).max() < 3e-5
        # the env's own step: that of an env without the copy, fed with the action the decision leaves
        robs_now = ora.observe()[:, cols]
        executed = np.where(got[:, None], np.clip(backup.reference(robs_now.astype(np.float32))[:, :7], -1, 1), act).astype(np.float32)
        p_obs, p_rew, p_done, _ = plain.step(torch.from_numpy(executed).cuda())
        A = np.array(env.scene.struct.acc_max)
        s1, p1 = env.get_state(), plain.get_state()
        assert np.abs((s1["a"] - p1["a"]) / A).max() < 1e-4 and np.abs(s1["q"] - p1["q"]).max() < 1e-6
        assert np.array_equal(done.cpu().numpy(), p_done.cpu().numpy()) or np.abs((s1["a"] - p1["a"]) / A).max() > 0
        n_risky += int(got.sum()); n_label += int(want["state_action"][:, R + 7].sum())
        if done.any():
            env.reset(mask=done)
    assert unexplained == 0, "%d labels differ without a threshold tie" % unexplained
    assert n_label > 20 and 0.01 * B * T < n_risky < 0.99 * B * T
    # risk_ignore_estimation_probability (safe_motions_base.py:1652-1655): with 1.0 no rating is acted on, with 0.35 about a third is dropped
    env.set_risk_rollout(backup_steps=K, ignore_probability=1.0)
    _, _, _, info = env.step(torch.from_numpy(act).cuda())
    assert float(info["values"][:, 11].sum()) == 0.0
    env.close(); plain.close()
    if risk is not None:
        risk.close()
    backup.close()


def test_ground_truth_of_network_modes_and_ignore_probability(built):
    """risk_store_ground_truth with a risk NETWORK (safe_motions_base.py:1605-1650): the decision is the network's, the copy only labels it -
    stepped with the action and `backup_steps` backup actions for a state-action network, with backup actions alone for a current-state
    network.  And risk_ignore_estimation_probability = 0.35 (README.md:131): the env's draws are the oracle's (Philox stream 15)."""
    import torch
    from oracle.oracle import OracleEnv
    from oracle.risk_rollout import is_action_risky
    from safemotionsrisk_amd import SPACE_ENV_CONFIG, VecSafeMotionsEnv, backup_policy, build_scene, risk_network
    B, K, thr = 1024, 3, 0.5
    for mode in (-1, 0):
        env = VecSafeMotionsEnv(B, device=0, seed=9, auto_reset=False, **SPACE_ENV_CONFIG)
        R = env.risk_observation_size
        cols = np.arange(R)
        rng = np.random.default_rng(31 + mode)
        risk = risk_network(*_weights(rng, [R + (7 if mode == -1 else 0), 256, 128, 1], scale=3.0))
        backup = backup_policy(*_weights(rng, [R, 256, 128, 14], scale=2.0))
        env.set_risk_networks(risk, backup, risk_threshold=thr, risk_state_config=0)
        env.set_risk_rollout(backup_steps=K, store_ground_truth=True, ignore_probability=0.35)
        env.reset()
        ora = OracleEnv(env.scene, B, seed=9, num_threads=16)
        ora_sh = OracleEnv(build_scene(dict(env.scene.env_config, trajectory_duration=(K + 1) * 0.1), auto_reset=False), B, seed=9, num_threads=16)
        dropped = kept = 0
        for t in range(3):
            st = env.get_state()
            ora.set_state(**{k: st[k] for k in env.STATE_KEYS})
            act = rng.uniform(-1, 1, (B, 7)).astype(np.float32)
            act[: B // 2, 1] = 1.0; act[: B // 2, 3] = -1.0
            want = is_action_risky(ora, ora_sh, act, risk.reference, backup.reference, cols, mode, K, thr, store_ground_truth=True, ignore_probability=0.35, seed=9)
            _, _, done, info = env.step(torch.from_numpy(act).cuda())
            got = info["values"][:, 11].cpu().numpy() > 0.5
            gt = env.risk_ground_truth["state_action"].cpu().numpy()
            near = np.abs(want["rating"] - thr) < 2e-4
            assert np.array_equal(got[~near], want["risky"][~near])
            agree = gt[:, R + 7] == want["state_action"][:, R + 7]
            assert agree.mean() > 0.99
            assert np.abs(gt[:, :R + 7] - want["state_action"][:, :R + 7]).max() < 3e-5 and (gt[:, R + 9] == 1.0).all()
            assert np.abs(gt[:, R + 8] - want["rating"]).max() < 2e-4
            rated = want["rating"] >= thr
            dropped += int((rated & ~want["risky"]).sum()); kept += int((rated & want["risky"]).sum())
            if done.any():
                env.reset(mask=done)
        assert 0.25 < dropped / max(dropped + kept, 1) < 0.45 and dropped + kept > 200
        env.close(); risk.close(); backup.close()


def test_env_config_with_a_risk_network_waits_for_its_weights(built, tmp_path):
    """risk_config_dir in the env_config (the shipped task params.json): the two JSON files are read, the env is built in the backup
    policy's scene, and step() refuses to run until set_risk_networks() has supplied the weights - with the network shape that
    risk_config.json describes, the threshold and mode of the env_config."""
    import json
    import torch
    from safemotionsrisk_amd import REACH_ENV_CONFIG, SPACE_ENV_CONFIG, VecSafeMotionsEnv, backup_policy, risk_network
    root = tmp_path / "trained_networks"
    (root / "risk_networks" / "state_action" / "space").mkdir(parents=True)
    (root / "backup_networks" / "space" / "checkpoint").mkdir(parents=True)
    (root / "risk_networks" / "state_action" / "space" / "risk_config.json").write_text(json.dumps(
        dict(action_size=7, checkpoint="backup_networks/space/checkpoint/checkpoint", observation_size=23)))
    (root / "backup_networks" / "space" / "params.json").write_text(json.dumps(dict(env_config=dict(SPACE_ENV_CONFIG, ball_machine_mode=True))))
    cfg = dict(REACH_ENV_CONFIG, risk_config_dir="risk_networks/state_action/space", risk_threshold=0.3, risk_state_config=0)
    env = VecSafeMotionsEnv(256, device=0, seed=1, auto_reset=True, trained_networks_dir=str(root), **cfg)
    assert env.risk_config["risk_threshold"] == 0.3 and env.risk_observation_size == 23
    env.reset()
    act = torch.zeros((256, 7), device="cuda")
    with pytest.raises(RuntimeError, match="set_risk_networks"):
        env.step(act)
    rng = np.random.default_rng(0)
    with pytest.raises(ValueError):                               # risk_config.json says state-action (30 inputs)
        env.set_risk_networks(risk_network(*_weights(rng, [23, 64, 1])), backup_policy(*_weights(rng, [23, 64, 14])))
    env.set_risk_networks(risk_network(*_weights(rng, [30, 256, 128, 1], scale=2.0)), backup_policy(*_weights(rng, [23, 256, 128, 14])))
    assert env._risk_threshold == 0.3
    risky = 0
    for _ in range(5):
        _, _, _, info = env.step(torch.from_numpy(rng.uniform(-1, 1, (256, 7)).astype(np.float32)).cuda())
        risky += int(info["values"][:, 11].sum())
    env.synchronize()
    assert 0 < risky < 5 * 256
    env.close()
