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
    from safemotionsrisk_amd.lib import SmrError
    with pytest.raises(SmrError):                                # the modes that step a background copy of the env are refused
        env.set_risk_networks(risk, backup, risk_state_config=3)
    env.close(); shadow.close()


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
