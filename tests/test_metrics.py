"""Episode metrics and the RLlib-style vector interface (host logic; CPU tensors and a fake env stand in for the GPU env)."""
import os
import numpy as np
import pytest
import torch

from safemotionsrisk_amd.env import VecSafeMotionsEnv
from safemotionsrisk_amd.metrics import EpisodeMetrics, RllibVectorEnv, step_scalar_names

COLS = VecSafeMotionsEnv.INFO_COLUMNS


def _random_rollout(B, T, rng, p_done=0.15):
    obs = rng.uniform(-1, 1, (T, B, 23)).astype(np.float32)
    rew = rng.uniform(0, 5, (T, B)).astype(np.float32)
    vals = rng.uniform(0, 1, (T, B, len(COLS))).astype(np.float32)
    vals[..., 8] = rng.random((T, B)) < 0.2                           # braking trajectory executed
    vals[..., 9] = rng.random((T, B)) < 0.1                           # target point reached
    vals[..., 10] = rng.choice([0, 0, 0, 1, 2, 3], (T, B))            # final state of a retired ball
    vals[..., 0][rng.random((T, B)) < 0.1] = 0.0                      # static collisions
    done = rng.random((T, B)) < p_done
    reason = np.where(done, rng.choice([1, 2, 4, 5], (T, B)), -1).astype(np.int8)
    length = np.zeros((T, B), np.float32)
    run = np.zeros(B)
    for t in range(T):
        run += 1
        length[t] = run
        run[done[t]] = 0
    vals[..., 7] = length
    return obs, rew, vals, done, reason


@pytest.mark.parametrize("auto_reset", [False, True])
def test_episode_metrics_match_a_per_env_python_fold(auto_reset):
    """the same fold CustomTrainCallbacks does per episode (train.py:47-79): append per step, then average / max / min"""
    rng = np.random.default_rng(0)
    B, T = 16, 60
    obs, rew, vals, done, reason = _random_rollout(B, T, rng)
    m = EpisodeMetrics(B, COLS, auto_reset=auto_reset, trajectory_length=21)
    names = step_scalar_names(COLS)
    assert len(names) == len(set(names)) and "joint_3_vel_abs" in names and "collision_rate_static_obstacles" in names
    hist = [[] for _ in range(B)]
    ev_hist = [[] for _ in range(B)]
    checked = 0
    for t in range(T):
        for e in range(B):
            ev_hist[e].append(vals[t, e, 8:11].copy())
        info = {"values": torch.from_numpy(vals[t]), "termination_reason": torch.from_numpy(reason[t])}
        fin = m.update(torch.from_numpy(obs[t]), torch.from_numpy(rew[t]), torch.from_numpy(done[t]), info)
        x = m.step_scalars(torch.from_numpy(obs[t]), torch.from_numpy(rew[t]), info["values"]).numpy()
        for e in range(B):
            hist[e].append((x[e], bool(done[t, e])))
        assert sorted(fin.env_index.tolist()) == sorted(np.nonzero(done[t])[0].tolist())
        for i, e in enumerate(fin.env_index.tolist()):
            rows = np.array([h[0] for h in hist[e]])
            obs_rows = rows[:-1] if auto_reset else rows            # auto_reset: the terminal obs belongs to the next episode
            for k, name in enumerate(names):
                r = obs_rows if k < m.n_obs else rows
                if len(r) == 0:
                    continue
                assert fin.metrics[name + "_average"][i].item() == pytest.approx(r[:, k].mean(), abs=1e-5)
                assert fin.metrics[name + "_max"][i].item() == pytest.approx(r[:, k].max(), abs=1e-6)
                assert fin.metrics[name + "_min"][i].item() == pytest.approx(r[:, k].min(), abs=1e-6)
            assert fin.episode_length[i].item() == len(rows) and fin.metrics["trajectory_length"][i].item() == 21
            assert fin.metrics["success_rate"][i].item() == (0.0 if reason[t, e] == 1 else 1.0)
            assert fin.metrics["collision_static_obstacles_termination_rate"][i].item() == float(reason[t, e] == 4)
            # episode-end statistics of the obstacle wrapper (ctlp.py:1141-1309) from the event columns of the episode's steps
            ev = np.array(ev_hist[e])
            assert fin.metrics["obstacles_time_influenced_by_braking_trajectory_collision"][i].item() == pytest.approx(ev[:, 0].mean(), abs=1e-6)
            assert fin.metrics["obstacles_episodes_without_influence_by_braking_trajectory"][i].item() == float(ev[:, 0].sum() == 0)
            assert fin.metrics["obstacles_num_target_points_reached"][i].item() == ev[:, 1].sum()
            assert fin.metrics["moving_object_hit_robot_total"][i].item() == (ev[:, 2] == 1).sum()
            assert fin.metrics["moving_object_missed_robot_total"][i].item() == (ev[:, 2] == 3).sum()
            n_final = (ev[:, 2] > 0).sum()
            frac = fin.metrics["moving_object_hit_obstacle_fraction"][i].item()
            assert (np.isnan(frac) and n_final == 0) or frac == pytest.approx((ev[:, 2] == 2).sum() / max(n_final, 1), abs=1e-6)
            hist[e] = []
            ev_hist[e] = []
            checked += 1
        if len(fin):
            assert set(fin.mean()) == set(fin.metrics)
    assert checked > 50


class _FakeEnv:
    """stands in for VecSafeMotionsEnv(auto_reset=False): deterministic tensors, episodes of 3 steps"""
    INFO_COLUMNS = COLS

    class _S:
        class struct:
            auto_reset = 0
            episode_steps = 20
    scene = _S()

    def __init__(self, B):
        from safemotionsrisk_amd.env import Box
        self.num_envs, self.device = B, None
        self.observation_space, self.action_space = Box(-1, 1, (23,)), Box(-1, 1, (7,))
        self.k = torch.zeros(B)

    def reset(self, mask=None):
        if mask is None:
            self.k[:] = 0
        else:
            self.k[mask.bool()] = 0
        return self.k.unsqueeze(1).repeat(1, 23) * 0.01

    def step(self, action):
        self.k += 1
        done = self.k >= 3
        vals = torch.zeros(self.num_envs, len(COLS))
        vals[:, 7] = self.k
        vals[:, 2] = 0.5
        info = {"values": vals, "termination_reason": torch.where(done, 2, -1).to(torch.int8)}
        return self.k.unsqueeze(1).repeat(1, 23) * 0.01, torch.ones(self.num_envs), done, info


def test_rllib_vector_interface_shapes_and_episode_end_info():
    v = RllibVectorEnv(_FakeEnv(4))
    obs = v.vector_reset()
    assert len(obs) == 4 and obs[0].shape == (23,)
    for step in range(3):
        obs, rew, done, infos = v.vector_step(np.zeros((4, 7), np.float32))
        assert len(obs) == len(rew) == len(done) == len(infos) == 4
        assert set(infos[0]) >= {"average", "max", "min"} and infos[0]["average"]["minimum_distance_to_moving_obstacles"] == 0.5
    assert all(done) and infos[1]["episode_length"] == 3 and infos[1]["trajectory_length"] == 21
    assert infos[1]["termination_reason"] == 2 and infos[1]["trajectory_successful"] == 1.0
    o = v.reset_at(2)
    assert o.shape == (23,) and float(o[0]) == 0.0
    with pytest.raises(ValueError):
        e = _FakeEnv(2)
        e.scene.struct.auto_reset = 1
        try:
            RllibVectorEnv(e)
        finally:
            e.scene.struct.auto_reset = 0


def test_risk_ground_truth_log_writes_the_reference_tables(tmp_path):
    """RiskGroundTruthLog: the rows smr_step leaves (state | action | risk | prediction | valid; state | risk | valid) become the
    reference's tables and CSV files (safe_motions_base.py:1413-1458, 1567-1586): invalid rows are dropped, mode 4 has two tables."""
    import pandas as pd
    import torch
    from safemotionsrisk_amd.metrics import RiskGroundTruthLog
    R, B = 23, 6
    rng = np.random.default_rng(0)
    sa = rng.uniform(-1, 1, (B, R + 10)).astype(np.float32)
    sa[:, R + 7] = [0, 1, 0, 0, 1, 0]; sa[:, R + 8] = sa[:, R + 7]; sa[:, R + 9] = 1
    st = rng.uniform(-1, 1, (B, R + 2)).astype(np.float32)
    st[:, R] = sa[:, R + 7]; st[:, R + 1] = [1, 0, 1, 1, 0, 1]
    log = RiskGroundTruthLog(R, simulate_backup_trajectory=True)
    log.add(dict(state_action=torch.from_numpy(sa), state=torch.from_numpy(st)))
    log.add(dict(state_action=torch.from_numpy(sa), state=torch.from_numpy(st)))
    assert len(log) == 12 and len(log.tables["state_risk"]["risk"]) == 8
    paths = log.write(str(tmp_path), 0, 4, pid=77, risk_config=dict(action_size=None, checkpoint="x", observation_size=None, config={"a": 1}))
    assert sorted(os.path.relpath(p, str(tmp_path)) for p in paths) == ["state_action_risk/episodes_0_to_4_risk_0.33_prediction_0.00_pid_77.csv",
                                                                      "state_risk/episodes_0_to_4_risk_0.00_prediction_0.00_pid_77.csv"]
    df = pd.read_csv(paths[0], index_col=0)
    assert list(df.columns) == ["state", "action", "risk"] and len(df) == 12 and df["risk"].sum() == 4
    assert len(eval(df["state"][0])) == R and len(eval(df["action"][0])) == 7
    assert "config" not in open(tmp_path / "risk_config.json").read() and len(log) == 0
    # a risk network's single table: valid rows only, with the prediction
    sa[:, R + 9] = [1, 1, 0, 1, 1, 1]
    net = RiskGroundTruthLog(R, simulate_backup_trajectory=False, state_action_network=False)
    net.add(dict(state_action=torch.from_numpy(sa), state=torch.from_numpy(st)))
    (p,) = net.write(str(tmp_path), 5, 9, pid=1)
    df = pd.read_csv(p, index_col=0)
    assert list(df.columns) == ["state", "risk", "risk_prediction"] and len(df) == 5 and "ground_truth_risk" in p
