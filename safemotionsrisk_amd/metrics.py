"""Episode metrics and the RLlib-style vector interface on top of VecSafeMotionsEnv (first half).

The reference returns, with every step, an ``info`` dictionary of per-step scalars grouped by the operator that
aggregates them over an episode (``info['average'|'max'|'min'][key]``: safe_motions_base.py:1350-1365,
observations.py:365-413, rewards.py:172-207,490-498), and ``CustomTrainCallbacks`` (train.py:40-117) folds them into one
custom metric ``<key>_<op>`` per finished episode plus ``episode_length``, ``success_rate`` and the
``*_termination_rate`` flags.  ``EpisodeMetrics`` is that pipeline for B environments at once: running sum / max / min
per (env, key) on the env's device, emitted for the envs that finished in a step.  Host logic only - it reads the
tensors ``step()`` returned and launches torch element-wise kernels; nothing here is on the hot path of the library.

``RllibVectorEnv`` exposes the three calls RLlib's ``VectorEnv`` is driven by (``vector_reset``, ``reset_at``,
``vector_step``; ray/rllib/env/vector_env.py) with per-env python ``info`` dictionaries shaped like the reference's.
Ray is not a dependency: the class is duck-typed, a maintainer registers it with ``VectorEnv.register`` or subclasses.
"""
from __future__ import annotations

import numpy as np

NJ = 7

# safe_motions_base.py:63-70
TERMINATION_JOINT_LIMITS = 1
TERMINATION_TRAJECTORY_LENGTH = 2
TERMINATION_SELF_COLLISION = 3
TERMINATION_COLLISION_WITH_STATIC_OBSTACLE = 4
TERMINATION_COLLISION_WITH_MOVING_OBSTACLE = 5

# train.py:106-117
TERMINATION_RATE_KEYS = {
    TERMINATION_TRAJECTORY_LENGTH: "trajectory_length_termination_rate",
    TERMINATION_JOINT_LIMITS: "joint_limit_violation_termination_rate",
    TERMINATION_SELF_COLLISION: "collision_self_termination_rate",
    TERMINATION_COLLISION_WITH_STATIC_OBSTACLE: "collision_static_obstacles_termination_rate",
    TERMINATION_COLLISION_WITH_MOVING_OBSTACLE: "collision_moving_obstacles_termination_rate",
}


#: event columns of the info matrix: folded into the episode-end statistics of ObstacleWrapperSim.get_info_and_print_stats
#: (ctlp.py:1141-1309) instead of the per-step average / max / min scalars
EVENT_COLUMNS = ("braking_trajectory_executed", "target_point_reached", "moving_object_final_state", "risky_action")
#: a property of the whole episode that travels in the info matrix: reported as it stands at the end of the episode
EPISODE_COLUMNS = {"starting_point_sampling_attempts": "obstacles_starting_point_sampling_attempts"}       # ctlp.py:1154


def step_scalar_names(info_columns):
    """Names of the per-step scalars, in the order of the columns of EpisodeMetrics' accumulators.
    obs-derived: observations.py:376-403 (joint_{j}_{pos,vel,acc}[_abs], joint_vel_norm); info-derived: the library's
    info columns except episode_length (rewards.py:490-498), collision rates (safe_motions_base.py:1290-1295), reward."""
    names = []
    for kind in ("pos", "vel", "acc"):
        names += ["joint_%d_%s" % (j, kind) for j in range(NJ)]
    for kind in ("pos", "vel", "acc"):
        names += ["joint_%d_%s_abs" % (j, kind) for j in range(NJ)]
    names += ["joint_vel_norm"]
    names += [c for c in info_columns if c != "episode_length" and c not in EVENT_COLUMNS and c not in EPISODE_COLUMNS]
    names += ["collision_rate_self", "collision_rate_static_obstacles", "collision_rate_moving_obstacles", "reward"]
    return names


class FinishedEpisodes:
    """Metrics of the episodes that ended in one step: ``env_index`` [n], ``episode_length`` [n], ``termination_reason``
    [n] and ``metrics[name]`` [n] for every ``<key>_<op>`` / rate of CustomTrainCallbacks.on_episode_end."""

    def __init__(self, env_index, episode_length, termination_reason, metrics):
        self.env_index, self.episode_length, self.termination_reason, self.metrics = env_index, episode_length, termination_reason, metrics

    def __len__(self):
        return int(self.env_index.shape[0])

    def mean(self):
        """What RLlib reports: the mean of every custom metric over these episodes (python floats)."""
        return {k: float(v.float().mean()) for k, v in self.metrics.items()} if len(self) else {}


class EpisodeMetrics:
    def __init__(self, num_envs, info_columns, device=None, auto_reset=True, trajectory_length=None):
        """trajectory_length: round(trajectory_duration / trajectory_time_step) + 1 (trajectory_manager.py:87), reported as is."""
        import torch
        self._t = torch
        self.trajectory_length = trajectory_length
        self.num_envs, self.auto_reset = int(num_envs), bool(auto_reset)
        self.info_columns = tuple(info_columns)
        self.names = step_scalar_names(self.info_columns)
        self._info_sel = [i for i, c in enumerate(self.info_columns) if c != "episode_length" and c not in EVENT_COLUMNS and c not in EPISODE_COLUMNS]
        self._episode_cols = [(i, EPISODE_COLUMNS[c]) for i, c in enumerate(self.info_columns) if c in EPISODE_COLUMNS]
        self._event_cols = [self.info_columns.index(c) if c in self.info_columns else None for c in EVENT_COLUMNS]
        self._len_col = self.info_columns.index("episode_length")
        self._d_cols = [self.info_columns.index(c) for c in ("minimum_distance_self_collision", "minimum_distance_to_static_obstacles",
                                                             "minimum_distance_to_moving_obstacles")]
        K = len(self.names)
        self.n_obs = 6 * NJ + 1                                    # the obs-derived scalars come first
        kw = dict(device=device, dtype=torch.float32)
        self.sum = torch.zeros((self.num_envs, K), **kw)
        self.max = torch.full((self.num_envs, K), -float("inf"), **kw)
        self.min = torch.full((self.num_envs, K), float("inf"), **kw)
        self.count = torch.zeros((self.num_envs, K), **kw)
        # episode-end statistics of the obstacle wrapper (ctlp.py:1141-1309): steps, steps with the braking trajectory executed,
        # target points reached, balls by final state (hit_robot, hit_obstacle, missed_robot)
        self.events = torch.zeros((self.num_envs, 8), **kw)      # ... steps with a risky action, first such step (0: none)
        self.link_min = None                                       # [B, n_bodies] running minimum of env.link_distances() (log_obstacle_data)

    def step_scalars(self, obs, reward, info_values):
        """[B, K] per-step scalars (the values of the reference's info['average'] entries of this step)."""
        t = self._t
        jo = obs[:, :3 * NJ].float()
        vel = jo[:, NJ:2 * NJ]
        d = info_values[:, self._d_cols]
        return t.cat([jo, jo.abs(), t.linalg.vector_norm(vel, dim=1, keepdim=True), info_values[:, self._info_sel].float(),
                      (d == 0).float(), reward.float().unsqueeze(1)], dim=1)

    def update(self, obs, reward, done, info, link_distances=None):
        """Fold one step in; returns FinishedEpisodes for the envs with done set (their accumulators restart).
        With auto_reset the obs of a finished env already belongs to its next episode, so the obs-derived scalars of
        terminal steps are left out there (every other scalar covers all steps).
        link_distances: ``env.link_distances()`` of this step ([n_bodies, B]) when the reference's ``log_obstacle_data`` statistics are
        wanted: the episode's minimum per body is reported as ``obstacles_link_data_table_body_<k>_closest_point_distance_set_min``
        (= ``..._actual_min``: the setpoints are the actual values of a kinematic env; ctlp.py:1279-1282, LinkData.export_metrics
        ctlp.py:5345-5358).  Sampled at the end of every step and exact below the query range of 0.102 m (the reference samples every
        1/240 s with a range of 10 m); its observed-point lists are empty for observed_link_point_scene = 0."""
        t = self._t
        if link_distances is not None:
            ld = link_distances.t().float()
            self.link_min = ld.clone() if self.link_min is None else t.minimum(self.link_min, ld)
        x = self.step_scalars(obs, reward, info["values"])
        valid = t.ones_like(x, dtype=t.bool)
        d = done.bool()
        if self.auto_reset:
            valid[:, :self.n_obs] = ~d.unsqueeze(1)
        xs = t.where(valid, x, t.zeros_like(x))
        self.sum += xs
        self.count += valid.float()
        self.max = t.where(valid, t.maximum(self.max, x), self.max)
        self.min = t.where(valid, t.minimum(self.min, x), self.min)
        v = info["values"]
        braked, reached, ball, risky = [v[:, c].float() if c is not None and c < v.shape[1] else t.zeros_like(reward, dtype=t.float32) for c in self._event_cols]
        first = t.where((risky > 0) & (self.events[:, 7] == 0), self.events[:, 0] + 1, t.zeros_like(risky))   # 1-based step of the first risky action
        self.events += t.stack([t.ones_like(braked), braked, reached, (ball == 1).float(), (ball == 2).float(), (ball == 3).float(), risky, first], dim=1)
        idx = d.nonzero(as_tuple=True)[0]
        if idx.numel() == 0:
            e = idx.new_zeros((0,))
            return FinishedEpisodes(idx, e, e, {})
        reason = info["termination_reason"][idx].long()
        length = info["values"][idx, self._len_col].long()
        cnt = self.count[idx].clamp(min=1)
        avg, mx, mn = self.sum[idx] / cnt, self.max[idx], self.min[idx]
        metrics = {}
        for k, name in enumerate(self.names):
            metrics[name + "_average"] = avg[:, k]
            metrics[name + "_max"] = mx[:, k]
            metrics[name + "_min"] = mn[:, k]
        metrics["episode_length"] = length.float()
        # trajectory_successful: True unless a joint limit was violated (actions.py:382-385, base.py:930,1368-1371)
        metrics["success_rate"] = (reason != TERMINATION_JOINT_LIMITS).float()
        if self.trajectory_length is not None:
            metrics["trajectory_length"] = t.full_like(metrics["episode_length"], float(self.trajectory_length))
        for code, name in TERMINATION_RATE_KEYS.items():
            metrics[name] = (reason == code).float()
        metrics.update(self._obstacle_stats(self.events[idx]))
        if self.link_min is not None:
            for k in range(self.link_min.shape[1]):
                metrics["obstacles_link_data_table_body_%d_closest_point_distance_set_min" % k] = self.link_min[idx, k]
                metrics["obstacles_link_data_table_body_%d_closest_point_distance_actual_min" % k] = self.link_min[idx, k]
            self.link_min[idx] = float("inf")
        for col, name in self._episode_cols:
            if col < info["values"].shape[1]:
                metrics[name] = info["values"][idx, col].float()
        self.events[idx] = 0
        self.sum[idx] = 0
        self.count[idx] = 0
        self.max[idx] = -float("inf")
        self.min[idx] = float("inf")
        return FinishedEpisodes(idx, length, reason, metrics)


    def _obstacle_stats(self, ev):
        """ObstacleWrapperSim.get_info_and_print_stats (ctlp.py:1141-1309), the part this path produces: the braking-trajectory
        influence (torque checks are physics: 0), target points reached, final states of the balls of the episode
        (obstacles_starting_point_sampling_attempts comes from its info column, obstacles_link_data_* from env.link_distances(), both
        in update()).  Not produced: the target-point sampling attempts of several robots (ctlp.py:1171-1176) and the per-pair
        obstacles_self_collision_data_* (the self distances exist per env, not per pair).
        minimum_distance_self_collision_{average,min,max} differ from the reference's logged values where the self distance is at or
        beyond the range its reward reads (SMR_INFO_D_SELF): the reference logs the closest-point distance up to
        0.102 m, this column holds the range there; reward and termination are unaffected."""
        t = self._t
        steps, braked, reached, hit_robot, hit_obstacle, missed, risky, first_risky = ev.unbind(dim=1)
        infl = braked / steps.clamp(min=1)
        zero = t.zeros_like(infl)
        out = {"obstacles_time_influenced_by_braking_trajectory": infl,
               "obstacles_time_influenced_by_braking_trajectory_collision": infl,
               "obstacles_time_influenced_by_braking_trajectory_torque": zero,
               "obstacles_episodes_without_influence_by_braking_trajectory": (infl == 0).float(),
               "obstacles_episodes_without_influence_by_braking_trajectory_collision": (infl == 0).float(),
               "obstacles_episodes_without_influence_by_braking_trajectory_torque": t.ones_like(infl),
               "obstacles_episodes_with_braking_timeout": zero,                  # a timed-out braking trajectory counts as a collision here
               "obstacles_num_target_points_reached": reached}
        if self._event_cols[3] is not None:                                       # actions.py:306-340, base.py:1378
            out["risky_action_rate_average"] = risky / steps.clamp(min=1)
            out["risk_network_first_risky_action_step"] = t.where(first_risky > 0, first_risky - 1, t.full_like(first_risky, float("nan")))
        if self._event_cols[2] is not None:
            total = hit_robot + hit_obstacle + missed
            out.update({"moving_object_hit_robot_total": hit_robot, "moving_object_hit_obstacle_total": hit_obstacle,
                        "moving_object_hit_robot_or_obstacle_total": hit_robot + hit_obstacle, "moving_object_missed_robot_total": missed,
                        "moving_object_hit_robot_r0": hit_robot, "moving_object_hit_obstacle_r0": hit_obstacle,
                        "moving_object_hit_robot_or_obstacle_r0": hit_robot + hit_obstacle, "moving_object_missed_robot_r0": missed})
            safe = total.clamp(min=1)
            nan = t.full_like(total, float("nan"))                               # the reference omits the fractions of an episode without a final state
            for k, x in (("hit_robot", hit_robot), ("hit_obstacle", hit_obstacle), ("hit_robot_or_obstacle", hit_robot + hit_obstacle),
                         ("missed_robot", missed)):
                out["moving_object_%s_fraction" % k] = t.where(total > 0, x / safe, nan)
        return out


class RllibVectorEnv:
    """vector_reset / reset_at / vector_step over a VecSafeMotionsEnv built with auto_reset=False (RLlib resets finished
    sub-environments itself through reset_at).  Observations, rewards and dones come back as python lists of numpy values;
    infos as one dictionary per env shaped like the reference's (``average`` / ``max`` / ``min`` sub-dicts of the step,
    and at the end of an episode ``episode_length``, ``trajectory_length``, ``termination_reason``,
    ``trajectory_successful``: safe_motions_base.py:1367-1377)."""

    def __init__(self, env):
        if getattr(env, "scene", None) is not None and env.scene.struct.auto_reset:
            raise ValueError("RllibVectorEnv needs an env built with auto_reset=False")
        self.env = env
        self.num_envs = env.num_envs
        self.observation_space, self.action_space = env.observation_space, env.action_space
        self._names = step_scalar_names(env.INFO_COLUMNS)
        self.trajectory_length = int(env.scene.struct.episode_steps) + 1                   # trajectory_manager.py:87
        self._metrics = EpisodeMetrics(env.num_envs, env.INFO_COLUMNS, device=getattr(env, "device", None), auto_reset=False)
        self._len_col = env.INFO_COLUMNS.index("episode_length")

    def get_sub_environments(self):
        return []

    def vector_reset(self):
        obs = self.env.reset()
        return list(obs.detach().cpu().numpy())

    def reset_at(self, index=None):
        import torch
        mask = torch.zeros(self.num_envs, dtype=torch.uint8)
        mask[0 if index is None else int(index)] = 1
        obs = self.env.reset(mask=mask)
        return obs[0 if index is None else int(index)].detach().cpu().numpy()

    def vector_step(self, actions):
        obs, reward, done, info = self.env.step(np.asarray(actions, dtype=np.float32))
        x = self._metrics.step_scalars(obs, reward, info["values"]).detach().cpu().numpy()
        o, r, d = obs.detach().cpu().numpy(), reward.detach().cpu().numpy(), done.detach().cpu().numpy().astype(bool)
        reason = info["termination_reason"].detach().cpu().numpy()
        length = info["values"][:, self._len_col].detach().cpu().numpy()
        infos = []
        for e in range(self.num_envs):
            step = dict(zip(self._names, x[e].tolist()))
            ie = {"average": step, "max": dict(step), "min": dict(step)}
            if d[e]:
                ie.update(episode_length=int(length[e]), trajectory_length=self.trajectory_length, termination_reason=int(reason[e]),
                          trajectory_successful=0.0 if reason[e] == TERMINATION_JOINT_LIMITS else 1.0)
            infos.append(ie)
        return list(o), r.tolist(), d.tolist(), infos


class RiskGroundTruthLog:
    """The tables of ``risk_store_ground_truth`` (safe_motions_base.py:1567-1586,1605-1650) and their files
    (``_store_risk_ground_truth_data``, safe_motions_base.py:1413-1458) for a batched env: ``add()`` after every step() takes the valid rows
    of ``env.risk_ground_truth`` (include/safemotions_hip.h, smr_set_risk_rollout), ``write()`` stores what has been collected the way
    the reference does - one CSV per table under ``<evaluation_dir>/state_action_risk`` and ``/state_risk`` (risk_state_config 4) or
    ``/ground_truth_risk`` (a risk network; columns state, [action,] risk, risk_prediction), named
    ``episodes_<a>_to_<b>_risk_<mean>_prediction_<mean>_pid_<pid>.csv``, list-valued cells as pandas writes the reference's dict of lists
    - and ``risk_config.json`` beside them.  This is what the reference's ``train_risk_network.py`` reads.  Host logic only."""

    def __init__(self, risk_observation_size, simulate_backup_trajectory, state_action_network=False):
        self.R, self.mode4, self.with_action = int(risk_observation_size), bool(simulate_backup_trajectory), bool(state_action_network) or bool(simulate_backup_trajectory)
        self.reset()

    def reset(self):
        self.tables = {"state_action_risk": dict(state=[], action=[], risk=[]), "state_risk": dict(state=[], risk=[])} if self.mode4 else \
                      {"ground_truth_risk": dict(state=[], **({"action": []} if self.with_action else {}), risk=[], risk_prediction=[])}

    def add(self, ground_truth):
        R = self.R
        sa = ground_truth["state_action"].detach().cpu().numpy()
        sa = sa[sa[:, R + 9] > 0.5]
        t = self.tables["state_action_risk" if self.mode4 else "ground_truth_risk"]
        t["state"] += sa[:, :R].tolist(); t["risk"] += sa[:, R + 7].tolist()
        if "action" in t:
            t["action"] += sa[:, R:R + 7].tolist()
        if "risk_prediction" in t:
            t["risk_prediction"] += sa[:, R + 8].tolist()
        if self.mode4:
            s = ground_truth["state"].detach().cpu().numpy()
            s = s[s[:, R + 1] > 0.5]
            self.tables["state_risk"]["state"] += s[:, :R].tolist(); self.tables["state_risk"]["risk"] += s[:, R].tolist()

    def __len__(self):
        return len(next(iter(self.tables.values()))["risk"])

    def write(self, evaluation_dir, start_episode, end_episode, pid=None, risk_config=None):
        """returns the paths written; the tables are emptied"""
        import json
        import os
        import pandas as pd
        pid = os.getpid() if pid is None else pid
        paths = []
        for name, table in self.tables.items():
            if not table["risk"]:
                continue
            d = os.path.join(evaluation_dir, name)
            os.makedirs(d, exist_ok=True)
            mean_risk = float(np.mean(table["risk"]))
            mean_pred = float(np.mean(table["risk_prediction"])) if "risk_prediction" in table else 0.0
            path = os.path.join(d, "episodes_{}_to_{}_risk_{:.2f}_prediction_{:.2f}_pid_{}.csv".format(start_episode, end_episode, mean_risk, mean_pred, pid))
            with open(path, "w") as f:
                pd.DataFrame(table).to_csv(path_or_buf=f)
            paths.append(path)
        cfg_path = os.path.join(evaluation_dir, "risk_config.json")
        if risk_config is not None and not os.path.isfile(cfg_path):
            with open(cfg_path, "w") as f:
                f.write(json.dumps({k: v for k, v in risk_config.items() if k != "config"}, sort_keys=True))
        self.reset()
        return paths
