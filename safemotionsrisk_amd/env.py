"""Batched drop-in for the reference's gym surface.

``VecSafeMotionsEnv(num_envs=B, **env_config)`` mirrors ``SafeMotionsEnvCollisionAvoidance(**env_config)`` and
``SafeMotionsEnv(**env_config)`` (safemotions/envs/safe_motions_env.py:38,19, constructed at train.py:620; which of the two reward
classes runs follows from ``collision_avoidance_mode``, as train.py:614-619 picks the class): same ``env_config`` keys, same
``TERMINATION_*`` constants, ``action_space`` / ``observation_space`` shapes, ``reset()`` -> obs and
``step(action)`` -> (obs, reward, done, info) - batched over B environments that live on one GPU, one
environment per wavefront lane.  All arithmetic runs in libsmr_hip.so; torch only provides device
buffers and the stream.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _lib
from .scene import NJ, build_scene


class Box:
    """Minimal stand-in for gym.spaces.Box (gym is not a dependency of the hot path)."""

    def __init__(self, low, high, shape, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = np.float32(low), np.float32(high), tuple(shape), dtype

    def __repr__(self):
        return "Box(%s, %s, %s, %s)" % (self.low, self.high, self.shape, np.dtype(self.dtype).name)


class VecSafeMotionsEnv:
    # base.py:63-70
    TERMINATION_UNSET = -1
    TERMINATION_SUCCESS = 0
    TERMINATION_JOINT_LIMITS = 1
    TERMINATION_TRAJECTORY_LENGTH = 2
    TERMINATION_SELF_COLLISION = 3
    TERMINATION_COLLISION_WITH_STATIC_OBSTACLE = 4
    TERMINATION_COLLISION_WITH_MOVING_OBSTACLE = 5

    STATE_KEYS = ("q", "v", "a", "planet_idx", "step_count", "moving_hit", "reset_count", "ball", "ball_i", "target",
                  "target_count")

    INFO_COLUMNS = ("minimum_distance_to_static_obstacles", "minimum_distance_self_collision",
                    "minimum_distance_to_moving_obstacles", "action_punishment", "self_collision_reward",
                    "static_obstacles_collision_reward", "moving_obstacles_collision_reward", "episode_length",
                    "braking_trajectory_executed", "target_point_reached", "moving_object_final_state", "risky_action",
                    "end_min_distance_punishment", "braking_trajectory_min_distance_punishment", "starting_point_sampling_attempts")

    def __init__(self, num_envs, device=0, seed=None, auto_reset=True, collision_checks_per_step=None,
                 contact_distance=None, return_info=True, env_index_base=0, trained_networks_dir=None, **env_config):
        """env_index_base: this handle owns the envs [env_index_base, env_index_base + num_envs) of a global batch that is
        sharded over several handles / GPUs (dist.py): every draw of an env is keyed by (seed, GLOBAL env index, its counters), so
        the shards together produce bit for bit what one handle over the whole batch produces.
        trained_networks_dir: where a relative ``risk_config_dir`` / checkpoint path of the env_config is looked up (the reference's
        ``safemotions/trained_networks``; scene.resolve_risk_config).  An env_config with a risk network reads its two JSON files
        and then WAITS for the weights: step() raises until set_risk_networks() has been called with arrays (policy.Mlp)."""
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("VecSafeMotionsEnv needs a GPU: the HIP library has no CPU path")
        self._torch = torch
        self._L = _lib.load()
        kw = {} if contact_distance is None else dict(contact_distance=contact_distance)
        self.scene = build_scene(env_config, collision_checks_per_step=collision_checks_per_step,
                                 auto_reset=auto_reset, trained_networks_dir=trained_networks_dir, **kw)
        self.num_envs = int(num_envs)
        self.device = torch.device("cuda", int(device))
        self.observation_size = self.scene.obs_size
        self.action_space = Box(-1, 1, (NJ,))                                   # actions.py:49-51
        self.observation_space = Box(-1, 1, (self.observation_size,))           # obs.py:116-117
        self.trajectory_time_step = float(self.scene.struct.dt)
        self._explicit_resets = 0
        self._return_info = bool(return_info)
        self._seed = 0 if seed is None else int(seed)                           # base.py:1704-1710
        self._device_index, self._env_index_base = int(device), int(env_index_base)
        if self._env_index_base < 0 or self._env_index_base + self.num_envs > 0xFFFFFFFF:
            raise ValueError("env_index_base + num_envs must fit 32 bits")
        self._h = None
        self._shadow_h, self._rollout = None, None    # the background copy of the env (set_risk_rollout)
        self.risk_ground_truth = None
        self._scene_kw, self._trained_networks_dir = dict(kw, collision_checks_per_step=collision_checks_per_step), trained_networks_dir
        self._risk_nets, self._risk_threshold, self._risk_state_config, self._profiling = None, 0.05, 0, False
        self.risk_config = self.scene.risk            # the env_config's risk network description (observation_size, action_size, ...) or None
        if self.risk_config is not None:
            self._risk_threshold, self._risk_state_config = self.risk_config["risk_threshold"], self.risk_config["risk_state_config"]
        self._create()
        if self.risk_config is not None and self.risk_config["observation_size"] not in (None, self.risk_observation_size):
            raise ValueError("risk_config.json: observation_size %d, the env's risk observation has %d columns (safe_motions_base.py:542-571)"
                             % (self.risk_config["observation_size"], self.risk_observation_size))
        B, O = self.num_envs, self.observation_size
        self.obs = torch.zeros((B, O), dtype=torch.float32, device=self.device)
        self.reward = torch.zeros(B, dtype=torch.float32, device=self.device)
        self.done = torch.zeros(B, dtype=torch.uint8, device=self.device)
        self.termination_reason = torch.full((B,), -1, dtype=torch.int8, device=self.device)
        self.info = torch.zeros((B, _lib.INFO_N), dtype=torch.float32, device=self.device) if return_info else None

    def _create(self):
        torch = self._torch
        with torch.cuda.device(self.device):
            self._stream = torch.cuda.current_stream(self.device)
            handle = C.c_void_p()
            rc = self._L.smr_create(C.cast(self.scene.pointer(), C.c_void_p), self.num_envs, C.c_uint32(self._env_index_base),
                                    self._device_index, C.c_uint64(self._seed & 0xFFFFFFFFFFFFFFFF),
                                    C.c_void_p(self._stream.cuda_stream), C.byref(handle))
            _lib.check(self._L, None, rc, "smr_create")
        self._h = handle

    # ---- gym surface ----
    def set_seed(self, seed=None):
        """base.py:1704-1710 seeds the global numpy / random generators.  Here the per-env Philox streams are keyed by
        (seed, env index, reset count): re-keying means a fresh handle - every env restarts from reset count 0 and must be
        reset() before the next step(), as after construction."""
        self._seed = 0 if seed is None else int(seed)
        self.close()
        self._create()
        self._explicit_resets = 0
        # the fresh handle knows nothing of what was configured on the old one: put it back (risk-aware stepping, profiling)
        rollout = self._rollout
        self._rollout = None
        if self._risk_nets is not None and self._risk_nets[1] is not None:
            self.set_risk_networks(*self._risk_nets, risk_threshold=self._risk_threshold, risk_state_config=self._risk_state_config)
        if rollout is not None:
            self.set_risk_rollout(**rollout)
        if self._profiling:
            self.set_profiling(self._profiling)

    def reset(self, mask=None):
        """obs = env.reset() for all envs (or those selected by a bool/uint8 mask tensor)."""
        t = self._torch
        m = None
        if mask is not None:
            m = mask.to(device=self.device, dtype=t.uint8).contiguous()
        rc = self._L.smr_reset(self._h, None if m is None else C.c_void_p(m.data_ptr()), C.c_void_p(self.obs.data_ptr()))
        _lib.check(self._L, self._h, rc, "smr_reset")
        self._explicit_resets += self.num_envs if m is None else int(m.sum().item())
        return self.obs

    def step(self, action):
        """(obs, reward, done, info) = env.step(action); action float32 [B, 7] on the env's device."""
        t = self._torch
        if not isinstance(action, t.Tensor):
            action = t.as_tensor(np.asarray(action, dtype=np.float32))
        a = action.to(device=self.device, dtype=t.float32).contiguous()
        if a.shape != (self.num_envs, NJ):
            raise ValueError("action must have shape (%d, %d), got %s" % (self.num_envs, NJ, tuple(a.shape)))
        if self.risk_config is not None and (self._risk_nets is None or self._risk_nets[1] is None):
            raise RuntimeError("this env_config names a risk network (%s): call set_risk_networks(risk, backup) with the weights as arrays "
                               "(policy.Mlp / Mlp.from_npz) before the first step - checkpoints and SavedModels are never deserialised here"
                               % self.risk_config["checkpoint"])
        rc = self._L.smr_step(self._h, C.c_void_p(a.data_ptr()), C.c_void_p(self.obs.data_ptr()),
                              C.c_void_p(self.reward.data_ptr()), C.c_void_p(self.done.data_ptr()),
                              C.c_void_p(self.termination_reason.data_ptr()),
                              None if self.info is None else C.c_void_p(self.info.data_ptr()))
        _lib.check(self._L, self._h, rc, "smr_step")
        info = {"termination_reason": self.termination_reason}
        if self.info is not None:
            info["columns"] = self.INFO_COLUMNS
            info["values"] = self.info
        return self.obs, self.reward, self.done, info

    def set_risk_networks(self, risk, backup, risk_threshold=None, risk_state_config=None):
        """Risk-aware stepping (actions.py:303-340, base.py:1498-1517,1598-1603; risk_threshold default base.py:222-223): inside
        step(), where the risk network's rating reaches risk_threshold the backup policy's deterministic action replaces the
        action.  A network over [risk observation, action] rates the state-action pair; one over the risk observation alone rates
        the state ``risk_state_config`` names: 0 the current one, 1 the next one with its kinematic columns forecast for the
        requested action, 2 with the planets / the ball forecast as well, 3 the state a background copy of the env reaches when it
        is stepped with the action (a collision there is risk 1), 4 no network at all: risk 1 if the copy, stepped with the action
        and then ``backup_trajectory_steps`` actions of the backup policy, ends in a collision (``risk`` may be None).  Modes 3 / 4
        and the ground truth need ``set_risk_rollout`` (an env_config that asks for them gets it here).  risk / backup: policy.Mlp
        objects on this env's device (None, None switches it off); info column ``risky_action`` reports it per step.  Defaults: what
        the env_config's risk keys said, else 0.05 and 0."""
        if risk_threshold is None:
            risk_threshold = self._risk_threshold
        if risk_state_config is None:
            risk_state_config = self._risk_state_config
        if risk is not None and self.risk_config is not None and self.risk_config["observation_size"] is not None:
            want = self.risk_observation_size + (NJ if self.risk_config["action_size"] is not None else 0)
            if risk.sizes[0] != want:
                raise ValueError("risk_config.json describes a risk network with %d inputs, the one given has %d" % (want, risk.sizes[0]))
        self._risk_nets, self._risk_threshold, self._risk_state_config = (risk, backup), float(risk_threshold), int(risk_state_config)   # re-applied by set_seed()
        rc = self._L.smr_set_risk_state_config(self._h, int(risk_state_config))
        _lib.check(self._L, self._h, rc, "smr_set_risk_state_config")
        rc = self._L.smr_set_risk_networks(self._h, None if risk is None else risk._h, None if backup is None else backup._h,
                                           C.c_float(float(risk_threshold)))
        _lib.check(self._L, self._h, rc, "smr_set_risk_networks")
        rc_cfg = self.risk_config or {}
        wants_copy = "backup_trajectory_steps" in rc_cfg or (int(risk_state_config) >= 3 and (risk is None or risk.sizes[0] == self.risk_observation_size))
        if backup is not None and wants_copy and self._rollout is None:
            self.set_risk_rollout(backup_steps=rc_cfg.get("backup_trajectory_steps"), store_ground_truth=rc_cfg.get("store_ground_truth", False),
                                  ignore_probability=rc_cfg.get("ignore_probability", 0.0))

    def set_risk_rollout(self, backup_steps=None, store_ground_truth=False, ignore_probability=0.0):
        """The background copy of the env (the reference's backup client, safe_motions_base.py:1518-1577,1605-1655,1801-1842): a second
        handle of the same scene, batch, seed and stream whose episodes are ``backup_steps + 1`` steps long; step() copies the envs'
        state into it and steps it ahead on the device - with the requested action, then ``backup_steps``
        (``risk_state_backup_trajectory_steps``; default: the length of this env's own episodes, as base.py:573-576 takes the backup
        policy's) actions of the backup policy.  ``store_ground_truth`` (``risk_store_ground_truth``): every step() then leaves
        ``self.risk_ground_truth`` = dict(state_action=[B, R + 7 + 3] rows (risk observation | action | risk | prediction | valid),
        state=[B, R + 2] rows (mode 4: the copy's next risk observation | risk | valid)) on the device (include/safemotions_hip.h,
        smr_set_risk_rollout; metrics.RiskGroundTruthLog collects them into the reference's tables).  ``ignore_probability``:
        ``risk_ignore_estimation_probability``.  backup_steps=False switches the copy off."""
        torch = self._torch
        if backup_steps is False:
            _lib.check(self._L, self._h, self._L.smr_set_risk_rollout(self._h, None, 0, C.c_float(float(ignore_probability)), None, None), "smr_set_risk_rollout")
            self._close_shadow()
            self._rollout, self.risk_ground_truth = None, None
            return
        s = self.scene.struct
        K = int(s.episode_steps) if backup_steps is None else int(backup_steps)
        if K < 0:
            raise ValueError("backup_steps must be >= 0")
        self._close_shadow()
        # the scene as resolve_risk_config left it (the backup policy's scene parameters put in, the risk keys taken out), episodes of K + 1 steps
        cfg = dict(self.scene.env_config, trajectory_duration=(K + 1) * float(s.dt))      # trajectory_length = backup_steps + 2, base.py:1819
        self._shadow_scene = build_scene(cfg, auto_reset=False, trained_networks_dir=self._trained_networks_dir, **self._scene_kw)
        assert int(self._shadow_scene.struct.episode_steps) == K + 1 and self._shadow_scene.obs_size == self.observation_size
        with torch.cuda.device(self.device):
            handle = C.c_void_p()
            rc = self._L.smr_create(C.cast(self._shadow_scene.pointer(), C.c_void_p), self.num_envs, C.c_uint32(self._env_index_base),
                                    self._device_index, C.c_uint64(self._seed & 0xFFFFFFFFFFFFFFFF), C.c_void_p(self._stream.cuda_stream), C.byref(handle))
            _lib.check(self._L, None, rc, "smr_create (background copy)")
        self._shadow_h = handle
        gt = None
        if store_ground_truth:
            R = self.risk_observation_size
            gt = dict(state_action=torch.zeros((self.num_envs, R + NJ + 3), dtype=torch.float32, device=self.device),
                      state=torch.zeros((self.num_envs, R + 2), dtype=torch.float32, device=self.device))
        rc = self._L.smr_set_risk_rollout(self._h, self._shadow_h, K, C.c_float(float(ignore_probability)),
                                          None if gt is None else C.c_void_p(gt["state_action"].data_ptr()), None if gt is None else C.c_void_p(gt["state"].data_ptr()))
        _lib.check(self._L, self._h, rc, "smr_set_risk_rollout")
        self.risk_ground_truth = gt
        self._rollout = dict(backup_steps=K, store_ground_truth=bool(store_ground_truth), ignore_probability=float(ignore_probability))   # re-applied by set_seed()

    def _close_shadow(self):
        if getattr(self, "_shadow_h", None):
            if getattr(self, "_h", None):
                self._L.smr_set_risk_rollout(self._h, None, 0, C.c_float(0.0), None, None)
            self._L.smr_destroy(self._shadow_h)
            self._shadow_h = None

    @property
    def episode_counter(self):
        """base.py:913-916: one count per reset() of an env - the explicit resets plus the restarts that happened inside step()
        (auto_reset; counted on the device, reading it waits for the stream)"""
        n = C.c_int64(0)
        _lib.check(self._L, self._h, self._L.smr_episode_count(self._h, C.byref(n)), "smr_episode_count")
        return self._explicit_resets + int(n.value)

    def link_distances(self):
        """[n_bodies, B] distances of the bodies to the static obstacles at the end of the last step (log_obstacle_data, ctlp.py:2639-2660;
        exact below the query range of 0.102 m, the range beyond it): what metrics.EpisodeMetrics folds into obstacles_link_data_*"""
        t = self._torch
        if getattr(self, "_link_dist", None) is None:
            self._link_dist = t.zeros((int(self.scene.struct.n_bodies), self.num_envs), dtype=t.float32, device=self.device)
        _lib.check(self._L, self._h, self._L.smr_link_distances(self._h, C.c_void_p(self._link_dist.data_ptr())), "smr_link_distances")
        return self._link_dist

    @property
    def risk_observation_size(self):
        """obs.py:419-431: the observation without its target-point columns"""
        s = self.scene.struct
        return self.observation_size - (3 * (int(s.obs_target_pos) + int(s.obs_target_rel)) if s.use_target_points else 0)

    def observe(self):
        rc = self._L.smr_observe(self._h, C.c_void_p(self.obs.data_ptr()))
        _lib.check(self._L, self._h, rc, "smr_observe")
        return self.obs

    def close(self):
        self._close_shadow()
        if getattr(self, "_h", None):
            self._L.smr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _lib.check(self._L, self._h, self._L.smr_synchronize(self._h), "smr_synchronize")

    # ---- state injection / checkpointing (base.py:1801-1891 counterpart) ----
    def get_state(self):
        B = self.num_envs
        st = dict(q=np.zeros((B, NJ), np.float32), v=np.zeros((B, NJ), np.float32), a=np.zeros((B, NJ), np.float32),
                  planet_idx=np.zeros((B, 2), np.int32), step_count=np.zeros(B, np.int32),
                  moving_hit=np.zeros(B, np.int32), reset_count=np.zeros(B, np.uint32),
                  ball=np.zeros((B, 10), np.float32), ball_i=np.zeros((B, 5), np.int32),
                  target=np.zeros((B, 5), np.float32), target_count=np.zeros(B, np.uint32))
        hs = _lib.HostState(*[st[k].ctypes.data for k in self.STATE_KEYS])
        _lib.check(self._L, self._h, self._L.smr_get_state(self._h, C.byref(hs)), "smr_get_state")
        return st

    def set_state(self, **kw):
        B = self.num_envs
        spec = dict(q=(np.float32, (B, NJ)), v=(np.float32, (B, NJ)), a=(np.float32, (B, NJ)),
                    planet_idx=(np.int32, (B, 2)), step_count=(np.int32, (B,)), moving_hit=(np.int32, (B,)),
                    reset_count=(np.uint32, (B,)), ball=(np.float32, (B, 10)), ball_i=(np.int32, (B, 5)),
                    target=(np.float32, (B, 5)), target_count=(np.uint32, (B,)))
        keep, ptrs = [], []
        for k in self.STATE_KEYS:
            if kw.get(k) is None:
                ptrs.append(None)
            else:
                arr = np.ascontiguousarray(np.broadcast_to(np.asarray(kw[k], dtype=spec[k][0]), spec[k][1]))
                keep.append(arr)
                ptrs.append(arr.ctypes.data)
        unknown = set(kw) - set(spec)
        if unknown:
            raise TypeError("unknown state fields: %s" % sorted(unknown))
        hs = _lib.HostState(*ptrs)
        _lib.check(self._L, self._h, self._L.smr_set_state(self._h, C.byref(hs)), "smr_set_state")

    # ---- profiling ----
    def set_profiling(self, enable):
        """HIP-event timing of the kernels: False / 0 off, True / 1 around every launch of every step, n > 1 of every n-th step (the
        events are barrier packets in the queue: timing every step costs 65 us per step at 65 536 envs)."""
        self._profiling = int(enable)
        _lib.check(self._L, self._h, self._L.smr_set_profiling(self._h, int(enable)), "smr_set_profiling")

    def get_profile(self):
        ms = (C.c_double * _lib.K_COUNT)()
        n = (C.c_int64 * _lib.K_COUNT)()
        _lib.check(self._L, self._h, self._L.smr_get_profile(self._h, ms, n), "smr_get_profile")
        return {name: dict(ms=ms[i], launches=int(n[i])) for i, name in enumerate(_lib.KERNEL_NAMES)}

    # ---- component entry points (parity tests) ----
    def test_fk(self, q):
        q = np.ascontiguousarray(q, dtype=np.float32).reshape(-1, NJ)
        n = len(q)
        pos = np.zeros((n, 8, 3), np.float32)
        rot = np.zeros((n, 8, 3, 3), np.float32)
        rc = self._L.smr_test_fk(self._h, q.ctypes.data, n, pos.ctypes.data, rot.ctypes.data)
        _lib.check(self._L, self._h, rc, "smr_test_fk")
        return pos, rot

    def test_safe_acc_range(self, p, v, a):
        p, v, a = [np.ascontiguousarray(x, dtype=np.float32).reshape(-1, NJ) for x in (p, v, a)]
        n = len(p)
        lo, hi, code = np.zeros((n, NJ), np.float32), np.zeros((n, NJ), np.float32), np.zeros((n, NJ), np.int32)
        rc = self._L.smr_test_safe_acc_range(self._h, p.ctypes.data, v.ctypes.data, a.ctypes.data, n,
                                             lo.ctypes.data, hi.ctypes.data, code.ctypes.data)
        _lib.check(self._L, self._h, rc, "smr_test_safe_acc_range")
        return lo, hi, code

    def test_distances(self, q, planet_idx):
        q = np.ascontiguousarray(q, dtype=np.float32).reshape(-1, NJ)
        idx = np.ascontiguousarray(planet_idx, dtype=np.int32).reshape(-1, 2)
        n = len(q)
        out = np.zeros((n, 4), np.float32)
        rc = self._L.smr_test_distances(self._h, q.ctypes.data, idx.ctypes.data, n, out.ctypes.data)
        _lib.check(self._L, self._h, rc, "smr_test_distances")
        return out


    def test_brake_plan(self):
        """(acc [segments + 1, 7, B], steps [B], braked [B]) of the last step's braking-trajectory plan (smr_test_brake_plan)"""
        B, n = self.num_envs, int(self.scene.struct.brake_timeout_steps) + 1
        acc, steps, braked = np.zeros((n, NJ, B), np.float32), np.zeros(B, np.int32), np.zeros(B, np.int32)
        rc = self._L.smr_test_brake_plan(self._h, acc.ctypes.data, steps.ctypes.data, braked.ctypes.data)
        _lib.check(self._L, self._h, rc, "smr_test_brake_plan")
        return acc, steps, braked

    def test_self_bounds(self, q):
        """[n][n_self_pairs] table lower bounds of the self-collision pairs' distances (-3e38: the pair has no table)"""
        q = np.ascontiguousarray(q, dtype=np.float32).reshape(-1, NJ)
        out = np.zeros((len(q), 16), np.float32)
        rc = self._L.smr_test_self_bounds(self._h, q.ctypes.data, len(q), out.ctypes.data)
        _lib.check(self._L, self._h, rc, "smr_test_self_bounds")
        return out[:, :int(self.scene.struct.n_self_pairs)]

    def test_ball_distances(self, q, ball, k):
        q = np.ascontiguousarray(q, dtype=np.float32).reshape(-1, NJ)
        ball = np.ascontiguousarray(ball, dtype=np.float32).reshape(-1, 10)
        k = np.ascontiguousarray(k, dtype=np.int32).reshape(-1)
        n = len(q)
        out = np.zeros((n, 4), np.float32)
        rc = self._L.smr_test_ball_distances(self._h, q.ctypes.data, ball.ctypes.data, k.ctypes.data, n, out.ctypes.data)
        _lib.check(self._L, self._h, rc, "smr_test_ball_distances")
        return out


#: name parity with the reference classes (safe_motions_env.py:38,19)
SafeMotionsEnvCollisionAvoidance = VecSafeMotionsEnv
SafeMotionsEnv = VecSafeMotionsEnv
