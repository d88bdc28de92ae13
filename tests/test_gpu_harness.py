"""GPU tests of the harness-level behaviour around the hot path: BASELINE configs[0] (one env, 1000 random steps), error
reporting through the C-ABI, re-seeding, and two real env shards in one process group.  Run on the GPU box: pytest -m gpu."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from parity_util import explain_mismatches

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STATE_KEYS = ("q", "v", "a", "planet_idx", "step_count", "moving_hit", "reset_count")
OBS_TOL = np.concatenate([np.full(7, 2e-6), np.full(7, 1e-5), np.full(7, 2e-5), np.full(2, 1e-6)])


def test_config1_single_env_1000_random_steps(built):
    """BASELINE configs[0] in its stated shape (evaluate.py:198-303 with --random_agent, base.py:1047-1048): ONE env, space/planet
    scene, 1000 steps of uniform(-1,1,7) actions from PCG64 seed 0, auto-reset.  Every step is checked against the oracle
    restarted from the env's state (one-step parity), episodes end after 20 steps or at a collision, limits hold."""
    import torch
    from oracle.oracle import OracleEnv
    from safemotionsrisk_amd import SafeMotionsEnvCollisionAvoidance, SPACE_ENV_CONFIG
    env = SafeMotionsEnvCollisionAvoidance(1, device=0, seed=0, auto_reset=True, **SPACE_ENV_CONFIG)
    s = env.scene.struct
    assert s.collision_checks_per_step == 24
    ora = OracleEnv(env.scene, 1, seed=0)
    obs = env.reset().cpu().numpy()
    assert obs.shape == (1, 23)
    rng = np.random.default_rng(0)
    lo, hi, V, A = (np.array(x) for x in (s.pos_lo, s.pos_hi, s.vel_max, s.acc_max))
    length, episodes, reasons, mism = 0, 0, [], 0
    for t in range(1000):
        st = env.get_state()
        assert st["step_count"][0] == length
        ora.set_state(**{k: st[k] for k in STATE_KEYS})
        act = rng.uniform(-1, 1, (1, 7)).astype(np.float32)
        obs, rew, done, info = env.step(torch.from_numpy(act).cuda())
        o_obs, o_rew, o_done, o_reason, o_info = ora.step(act.astype(np.float64))
        obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        reason = info["termination_reason"].cpu().numpy()
        vals = info["values"].cpu().numpy()
        length += 1
        assert vals[0, 7] == length and np.abs(obs).max() <= 1.0
        if done[0] != o_done[0] or reason[0] != o_reason[0]:
            # both envs auto-reset: the end acceleration of the step's cubic is recomputed from the oracle's range (A2, actions.py:496-498)
            r_lo, r_hi, _ = ora.safe_acc_range(*(st[k][0].astype(np.float64) for k in ("q", "v", "a")))
            a1 = r_lo + 0.5 * (act[0].astype(np.float64) + 1.0) * (r_hi - r_lo)
            mism += 0 if explain_mismatches(ora, env.scene, st, dict(a=a1[None]), np.array([True]))[0] else 1
        else:
            assert abs(rew[0] - o_rew[0]) < 5e-5
            if not done[0]:
                assert (np.abs(obs[0] - o_obs[0]) < OBS_TOL).all()
        if done[0]:
            assert length <= 20 and (reason[0] == 2) == (length == 20 and reason[0] not in (4, 5)) and reason[0] in (2, 4, 5)
            episodes += 1; reasons.append(int(reason[0])); length = 0
            st = env.get_state()                              # the next episode's start state (auto-reset inside step)
            assert st["step_count"][0] == 0 and (st["q"] >= lo - 1e-6).all() and (st["q"] <= hi + 1e-6).all()
            assert (np.abs(st["v"]) <= V * 1.0001).all() and (np.abs(st["a"]) <= A * 1.0001).all()
        else:
            assert reason[0] == -1
    env.synchronize()
    assert mism == 0 and 50 <= episodes <= 1000 and 2 in reasons and 5 in reasons
    env.close()


def test_sampling_exhaustion_is_reported_through_the_abi(built):
    """ctlp.py:2127-2136 raises ValueError when no collision-free start position is found within the attempt limit; the kernels'
    bounded loops latch a device flag that the next synchronising call reports as SMR_E_SAMPLING (-> SamplingError, a ValueError)."""
    from safemotionsrisk_amd import VecSafeMotionsEnv, SPACE_ENV_CONFIG
    from safemotionsrisk_amd.lib import SamplingError
    env = VecSafeMotionsEnv(8, device=0, seed=0, auto_reset=False, **SPACE_ENV_CONFIG)
    env.close()
    for c in range(3):                                        # an end-effector box no pose can reach
        env.scene.struct.start_box[c][0], env.scene.struct.start_box[c][1] = 5.0, 5.1
    env._create()
    env.reset()
    with pytest.raises(ValueError) as exc:
        env.synchronize()
    assert isinstance(exc.value, SamplingError) and "start position" in str(exc.value)
    env.synchronize()                                         # reported once, then cleared
    env.close()


def test_step_before_reset_is_an_error_and_set_seed_rekeys(built):
    import torch
    from safemotionsrisk_amd import SafeMotionsEnv, VecSafeMotionsEnv, SPACE_ENV_CONFIG
    from safemotionsrisk_amd.lib import SmrError
    assert SafeMotionsEnv is VecSafeMotionsEnv
    env = VecSafeMotionsEnv(256, device=0, seed=3, auto_reset=True, **SPACE_ENV_CONFIG)
    act = torch.zeros((256, 7), device="cuda")
    with pytest.raises(SmrError):
        env.step(act)                                         # auto-reset needs the start states reset() prepares
    o3 = env.reset().cpu().numpy().copy()
    env.set_seed(4)
    with pytest.raises(SmrError):
        env.step(act)
    o4 = env.reset().cpu().numpy().copy()
    env.set_seed(3)
    assert np.array_equal(env.reset().cpu().numpy(), o3) and not np.array_equal(o3, o4)
    for _ in range(25):
        env.step(act)
    env.synchronize()
    env.close()


def _global_actions(n_envs, steps):
    return np.random.default_rng(100).uniform(-1, 1, (steps, n_envs, 7)).astype(np.float32)


def _shard_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from safemotionsrisk_amd import VecSafeMotionsEnv, SPACE_ENV_CONFIG
    from safemotionsrisk_amd.dist import ShardedRollout
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B = 512
        env = VecSafeMotionsEnv(B, device=0, seed=7, auto_reset=True, env_index_base=rank * B, **SPACE_ENV_CONFIG)
        roll = ShardedRollout(env, host_gather=True)          # both ranks share the one GPU of the box: RCCL needs one device per rank
        parts = roll.reset()
        rec = [roll.concatenated(parts)[0].numpy().copy()]
        acts = _global_actions(world * B, 25)                  # this rank's block of the global batch's actions
        for t in range(25):
            parts = roll.step(torch.from_numpy(acts[t, rank * B:(rank + 1) * B]).cuda())
            obs, rew, done = roll.concatenated(parts)
            rec.append(np.concatenate([obs.numpy(), rew.numpy()[:, None], done.numpy()[:, None].astype(np.float32)], axis=1))
            # this rank's slice of the gathered batch is what its env wrote
            assert torch.equal(parts[rank][0], env.obs.cpu()) and torch.equal(parts[rank][2], env.done.cpu())
        env.synchronize()
        env.close()
        np.savez(out_path % rank, *rec)
    finally:
        dist.destroy_process_group()


def test_two_real_shards_through_one_process_group(built, tmp_path):
    """two ranks, each stepping a real VecSafeMotionsEnv shard (env_index_base = rank x B) on the GPU, one packed all-gather per
    step: both ranks end up with the same concatenated (obs, reward, done) batch - and that batch is BIT FOR BIT the one a single
    handle over all 2 B envs produces (resets, auto-resets and 25 steps): shards are a partition of the global batch, the results do
    not depend on the world size ("envs shard trivially ... only to concatenate the rollout batch")."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "shard%d.npz")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
    assert [p.exitcode for p in procs] == [0, 0]
    a, b = np.load(out % 0), np.load(out % 1)
    for k in a.files:
        assert np.array_equal(a[k], b[k])
    last = a[a.files[-1]]
    assert last.shape == (1024, 25) and not np.array_equal(last[:512], last[512:])
    assert sum(a[k][:, -1].sum() for k in a.files[1:]) > 512              # episodes ended in both shards
    # the same global batch through ONE handle
    import torch
    from safemotionsrisk_amd import VecSafeMotionsEnv, SPACE_ENV_CONFIG
    env = VecSafeMotionsEnv(1024, device=0, seed=7, auto_reset=True, **SPACE_ENV_CONFIG)
    assert np.array_equal(env.reset().cpu().numpy(), a[a.files[0]])
    acts = _global_actions(1024, 25)
    for t in range(25):
        obs, rew, done, _ = env.step(torch.from_numpy(acts[t]).cuda())
        one = np.concatenate([obs.cpu().numpy(), rew.cpu().numpy()[:, None], done.cpu().numpy()[:, None].astype(np.float32)], axis=1)
        assert np.array_equal(one, a[a.files[1 + t]]), "step %d: the shards' batch differs from the single handle's" % t
    env.synchronize()
    env.close()


def _overlap_worker(port, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from safemotionsrisk_amd import VecSafeMotionsEnv, SPACE_ENV_CONFIG
    from safemotionsrisk_amd.dist import ShardedRollout
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        B, T = 1024, 12
        env = VecSafeMotionsEnv(B, device=0, seed=11, auto_reset=True, **SPACE_ENV_CONFIG)
        ref = VecSafeMotionsEnv(B, device=0, seed=11, auto_reset=True, **SPACE_ENV_CONFIG)
        roll = ShardedRollout(env, overlap=True)
        roll.reset(); ref.reset()
        gen = torch.Generator(device="cuda"); gen.manual_seed(5)
        acts = [torch.rand((B, 7), device="cuda", generator=gen) * 2 - 1 for _ in range(T)]
        ok, held = True, None
        for t in range(T):
            parts = roll.step(acts[t])                     # the gather of this step is in flight while the next one is issued
            o, r, d, _ = ref.step(acts[t])
            expect = (o.clone(), r.clone(), d.clone())
            if held is not None:                           # read a step's batch one step later, before its slot is written again
                roll.wait()
                torch.cuda.synchronize()
                (po, pr, pd), (eo, er, ed) = held[0][0], held[1]
                ok &= bool(torch.equal(po, eo) and torch.equal(pr, er) and torch.equal(pd, ed))
            held = (parts, expect)
        roll.wait(); torch.cuda.synchronize()
        ok &= bool(torch.equal(held[0][0][0], held[1][0]))
        # the self-check bench.py --gpus N runs before its timed region (checksums through RCCL, int64 on the device)
        ok &= roll.verify([acts[0], acts[1]]) == []
        env.synchronize(); ref.synchronize()
        env.close(); ref.close()
        with open(out_path, "w") as f:
            f.write("ok" if ok else "mismatch")
    finally:
        dist.destroy_process_group()


def test_overlapped_gather_on_the_device(built, tmp_path):
    """ShardedRollout(overlap=True) through RCCL (a one-rank group: the box has one GPU): the env alternates between two packed
    buffers, the all-gather of a step runs asynchronously beside the next step's kernels, and what is read one step later equals
    the batch of an identical env stepped without the rollout wrapper."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "overlap.txt")
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_overlap_worker, args=(port, out))
    p.start()
    p.join(timeout=300)
    assert p.exitcode == 0 and open(out).read() == "ok"


def test_bench_refuses_more_ranks_than_gpus(built):
    """`python bench.py --gpus 2` without a launcher starts its own two ranks; on a one-GPU box rank 1 has no device and the
    command must fail loudly instead of printing a one-GPU line."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--envs-per-gpu", "1024"], capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "n_gpus" not in p.stdout
