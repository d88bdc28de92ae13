"""N > 1 path on CPU: two gloo ranks, one packed all-gather of (obs, reward, done) per step."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class StubEnv:
    """Stands in for VecSafeMotionsEnv on a CPU: deterministic outputs that depend on rank and step."""

    def __init__(self, rank, num_envs, obs_size):
        self.rank, self.num_envs, self.observation_size, self.t = rank, num_envs, obs_size, 0
        self.obs = torch.zeros(num_envs, obs_size)
        self.reward = torch.zeros(num_envs)
        self.done = torch.zeros(num_envs, dtype=torch.uint8)

    def _fill(self):
        base = 1000 * self.rank + self.t
        self.obs.copy_(torch.arange(self.num_envs * self.observation_size, dtype=torch.float32).view(self.num_envs, -1) + base)
        self.reward.copy_(torch.arange(self.num_envs, dtype=torch.float32) * 0.5 + base)
        self.done.copy_(((torch.arange(self.num_envs) + self.t + self.rank) % 3 == 0).to(torch.uint8))

    def reset(self):
        self.t = 0
        self._fill()

    def step(self, action):
        self.t += 1
        self._fill()


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from safemotionsrisk_amd.dist import ShardedRollout, shard_range
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(10, rank, world)
        env = StubEnv(rank, hi - lo, 23)
        roll = ShardedRollout(env)
        parts = roll.reset()
        ok = len(parts) == world
        for t in range(1, 4):
            parts = roll.step(torch.zeros(env.num_envs, 7))
            for r in range(world):
                ref = StubEnv(r, env.num_envs, 23)
                ref.t = t
                ref._fill()
                ok &= torch.equal(parts[r][0], ref.obs) and torch.equal(parts[r][1], ref.reward) and torch.equal(parts[r][2], ref.done)
        # overlapped mode: the gather of a step is still in flight when the next step is issued; two slots alternate
        env2 = StubEnv(rank, hi - lo, 23)
        roll2 = ShardedRollout(env2, overlap=True)
        roll2.reset()
        held = []
        for t in range(1, 6):
            held.append((t, roll2.step(torch.zeros(env2.num_envs, 7))))
            if len(held) == 2:                          # read a step's batch one step later, as a pipelined learner would
                roll2.wait()
                tt, pp = held.pop(0)
                for r in range(world):
                    ref = StubEnv(r, env2.num_envs, 23)
                    ref.t = tt
                    ref._fill()
                    ok &= torch.equal(pp[r][0], ref.obs) and torch.equal(pp[r][1], ref.reward) and torch.equal(pp[r][2], ref.done)
        roll2.wait()
        # a consumer that reads the returned batch at once, without wait(): the first access waits for the collective itself
        batch = roll2.step(torch.zeros(env2.num_envs, 7))
        ok &= type(batch).__name__ == "GatheredBatch" and len(batch) == world
        for r in range(world):
            ref = StubEnv(r, env2.num_envs, 23)
            ref.t = 6
            ref._fill()
            ok &= torch.equal(batch[r][0], ref.obs) and torch.equal(batch[r][2], ref.done)
        ok &= batch.work is None                        # consumed by the read
        ok &= env2.obs.data_ptr() in (roll2._locals[0].flat.data_ptr(), roll2._locals[1].flat.data_ptr())
        # the self-check bench.py runs before its timed region: sound exchanges pass in both modes ...
        for overlap in (False, True):
            env3 = StubEnv(rank, hi - lo, 23)
            roll3 = ShardedRollout(env3, overlap=overlap)
            roll3.reset()
            ok &= roll3.verify([torch.zeros(env3.num_envs, 7)] * 2) == []
        # ... and a rank whose gathered copy of rank 1's slice differs from what rank 1 wrote is reported (here: rank 0 corrupts its copy)
        env4 = StubEnv(rank, hi - lo, 23)
        roll4 = ShardedRollout(env4, overlap=True)
        roll4.reset()
        orig = roll4.dist.all_gather_into_tensor

        class Corrupting:
            def __init__(self, d): self.d = d
            def __getattr__(self, k): return getattr(self.d, k)
            def all_gather_into_tensor(self, out, inp, group=None, async_op=False):
                w = orig(out, inp, group=group, async_op=False)
                if rank == 0 and out.dtype == torch.uint8:
                    out[inp.numel() + 5] ^= 0x40                       # one bit of rank 1's observations, as received by rank 0
                class Done:
                    def wait(self): return True
                return Done() if async_op else w
        roll4.dist = Corrupting(roll4.dist)
        found = roll4.verify([torch.zeros(env4.num_envs, 7)] * 2)
        ok &= (found == [(0, 1), (1, 1)]) if rank == 0 else (found == [])
        obs, rew, done = roll.concatenated(parts)
        ok &= obs.shape == (world * env.num_envs, 23) and rew.shape == (world * env.num_envs,) and done.dtype == torch.uint8
        # the env writes straight into the packed buffer (no staging copy)
        ok &= env.obs.data_ptr() == roll.local.flat.data_ptr()
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_packed_all_gather_two_ranks():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, True), (1, True)]


def test_shard_range_partitions_everything():
    from safemotionsrisk_amd.dist import shard_range
    for total, world in [(524288, 8), (10, 3), (7, 8)]:
        spans = [shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
