"""Rollout sharding over the GPUs of one node (BASELINE config 5).

Environments are independent, so rank g of G owns the contiguous block [g*B_g, (g+1)*B_g) of the global batch with
its own handle and stream (``VecSafeMotionsEnv(B_g, env_index_base=g * B_g)``: every draw of an env is keyed by
(seed, global env index, its counters), so the shards are a partition of the global batch and their concatenation does
not depend on the world size); there is no collective on the data path of step().  The only exchange is
the concatenation of the rollout batch: ONE all-gather per step of a packed byte buffer
    [ obs float32 [B_g][O] | reward float32 [B_g] | done uint8 [B_g] | pad to 16 B ]
(backend "nccl" is RCCL on ROCm; 7 xGMI links per GPU, so the gather is per-link bound: one call with
the three arrays packed instead of three).  The reference has no counterpart (Ray actor RPC, train.py:598-648).
"""
from __future__ import annotations


def shard_range(total, rank, world):
    """Contiguous block of `total` envs owned by `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class PackedRollout:
    """obs/reward/done buffers of one shard laid out in a single flat byte tensor."""

    def __init__(self, torch, num_envs, obs_size, device):
        self.torch, self.B, self.O = torch, int(num_envs), int(obs_size)
        self.obs_bytes = self.B * self.O * 4
        self.rew_bytes = self.B * 4
        raw = self.obs_bytes + self.rew_bytes + self.B
        self.nbytes = (raw + 15) // 16 * 16
        self.flat = torch.zeros(self.nbytes, dtype=torch.uint8, device=device)
        self.obs, self.reward, self.done = self.views(self.flat)

    def views(self, flat):
        t = self.torch
        obs = flat[:self.obs_bytes].view(t.float32).view(self.B, self.O)
        rew = flat[self.obs_bytes:self.obs_bytes + self.rew_bytes].view(t.float32)
        done = flat[self.obs_bytes + self.rew_bytes:self.obs_bytes + self.rew_bytes + self.B]
        return obs, rew, done


class GatheredBatch:
    """What ShardedRollout.step() returns in overlap mode: the per-rank (obs, reward, done) views of a gather that may still be in
    flight.  Any read access (indexing, iteration, len) first makes the current stream wait for the collective, so a consumer
    cannot see a partly written batch by accident; ``work`` is the collective's handle for callers that schedule the wait themselves."""

    def __init__(self, parts, work):
        self._parts, self.work = parts, work

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        return self._parts

    def __getitem__(self, i):
        return self.wait()[i]

    def __iter__(self):
        return iter(self.wait())

    def __len__(self):
        return len(self._parts)


class ShardedRollout:
    """step() the local shard, then all-gather (obs, reward, done) of every shard in one collective.

    `env` is any object with num_envs, observation_size, obs/reward/done tensors and step()/reset()
    (VecSafeMotionsEnv on a GPU).  `group` is a torch.distributed process group (None: default group).
    """

    def __init__(self, env, group=None, equal_shards=True, host_gather=False, overlap=False):
        """host_gather: the consumer of the rollout lives on the host (a learner that wants numpy batches, or a process group
        without a device backend such as gloo): the packed shard is copied to pinned host memory once per step and the
        all-gather runs on host buffers; the gathered views are then CPU tensors.

        overlap: the gather of step t runs beside the kernels of step t + 1 (xGMI is point to point: 7 x 6.4 MB arrive per GPU and
        step at 8 x 65 536 envs - a quarter of the step time if the step waited for it).  Two packed buffers alternate: the env
        writes slot t & 1 while the collective still reads the other one; step() returns the views of its gathered slot at once,
        they are complete after wait() (or after the step() after next, which waits for the slot before it is written again)."""
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.env, self.group = torch, dist, env, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.local = PackedRollout(torch, env.num_envs, env.observation_size, env.obs.device)
        # the env writes straight into the packed views: no staging copy before the collective
        env.obs, env.reward, env.done = self.local.obs, self.local.reward, self.local.done
        if not equal_shards:
            raise NotImplementedError("all_gather_into_tensor needs equal shards; pad num_envs to a multiple of world")
        self.host_gather = bool(host_gather)
        self.overlap = bool(overlap) and not self.host_gather
        out_device = "cpu" if self.host_gather else env.obs.device
        self.gathered = torch.zeros(self.world * self.local.nbytes, dtype=torch.uint8, device=out_device)
        if self.overlap:
            self._locals = [self.local, PackedRollout(torch, env.num_envs, env.observation_size, env.obs.device)]
            self._gathereds = [self.gathered, torch.zeros_like(self.gathered)]
            self._work = [None, None]
            self._slot = 0
        self._host = None
        if self.host_gather:
            self._host = torch.zeros(self.local.nbytes, dtype=torch.uint8)
            if env.obs.device.type == "cuda":
                self._host = self._host.pin_memory()

    def _gather(self):
        if self.host_gather:
            self._host.copy_(self.local.flat)                       # device -> pinned host (synchronises with the env's stream)
            self.dist.all_gather_into_tensor(self.gathered, self._host, group=self.group)
        else:
            self.dist.all_gather_into_tensor(self.gathered, self.local.flat, group=self.group)
        n = self.local.nbytes
        return [self.local.views(self.gathered[r * n:(r + 1) * n]) for r in range(self.world)]

    def reset(self):
        if self.overlap:
            self.wait()
            self.env.obs, self.env.reward, self.env.done = self.local.obs, self.local.reward, self.local.done
            self._slot = 1                              # reset() used slot 0 synchronously
            self._work = [None, None]
        self.env.reset()
        return self._gather()

    def step(self, action):
        """action: float32 [B_g, 7] for the local shard.  Returns [(obs, reward, done)] per rank (overlap: a GatheredBatch, whose
        first read access waits for the collective)."""
        if not self.overlap:
            self.env.step(action)
            return self._gather()
        i = self._slot
        if self._work[i] is not None:
            self._work[i].wait()                        # the gather that read this slot two steps ago (long done)
        loc = self._locals[i]
        self.env.obs, self.env.reward, self.env.done = loc.obs, loc.reward, loc.done
        self.env.step(action)
        self._work[i] = self.dist.all_gather_into_tensor(self._gathereds[i], loc.flat, group=self.group, async_op=True)
        self._slot = i ^ 1
        n = loc.nbytes
        return GatheredBatch([loc.views(self._gathereds[i][r * n:(r + 1) * n]) for r in range(self.world)], self._work[i])

    def wait(self):
        """overlap: every gather in flight has landed (device backends: the current stream waits, not the host)"""
        if self.overlap:
            for w in self._work:
                if w is not None:
                    w.wait()

    def _checksums(self, obs, rew, done):
        """exact, order-independent sums of the raw bytes of one shard's (obs, reward, done): int64 [3]"""
        t = self.torch
        return t.stack([obs.contiguous().view(t.int32).sum(dtype=t.int64), rew.contiguous().view(t.int32).sum(dtype=t.int64),
                        done.sum(dtype=t.int64)])

    def verify(self, actions):
        """Self-check of the exchange, meant to run before anything is timed or consumed: two steps (in overlap mode the gather of the
        first is in flight while the second runs), then every rank's checksums of its LOCAL (obs, reward, done) are exchanged in one small
        all-gather of 3 x int64 per rank and compared with the checksums of the slice every other rank received.  Returns the list of
        (step, rank) pairs whose slice on THIS rank differs from what that rank wrote (empty: the exchange is sound).  A broken
        overlap - a slot written while its collective still reads it, a gather that has not landed when it is read - shows up here
        as a mismatch instead of as a silently wrong batch."""
        t, bad = self.torch, []
        pending = []
        for a in actions[:2]:
            loc = self._locals[self._slot] if self.overlap else self.local
            pending.append((self.step(a), loc))
            if not self.overlap:                        # (synchronous mode re-uses one buffer: check each step before the next overwrites it)
                bad += self._compare(len(pending) - 1, *pending[-1])
        if self.overlap:
            self.wait()
            for k, (batch, loc) in enumerate(pending):
                bad += self._compare(k, batch, loc)
        return bad

    def _compare(self, k, batch, loc):
        t = self.torch
        parts = batch.wait() if isinstance(batch, GatheredBatch) else batch
        mine = self._checksums(loc.obs, loc.reward, loc.done)
        if self.host_gather:
            mine = mine.cpu()
        theirs = t.zeros(self.world * 3, dtype=t.int64, device=mine.device)
        self.dist.all_gather_into_tensor(theirs, mine, group=self.group)
        theirs = theirs.view(self.world, 3).cpu()
        return [(k, r) for r in range(self.world) if not t.equal(self._checksums(*parts[r]).cpu(), theirs[r])]

    def concatenated(self, parts):
        t = self.torch
        return (t.cat([p[0] for p in parts]), t.cat([p[1] for p in parts]), t.cat([p[2] for p in parts]))
