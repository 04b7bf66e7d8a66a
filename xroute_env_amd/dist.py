"""Multi-GPU sharding of the env batch: one process per GPU, envs statically partitioned, and ONE small
RCCL collective per step — the batched-env gather of compact per-env results.

Regions are independent, so the data path has no exchange step inside a step.  Each rank
owns a contiguous block of env ids for the whole run; after `step` the learner-facing record of every env
(reward, metric deltas, done, nets left: 6 x f64 = 48 B/env, ~200 KB at 4096 envs) is all-gathered.
fp32 observations are NOT gathered (2.5 MB/env): they stay on the GPU that owns the env.

`torch.distributed` backend "nccl" is RCCL on ROCm; the same code runs on "gloo" with CPU tensors, which
is how the world_size-2 tests exercise it without GPUs.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist

RECORD_FIELDS = ("reward", "d_violation", "d_wirelength", "d_via", "done", "nlegal")
RECORD_WIDTH = len(RECORD_FIELDS)


def shard_range(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block partition of env ids; the first (n_total % world) ranks get one more."""
    if world < 1 or not (0 <= rank < world) or n_total < 0:
        raise ValueError("bad shard arguments")
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def env_seed(config: int, env_id: int) -> int:
    """Generator seed of a global env id (seed = 1000*config + env id)."""
    return 1000 * config + env_id


def pack_records(reward: torch.Tensor, delta: torch.Tensor, done: torch.Tensor, nlegal: torch.Tensor,
                 out: torch.Tensor = None) -> torch.Tensor:
    """[B, 6] float64 record (all values are small integers or halves: exact in f64)."""
    b = reward.shape[0]
    if out is None:
        out = torch.empty((b, RECORD_WIDTH), dtype=torch.float64, device=reward.device)
    out[:, 0] = reward
    out[:, 1:4] = delta.to(torch.float64)
    out[:, 4] = done.to(torch.float64)
    out[:, 5] = nlegal.to(torch.float64)
    return out


def gather_records(local: torch.Tensor, out: torch.Tensor = None, group=None) -> torch.Tensor:
    """All-gather the per-env records of every rank, in rank (= env id) order.  Equal shard sizes use
    one all_gather_into_tensor; ragged shards fall back to all_gather on padded blocks."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local if out is None else out.copy_(local)
    world = dist.get_world_size(group)
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    if len(set(sizes)) == 1:
        if out is None:
            out = torch.empty((world * sizes[0], local.shape[1]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    m = max(sizes)
    pad = torch.zeros((m, local.shape[1]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    blocks = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(blocks, pad, group=group)
    return torch.cat([blk[:s] for blk, s in zip(blocks, sizes)], dim=0)


def gather_records_fixed(local: torch.Tensor, out: torch.Tensor, group=None) -> torch.Tensor:
    """Hot-loop variant for equal shards: no size exchange, no allocation, one RCCL call."""
    dist.all_gather_into_tensor(out, local, group=group)
    return out


class ShardedVectorEnv:
    """The rank-local slice of a global batch of `n_total` envs of BASELINE config `config`."""

    def __init__(self, config: int, n_total: int, device=None, with_observation: bool = True, **batch_kw):
        from .envs.vector_env import XRouteVectorEnv
        from .regions import CONFIGS, generate_region
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.lo, self.hi = shard_range(n_total, self.world, self.rank)
        regions = [generate_region(env_seed(config, e), **CONFIGS[config]) for e in range(self.lo, self.hi)]
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.env = XRouteVectorEnv(regions, device=device, with_observation=with_observation, **batch_kw)
        self.n_local = self.hi - self.lo
        self.n_total = n_total
        self._rec = torch.empty((self.n_local, RECORD_WIDTH), dtype=torch.float64, device=self.env.device)
        equal = n_total % self.world == 0
        self._all = torch.empty((n_total, RECORD_WIDTH), dtype=torch.float64, device=self.env.device) if equal else None

    def reset(self):
        return self.env.reset()

    def step(self, actions: torch.Tensor):
        """Local step + global gather.  Returns (obs_local, records_all [n_total, 6], info_local)."""
        obs, reward, done, info = self.env.step(actions)
        pack_records(reward, info["delta"], done, info["nlegal"], self._rec)
        if self.world == 1:
            return obs, self._rec, info
        if self._all is not None:
            return obs, gather_records_fixed(self._rec, self._all), info
        return obs, gather_records(self._rec), info
