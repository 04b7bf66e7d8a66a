"""Multi-GPU sharding of the env batch: one process per GPU, envs statically partitioned, and ONE small
RCCL collective per step — the batched-env gather of compact per-env results (plus, in the learner flow of BASELINE
config 4, one broadcast of the actions going the other way).

Regions are independent, so the data path has no exchange step inside a step.  Each rank
owns a contiguous block of env ids for the whole run; after `step` the learner-facing record of every env
is all-gathered: the 48-byte `xr_step_record` the step kernels write themselves (reward f64, metric deltas and
cumulative metrics i32[3] each, nets left, step counter, path length, done, status: include/xroute_hip.h) — 196 KB at 4096
envs, latency-bound, no packing kernels.  fp32 observations are NOT gathered (2.5 MB/env): they stay on the GPU that
owns the env.

Learner flow (reference caller loop baseline/PPO/train_PPO.py:96-102: `action = agent.select_action(state)`;
`state, done, ... = game.step(action)`): records (+ legal-net bitmasks) are gathered, the policy runs on rank 0 over ALL
envs, the chosen actions travel back as one i32[n_total] broadcast and every rank steps its own slice.

`torch.distributed` backend "nccl" is RCCL on ROCm; the same code runs on "gloo" with CPU tensors, which
is how the world_size-2 tests exercise it without GPUs.
"""
from __future__ import annotations

import os
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist

RECORD_BYTES = 48
# int32 column view of a record row ([B, 48] uint8 -> [B, 12] int32; reward = float64 view column 0)
REC_DELTA, REC_CUM, REC_NLEGAL, REC_ENV_STEPS, REC_PATH_LEN, REC_FLAGS = slice(2, 5), slice(5, 8), 8, 9, 10, 11
RECORD_FIELDS = ("reward", "delta", "cum", "nlegal", "env_steps", "path_len", "done", "status")


def collectives_on(group=None) -> bool:
    """True when the collective code paths run: a process group exists and holds more than one rank — or exactly one with
    XR_FORCE_COLLECTIVES=1, the switch that makes a ONE-rank job take every N > 1 branch (init with a device id, the async all_gather
    pairs, verify_gather, broadcast, all_reduce, the learner's gather) on the real backend.  With one GPU that is the only way to execute
    the RCCL code path before an 8-GPU node does (tests/test_gpu_bench_contract.py: the forced-nccl lines)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("XR_FORCE_COLLECTIVES") == "1"


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


def unpack_records(rec: torch.Tensor) -> dict:
    """Views (no copies) of a [n, 48] uint8 record tensor: reward f64[n], delta / cum i32[n,3], nlegal, env_steps,
    path_len i32[n], done u8[n], status i32[n]."""
    if rec.dtype != torch.uint8 or rec.dim() != 2 or rec.shape[1] != RECORD_BYTES or not rec.is_contiguous():
        raise ValueError("records must be a contiguous [n, 48] uint8 tensor")
    i32 = rec.view(torch.int32)
    flags = i32[:, REC_FLAGS]
    return {"reward": rec.view(torch.float64)[:, 0], "delta": i32[:, REC_DELTA], "cum": i32[:, REC_CUM],
            "nlegal": i32[:, REC_NLEGAL], "env_steps": i32[:, REC_ENV_STEPS], "path_len": i32[:, REC_PATH_LEN],
            "done": (flags & 0xFF).to(torch.uint8), "status": (flags >> 16) & 0xFFFF}


def pack_records(reward, delta, done, nlegal, cum=None, env_steps=None, path_len=None, status=None,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Host-side packer with the kernels' layout (tests and CPU stand-ins; the GPU path never packs: the step kernels
    write the records)."""
    n = reward.shape[0]
    if out is None:
        out = torch.zeros((n, RECORD_BYTES), dtype=torch.uint8, device=reward.device)
    i32 = out.view(torch.int32)
    out.view(torch.float64)[:, 0] = reward.to(torch.float64)
    i32[:, REC_DELTA] = delta.to(torch.int32)
    i32[:, REC_CUM] = 0 if cum is None else cum.to(torch.int32)
    i32[:, REC_NLEGAL] = nlegal.to(torch.int32)
    i32[:, REC_ENV_STEPS] = 0 if env_steps is None else env_steps.to(torch.int32)
    i32[:, REC_PATH_LEN] = 0 if path_len is None else path_len.to(torch.int32)
    st = torch.zeros(n, dtype=torch.int32, device=reward.device) if status is None else status.to(torch.int32)
    i32[:, REC_FLAGS] = (done.to(torch.int32) & 0xFF) | (st << 16)
    return out


def gather_rows(local: torch.Tensor, out: Optional[torch.Tensor] = None, group=None) -> torch.Tensor:
    """All-gather per-env rows of every rank, in rank (= env id) order.  Equal shard sizes use one
    all_gather_into_tensor; ragged shards fall back to all_gather on padded blocks."""
    if not collectives_on(group):
        return local if out is None else out.copy_(local)
    world = dist.get_world_size(group)
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    if len(set(sizes)) == 1:
        if out is None:
            out = torch.empty((world * sizes[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    m = max(sizes)
    pad = torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    blocks = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(blocks, pad, group=group)
    return torch.cat([blk[:s] for blk, s in zip(blocks, sizes)], dim=0)


gather_records = gather_rows


def gather_records_fixed(local: torch.Tensor, out: torch.Tensor, group=None) -> torch.Tensor:
    """Hot-loop variant for equal shards: no size exchange, no allocation, one RCCL call."""
    dist.all_gather_into_tensor(out, local, group=group)
    return out


def rows_checksum(rows: torch.Tensor) -> torch.Tensor:
    """Position-weighted int64 checksum of a [n, w] uint8 tensor (a swap of two rows or two bytes changes it); device-only."""
    flat = rows.reshape(-1).to(torch.int64)
    w = (torch.arange(flat.numel(), device=rows.device, dtype=torch.int64) % 65521) + 1
    return (flat * w).sum().reshape(1)


def verify_gather(local: torch.Tensor, gathered: torch.Tensor, lo: int, group=None) -> dict:
    """Self-certification of the batched-env gather: every rank sends (rank, lo, rows, checksum of its LOCAL records); every rank
    then recomputes the checksum of the matching slice of what the collective delivered to IT.  Returns
    {"ranks_seen": distinct ranks that answered, "gather_verified": every slice on every rank equals its owner's local rows,
     "rows": total rows}.  One tiny all_gather + one all_reduce; call it outside the timed region."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    dev = local.device
    mine = torch.cat([torch.tensor([rank, lo, local.shape[0]], dtype=torch.int64, device=dev), rows_checksum(local)])
    if not collectives_on(group):
        ok = bool(torch.equal(local, gathered[lo:lo + local.shape[0]]))
        return {"ranks_seen": 1, "gather_verified": ok, "rows": int(local.shape[0])}
    table = torch.empty((world, 4), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(table, mine.reshape(1, 4), group=group)
    t = table.cpu().tolist()
    ok = True
    for r_, lo_, n_, sum_ in t:
        if lo_ < 0 or lo_ + n_ > gathered.shape[0] or int(rows_checksum(gathered[lo_:lo_ + n_]).item()) != sum_:
            ok = False
    covered = sorted((lo_, lo_ + n_) for _, lo_, n_, _ in t)
    ok = ok and covered[0][0] == 0 and covered[-1][1] == gathered.shape[0] and all(a[1] == b_[0] for a, b_ in zip(covered, covered[1:]))
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return {"ranks_seen": len({r_ for r_, *_ in t}), "gather_verified": bool(flag.item() == 1), "rows": int(sum(n_ for _, _, n_, _ in t))}


def first_legal_policy(records: dict, legal: torch.Tensor) -> torch.Tensor:
    """A deterministic stand-in learner policy: the lowest legal net of every env (0 when none).  legal: int64[n, words],
    bit n-1 of the row <=> net n in netSet."""
    n, words = legal.shape
    bits = torch.arange(64, device=legal.device, dtype=torch.int64)
    m = ((legal.unsqueeze(-1) >> bits) & 1).reshape(n, words * 64)            # [n, 64*words] 0/1
    has = m.any(dim=1)
    first = torch.argmax(m, dim=1).to(torch.int32) + 1
    return torch.where(has, first, torch.zeros_like(first))


def random_legal_policy(records: dict, legal: torch.Tensor, seed: int, env_ids: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Random net-order policy over gathered state, in torch ops on the learner's device: a uniformly chosen legal net
    of every env (0 when none) from a counter-based hash of (seed, env, env_steps) — reproducible, no host sync.
    `env_ids` (int64 [n]): the GLOBAL id of every row (default: its index) — a rank that evaluates this on its own shard then chooses
    what a one-process run chooses for the same env."""
    n, words = legal.shape
    dev = legal.device
    bits = torch.arange(64, device=dev, dtype=torch.int64)
    m = ((legal.unsqueeze(-1) >> bits) & 1).reshape(n, words * 64)            # [n, 64*words] 0/1
    k = m.sum(dim=1)
    ids = torch.arange(n, device=dev, dtype=torch.int64) if env_ids is None else env_ids.to(device=dev, dtype=torch.int64)
    x = (ids * 0x100000001B3 + records["env_steps"].to(torch.int64)) ^ int(seed)
    x = (x ^ (x >> 30) & 0x3FFFFFFFF) * 0x3F58476D1CE4E5B9
    x = (x ^ ((x >> 27) & 0x1FFFFFFFFF)) * 0x14D049BB133111EB
    j = (x & 0x7FFFFFFF) % torch.clamp(k, min=1)
    csum = torch.cumsum(m, dim=1)
    pick = torch.argmax(((csum == (j + 1).unsqueeze(1)) & (m == 1)).to(torch.int8), dim=1).to(torch.int32) + 1
    return torch.where(k > 0, pick, torch.zeros_like(pick))


class ShardedVectorEnv:
    """The rank-local slice of a global batch of `n_total` envs of BASELINE config `config`.

    `env_factory(regions, device, **kw)` builds the local vector env (default: XRouteVectorEnv on the MI355X); the tests
    pass a CPU stand-in with the same surface, so the collective logic below is exactly what runs on RCCL."""

    def __init__(self, config: int, n_total: int, device=None, with_observation: bool = True,
                 env_factory: Optional[Callable] = None, group=None, **batch_kw):
        from .regions import CONFIGS, generate_region
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.multi = collectives_on(group)
        self.lo, self.hi = shard_range(n_total, self.world, self.rank)
        regions = [generate_region(env_seed(config, e), **CONFIGS[config]) for e in range(self.lo, self.hi)]
        if env_factory is None:
            from .envs.vector_env import XRouteVectorEnv
            if device is None:
                device = torch.device("cuda", torch.cuda.current_device())
            env_factory = XRouteVectorEnv
        self.env = env_factory(regions, device=device, with_observation=with_observation, **batch_kw)
        self.device = self.env.device
        self.n_local = self.hi - self.lo
        self.n_total = n_total
        self.equal = n_total % self.world == 0
        self._all = torch.empty((n_total, RECORD_BYTES), dtype=torch.uint8, device=self.device) if self.equal else None
        self._actions_all = torch.zeros(n_total, dtype=torch.int32, device=self.device)
        self._legal_all = None          # [n_total, legal_words] int64, allocated at the first exchange (equal shards: one collective, no host wait)

    def reset(self):
        return self.env.reset()

    def _gather(self, rows: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if not self.multi:
            return rows
        if self.equal and out is not None:
            return gather_records_fixed(rows, out, self.group)
        return gather_rows(rows, group=self.group)

    def step(self, actions: torch.Tensor):
        """Local step + global gather.  Returns (obs_local, records_all uint8[n_total, 48], info_local)."""
        obs, reward, done, info = self.env.step(actions)
        return obs, self._gather(info["record"], self._all), info

    def compact_exchange(self, learner_batch=None) -> "CompactStateExchange":
        """The compact-state gather of a central learner for this sharded env (CompactStateExchange below): every rank's envs packed to
        ~1 KB rows, one all_gather, expansion to head rows on the learner — for a policy that reads the grid (the DQN / PPO counterparts),
        where `learner_step`'s records + legal masks are enough for policies that do not.  `learner_batch` (learner rank): a RegionBatch
        holding the regions of ALL ranks in global env order (region of global env g at index g)."""
        return CompactStateExchange(self.env.batch, self.n_total, self.lo, region_base=self.lo, learner_batch=learner_batch, group=self.group)

    # ---- learner flow (BASELINE config 4) ------------------------------------------------------------------------
    def learner_reset(self, policy: Callable = first_legal_policy):
        """reset + the first action exchange.  Returns (obs_local, actions_local)."""
        obs, info = self.env.reset()
        return obs, self._exchange(info, policy)

    def learner_step(self, actions_local: torch.Tensor, policy: Callable = first_legal_policy):
        """One step of the config 4 loop: step the local slice with the actions the learner sent, gather every env's
        record and legal set, let rank 0 choose the next action of ALL envs, broadcast them (i32[n_total]) and return
        (obs_local, records_all, next_actions_local, info_local)."""
        obs, reward, done, info = self.env.step(actions_local)
        nxt = self._exchange(info, policy)
        return obs, self._records_all, nxt, info

    def _exchange(self, info: dict, policy: Callable) -> torch.Tensor:
        rec_all = self._gather(info["record"], self._all)
        if self.equal and self.multi and (self._legal_all is None or self._legal_all.shape[1] != info["legal"].shape[1]):
            self._legal_all = torch.empty((self.n_total, info["legal"].shape[1]), dtype=info["legal"].dtype, device=self.device)
        legal_all = self._gather(info["legal"], self._legal_all)
        self._records_all = rec_all
        if self.rank == 0:
            self._actions_all.copy_(policy(unpack_records(rec_all), legal_all).to(torch.int32))
        if self.multi:
            dist.broadcast(self._actions_all, src=0, group=self.group)      # actions travel the other way
        return self._actions_all[self.lo:self.hi].contiguous()


class CompactStateExchange:
    """The compact-state gather of a CENTRAL learner (BASELINE config 4's `--learner` placement): every rank packs what planes 0..1
    of its envs' observations are functions of (`RegionBatch.pack_state`: region, nets left, legal bitmask, one occupancy bit per node), ONE
    collective carries the rows TO THE LEARNER, which expands them to the fp32 head rows its policy reads (`RegionBatch.expand_state`,
    byte-identical to what `step_compact` writes).  fp32 planes are never gathered.

    The collective is a `gather` to the learner rank (round 6; it was an all_gather): only the learner reads the rows, so every other rank
    sends its n_local x row_bytes once over its own xGMI link to the learner and receives nothing — 1/N of what an all_gather moves per
    link and (N - 1)/N less in total (`bytes_per_link`, `bytes_per_step`).  north_star keeps RCCL "only for the batched-env gather": the
    all_gather that remains in the design is the one of the 48-byte records.

    batch          the rank's RegionBatch (its shard)
    n_total        env slots over all ranks; `lo` = global id of this rank's first env (contiguous shards, `shard_range`)
    region_base    what turns the shard's local region index into an index of the learner's region table (0 when every rank loaded the same regions)
    learner_batch  on the learner rank: a RegionBatch whose region table holds every region of the job (default: `batch`)"""

    def __init__(self, batch, n_total: int, lo: int, region_base: int = 0, learner_batch=None, group=None, learner_rank: int = 0):
        self.batch, self.learner = batch, (learner_batch if learner_batch is not None else batch)
        self.n_total, self.lo, self.region_base, self.group = int(n_total), int(lo), int(region_base), group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.multi = collectives_on(group)
        self.learner_rank = int(learner_rank)
        rb = max(batch.state_row_bytes(), self.learner.state_row_bytes())
        if self.multi:                           # one row size for everybody (ranks may hold regions with different net counts)
            t = torch.tensor([rb], dtype=torch.int64, device=batch.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            rb = int(t.item())
        self.row_bytes = rb
        self.rows_local = torch.empty((batch.n_envs, rb), dtype=torch.uint8, device=batch.device)
        self.equal = self.multi and self.n_total == self.world * batch.n_envs
        on_learner = self.rank == self.learner_rank
        self.rows_all = torch.empty((self.n_total, rb), dtype=torch.uint8, device=batch.device) if self.equal and on_learner else None
        self._sizes = None                       # ragged shards: rows of every rank (exchanged once)

    @property
    def bytes_per_step(self) -> int:
        """rows the learner holds after the gather, in bytes (its own shard included: that part never leaves its GPU)"""
        return self.n_total * self.row_bytes

    @property
    def bytes_per_link(self) -> int:
        """bytes one non-learner rank puts on its link to the learner per step (the all_gather it replaces: (N - 1) / N x n_total x row_bytes per ring link)"""
        return self.batch.n_envs * self.row_bytes

    def gather(self, to_all: bool = False) -> Optional[torch.Tensor]:
        """pack this rank's envs and send the rows to the learner: uint8 [n_total, row_bytes] in global env order on the learner rank, None on the
        others.  `to_all=True`: the round-5 form (an all_gather, every rank gets the rows)."""
        self.batch.pack_state(self.rows_local, region_base=self.region_base)
        if not self.multi:
            return self.rows_local
        if to_all:
            return gather_rows(self.rows_local, group=self.group)
        on_learner = self.rank == self.learner_rank
        if self.rows_local.is_cuda and dist.get_backend(self.group) == "gloo":
            # (gloo has no `gather` for device tensors — the two-ranks-on-one-GPU functional tests run on it: its all_gather, the learner keeps the rows)
            rows = gather_rows(self.rows_local, group=self.group)
            return rows if on_learner else None
        if self.equal:
            # (chunks of a contiguous [n_total, rb] tensor are contiguous views: the learner receives straight into global env order)
            dist.gather(self.rows_local, gather_list=list(self.rows_all.chunk(self.world)) if on_learner else None, dst=self.learner_rank, group=self.group)
            return self.rows_all if on_learner else None
        if self._sizes is None:                  # ragged shards: sizes once, then padded blocks
            n = torch.tensor([self.rows_local.shape[0]], dtype=torch.int64, device=self.rows_local.device)
            sizes = [torch.zeros_like(n) for _ in range(self.world)]
            dist.all_gather(sizes, n, group=self.group)
            self._sizes = [int(v.item()) for v in sizes]
        m = max(self._sizes)
        pad = torch.zeros((m, self.row_bytes), dtype=torch.uint8, device=self.rows_local.device)
        pad[: self.rows_local.shape[0]] = self.rows_local
        blocks = [torch.empty_like(pad) for _ in range(self.world)] if on_learner else None
        dist.gather(pad, gather_list=blocks, dst=self.learner_rank, group=self.group)
        return torch.cat([blk[:n_] for blk, n_ in zip(blocks, self._sizes)], dim=0) if on_learner else None

    def expand(self, rows: torch.Tensor, head_out=None, nlegal_out=None, region_out=None):
        """learner rank: rows -> (head fp32 [n, 2 * n_max], nlegal int32 [n], region int32 [n]); rows that do not parse are flagged -1"""
        return self.learner.expand_state(rows, head_out, nlegal_out, region_out)
