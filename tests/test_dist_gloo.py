"""N > 1 path on CPU: world_size-2 gloo processes shard the env batch, step their shard (the oracle stands
in for the GPU env here — tests may use it) and exchange exactly what the RCCL path exchanges: an all-gather of the
48-byte per-env records (+ legal bitmasks) and, in the learner flow of BASELINE config 4, one broadcast of the actions
chosen on rank 0.  The gathered results must equal the single-process run of the whole batch, env for env."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from xroute_env_amd.dist import (RECORD_BYTES, ShardedVectorEnv, first_legal_policy, gather_records, pack_records,
                                 shard_range, unpack_records)

N_TOTAL = 11          # ragged on purpose (6 + 5)
STEPS = 7
CONFIG = 9


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleVectorEnv:
    """CPU stand-in with XRouteVectorEnv's surface (reset / step -> obs, reward, done, info with `record` and `legal`),
    backed by the oracle.  Lets the collective logic of ShardedVectorEnv run under gloo without a GPU."""

    def __init__(self, regions, device=None, with_observation=False, **kw):
        from oracle import xr_oracle as orc
        self.ob = orc.OracleBatch(regions)
        self.n_envs = len(regions)
        self.device = torch.device("cpu")
        self.k_max = max(r.n_nets for r in regions)
        self.words = max(1, (self.k_max + 63) // 64)

    def _info(self, reward, delta, done):
        nleg = np.array([e.nlegal() for e in self.ob.envs], np.int32)
        cum = np.stack([e.cum() for e in self.ob.envs])
        steps = np.array([e.steps() for e in self.ob.envs], np.int32)
        rec = pack_records(torch.from_numpy(reward), torch.from_numpy(delta), torch.from_numpy(done), torch.from_numpy(nleg),
                           cum=torch.from_numpy(cum), env_steps=torch.from_numpy(steps))
        legal = np.zeros((self.n_envs, self.words), np.uint64)
        for i, e in enumerate(self.ob.envs):
            for n in e.legal():
                legal[i, (n - 1) >> 6] |= np.uint64(1) << np.uint64((n - 1) & 63)
        return {"record": rec, "legal": torch.from_numpy(legal.view(np.int64)), "nlegal": torch.from_numpy(nleg),
                "delta": torch.from_numpy(delta)}

    def reset(self):
        for e in self.ob.envs:
            e.reset()
        z = np.zeros(self.n_envs)
        return None, self._info(z, np.zeros((self.n_envs, 3), np.int32), np.zeros(self.n_envs, np.uint8))

    def step(self, actions):
        r = self.ob.step(actions.cpu().numpy(), threads=1, auto_reset=True)
        info = self._info(r["reward"], r["delta"], r["done"])
        return None, torch.from_numpy(r["reward"]), torch.from_numpy(r["done"]), info


class _FakeStateBatch:
    """What dist.CompactStateExchange needs of a RegionBatch (state_row_bytes / pack_state / expand_state), on CPU tensors: the collective
    logic runs under gloo without a GPU; the kernels behind the real methods are covered by tests/test_gpu_learner_state.py."""

    def __init__(self, n_envs, words):
        self.n_envs, self.device, self._rb = n_envs, torch.device("cpu"), 16 + 8 * words

    def state_row_bytes(self):
        return self._rb

    def pack_state(self, out, region_base=0):
        out.zero_()
        v = out.view(torch.int32)
        v[:, 0] = torch.arange(self.n_envs, dtype=torch.int32) + region_base
        v[:, 1] = 7
        return out

    def expand_state(self, rows, *unused):
        return rows.view(torch.int32)[:, 0].clone(), None, None


DIMS = dict(dims=(8, 7, 3), k_range=(2, 4))


def _make_env(n_total):
    import xroute_env_amd.regions as rg
    rg.CONFIGS[CONFIG] = DIMS                      # a tiny test-only config id
    return ShardedVectorEnv(CONFIG, n_total, env_factory=OracleVectorEnv, with_observation=False)


def _run_learner(n_total, steps):
    env = _make_env(n_total)
    _, acts = env.learner_reset(first_legal_policy)
    recs, sent = [], []
    for _ in range(steps):
        sent.append(acts.clone())
        _, rec_all, acts, _ = env.learner_step(acts, first_legal_policy)
        recs.append(rec_all.clone())
    return recs, sent, (env.lo, env.hi)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    recs, sent, (lo, hi) = _run_learner(N_TOTAL, STEPS)
    # plain step + gather (no learner), ragged shards
    env = _make_env(N_TOTAL)
    env.reset()
    plain = []
    for it in range(3):
        a = torch.tensor([1 + ((lo + i + it) % 2) for i in range(hi - lo)], dtype=torch.int32)
        plain.append(env.step(a)[1].clone())
    # equal-shard fast path (all_gather_into_tensor)
    eq_lo, eq_hi = shard_range(10, world, rank)
    eq = torch.full((eq_hi - eq_lo, RECORD_BYTES), rank, dtype=torch.uint8)
    eq_all = gather_records(eq)
    # self-certification of the gather (bench.py's N > 1 line): equal shards, ragged shards, and a slice corrupted on ONE rank only
    from xroute_env_amd.dist import verify_gather
    local = torch.arange((eq_hi - eq_lo) * RECORD_BYTES, dtype=torch.int64).reshape(-1, RECORD_BYTES).add(37 * rank).to(torch.uint8)
    cert = [verify_gather(local, gather_records(local), eq_lo)]
    rl = plain[-1][lo:hi].clone()                                  # ragged: this rank's rows of the last plain gather
    cert.append(verify_gather(rl, gather_records(rl), lo))
    bad = gather_records(local).clone()
    if rank == 1:
        bad[0, 3] ^= 0x40                                          # rank 1 received a wrong byte in rank 0's slice
    cert.append(verify_gather(local, bad, eq_lo))
    # the compact-state gather of a central learner (dist.CompactStateExchange) over a stand-in batch: ranks with DIFFERENT row sizes agree on
    # the largest, rows arrive in global env order for equal and for ragged shards, the learner sees every env
    from xroute_env_amd.dist import CompactStateExchange
    xch_out = []
    for n_tot in (10, N_TOTAL):
        l2, h2 = shard_range(n_tot, world, rank)
        fb = _FakeStateBatch(h2 - l2, words=3 + 2 * rank)
        xch = CompactStateExchange(fb, n_tot, l2, region_base=l2)
        rows = xch.gather()                          # to the learner (rank 0) only: the other ranks send and receive nothing
        assert (rows is None) == (rank != 0)
        everywhere = xch.gather(to_all=True)         # round 5's all_gather form is still there
        if rank == 0:
            assert torch.equal(rows, everywhere)
        rows = everywhere
        xch_out.append((xch.row_bytes, xch.bytes_per_step, rows.view(torch.int32)[:, :2].clone().numpy(), xch.expand(rows)[0].numpy(), xch.bytes_per_link))
    q.put((rank, [r.numpy() for r in recs], [s.numpy() for s in sent], (lo, hi), [p.numpy() for p in plain], eq_all.numpy(), cert, xch_out))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    for n in (0, 1, 7, 4096, 4097):
        for w in (1, 2, 3, 8):
            cover = []
            for r in range(w):
                lo, hi = shard_range(n, w, r)
                assert 0 <= lo <= hi <= n
                cover += list(range(lo, hi))
            assert cover == list(range(n))
            sizes = [shard_range(n, w, r)[1] - shard_range(n, w, r)[0] for r in range(w)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_record_pack_unpack_roundtrip():
    n = 5
    reward = torch.tensor([-0.0, -1.5, -500.0, -12345.5, 0.0], dtype=torch.float64)
    delta = torch.arange(15, dtype=torch.int32).reshape(n, 3)
    done = torch.tensor([0, 1, 0, 1, 1], dtype=torch.uint8)
    nleg = torch.tensor([3, 0, 7, 0, 0], dtype=torch.int32)
    rec = pack_records(reward, delta, done, nleg, cum=delta * 2, env_steps=nleg + 1, path_len=nleg + 2,
                       status=torch.tensor([0, 8, 2, 1, 4]))
    assert rec.shape == (n, RECORD_BYTES) and rec.dtype == torch.uint8
    u = unpack_records(rec)
    assert torch.equal(u["reward"], reward) and torch.equal(u["delta"], delta) and torch.equal(u["cum"], delta * 2)
    assert torch.equal(u["nlegal"], nleg) and torch.equal(u["done"], done) and u["status"].tolist() == [0, 8, 2, 1, 4]
    # layout == the C struct (include/xroute_hip.h xr_step_record), checked through the numpy dtype the GPU path uses
    from xroute_env_amd._lib import XrStepRecord
    import ctypes
    assert ctypes.sizeof(XrStepRecord) == RECORD_BYTES
    r0 = XrStepRecord.from_buffer_copy(rec[3].numpy().tobytes())
    assert r0.reward == -12345.5 and list(r0.delta) == [9, 10, 11] and r0.nlegal == 0 and r0.done == 1 and r0.status == 1


def test_first_legal_policy():
    legal = torch.tensor([[0b0110, 0], [0, 1 << 5], [0, 0]], dtype=torch.int64)
    assert first_legal_policy({}, legal).tolist() == [2, 64 + 6, 0]


def test_two_rank_learner_flow_equals_single_process():
    """BASELINE config 4 loop (gather records + legal sets -> rank-0 policy -> broadcast i32[B] actions -> local step) on
    two gloo ranks == the same loop in one process."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        item = q.get(timeout=180)
        got[item[0]] = item[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single_recs, single_sent, _ = _run_learner(N_TOTAL, STEPS)
    for rank in (0, 1):
        recs, sent, (lo, hi), plain, eq_all, cert, xch_out = got[rank]
        for (rb, per_step, hdr, regions_seen, per_link), n_tot in zip(xch_out, (10, N_TOTAL)):
            n_loc = shard_range(n_tot, 2, rank)[1] - shard_range(n_tot, 2, rank)[0]
            assert rb == 16 + 8 * 5                                             # the larger of the two ranks' row sizes, on both ranks
            assert per_link == n_loc * rb and per_step == n_tot * rb             # a gather to the learner: a rank's rows cross ONE link once
            assert hdr[:, 0].tolist() == list(range(n_tot)) and (hdr[:, 1] == 7).all() and regions_seen.tolist() == list(range(n_tot))
        assert len(recs) == STEPS
        assert cert[0] == {"ranks_seen": 2, "gather_verified": True, "rows": 10}
        assert cert[1] == {"ranks_seen": 2, "gather_verified": True, "rows": N_TOTAL}
        assert cert[2]["gather_verified"] is False and cert[2]["ranks_seen"] == 2       # BOTH ranks learn that one of them saw a bad slice
        for g, s in zip(recs, single_recs):                     # every rank holds every env's record
            assert g.shape == (N_TOTAL, RECORD_BYTES) and np.array_equal(g, s.numpy())
        for a, s in zip(sent, single_sent):                     # the broadcast actions are the rank's slice of the learner's
            assert np.array_equal(a, s.numpy()[lo:hi])
        assert eq_all.shape == (10, RECORD_BYTES) and (eq_all[:5] == 0).all() and (eq_all[5:] == 1).all()
    # the plain gather of ragged shards is in env order on both ranks
    for p0, p1 in zip(got[0][3], got[1][3]):
        assert p0.shape == (N_TOTAL, RECORD_BYTES) and np.array_equal(p0, p1)
    # something was actually routed (non-zero rewards) and episodes ended and restarted
    u = unpack_records(torch.from_numpy(np.ascontiguousarray(np.stack(got[0][0]).reshape(-1, RECORD_BYTES))))
    assert (u["reward"] < 0).any() and (u["done"] == 1).any()
