"""N > 1 path on CPU: world_size-2 gloo processes shard the env batch, step their shard (the oracle stands
in for the GPU env here — tests may use it) and all-gather the compact per-env records.  The gathered
result must equal the single-process run of the whole batch, env for env."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from xroute_env_amd.dist import RECORD_WIDTH, env_seed, gather_records, pack_records, shard_range

N_TOTAL = 11          # ragged on purpose (6 + 5)
STEPS = 5
DIMS = dict(dims=(8, 7, 3), k_range=(2, 4))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_shard(lo, hi, steps):
    from oracle import xr_oracle as orc
    from xroute_env_amd.regions import generate_region
    regions = [generate_region(env_seed(9, e), **DIMS) for e in range(lo, hi)]
    ob = orc.OracleBatch(regions)
    out = []
    for it in range(steps):
        # the random policy hashes (seed, local index, step count): make the seed carry the global offset
        acts = np.zeros(hi - lo, np.int32)
        for i, e in enumerate(ob.envs):
            legal = e.legal()
            acts[i] = legal[(lo + i + it) % len(legal)] if len(legal) else 0
        r = ob.step(acts, threads=1, auto_reset=True)
        nleg = np.array([e.nlegal() for e in ob.envs])
        rec = pack_records(torch.from_numpy(r["reward"]), torch.from_numpy(r["delta"]), torch.from_numpy(r["done"]),
                           torch.from_numpy(nleg))
        out.append(rec)
    return out


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(N_TOTAL, world, rank)
    recs = _run_shard(lo, hi, STEPS)
    gathered = [gather_records(r) for r in recs]
    # equal-shard fast path as well: pad the local block to the max shard and use all_gather_into_tensor
    eq_lo, eq_hi = shard_range(10, world, rank)
    eq = torch.full((eq_hi - eq_lo, RECORD_WIDTH), float(rank), dtype=torch.float64)
    eq_all = gather_records(eq)
    if rank == 0:
        q.put(([g.numpy() for g in gathered], eq_all.numpy()))
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


def test_two_rank_gather_equals_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    gathered, eq_all = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single = _run_shard(0, N_TOTAL, STEPS)
    assert len(gathered) == STEPS
    for g, s in zip(gathered, single):
        assert g.shape == (N_TOTAL, RECORD_WIDTH)
        assert np.array_equal(g, s.numpy())
    assert eq_all.shape == (10, RECORD_WIDTH)
    assert (eq_all[:5] == 0).all() and (eq_all[5:] == 1).all()
