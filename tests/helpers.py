"""Shared helpers for the parity tests (fixture loading, reference-`data` conversion)."""
import hashlib
import json
import os

import numpy as np

from xroute_env_amd.regions import ACCESS, BLOCKAGE, NORMAL, Region, pack_records, records_from_entries

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load_g1():
    z = np.load(os.path.join(GOLDEN, "g1_build3dgrid.npz"))
    cases = []
    for i in range(int(z["n_cases"])):
        p = f"c{i}_"
        cases.append({k[len(p):]: z[k] for k in z.files if k.startswith(p)})
    return cases


def g1_data(case) -> list:
    """Rebuild the reference `data` list of a G1 case."""
    nodes = [[[int(v) for v in m], [int(v) for v in p], [int(v) for v in t]]
             for m, p, t in zip(case["maze"], case["point"], case["info"])]
    return [[int(v) for v in case["dims"]], nodes, [int(v) for v in case["metrics"]],
            [int(v) for v in case["nets"]]]


def g1_records(case) -> np.ndarray:
    """Dense packed records of a G1 case (nodes absent from the list = unused NORMAL; a vertex listed twice = the reference's
    per-entry OR, regions.records_from_entries)."""
    X, Y, Z = (int(v) for v in case["dims"])
    n = X * Y * Z
    if not len(case["maze"]):
        return records_from_entries(n, [], [], [], [])
    m = case["maze"].astype(np.int64)
    info = case["info"].astype(np.int64)
    f = (m[:, 0] * Y + m[:, 1]) * Z + m[:, 2]
    return records_from_entries(n, f, info[:, 1], info[:, 0], info[:, 2])


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


class OracleOrderSimulator:
    """CPU-oracle stand-in with xroute_env_amd.envs.order_contracts.OrderSimulator's interface: the checker the
    whole-order contracts (A3CGame / Route / OrderVectorEnv) are compared against.  Test infrastructure only."""

    def __init__(self, regions, n_envs=None, **oracle_kw):
        import torch
        from xroute_env_amd.proto import region_wire_fields
        self.torch = torch
        self.regions = list(regions)
        self.n_envs = int(n_envs if n_envs is not None else len(self.regions))
        self.device = torch.device("cpu")
        self.stride = max(max(r.n_nets for r in self.regions), 1)
        self.env_region = np.arange(self.n_envs) % len(self.regions)
        self.orders = torch.zeros((self.n_envs, self.stride), dtype=torch.int32)
        self.net_stats = torch.zeros((self.n_envs, self.stride, 4), dtype=torch.int32)
        self.batch = self                       # .batch.reset() of the GPU simulator
        self._wire = region_wire_fields
        self._kw = oracle_kw
        self.last = None

    def reset(self):
        pass

    def fields(self, r):
        return self._wire(self.regions[r])

    def assign(self, env_region):
        self.env_region = np.asarray(env_region, np.int64) % len(self.regions)

    def default_orders(self):
        o = self.torch.zeros((self.n_envs, self.stride), dtype=self.torch.int32)
        for e, r in enumerate(self.env_region):
            k = self.regions[r].n_nets
            o[e, :k] = self.torch.arange(1, k + 1, dtype=self.torch.int32)
        return o

    def route(self, orders, with_stats=True):
        from oracle.xr_oracle import OracleEnv
        cum = self.torch.zeros((self.n_envs, 3), dtype=self.torch.int32)
        self.last = []
        for e, r in enumerate(self.env_region):
            env = OracleEnv(self.regions[r], **self._kw)
            env.reset()
            status, plen = 0, 0
            for a in orders[e].tolist():
                if a <= 0 or env.nlegal() == 0:
                    break
                res = env.step(int(a))
                status |= res["status"]
                if not (res["status"] & 1):
                    plen += res["path_len"]
                    if with_stats:
                        self.net_stats[e, a - 1, :3] = self.torch.as_tensor(res["delta"])
                        self.net_stats[e, a - 1, 3] += 1
            cum[e] = self.torch.as_tensor(env.cum())
            self.last.append(dict(status=status, path_len=plen, owner=env.owner().copy(), done=env.nlegal() == 0))
        return cum
