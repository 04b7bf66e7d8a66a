"""Shared helpers for the parity tests (fixture loading, reference-`data` conversion)."""
import hashlib
import json
import os

import numpy as np

from xroute_env_amd.regions import ACCESS, BLOCKAGE, NORMAL, Region, pack_records

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
    """Dense packed records of a G1 case (nodes absent from the list = unused NORMAL)."""
    X, Y, Z = (int(v) for v in case["dims"])
    n = X * Y * Z
    ntype = np.full(n, NORMAL, np.int64)
    used = np.zeros(n, np.int64)
    net = np.full(n, -1, np.int64)
    pin = np.full(n, -1, np.int64)
    if len(case["maze"]):
        m = case["maze"].astype(np.int64)
        info = case["info"].astype(np.int64)
        f = (m[:, 0] * Y + m[:, 1]) * Z + m[:, 2]
        t = info[:, 1]
        ntype[f] = np.where(t == -1, BLOCKAGE, np.where(t == 0, NORMAL, ACCESS))
        used[f] = info[:, 0]
        net[f] = np.where(t >= 1, t - 1, -1)
        pin[f] = np.where(t >= 1, info[:, 2] - 1, -1)
    return pack_records(ntype, used, net, pin)


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)
