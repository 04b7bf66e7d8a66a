"""Drop-in for the reference's observation builder, computed on the MI355X.

    from xroute_env_amd.build_3Dgrid import build_3Dgrid
    observation, netSet, violation, wirelength, via = build_3Dgrid(data, routed_nets, bool_inference)

Same signature, argument meaning and return tuple as reference baseline/build_3Dgrid.py:224-270.
The host part only restates the *set logic* (which nets are in netSet); every tensor element is
written by the HIP kernel `xr_obs_records_kernel` through `xr_observation_from_records`.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Optional

import numpy as np
import torch

from . import _lib
from .regions import ACCESS, BLOCKAGE, NORMAL, pack_records, records_from_entries


def get_grid_size(data):
    return data[0]


def data_to_records(data) -> np.ndarray:
    """Reference `data` list -> dense packed node records in flat order f=(x*Y+y)*Z+z.
    Vertices the list does not mention are unused NORMAL nodes (the reference never touches them,
    baseline/build_3Dgrid.py:18-43); a vertex listed more than once is an obstacle / an access point if ANY of its entries
    says so, as in the reference's per-entry loop (regions.records_from_entries)."""
    X, Y, Z = (int(v) for v in data[0])
    n = X * Y * Z
    if not len(data[1]):
        return records_from_entries(n, [], [], [], [])
    maze = np.array([v[0] for v in data[1]], dtype=np.int64).reshape(-1, 3)
    info = np.array([v[2] for v in data[1]], dtype=np.int64).reshape(-1, 3)
    f = (maze[:, 0] * Y + maze[:, 1]) * Z + maze[:, 2]
    t = info[:, 1]
    if ((t < -1)).any():
        raise AssertionError("Net must be -1, 0 or >= 1")      # build_3Dgrid.py:32 asserts the same
    return records_from_entries(n, f, t, info[:, 0], info[:, 2])      # `bool_occupy == 1` (:24,:34)


def legal_nets(records: np.ndarray, routed_nets: Iterable[int], bool_inference: bool,
               net_list: Optional[Iterable[int]]) -> np.ndarray:
    """netSet, ascending: nets that own an access point, minus routed nets in training mode
    (build_3Dgrid.py:46-55), intersected with data[3] in inference mode (:243-250)."""
    rec = records.astype(np.int64)
    is_ap = (rec & 3) == ACCESS
    nets = np.unique((rec[is_ap] >> 3) & 0x3FFF)
    if not bool_inference:
        routed = np.fromiter((int(v) for v in routed_nets), dtype=np.int64)
        nets = nets[~np.isin(nets, routed)]
    else:
        keep = np.fromiter((int(v) for v in (net_list if net_list is not None else [])), dtype=np.int64)
        nets = nets[np.isin(nets, keep)]
    return nets.astype(np.int32)


def observation_from_records(records: np.ndarray, dims, nets: np.ndarray, device="cuda:0") -> torch.Tensor:
    """[1, 2+7K, Z, Y, X] fp32 observation on `device`, written by the HIP kernel."""
    dev = torch.device(device)
    if dev.type != "cuda" or not torch.cuda.is_available():
        raise RuntimeError("build_3Dgrid needs a HIP device (MI355X); there is no CPU fallback")
    X, Y, Z = (int(v) for v in dims)
    n = X * Y * Z
    k = int(len(nets))
    L = _lib.lib()
    with torch.cuda.device(dev):
        rec_d = torch.from_numpy(records.view(np.int32).copy()).to(dev)
        nets_d = torch.from_numpy(np.ascontiguousarray(nets, np.int32)).to(dev) if k else \
            torch.zeros(1, dtype=torch.int32, device=dev)
        out = torch.empty((1, 2 + 7 * k, Z, Y, X), dtype=torch.float32, device=dev)
        if n > 0:
            _lib.check(L.xr_observation_from_records(C.c_void_p(rec_d.data_ptr()), X, Y, Z,
                                                     C.c_void_p(nets_d.data_ptr()), k, C.c_void_p(out.data_ptr()),
                                                     C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
            torch.cuda.current_stream(dev).synchronize()   # rec_d / nets_d die with this frame
    return out


def build_3Dgrid(data, routed_nets, bool_inference=False, device="cuda:0", return_device=False):
    """reference baseline/build_3Dgrid.py:224-270.  Returns (observation [1,C,D,H,W] fp32 — a CPU
    tensor like the reference's unless return_device — , netSet, violation, wirelength, via)."""
    records = data_to_records(data)
    nets = legal_nets(records, routed_nets, bool_inference, data[3] if len(data) > 3 else None)
    obs = observation_from_records(records, data[0], nets, device)
    if not return_device:
        obs = obs.cpu()
    net_set = set(int(v) for v in nets)
    return obs, net_set, data[2][0], data[2][1], data[2][2]
