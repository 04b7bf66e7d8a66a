"""ctypes binding of oracle/libxr_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see oracle/xr_oracle.h for the rule and the parity status: observation half pinned to
tests/golden, router half "parity unpinned" = this repo's XR-Maze v1 spec).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libxr_oracle.so")
    src = os.path.join(_HERE, "xr_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libxr_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    L = C.CDLL(build())
    vp, i32p, u32p, f32p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_float)
    L.xro_legal_nets.argtypes = [vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, vp, C.c_int]
    L.xro_legal_nets.restype = C.c_int
    L.xro_build_observation.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp]
    L.xro_build_observation.restype = C.c_int
    L.xro_reward.argtypes = [C.c_int64, C.c_int64, C.c_int64]
    L.xro_reward.restype = C.c_double
    L.xro_env_create.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int]
    L.xro_env_create.restype = vp
    L.xro_env_set_v2.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.xro_env_set_v2.restype = None
    L.xro_env_set_guides.argtypes = [vp, vp, vp]
    L.xro_env_set_guides.restype = C.c_int
    L.xro_env_destroy.argtypes = [vp]
    L.xro_env_destroy.restype = None
    L.xro_env_reset.argtypes = [vp]
    L.xro_env_reset.restype = None
    L.xro_env_step.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, vp]
    L.xro_env_step.restype = C.c_int
    L.xro_env_nlegal.argtypes = [vp]
    L.xro_env_nlegal.restype = C.c_int
    L.xro_env_legal.argtypes = [vp, vp, C.c_int]
    L.xro_env_legal.restype = C.c_int
    L.xro_env_cum.argtypes = [vp, vp]
    L.xro_env_cum.restype = None
    L.xro_env_owner.argtypes = [vp, vp]
    L.xro_env_owner.restype = None
    L.xro_env_hash.argtypes = [vp]
    L.xro_env_hash.restype = C.c_uint64
    L.xro_env_observation.argtypes = [vp, vp]
    L.xro_env_observation.restype = C.c_int
    L.xro_env_n_nodes.argtypes = [vp]
    L.xro_env_n_nodes.restype = C.c_int
    L.xro_env_distance_field.argtypes = [vp, C.c_int, vp]
    L.xro_env_distance_field.restype = C.c_int
    L.xro_env_steps.argtypes = [vp]
    L.xro_env_steps.restype = C.c_int64
    L.xro_max_threads.restype = C.c_int
    L.xro_batch_step.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    L.xro_batch_step.restype = C.c_int64
    L.xro_batch_observation.argtypes = [vp, C.c_int, vp, C.c_int64, C.c_int]
    L.xro_batch_observation.restype = C.c_int
    L.xro_batch_random_actions.argtypes = [vp, C.c_int, C.c_uint64, vp]
    L.xro_batch_random_actions.restype = None
    _LIB = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def legal_nets(rec, routed=(), nets_filter=(), inference=False):
    rec = np.ascontiguousarray(rec, np.uint32)
    routed = np.ascontiguousarray(sorted(routed), np.int32)
    flt = np.ascontiguousarray(list(nets_filter), np.int32)
    out = np.zeros(0x4000, np.int32)
    k = lib().xro_legal_nets(_p(rec), rec.size, _p(routed), routed.size, _p(flt), flt.size,
                             int(bool(inference)), _p(out), out.size)
    return out[:k].copy()


def build_observation(dims, rec, nets):
    X, Y, Z = (int(v) for v in dims)
    rec = np.ascontiguousarray(rec, np.uint32)
    nets = np.ascontiguousarray(nets, np.int32)
    out = np.empty((2 + 7 * nets.size, Z, Y, X), np.float32)
    lib().xro_build_observation(X, Y, Z, _p(rec), _p(nets), nets.size, _p(out))
    return out


def reward(dv, dw, dvia):
    return lib().xro_reward(int(dv), int(dw), int(dvia))


class OracleEnv:
    """One env on the CPU oracle (Game bookkeeping + XR-Maze v1)."""

    def __init__(self, region, via_cost=800, drc_cost=8, drc_unit=400, guide_cost=0, guide_margin=0, maze_end_iter=1):
        self.region = region
        X, Y, Z = region.dims
        self._keep = [np.ascontiguousarray(region.xs, np.int32), np.ascontiguousarray(region.ys, np.int32),
                      np.ascontiguousarray(region.layer_dir, np.uint8),
                      np.ascontiguousarray(region.nodes, np.uint32),
                      np.ascontiguousarray(region.metrics0, np.int32)]
        self.h = lib().xro_env_create(X, Y, Z, _p(self._keep[0]), _p(self._keep[1]), _p(self._keep[2]),
                                      _p(self._keep[3]), int(region.n_nets), _p(self._keep[4]),
                                      via_cost, drc_cost, drc_unit)
        if not self.h:
            raise MemoryError("xro_env_create failed")
        if guide_cost or maze_end_iter != 1:
            lib().xro_env_set_v2(self.h, int(guide_cost), int(guide_margin), int(maze_end_iter))
        if getattr(region, "guide_off", None) is not None:
            off = np.ascontiguousarray(region.guide_off, np.int32)
            box = np.ascontiguousarray(region.guide_box, np.int16).reshape(-1, 6)
            if off.size != region.n_nets + 1 or off[-1] != box.shape[0]:
                raise ValueError("region guides: guide_off must have n_nets + 1 entries and end at the number of boxes")
            if lib().xro_env_set_guides(self.h, _p(off), _p(box)) != 0:
                raise ValueError("region guides: more than 8 boxes for one net")
        self.n = lib().xro_env_n_nodes(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            try:
                lib().xro_env_destroy(self.h)
            except Exception:          # interpreter shutdown: module globals may already be gone
                pass
            self.h = None

    def reset(self):
        lib().xro_env_reset(self.h)

    def step(self, action, path_cap=None):
        cap = self.n if path_cap is None else path_cap
        delta = np.zeros(3, np.int32)
        done = C.c_int(0)
        path = np.zeros(max(cap, 1), np.int32)
        plen = C.c_int(0)
        st = lib().xro_env_step(self.h, int(action), _p(delta), C.byref(done), _p(path), cap, C.byref(plen))
        return dict(status=st, delta=delta, done=bool(done.value), path=path[:min(plen.value, cap)].copy(),
                    path_len=plen.value)

    def legal(self):
        out = np.zeros(self.region.n_nets + 1, np.int32)
        k = lib().xro_env_legal(self.h, _p(out), out.size)
        return out[:k].copy()

    def nlegal(self):
        return lib().xro_env_nlegal(self.h)

    def cum(self):
        c = np.zeros(3, np.int32)
        lib().xro_env_cum(self.h, _p(c))
        return c

    def owner(self):
        o = np.zeros(max(self.n, 1), np.int16)
        lib().xro_env_owner(self.h, _p(o))
        return o[:self.n]

    def hash(self):
        return int(lib().xro_env_hash(self.h))

    def steps(self):
        return int(lib().xro_env_steps(self.h))

    def observation(self):
        X, Y, Z = self.region.dims
        out = np.empty((2 + 7 * self.nlegal(), Z, Y, X), np.float32)
        lib().xro_env_observation(self.h, _p(out))
        return out

    def distance_field(self, action):
        d = np.zeros(max(self.n, 1), np.uint32)
        rc = lib().xro_env_distance_field(self.h, int(action), _p(d))
        if rc != 0:
            raise ValueError("net has no access points")
        return d[:self.n]


class OracleBatch:
    """Many oracle envs stepped with OpenMP over envs (the timed CPU baseline)."""

    def __init__(self, regions, via_cost=800, drc_cost=8, drc_unit=400, **v2):
        self.envs = [OracleEnv(r, via_cost, drc_cost, drc_unit, **v2) for r in regions]
        self.handles = (C.c_void_p * len(self.envs))(*[e.h for e in self.envs])
        self.n = len(self.envs)

    def random_actions(self, seed):
        a = np.zeros(self.n, np.int32)
        lib().xro_batch_random_actions(self.handles, self.n, C.c_uint64(seed), _p(a))
        return a

    def step(self, actions, threads=1, auto_reset=True):
        actions = np.ascontiguousarray(actions, np.int32)
        delta = np.zeros((self.n, 3), np.int32)
        done = np.zeros(self.n, np.uint8)
        rew = np.zeros(self.n, np.float64)
        real = lib().xro_batch_step(self.handles, _p(actions), self.n, threads, int(auto_reset), _p(delta),
                                    _p(done), _p(rew))
        return dict(real_steps=int(real), delta=delta, done=done, reward=rew)

    def observation(self, out, stride, threads=1):
        lib().xro_batch_observation(self.handles, self.n, _p(out), int(stride), threads)

    def max_threads(self):
        return lib().xro_max_threads()
