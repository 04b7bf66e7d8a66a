"""Observation / action spaces of the env façade.

The reference registers a gymnasium id (`xroute_env/ordering-training-v0`, xroute_env/__init__.py:3-6) but defines no
`observation_space` / `action_space` anywhere (its env classes are empty, xroute_env/envs/*.py).  The build's choice
(SURVEY §8b): the reference observation has a variable channel count (2 + 7K, K = nets left), so the FIXED-shape form is

    Dict{ "grid": Box(float32, [Cmax, Z, Y, X]),   Cmax = 2 + 7 * Kmax: the reference tensor, zero-padded in the channel axis
          "legal_mask": MultiBinary(Kmax) }         bit n-1 <=> net n is in netSet;   actions: Discrete(Kmax, start=1)

and the per-episode Box (exact reference shape, changes every step) stays available for callers that want the reference tensor
as is.  gymnasium is optional (absent from the build image): without it the minimal stand-ins below carry the same fields."""
from __future__ import annotations

import numpy as np


class _Space:
    def contains(self, x) -> bool:          # pragma: no cover - overridden
        raise NotImplementedError

    def __contains__(self, x):
        return self.contains(x)


class Box(_Space):
    def __init__(self, low, high, shape, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = float(low), float(high), tuple(int(v) for v in shape), np.dtype(dtype)

    def contains(self, x):
        a = np.asarray(x.cpu() if hasattr(x, "cpu") else x)
        return tuple(a.shape) == self.shape and bool((a >= self.low).all() and (a <= self.high).all())

    def __repr__(self):
        return f"Box({self.low}, {self.high}, {self.shape}, {self.dtype})"


class Discrete(_Space):
    def __init__(self, n, start=0):
        self.n, self.start = int(n), int(start)

    def contains(self, x):
        return self.start <= int(x) < self.start + self.n

    def __repr__(self):
        return f"Discrete({self.n}, start={self.start})"


class MultiBinary(_Space):
    def __init__(self, n):
        self.n = n
        self.shape = tuple(n) if isinstance(n, (tuple, list)) else (int(n),)

    def contains(self, x):
        a = np.asarray(x.cpu() if hasattr(x, "cpu") else x)
        return tuple(a.shape) == self.shape and bool(((a == 0) | (a == 1)).all())

    def __repr__(self):
        return f"MultiBinary({self.n})"


class Dict(_Space):
    def __init__(self, spaces):
        self.spaces = dict(spaces)

    def __getitem__(self, k):
        return self.spaces[k]

    def contains(self, x):
        return set(x.keys()) == set(self.spaces.keys()) and all(self.spaces[k].contains(x[k]) for k in self.spaces)

    def __repr__(self):
        return "Dict(" + ", ".join(f"{k}: {v!r}" for k, v in self.spaces.items()) + ")"


def backend():
    """gymnasium.spaces when it is importable, else this module."""
    try:
        from gymnasium import spaces as gs
        return gs
    except Exception:
        import sys
        return sys.modules[__name__]


def fixed_spaces(dims, kmax: int, batch: int = 0, sp=None, row: int = 0):
    """(observation_space, action_space) of the fixed-shape form for regions of `dims` = (X, Y, Z) with at most `kmax` nets.
    batch > 0: the vector env's batched spaces (grid rows are flat: [batch, row], the layout of the device buffer: row =
    the buffer's env stride >= Cmax * N — xr_batch_sizes rounds it up to whole 128-byte lines; 0: exactly Cmax * N)."""
    sp = sp or backend()
    X, Y, Z = (int(v) for v in dims)
    kmax = max(int(kmax), 1)
    cmax = 2 + 7 * kmax
    if batch:
        if row and row < cmax * X * Y * Z:
            raise ValueError("row stride shorter than the observation")
        grid = sp.Box(low=0.0, high=float(kmax), shape=(batch, int(row) if row else cmax * X * Y * Z), dtype=np.float32)
        mask = sp.MultiBinary((batch, kmax))
    else:
        grid = sp.Box(low=0.0, high=float(kmax), shape=(cmax, Z, Y, X), dtype=np.float32)
        mask = sp.MultiBinary(kmax)
    return sp.Dict({"grid": grid, "legal_mask": mask}), sp.Discrete(kmax, start=1)


def episode_spaces(obs_shape, kmax: int, sp=None):
    """The per-episode form: a Box of exactly the reference tensor's current shape."""
    sp = sp or backend()
    kmax = max(int(kmax), 1)
    return sp.Box(low=0.0, high=float(kmax), shape=tuple(int(v) for v in obs_shape), dtype=np.float32), sp.Discrete(kmax, start=1)
