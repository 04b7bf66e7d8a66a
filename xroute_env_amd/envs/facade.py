"""Gym-style single-env façade: XRouteEnv and the three named envs the reference reserves
(OrderingTrainingEnv, OrderingEvaluationEnv, StaticRegionEnv).

The reference reserves the names only: `XRouteEnv.step` is `pass` and the three env classes are
empty (reference xroute_env/envs/core.py:3-8, ordering_training_env.py:4-5, ...), so the behaviour
here is the `Game` contract (baseline/baseline_utils.py:383-481) wrapped in the gymnasium calling
convention.  gymnasium itself is optional: without it a minimal stand-in base class is used.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np

from . import spaces as xr_spaces

try:                                    # gymnasium is not installed in the build image
    import gymnasium as gym
    _HAVE_GYM = True
except Exception:                       # pragma: no cover - exercised when gymnasium is absent
    gym = None
    _HAVE_GYM = False


class _EnvBase:
    metadata = {"render_modes": []}
    observation_space = None
    action_space = None

    def close(self):
        pass


EnvBase = gym.Env if _HAVE_GYM else _EnvBase


class XRouteEnv(EnvBase):
    """reset(seed, options) -> (obs, info);  step(action) -> (obs, reward, terminated, truncated, info).

    * action: 1-based net id, must be in info["legal_actions"] (the reference's netSet);
    * obs: fp32 [2+7K, Z, Y, X] in the reference layout (K shrinks as nets are routed; `observation_space` is the Box of the
      current shape); with pad_channels=True it is zero-padded to the episode's initial channel count; with fixed_shape=True
      it is {"grid": [2+7*Kmax, Z, Y, X] (zero-padded), "legal_mask": [Kmax]} and the spaces are the fixed
      Dict{grid: Box, legal_mask: MultiBinary(Kmax)} / Discrete(Kmax, start=1) of envs/spaces.py, defined at construction;
    * reward = -(500*d_violation + 4*d_via + 0.5*d_wirelength)  (baseline/DQN/train_DQN.py:98-99).
    """

    def __init__(self, regions: Sequence, device="cuda:0", pad_channels: bool = False, max_route_count: int = 10,
                 fixed_shape: bool = False, **router_kw):
        from ..game import Game
        regions = list(regions)
        self.game = Game(regions=regions, device=device, max_route_count=max_route_count, **router_kw)
        self.pad_channels = pad_channels
        self.fixed_shape = fixed_shape
        self._c0 = None
        self.kmax = max(max(int(r.n_nets) for r in regions), 1)
        self.observation_space = None            # per-episode form: a Box of the current tensor's shape, set by reset()
        self.action_space = None
        if fixed_shape:
            # Dict{grid: Box[Cmax, Z, Y, X], legal_mask: MultiBinary(Kmax)} — defined before the first reset (spaces.py)
            dims = set(tuple(int(v) for v in r.dims) for r in regions)
            if len(dims) != 1:
                raise ValueError("fixed_shape=True needs regions of one size (the grid Box has one shape); use the per-episode form")
            self.dims = dims.pop()
            self.observation_space, self.action_space = xr_spaces.fixed_spaces(self.dims, self.kmax)

    def _spaces(self, obs):
        if self.fixed_shape:
            return
        kmax = max(self.game.action_space) if self.game.action_space else 1
        self.observation_space, self.action_space = xr_spaces.episode_spaces(obs.shape, kmax)

    def _fmt(self, obs):
        import torch
        obs = obs[0]
        if self.fixed_shape:
            cmax = 2 + 7 * self.kmax
            grid = torch.zeros((cmax,) + tuple(obs.shape[1:]), dtype=obs.dtype, device=obs.device)
            grid[:obs.shape[0]] = obs
            mask = np.zeros(self.kmax, np.int8)
            for n in self.game.legal_action_set:
                mask[n - 1] = 1
            return {"grid": grid, "legal_mask": mask}
        if self.pad_channels and obs.shape[0] < self._c0:
            pad = torch.zeros((self._c0 - obs.shape[0],) + tuple(obs.shape[1:]), dtype=obs.dtype, device=obs.device)
            obs = torch.cat([obs, pad], dim=0)
        return obs

    def reset(self, *, seed: Optional[int] = None, options: Optional[dict] = None):
        obs, tries = self.game.reset()
        self._c0 = obs.shape[1]
        out = self._fmt(obs)
        self._spaces(out)
        return out, {"legal_actions": sorted(self.game.legal_action_set), "reset_try_time": tries}

    def step(self, action):
        from ..game import reward_from_deltas
        obs, done, dv, dw, dvia = self.game.step(int(action))
        info = {"legal_actions": sorted(self.game.legal_action_set), "violation": dv, "wirelength": dw, "via": dvia}
        out = self._fmt(obs)
        if not self.fixed_shape and not self.pad_channels:
            self._spaces(out)         # the reference tensor loses 7 channels per routed net: the per-episode Box follows its shape
        return out, reward_from_deltas(dv, dw, dvia), bool(done), False, info


class OrderingTrainingEnv(XRouteEnv):
    """`xroute_env/ordering-training-v0` (reference xroute_env/__init__.py:3-6): net-ordering training
    on rotating regions — Game semantics, 10 replays per region then the next."""


class OrderingEvaluationEnv(XRouteEnv):
    """Evaluation flavour (reference xroute_env/envs/ordering_evaluation_env.py:4-5 is an empty class;
    the reference's evaluation servers build observations in inference mode,
    baseline/DQN/test_DQN.py:62): every region is played once, in order, no replays."""

    def __init__(self, regions, **kw):
        kw.setdefault("max_route_count", 1)
        super().__init__(regions, **kw)


# the static regions the reference sketches (xroute_env/__init__.py:13-23), extracted from its own benchmark inputs by
# tools/extract_regions.py --static-region1 (data derived from the ispd18_test1 LEF / DEF / guide files, like tests/golden's region pack)
STATIC_REGIONS = [{"benchmark": "region1", "from": "ISPD-2018 test1", "size": "1x1", "position": [(199500, 245100), (205200, 250800)]}]


def load_static_region(key):
    """A static region by the reference's description: its benchmark name ("region1") or the dict of `static_regions`
    (matched by "benchmark").  Returns a `Region` (worker model: the 1x1 GCell is the routeBox, +2000 DBU of routing resource)."""
    import os
    from ..lefdef import load_region_pack
    name = key.get("benchmark") if isinstance(key, dict) else str(key)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "static_regions.npz")
    for reg in load_region_pack(path):
        if reg.name == name:
            return reg
    raise KeyError(f"no static region {name!r} (known: {[r['benchmark'] for r in STATIC_REGIONS]})")


class StaticRegionEnv(XRouteEnv):
    """One fixed region replayed forever (reference xroute_env/__init__.py:13-33 sketches
    `static-{benchmark}-v0` registrations with a `region` kwarg — a dict with "benchmark", "position" ...; the class itself is empty).
    `region`: a `Region`, the reference's dict, or the benchmark name."""

    def __init__(self, region, **kw):
        if isinstance(region, (dict, str)):
            region = load_static_region(region)
        super().__init__([region], **kw)
