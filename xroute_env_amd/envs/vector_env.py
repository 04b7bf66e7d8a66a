"""Batched env: B regions stepped per call on one MI355X, results left on the device.

This is the data-parallel form of Game.step the north star asks for.  Nothing here synchronises with
the host: actions come in as a device tensor, observation / reward / done / legal masks stay device
tensors.  Envs that finished an episode are re-initialised by the next step (gym "next-step"
autoreset), with the reference's region rotation (examples/launch_training.py:28-54).
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from ..batch import RegionBatch
from ..dist import RECORD_BYTES, unpack_records


class XRouteVectorEnv:
    def __init__(self, regions: Sequence, n_envs: Optional[int] = None, device="cuda:0", with_observation: bool = True,
                 dict_observation: bool = False, **batch_kw):
        """dict_observation: reset() / step() return the observation as the advertised Dict space's member
        {"grid": [B, stride] fp32, "legal_mask": [B, Kmax] u8} instead of the bare grid tensor (the default: the hot loop's
        consumers — agents.dqn_actions / ppo_actions — take the grid buffer and the legal bitmasks of `info` as they are)."""
        self.batch = RegionBatch(regions, n_envs=n_envs, device=device, auto_reset=True, **batch_kw)
        self.n_envs = self.batch.n_envs
        self.device = self.batch.device
        self.with_observation = with_observation
        self.dict_observation = bool(dict_observation)
        # (zeros, not empty: a row beyond its env's (2+7K)*N floats then only ever holds zeros or stale 0/1 planes of earlier steps —
        #  inside the Box's bounds; consumers go by nlegal / legal_mask)
        self.obs = self.batch.alloc_observation().zero_() if with_observation else None
        # the 48-byte xr_step_record of every env, written by the step / reset kernels themselves: ONE copy per step;
        # reward / done / delta / nlegal below are views into it
        self.record = torch.empty((self.n_envs, RECORD_BYTES), dtype=torch.uint8, device=self.device)
        rec = unpack_records(self.record)
        self.reward, self.delta, self.nlegal, self.cum = rec["reward"], rec["delta"], rec["nlegal"], rec["cum"]
        self.done = torch.empty(self.n_envs, dtype=torch.uint8, device=self.device)
        self.legal = torch.empty((self.n_envs, self.batch.legal_words), dtype=torch.int64, device=self.device)
        self.region = torch.empty(self.n_envs, dtype=torch.int32, device=self.device)
        # fixed-shape spaces (envs/spaces.py): grid rows are the device buffer's own flat rows [B, stride >= (2+7*Kmax)*N]
        from . import spaces as xr_spaces
        self.kmax = max(self.batch.k_max, 1)
        dims = set(tuple(int(v) for v in r.dims) for r in self.batch.regions)
        self.single_observation_space = self.single_action_space = self.observation_space = self.action_space = None
        if len(dims) == 1:
            d = dims.pop()
            self.single_observation_space, self.single_action_space = xr_spaces.fixed_spaces(d, self.kmax)
            self.observation_space, _ = xr_spaces.fixed_spaces(d, self.kmax, batch=self.n_envs, row=int(self.batch.obs_env_stride))
            self.action_space = self.single_action_space

    def observation_dict(self) -> dict:
        """The current observation as a member of `observation_space`: the grid buffer itself (no copy) + the legal mask."""
        return {"grid": self.obs, "legal_mask": self.legal_mask()}

    def legal_mask(self) -> torch.Tensor:
        """uint8 [B, Kmax] (the `legal_mask` of the Dict space): bit n-1 of an env's row <=> net n is in its netSet; a device op on
        the bitmasks of the last step / reset."""
        bits = torch.arange(64, device=self.device, dtype=torch.int64)
        m = ((self.legal.unsqueeze(-1) >> bits) & 1).reshape(self.n_envs, -1)
        return m[:, :self.kmax].to(torch.uint8)

    def _collect(self, observe: bool):
        b = self.batch
        if observe and self.with_observation:
            b.observation(self.obs)
        b.fetch("record", self.record)
        b.fetch("done", self.done)
        b.fetch("legal", self.legal)
        b.fetch("region", self.region)          # the region every slot is playing (key of agents.NetVectorCache)
        return (self.observation_dict() if self.dict_observation and self.with_observation else self.obs), self.reward, self.done, {"delta": self.delta, "nlegal": self.nlegal, "legal": self.legal,
                                                  "region": self.region, "record": self.record, "cum": self.cum}

    def reset(self):
        self.batch.reset(rotate=True)
        obs, _, _, info = self._collect(observe=True)
        return obs, info

    def step(self, actions: torch.Tensor):
        if self.with_observation:
            # route + observation of every env; self.obs is this env's own persistent buffer, so the in-place form applies:
            # only the planes that change are written (byte-identical to a full write; do not write into `obs` yourself)
            self.batch.step(actions, self.obs, inplace=True)
            return self._collect(observe=False)
        self.batch.step(actions)
        return self._collect(observe=False)

    def random_actions(self, seed: int, out: Optional[torch.Tensor] = None):
        return self.batch.random_actions(seed, out)
