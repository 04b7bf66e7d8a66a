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


class XRouteVectorEnv:
    def __init__(self, regions: Sequence, n_envs: Optional[int] = None, device="cuda:0", with_observation: bool = True,
                 **batch_kw):
        self.batch = RegionBatch(regions, n_envs=n_envs, device=device, auto_reset=True, **batch_kw)
        self.n_envs = self.batch.n_envs
        self.device = self.batch.device
        self.with_observation = with_observation
        self.obs = self.batch.alloc_observation() if with_observation else None
        self.reward = torch.empty(self.n_envs, dtype=torch.float64, device=self.device)
        self.done = torch.empty(self.n_envs, dtype=torch.uint8, device=self.device)
        self.delta = torch.empty((self.n_envs, 3), dtype=torch.int32, device=self.device)
        self.nlegal = torch.empty(self.n_envs, dtype=torch.int32, device=self.device)
        self.legal = torch.empty((self.n_envs, self.batch.legal_words), dtype=torch.int64, device=self.device)
        self.region = torch.empty(self.n_envs, dtype=torch.int32, device=self.device)

    def _collect(self, observe: bool):
        b = self.batch
        if observe and self.with_observation:
            b.observation(self.obs)
        b.fetch("reward", self.reward)
        b.fetch("done", self.done)
        b.fetch("delta", self.delta)
        b.fetch("nlegal", self.nlegal)
        b.fetch("legal", self.legal)
        b.fetch("region", self.region)          # the region every slot is playing (key of agents.NetVectorCache)
        return self.obs, self.reward, self.done, {"delta": self.delta, "nlegal": self.nlegal, "legal": self.legal,
                                                  "region": self.region}

    def reset(self):
        self.batch.reset(rotate=True)
        obs, _, _, info = self._collect(observe=True)
        return obs, info

    def step(self, actions: torch.Tensor):
        if self.with_observation:
            self.batch.step(actions, self.obs)          # one fused launch: route + observation of every env
            return self._collect(observe=False)
        self.batch.step(actions)
        return self._collect(observe=False)

    def random_actions(self, seed: int, out: Optional[torch.Tensor] = None):
        return self.batch.random_actions(seed, out)
