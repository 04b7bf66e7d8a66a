"""RegionBatch — Python host layer over the `xr_batch_*` C ABI (include/xroute_hip.h).

A batch holds B env slots on one MI355X.  torch is plumbing only: it allocates the caller-side
device buffers (actions, observation, fetched results) and supplies the HIP stream; all compute is in
libxroute_hip.so.  One process per GPU; see xroute_env_amd/dist.py for the sharded launcher.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .regions import Region


RECORD_DTYPE = np.dtype([("reward", "<f8"), ("delta", "<i4", (3,)), ("cum", "<i4", (3,)), ("nlegal", "<i4"),
                         ("env_steps", "<i4"), ("path_len", "<i4"), ("done", "u1"), ("pad", "u1"), ("status", "<u2")])
assert RECORD_DTYPE.itemsize == _lib.RECORD_BYTES


def _require_gpu(device) -> torch.device:
    dev = torch.device(device)
    if dev.type != "cuda" or not torch.cuda.is_available():
        raise RuntimeError("xroute_env_amd needs a HIP device (MI355X): torch.cuda.is_available() is False "
                           "and there is no CPU fallback")
    return dev


def _stream_ptr(dev) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class RegionBatch:
    """B env slots playing a set of regions.  Mirrors, per env, the reference's Game
    (baseline/baseline_utils.py:383-481): reset() / step(actions) / observation()."""

    def __init__(self, regions: Sequence[Region], n_envs: Optional[int] = None, device="cuda:0",
                 auto_reset: bool = False, via_cost: int = 800, drc_cost: int = 8, drc_unit: int = 400,
                 max_route_count: int = 10, path_cap: int = 0, block_threads: int = 0,
                 force_scratch_field: bool = False, obs_mode: int = 0, obs_writer_blocks: int = 0,
                 obs_split_permille: int = 0, router: int = 0, dial_mult: int = 0,
                 stream_per_region: bool = False, obs_helper_blocks: int = 0, launch_order: int = 0,
                 guide_cost: int = 0, guide_margin: int = 0, maze_end_iter: int = 1, debug_round_cap: int = 0,
                 window: int = 0):
        self.device = _require_gpu(device)
        self.L = _lib.lib()
        self.regions = list(regions)
        self.n_envs = int(n_envs if n_envs is not None else len(self.regions))
        cfg = _lib.default_config()
        cfg.device = self.device.index if self.device.index is not None else torch.cuda.current_device()
        cfg.n_envs = self.n_envs
        cfg.via_cost, cfg.drc_cost, cfg.drc_unit = via_cost, drc_cost, drc_unit
        cfg.max_route_count = max_route_count
        cfg.auto_reset = int(auto_reset)
        cfg.path_cap = path_cap
        cfg.block_threads = block_threads
        cfg.force_scratch_field = int(force_scratch_field)
        cfg.obs_mode = int(obs_mode)                    # 0 default, 1 fused single launch, 2 split (route || net-plane writer)
        cfg.obs_writer_blocks = int(obs_writer_blocks)
        cfg.obs_split_permille = int(obs_split_permille)   # split form: share of the net planes the writer kernel takes
        cfg.router = int(router)                        # 0 auto (frontier router; the full-rewrite queue launch of >= 4096 slots takes the sweeps), 1 line-segment sweeps, 2 frontier (required)
        cfg.dial_mult = int(dial_mult)
        cfg.launch_order = int(launch_order)             # route-only launches: 0 auto, 1 slot order, 2 longest predicted route first
        cfg.obs_helper_blocks = int(obs_helper_blocks)   # queue form: LDS-free helper writers beside the step kernel (0 = none = the default; < 0 is rejected)
        cfg.debug_round_cap = int(debug_round_cap)       # 0 default (1024 + N rounds per search); tests force XR_ENV_ROUTER_ABORT with 1
        cfg.guide_cost, cfg.guide_margin, cfg.maze_end_iter = int(guide_cost), int(guide_margin), int(maze_end_iter)   # XR-Maze v2
        cfg.window = int(window)                         # regions too large for LDS: > 0 = LDS-window router first (at most that many tracks); 0 = off (measured no faster)
        cfg.stream_per_region = int(stream_per_region)   # one single-workgroup launch per env slot on a pool of streams (<= 64 slots)
        self.cfg = cfg
        self.region_epoch = 0      # bumped whenever the HOST changes which region a slot plays (assign, reset(rotate), load_state_dict)
        self._h = C.c_void_p()
        _lib.check(self.L.xr_batch_create(C.byref(cfg), C.byref(self._h)))
        with torch.cuda.device(self.device):
            self._load()

    # ------------------------------------------------------------------------------------------
    def _load(self):
        descs = (_lib.XrRegionDesc * len(self.regions))()
        keep = []
        for i, r in enumerate(self.regions):
            xs = np.ascontiguousarray(r.xs, np.int32)
            ys = np.ascontiguousarray(r.ys, np.int32)
            ld = np.ascontiguousarray(r.layer_dir, np.uint8)
            nd = np.ascontiguousarray(r.nodes, np.uint32)
            if nd.size != r.n_nodes:
                raise ValueError(f"region {i}: {nd.size} node records for dims {r.dims}")
            keep += [xs, ys, ld, nd]
            d = descs[i]
            d.dim_x, d.dim_y, d.dim_z = (int(v) for v in r.dims)
            d.xs_host, d.ys_host = xs.ctypes.data, ys.ctypes.data
            d.layer_dir_host, d.nodes_host = ld.ctypes.data, nd.ctypes.data
            d.n_nets = int(r.n_nets)
            for j in range(3):
                d.metrics0[j] = int(r.metrics0[j])
        _lib.check(self.L.xr_batch_load_regions(self._h, descs, len(self.regions), _stream_ptr(self.device)))
        self._load_guides()
        sz = [C.c_int32() for _ in range(6)]
        stride = C.c_int64()
        _lib.check(self.L.xr_batch_sizes(self._h, *[C.byref(s) for s in sz], C.byref(stride)))
        (_, self.n_regions, self.n_max, self.k_max, self.legal_words, self.path_cap) = (s.value for s in sz)
        self.obs_env_stride = stride.value

    def _load_guides(self):
        """XR-Maze v2: regions that carry global-route guide boxes (`Region.guide_off / guide_box`, lefdef.RegionExtractor) hand
        them to the device (xr_batch_load_guides); the others keep the default guide (bounding box of the net's access points)."""
        if not any(getattr(r, "guide_off", None) is not None for r in self.regions):
            return
        n = len(self.regions)
        offs, boxes, keep = (C.c_void_p * n)(), (C.c_void_p * n)(), []
        for i, r in enumerate(self.regions):
            if getattr(r, "guide_off", None) is None:
                continue
            off = np.ascontiguousarray(r.guide_off, np.int32)
            box = np.ascontiguousarray(r.guide_box, np.int16).reshape(-1, 6)
            if off.size != r.n_nets + 1 or int(off[-1]) != box.shape[0]:
                raise ValueError(f"region {i}: guide_off must have n_nets + 1 entries and end at the number of boxes")
            if box.shape[0] == 0:
                box = np.zeros((1, 6), np.int16)
            keep += [off, box]
            offs[i], boxes[i] = off.ctypes.data, box.ctypes.data
        _lib.check(self.L.xr_batch_load_guides(self._h, offs, boxes, _stream_ptr(self.device)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.L.xr_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------
    def assign(self, env_region: Sequence[int]):
        a = np.ascontiguousarray(env_region, np.int32)
        if a.size != self.n_envs:
            raise ValueError("env_region must have n_envs entries")
        _lib.check(self.L.xr_batch_assign(self._h, a.ctypes.data))
        self.region_epoch += 1

    def reset(self, mask: Optional[torch.Tensor] = None, rotate: bool = False):
        """Game.reset for the masked envs (all when mask is None)."""
        ptr = None
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            if mask.numel() != self.n_envs:
                raise ValueError("mask must have n_envs entries")
            ptr = C.c_void_p(mask.data_ptr())
        with torch.cuda.device(self.device):
            _lib.check(self.L.xr_batch_reset(self._h, ptr, int(rotate), _stream_ptr(self.device)))
        self.region_epoch += int(bool(rotate))

    def step(self, actions: torch.Tensor, obs_out: Optional[torch.Tensor] = None, inplace: bool = False):
        """Game.step for every env: actions int32[B] on the device, 1-based net ids.  With `obs_out`
        ([n_envs, stride] fp32) the same launch also writes every env's observation of the new state
        (xr_batch_step_observe).  `inplace=True` (xr_batch_step_observe_inplace): `obs_out` is the buffer that received this
        batch's previous observation and has not been written to since — only the planes that change are written."""
        if actions.device != self.device or actions.dtype != torch.int32 or not actions.is_contiguous() \
                or actions.numel() != self.n_envs:
            raise ValueError("actions must be a contiguous int32 tensor of n_envs entries on the batch device")
        with torch.cuda.device(self.device):
            if obs_out is None:
                _lib.check(self.L.xr_batch_step(self._h, C.c_void_p(actions.data_ptr()), _stream_ptr(self.device)))
            else:
                if obs_out.device != self.device or obs_out.dtype != torch.float32 or not obs_out.is_contiguous() \
                        or obs_out.dim() != 2 or obs_out.shape[0] < self.n_envs:
                    raise ValueError("obs_out must be a contiguous fp32 [n_envs, stride] tensor on the batch device")
                fn = self.L.xr_batch_step_observe_inplace if inplace else self.L.xr_batch_step_observe
                _lib.check(fn(self._h, C.c_void_p(actions.data_ptr()), C.c_void_p(obs_out.data_ptr()), obs_out.shape[1],
                              _stream_ptr(self.device)))
        return obs_out

    def alloc_head(self) -> torch.Tensor:
        """[n_envs, 2*n_max] fp32 buffer for the compact-consumer step (planes 0..1 of every env)."""
        return torch.empty((self.n_envs, 2 * self.n_max), dtype=torch.float32, device=self.device)

    def step_compact(self, actions: torch.Tensor, head_out: torch.Tensor):
        """Game.step for every env + only the two observation planes that change (obstacles, net order):
        xr_batch_step_compact.  The net planes come once per (region, net) from `net_planes`."""
        if actions.device != self.device or actions.dtype != torch.int32 or not actions.is_contiguous() \
                or actions.numel() != self.n_envs:
            raise ValueError("actions must be a contiguous int32 tensor of n_envs entries on the batch device")
        if head_out.device != self.device or head_out.dtype != torch.float32 or not head_out.is_contiguous() \
                or head_out.dim() != 2 or head_out.shape[0] < self.n_envs:
            raise ValueError("head_out must be a contiguous fp32 [n_envs, stride] tensor on the batch device")
        with torch.cuda.device(self.device):
            _lib.check(self.L.xr_batch_step_compact(self._h, C.c_void_p(actions.data_ptr()), C.c_void_p(head_out.data_ptr()),
                                                    head_out.shape[1], _stream_ptr(self.device)))
        return head_out

    def net_planes(self, region: torch.Tensor, net: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The 7 static planes of (region index, 1-based net id) pairs: [n, 7*n_max] fp32 (xr_batch_net_planes)."""
        region = region.to(device=self.device, dtype=torch.int32).contiguous()
        net = net.to(device=self.device, dtype=torch.int32).contiguous()
        n = region.numel()
        if net.numel() != n:
            raise ValueError("region and net must have the same length")
        if out is None:
            out = torch.empty((n, 7 * self.n_max), dtype=torch.float32, device=self.device)
        if out.device != self.device or out.dtype != torch.float32 or not out.is_contiguous() or out.dim() != 2 or out.shape[0] < n:
            raise ValueError("out must be a contiguous fp32 [n, stride] tensor on the batch device")
        with torch.cuda.device(self.device):
            _lib.check(self.L.xr_batch_net_planes(self._h, C.c_void_p(region.data_ptr()), C.c_void_p(net.data_ptr()), n,
                                                  C.c_void_p(out.data_ptr()), out.shape[1], _stream_ptr(self.device)))
        return out

    # ---- compact state for a central learner (gather compact state, expand on the learner GPU) ----------------
    def state_row_bytes(self) -> int:
        """Bytes of one packed state row of this batch (region, nets left, legal bitmask, one occupancy bit per node)."""
        n = C.c_int64()
        _lib.check(self.L.xr_batch_state_row_bytes(self._h, C.byref(n)))
        return int(n.value)

    def pack_state(self, out: Optional[torch.Tensor] = None, region_base: int = 0, row_bytes: Optional[int] = None) -> torch.Tensor:
        """One packed row per env slot (uint8 [n_envs, row_bytes]): what planes 0..1 of its observation are functions of — ready for
        ONE all_gather to a learner.  `region_base` turns the local region index into an index of the learner's region table."""
        rb = int(row_bytes) if row_bytes is not None else (int(out.shape[1]) if out is not None else self.state_row_bytes())
        if out is None:
            out = torch.empty((self.n_envs, rb), dtype=torch.uint8, device=self.device)
        if out.dtype != torch.uint8 or not out.is_contiguous() or tuple(out.shape) != (self.n_envs, rb):
            raise ValueError("out must be a contiguous uint8 [n_envs, row_bytes] tensor")
        with torch.cuda.device(self.device):
            _lib.check(self.L.xr_batch_pack_state(self._h, C.c_void_p(out.data_ptr()), rb, int(region_base), _stream_ptr(self.device)))
        return out

    def ingest_state(self, owner: torch.Tensor, legal: torch.Tensor, cum: torch.Tensor):
        """Adopt a new state of every env slot produced by an EXTERNAL simulator (the client half of the reference's Game.step without the
        route, baseline/baseline_utils.py:420-438 — BASELINE config 2): owner int16 [n_envs, n_max], legal int64 / uint64 [n_envs, legal_words],
        cum int32 [n_envs, 3].  The records then carry the metric deltas, the reward and `done`; `observation()` builds the grid."""
        if owner.dtype != torch.int16 or tuple(owner.shape) != (self.n_envs, self.n_max) or not owner.is_contiguous():
            raise ValueError("owner must be a contiguous int16 [n_envs, n_max] tensor")
        if legal.dtype not in (torch.int64, torch.uint64) or tuple(legal.shape) != (self.n_envs, self.legal_words) or not legal.is_contiguous():
            raise ValueError("legal must be a contiguous 64-bit [n_envs, legal_words] tensor")
        if cum.dtype != torch.int32 or tuple(cum.shape) != (self.n_envs, 3) or not cum.is_contiguous():
            raise ValueError("cum must be a contiguous int32 [n_envs, 3] tensor")
        with torch.cuda.device(self.device):
            _lib.check(self.L.xr_batch_ingest_state(self._h, C.c_void_p(owner.data_ptr()), C.c_void_p(legal.data_ptr()), C.c_void_p(cum.data_ptr()),
                                                    _stream_ptr(self.device)))

    def expand_state(self, rows: torch.Tensor, head_out: Optional[torch.Tensor] = None, nlegal_out: Optional[torch.Tensor] = None,
                     region_out: Optional[torch.Tensor] = None):
        """Learner side: packed rows (uint8 [n, row_bytes], e.g. the all_gather of every rank's `pack_state`) -> (head [n, stride] fp32 with
        planes 0..1 of every env exactly as `step_compact` writes them, nlegal int32 [n], region int32 [n]).  This batch supplies the
        region table (it must hold every region the rows name); rows that do not parse are flagged nlegal = region = -1."""
        if rows.dtype != torch.uint8 or rows.dim() != 2 or not rows.is_contiguous():
            raise ValueError("rows must be a contiguous uint8 [n, row_bytes] tensor")
        n = int(rows.shape[0])
        if head_out is None:
            head_out = torch.empty((n, 2 * self.n_max), dtype=torch.float32, device=self.device)
        if nlegal_out is None:
            nlegal_out = torch.empty(n, dtype=torch.int32, device=self.device)
        if region_out is None:
            region_out = torch.empty(n, dtype=torch.int32, device=self.device)
        if head_out.dtype != torch.float32 or head_out.shape[0] != n or head_out.stride(1) != 1:
            raise ValueError("head_out must be fp32 [n, stride] with unit inner stride")
        with torch.cuda.device(self.device):
            _lib.check(self.L.xr_batch_expand_state(self._h, C.c_void_p(rows.data_ptr()), int(rows.shape[1]), n, C.c_void_p(head_out.data_ptr()),
                                                    int(head_out.stride(0)), C.c_void_p(nlegal_out.data_ptr()), C.c_void_p(region_out.data_ptr()),
                                                    _stream_ptr(self.device)))
        return head_out, nlegal_out, region_out

    def route_occupancy(self):
        """(resident workgroups per CU, LDS bytes per workgroup) of the step kernel for the loaded regions."""
        n, lds = C.c_int32(0), C.c_int64(0)
        _lib.check(self.L.xr_batch_route_occupancy(self._h, C.byref(n), C.byref(lds)))
        return int(n.value), int(lds.value)

    def observe_timing(self):
        """(mode, writer_ms) of the last step(actions, obs_out): mode 1 = fused launch, 2 = split (route kernel and
        net-plane writer running concurrently); writer_ms = HIP-event duration of the writer kernel (0 when fused)."""
        mode, ms = C.c_int32(0), C.c_float(0.0)
        _lib.check(self.L.xr_batch_observe_timing(self._h, C.byref(mode), C.byref(ms)))
        return int(mode.value) & ~32, float(ms.value)       # (form | 16 when the in-place path ran; bit 5: see observe_info)

    def observe_info(self) -> dict:
        """The last step(actions, obs_out) in words: form (1 fused, 2 split, 3 queue), whether the in-place path ran, and
        whether the auto router ran the line-segment sweeps in that launch (full rewrite of a batch of >= 4096 slots)."""
        mode, ms = C.c_int32(0), C.c_float(0.0)
        _lib.check(self.L.xr_batch_observe_timing(self._h, C.byref(mode), C.byref(ms)))
        m = int(mode.value)
        return {"form": m & 15, "inplace": bool(m & 16), "sweeps": bool(m & 32), "writer_ms": float(ms.value)}

    def route_order(self, orders: torch.Tensor, net_stats: Optional[torch.Tensor] = None):
        """Whole-order re-route (xr_batch_route_order): every env restarts its region and routes
        orders[e, :] (int32 [B, stride >= k_max], 1-based ids, 0-terminated) in that order, one launch.
        net_stats: optional int32 [B, stride, 4] (per net: d_vio, d_wl, d_via, route counter)."""
        if orders.device != self.device or orders.dtype != torch.int32 or not orders.is_contiguous() \
                or orders.dim() != 2 or orders.shape[0] != self.n_envs:
            raise ValueError("orders must be a contiguous int32 [n_envs, stride] tensor on the batch device")
        sp = None
        if net_stats is not None:
            if net_stats.device != self.device or net_stats.dtype != torch.int32 or not net_stats.is_contiguous() \
                    or tuple(net_stats.shape) != (self.n_envs, orders.shape[1], 4):
                raise ValueError("net_stats must be a contiguous int32 [n_envs, stride, 4] tensor on the batch device")
            sp = C.c_void_p(net_stats.data_ptr())
        with torch.cuda.device(self.device):
            _lib.check(self.L.xr_batch_route_order(self._h, C.c_void_p(orders.data_ptr()), int(orders.shape[1]), sp,
                                                   _stream_ptr(self.device)))
        return net_stats

    def random_actions(self, seed: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if out is None:
            out = torch.empty(self.n_envs, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.L.xr_batch_random_actions(self._h, C.c_void_p(out.data_ptr()),
                                                      C.c_uint64(seed & (2 ** 64 - 1)), _stream_ptr(self.device)))
        return out

    def alloc_observation(self, n_envs: Optional[int] = None, env_stride: Optional[int] = None) -> torch.Tensor:
        n = self.n_envs if n_envs is None else n_envs
        stride = self.obs_env_stride if env_stride is None else env_stride
        return torch.empty((n, stride), dtype=torch.float32, device=self.device)

    def observation(self, out: Optional[torch.Tensor] = None, env_lo: int = 0, env_hi: Optional[int] = None):
        """build_3Dgrid of the current state of envs [env_lo, env_hi) into `out` ([n, stride] fp32).
        Env i's observation is out[i, :(2+7K_i)*N_i] viewed as [2+7K_i, Z, Y, X]."""
        env_hi = self.n_envs if env_hi is None else env_hi
        if out is None:
            out = self.alloc_observation(env_hi - env_lo)
        if out.device != self.device or out.dtype != torch.float32 or not out.is_contiguous() or out.dim() != 2 \
                or out.shape[0] < env_hi - env_lo:
            raise ValueError("out must be a contiguous fp32 [n_envs, stride] tensor on the batch device")
        with torch.cuda.device(self.device):
            _lib.check(self.L.xr_batch_observation(self._h, C.c_void_p(out.data_ptr()), out.shape[1], env_lo, env_hi,
                                                   _stream_ptr(self.device)))
        return out

    def env_observation(self, e: int, nlegal: Optional[int] = None) -> torch.Tensor:
        """Reference-shaped [1, 2+7K, Z, Y, X] observation of one env (device tensor)."""
        if nlegal is None:
            nlegal = int(self.fetch("nlegal")[e].item())
        reg = self.regions[int(self.fetch("region")[e].item())]
        X, Y, Z = reg.dims
        buf = self.observation(env_lo=e, env_hi=e + 1)
        c = 2 + 7 * nlegal
        return buf[0, : c * reg.n_nodes].view(1, c, Z, Y, X)

    # ------------------------------------------------------------------------------------------
    _FETCH = {
        "cum": (_lib.XR_FETCH_CUM, torch.int32, lambda s: (s.n_envs, 3)),
        "delta": (_lib.XR_FETCH_DELTA, torch.int32, lambda s: (s.n_envs, 3)),
        "reward": (_lib.XR_FETCH_REWARD, torch.float64, lambda s: (s.n_envs,)),
        "done": (_lib.XR_FETCH_DONE, torch.uint8, lambda s: (s.n_envs,)),
        "nlegal": (_lib.XR_FETCH_NLEGAL, torch.int32, lambda s: (s.n_envs,)),
        "status": (_lib.XR_FETCH_STATUS, torch.int32, lambda s: (s.n_envs,)),
        "legal": (_lib.XR_FETCH_LEGAL, torch.int64, lambda s: (s.n_envs, s.legal_words)),
        "path_len": (_lib.XR_FETCH_PATH_LEN, torch.int32, lambda s: (s.n_envs,)),
        "path": (_lib.XR_FETCH_PATH, torch.int32, lambda s: (s.n_envs, s.path_cap)),
        "owner": (_lib.XR_FETCH_OWNER, torch.int16, lambda s: (s.n_envs, s.n_max)),
        "hash": (_lib.XR_FETCH_HASH, torch.int64, lambda s: (s.n_envs,)),
        "region": (_lib.XR_FETCH_REGION, torch.int32, lambda s: (s.n_envs,)),
        "steps": (_lib.XR_FETCH_STEPS, torch.int64, lambda s: (1,)),
        "sweeps": (_lib.XR_FETCH_SWEEPS, torch.int32, lambda s: (s.n_envs,)),
        "phases": (_lib.XR_FETCH_PHASES, torch.int64, lambda s: (s.n_envs, 8)),
        "units": (_lib.XR_FETCH_UNITS, torch.int32, lambda s: (1,)),
        "route_order": (_lib.XR_FETCH_ROUTE_ORDER, torch.int32, lambda s: (s.n_envs,)),
        "touched": (_lib.XR_FETCH_TOUCHED, torch.int32, lambda s: (s.n_envs,)),
        "record": (_lib.XR_FETCH_RECORD, torch.uint8, lambda s: (s.n_envs, _lib.RECORD_BYTES)),
        "replay": (_lib.XR_FETCH_REPLAY, torch.int32, lambda s: (s.n_envs,)),
        "env_steps": (_lib.XR_FETCH_ENV_STEPS, torch.int64, lambda s: (s.n_envs,)),
    }
    # the arrays that ARE the env state (xr_batch_store accepts exactly these)
    # (restore order: `region` first — xr_batch_store checks legal bits and nets-left counts against the region each slot plays)
    _STATE = ("region", "owner", "legal", "nlegal", "cum", "delta", "reward", "done", "status", "path_len", "hash", "replay",
              "env_steps", "record", "steps")

    def fetch(self, what: str, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Copy one result array into a device tensor (async on the current stream)."""
        sel, dtype, shape = self._FETCH[what]
        if out is None:
            out = torch.empty(shape(self), dtype=dtype, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.L.xr_batch_fetch(self._h, sel, C.c_void_p(out.data_ptr()),
                                             out.numel() * out.element_size(), _stream_ptr(self.device)))
        return out

    def fetch_host(self, what: str, out: torch.Tensor) -> torch.Tensor:
        """Copy one result array straight into a PINNED host tensor (async on the current stream; synchronise the stream
        before reading it).  One transfer instead of device copy + .cpu(): the small-batch path."""
        sel, dtype, shape = self._FETCH[what]
        if out.device.type != "cpu" or not out.is_pinned() or out.dtype != dtype or not out.is_contiguous():
            raise ValueError("out must be a contiguous pinned host tensor of the array's dtype")
        with torch.cuda.device(self.device):
            _lib.check(self.L.xr_batch_fetch(self._h, sel, C.c_void_p(out.data_ptr()),
                                             out.numel() * out.element_size(), _stream_ptr(self.device)))
        return out

    def state_dict(self) -> dict:
        """Env checkpoint: every array that is state of the env slots, as CPU tensors (+ the sizes a restore must match).  The
        reference never checkpoints its env — the state lives in the simulator process; only its agents are saved
        (baseline/DQN/DQN.py:236-242)."""
        d = {k: self.fetch(k).cpu() for k in self._STATE}
        d["_meta"] = torch.tensor([self.n_envs, self.n_regions, self.n_max, self.legal_words], dtype=torch.int64)
        d["_fingerprint"] = torch.from_numpy(np.frombuffer(self.fingerprint(), np.uint8).copy())
        return d

    def fingerprint(self) -> bytes:
        """sha256 over what the env state is a state OF: every region's content (dims, tracks, layer directions, node records, initial
        metrics, guide boxes) and the config fields that change what a step computes (costs, rotation period, XR-Maze v2 knobs)."""
        import hashlib
        h = hashlib.sha256()
        c = self.cfg
        h.update(np.array([c.via_cost, c.drc_cost, c.drc_unit, c.max_route_count, c.guide_cost, c.guide_margin, c.maze_end_iter,
                           len(self.regions)], np.int64).tobytes())
        for r in self.regions:
            h.update(np.array([*r.dims, r.n_nets, *[int(v) for v in r.metrics0]], np.int64).tobytes())
            for a, dt in ((r.xs, np.int32), (r.ys, np.int32), (r.layer_dir, np.uint8), (r.nodes, np.uint32)):
                h.update(np.ascontiguousarray(a, dt).tobytes())
            if getattr(r, "guide_off", None) is not None:
                h.update(np.ascontiguousarray(r.guide_off, np.int32).tobytes())
                h.update(np.ascontiguousarray(r.guide_box, np.int16).tobytes())
        return h.digest()

    def load_state_dict(self, d: dict, host_checks: bool = True):
        """Restore a state_dict() into a batch created with the same regions and config: it continues bit-identically.
        All-or-nothing: a dict that is refused — here on the host, or by xr_batch_store (`host_checks=False` leaves the range checks to
        the library: the rollback path) — leaves the batch in the state it had."""
        meta = [int(v) for v in d["_meta"]]
        if meta != [self.n_envs, self.n_regions, self.n_max, self.legal_words]:
            raise ValueError(f"state of a different batch: (n_envs, n_regions, n_max, legal_words) = {meta}")
        if "_fingerprint" in d and bytes(d["_fingerprint"].numpy().tobytes()) != self.fingerprint():
            raise ValueError("state of a different batch: the regions or the step-relevant config (costs, rotation, XR-Maze v2 knobs) differ")
        #  nothing is stored until EVERYTHING has been checked: shapes of all arrays, then on the host what xr_batch_store would
        # refuse later (a region index out of range, legal bits beyond the region's nets, a nets-left count that is not the popcount) — and
        # should a store fail all the same, the batch is rolled back to the state it had, so a caller never holds a half-restored batch.
        optional = ("steps",)                  # a dump written before `steps` was part of the state keeps the batch's own counter
        missing = [k for k in self._STATE if k not in d and k not in optional]
        if missing:
            raise ValueError(f"state dict lacks {missing} (written by an older version of RegionBatch.state_dict?)")
        staged = {}
        for k in self._STATE:
            if k not in d:
                continue
            sel, dtype, shape = self._FETCH[k]
            t = torch.as_tensor(d[k]).to(dtype=dtype).contiguous()
            if tuple(t.shape) != tuple(shape(self)):
                raise ValueError(f"state array {k}: shape {tuple(t.shape)} != {tuple(shape(self))}")
            staged[k] = t
        if host_checks:
            self._check_state_ranges(staged)
        backup = {k: self.fetch(k) for k in staged}
        self.region_epoch += 1
        with torch.cuda.device(self.device):
            try:
                self._store_arrays({k: t.to(self.device) for k, t in staged.items()})
            except Exception:
                self._store_arrays(backup)
                raise

    def _check_state_ranges(self, staged: dict):
        """What xr_batch_store range-checks (csrc/xr_batch.cpp), on the host and BEFORE anything is stored; same error convention."""
        reg = staged["region"].cpu().numpy().astype(np.int64)
        if reg.size and (reg.min() < 0 or reg.max() >= self.n_regions):
            raise _lib.XRouteError(_lib.XR_ERR_RANGE, "state array region: index outside the batch's regions")
        words = staged["legal"].cpu().numpy().view(np.uint64).reshape(self.n_envs, self.legal_words)
        nn = np.array([int(r.n_nets) for r in self.regions], np.int64)[reg]
        bit0 = np.arange(self.legal_words, dtype=np.int64)[None, :] * 64
        nbits = np.clip(nn[:, None] - bit0, 0, 64)
        allowed = np.where(nbits >= 64, np.uint64(0xFFFFFFFFFFFFFFFF), (np.uint64(1) << nbits.astype(np.uint64)) - np.uint64(1))
        if np.any(words & ~allowed):
            raise _lib.XRouteError(_lib.XR_ERR_RANGE, "state array legal: bits beyond the nets of the region a slot plays")
        pop = np.unpackbits(words.view(np.uint8), axis=1).sum(axis=1)
        if not np.array_equal(pop, staged["nlegal"].cpu().numpy().reshape(-1).astype(np.int64)):
            raise _lib.XRouteError(_lib.XR_ERR_RANGE, "state array nlegal: not the number of legal bits")

    def _store_arrays(self, arrays: dict):
        with torch.cuda.device(self.device):
            for k in self._STATE:              # (restore order matters: see _STATE)
                if k not in arrays:
                    continue
                sel = self._FETCH[k][0]
                t = arrays[k]
                _lib.check(self.L.xr_batch_store(self._h, sel, C.c_void_p(t.data_ptr()), t.numel() * t.element_size(),
                                                 _stream_ptr(self.device)))
            torch.cuda.current_stream(self.device).synchronize()          # the staging tensors die with the caller's frame

    def legal_sets(self) -> List[set]:
        """netSet of every env as Python sets of 1-based ids (host sync)."""
        words = self.fetch("legal").cpu().numpy().view(np.uint64)
        out = []
        for e in range(self.n_envs):
            s = set()
            for w in range(self.legal_words):
                m = int(words[e, w])
                while m:
                    b = (m & -m).bit_length() - 1
                    s.add(w * 64 + b + 1)
                    m &= m - 1
            out.append(s)
        return out

    def records(self) -> np.ndarray:
        """The packed per-env result records (xr_step_record) as a host structured array: ONE device->host copy."""
        raw = self.fetch("record").cpu().numpy()
        return raw.view(RECORD_DTYPE).reshape(self.n_envs)

    def total_steps(self) -> int:
        return int(self.fetch("steps").item())
