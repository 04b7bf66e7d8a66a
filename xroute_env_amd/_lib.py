"""ctypes binding of libxroute_hip.so (C ABI: include/xroute_hip.h).

The library is the product's only compute path.  If it is missing this module raises — there is
no CPU fallback anywhere in the package.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("XR_LIB", "libxroute_hip.so"))   # XR_LIB: experiment builds

XR_OK = 0
ABI_VERSION = 9
XR_ERR_INVALID, XR_ERR_NOMEM, XR_ERR_HIP, XR_ERR_STATE, XR_ERR_RANGE, XR_ERR_PARSE = -1, -2, -3, -4, -5, -6

XR_ENV_OK, XR_ENV_BAD_ACTION, XR_ENV_UNREACHABLE, XR_ENV_PATH_TRUNC, XR_ENV_WAS_RESET, XR_ENV_ROUTER_ABORT = 0, 1, 2, 4, 8, 16
XR_OWNER_FOREIGN = 0x7FFF

(XR_FETCH_CUM, XR_FETCH_DELTA, XR_FETCH_REWARD, XR_FETCH_DONE, XR_FETCH_NLEGAL, XR_FETCH_STATUS,
 XR_FETCH_LEGAL, XR_FETCH_PATH_LEN, XR_FETCH_PATH, XR_FETCH_OWNER, XR_FETCH_HASH, XR_FETCH_REGION,
 XR_FETCH_STEPS, XR_FETCH_SWEEPS, XR_FETCH_PHASES, XR_FETCH_RECORD, XR_FETCH_TOUCHED, XR_FETCH_UNITS,
 XR_FETCH_ROUTE_ORDER, XR_FETCH_REPLAY, XR_FETCH_ENV_STEPS) = range(21)

# every symbol include/xroute_hip.h declares (tests check the library exports all of them)
SYMBOLS = [
    "xr_abi_version", "xr_last_error", "xr_config_default", "xr_device_count",
    "xr_batch_create", "xr_batch_destroy", "xr_batch_load_regions", "xr_batch_assign", "xr_batch_sizes",
    "xr_batch_reset", "xr_batch_step", "xr_batch_step_observe", "xr_batch_step_observe_inplace", "xr_batch_step_compact", "xr_batch_net_planes", "xr_batch_route_order", "xr_batch_observe_timing", "xr_batch_route_occupancy", "xr_batch_random_actions", "xr_batch_observation", "xr_batch_fetch", "xr_batch_store", "xr_batch_load_guides",
    "xr_batch_state_row_bytes", "xr_batch_pack_state", "xr_batch_expand_state", "xr_batch_ingest_state",
    "xr_agent_obstacle_tower_weights", "xr_agent_obstacle_tower", "xr_agent_net_tower_weights", "xr_agent_matrix_mode", "xr_batch_net_vectors", "xr_agent_actor_weights", "xr_agent_actor", "xr_agent_actor_sample",
    "xr_observation_from_records", "xr_proto_decode", "xr_proto_encode_response", "xr_proto_encode_request",
]


class XrConfig(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32), ("n_envs", C.c_int32),
                ("via_cost", C.c_int32), ("drc_cost", C.c_int32), ("drc_unit", C.c_int32),
                ("max_route_count", C.c_int32), ("auto_reset", C.c_int32), ("path_cap", C.c_int32),
                ("block_threads", C.c_int32), ("force_scratch_field", C.c_int32), ("obs_mode", C.c_int32),
                ("w_violation", C.c_double), ("w_via", C.c_double), ("w_wirelength", C.c_double),
                ("obs_writer_blocks", C.c_int32), ("router", C.c_int32), ("dial_mult", C.c_int32),
                ("guide_cost", C.c_int32), ("guide_margin", C.c_int32), ("maze_end_iter", C.c_int32),
                ("stream_per_region", C.c_int32), ("obs_helper_blocks", C.c_int32), ("obs_split_permille", C.c_int32),
                ("launch_order", C.c_int32), ("debug_round_cap", C.c_int32), ("window", C.c_int32)]


class XrStepRecord(C.Structure):          # include/xroute_hip.h xr_step_record (48 bytes)
    _fields_ = [("reward", C.c_double), ("delta", C.c_int32 * 3), ("cum", C.c_int32 * 3), ("nlegal", C.c_int32),
                ("env_steps", C.c_int32), ("path_len", C.c_int32), ("done", C.c_uint8), ("pad", C.c_uint8),
                ("status", C.c_uint16)]


RECORD_BYTES = 48
XR_ROUTER_SWEEP, XR_ROUTER_DIAL = 1, 2


class XrRegionDesc(C.Structure):
    _fields_ = [("dim_x", C.c_int32), ("dim_y", C.c_int32), ("dim_z", C.c_int32),
                ("xs_host", C.c_void_p), ("ys_host", C.c_void_p), ("layer_dir_host", C.c_void_p),
                ("nodes_host", C.c_void_p), ("n_nets", C.c_int32), ("metrics0", C.c_int32 * 3)]


class XRouteError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libxroute_hip error {code}: {msg}")
        self.code = code


_LIB = None


def lib():
    """Load libxroute_hip.so; raise loudly when it has not been built (no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `make -C xroute_env_amd/csrc` "
            "(or __graft_entry__.build()). xroute_env_amd has no CPU fallback.")
    # torch first: the library must bind to the SAME libamdhip64 instance torch uses (streams and
    # device pointers are shared with it); loading ours first would pull in a second HIP runtime.
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.xr_abi_version.restype = C.c_int32
    L.xr_last_error.restype = C.c_char_p
    L.xr_config_default.argtypes = [C.POINTER(XrConfig)]
    L.xr_config_default.restype = None
    L.xr_device_count.argtypes = [C.POINTER(C.c_int32)]
    L.xr_batch_create.argtypes = [C.POINTER(XrConfig), C.POINTER(vp)]
    L.xr_batch_destroy.argtypes = [vp]
    L.xr_batch_load_regions.argtypes = [vp, C.POINTER(XrRegionDesc), C.c_int32, vp]
    L.xr_batch_assign.argtypes = [vp, vp]
    L.xr_batch_sizes.argtypes = [vp] + [C.POINTER(C.c_int32)] * 6 + [C.POINTER(C.c_int64)]
    L.xr_batch_reset.argtypes = [vp, vp, C.c_int32, vp]
    L.xr_batch_step.argtypes = [vp, vp, vp]
    L.xr_batch_step_observe.argtypes = [vp, vp, vp, C.c_int64, vp]
    L.xr_batch_step_observe_inplace.argtypes = [vp, vp, vp, C.c_int64, vp]
    L.xr_batch_step_compact.argtypes = [vp, vp, vp, C.c_int64, vp]
    L.xr_batch_net_planes.argtypes = [vp, vp, vp, C.c_int32, vp, C.c_int64, vp]
    L.xr_batch_route_order.argtypes = [vp, vp, C.c_int32, vp, vp]
    L.xr_batch_observe_timing.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_float)]
    L.xr_batch_route_occupancy.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    L.xr_batch_random_actions.argtypes = [vp, vp, C.c_uint64, vp]
    L.xr_batch_observation.argtypes = [vp, vp, C.c_int64, C.c_int32, C.c_int32, vp]
    L.xr_batch_fetch.argtypes = [vp, C.c_int32, vp, C.c_size_t, vp]
    L.xr_batch_store.argtypes = [vp, C.c_int32, vp, C.c_size_t, vp]
    L.xr_batch_load_guides.argtypes = [vp, vp, vp, vp]
    L.xr_batch_state_row_bytes.argtypes = [vp, C.POINTER(C.c_int64)]
    L.xr_batch_pack_state.argtypes = [vp, vp, C.c_int64, C.c_int32, vp]
    L.xr_batch_expand_state.argtypes = [vp, vp, C.c_int64, C.c_int32, vp, C.c_int64, vp, vp, vp]
    L.xr_batch_ingest_state.argtypes = [vp, vp, vp, vp, vp]
    L.xr_agent_obstacle_tower_weights.argtypes = []
    L.xr_agent_obstacle_tower.argtypes = [vp, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, vp, C.c_int32, vp]
    L.xr_agent_net_tower_weights.argtypes = []
    L.xr_agent_matrix_mode.argtypes = []
    L.xr_batch_net_vectors.argtypes = [vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp, C.c_int32, vp]
    L.xr_agent_actor_weights.argtypes = []
    L.xr_agent_actor.argtypes = [vp, vp, C.c_int64, C.c_int32, vp, vp, vp, vp, C.c_int32, vp, C.c_int32, C.c_int32, vp, vp, vp]
    L.xr_agent_actor_sample.argtypes = [vp, vp, C.c_int64, C.c_int32, vp, vp, vp, vp, C.c_int32, vp, C.c_int32, C.c_int32, vp, vp, vp, C.c_uint64, vp]
    L.xr_observation_from_records.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, vp, C.c_int32, vp, vp]
    L.xr_proto_decode.argtypes = [vp, C.c_size_t, vp, vp, vp, C.c_int64, vp, C.c_int64]
    L.xr_proto_encode_response.argtypes = [C.c_int32, vp, C.POINTER(C.c_size_t)]
    L.xr_proto_encode_request.argtypes = [C.c_int32, C.c_int32, C.c_int32, vp, C.c_int32, vp, C.c_int32, vp,
                                          C.c_int32, vp, C.POINTER(C.c_size_t)]
    for name in SYMBOLS:
        fn = getattr(L, name)
        if name not in ("xr_last_error", "xr_config_default"):
            fn.restype = C.c_int32
    if L.xr_abi_version() != ABI_VERSION:
        raise RuntimeError("libxroute_hip.so ABI version mismatch")
    _LIB = L
    return L


def check(code: int):
    if code != XR_OK:
        raise XRouteError(code, lib().xr_last_error().decode(errors="replace"))


def default_config() -> XrConfig:
    cfg = XrConfig()
    lib().xr_config_default(C.byref(cfg))
    return cfg
