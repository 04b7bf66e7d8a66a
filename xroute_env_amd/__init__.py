"""xroute_env_amd — MI355X-native implementation of xroute_env's batched env.step() hot path.

    from xroute_env_amd import Game, build_3Dgrid, handle_messange      # the reference's names
    from xroute_env_amd import RegionBatch, XRouteVectorEnv              # the batched form

Compute lives in libxroute_hip.so (hand-written gfx950 kernels behind the C ABI of
include/xroute_hip.h); there is no CPU fallback.  Submodules import torch lazily so that the
region model and the wire codec are usable in plain host tools.
"""
ENV_ID = "xroute_env/ordering-training-v0"      # reference xroute_env/__init__.py:3-6

_LAZY = {
    "Game": ("game", "Game"),
    "reward_from_deltas": ("game", "reward_from_deltas"),
    "build_3Dgrid": ("build_3Dgrid", "build_3Dgrid"),
    "handle_messange": ("proto", "handle_messange"),
    "RegionBatch": ("batch", "RegionBatch"),
    "XRouteVectorEnv": ("envs.vector_env", "XRouteVectorEnv"),
    "OrderingTrainingEnv": ("envs.facade", "OrderingTrainingEnv"),
    "OrderingEvaluationEnv": ("envs.facade", "OrderingEvaluationEnv"),
    "StaticRegionEnv": ("envs.facade", "StaticRegionEnv"),
    "XRouteEnv": ("envs.facade", "XRouteEnv"),
    "A3CGame": ("envs.order_contracts", "A3CGame"),
    "Route": ("envs.order_contracts", "Route"),
    "OrderVectorEnv": ("envs.order_contracts", "OrderVectorEnv"),
    "Region": ("regions", "Region"),
    "generate_region": ("regions", "generate_region"),
    "config_regions": ("regions", "config_regions"),
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        mod, attr = _LAZY[name]
        return getattr(importlib.import_module(f"{__name__}.{mod}"), attr)
    raise AttributeError(name)


def register_gym():
    """Register `xroute_env/ordering-training-v0` when gymnasium is installed."""
    try:
        from gymnasium.envs.registration import register
    except Exception:
        return False
    register(id=ENV_ID, entry_point="xroute_env_amd.envs:OrderingTrainingEnv")
    # the static-region ids the reference sketches (xroute_env/__init__.py:25-33): `xroute_env/static-region1-v0`
    from .envs.facade import STATIC_REGIONS
    for region in STATIC_REGIONS:
        register(id="xroute_env/static-{}-v0".format(region["benchmark"]), entry_point="xroute_env_amd.envs:StaticRegionEnv",
                 kwargs={"region": region})
    return True
