"""The C-ABI library loads on a CPU-only box and exports every symbol include/xroute_hip.h declares.
No compute call is made here (there is no GPU in the build container)."""
import ctypes as C
import os
import re

import pytest

from xroute_env_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "xroute_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(xr_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported():
    L = _lib.lib()
    declared = _declared_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(L, name), f"libxroute_hip.so does not export {name}"
    assert sorted(_lib.SYMBOLS) == declared


def test_abi_version_and_config_defaults():
    L = _lib.lib()
    assert L.xr_abi_version() == 9
    cfg = _lib.default_config()
    assert cfg.struct_size == C.sizeof(_lib.XrConfig)
    assert (cfg.via_cost, cfg.drc_cost, cfg.drc_unit, cfg.max_route_count) == (800, 8, 400, 10)
    assert (cfg.w_violation, cfg.w_via, cfg.w_wirelength) == (500.0, 4.0, 0.5)


def test_argument_errors_without_gpu():
    """Error convention: negative status + thread-local message; never a crash."""
    L = _lib.lib()
    assert L.xr_batch_create(None, None) == _lib.XR_ERR_INVALID
    assert b"null" in L.xr_last_error()
    cfg = _lib.default_config()
    cfg.struct_size = 4
    h = C.c_void_p()
    assert L.xr_batch_create(C.byref(cfg), C.byref(h)) == _lib.XR_ERR_INVALID
    assert b"ABI" in L.xr_last_error()
    assert L.xr_batch_step(None, None, None) == _lib.XR_ERR_INVALID
    assert L.xr_batch_fetch(None, 0, None, 0, None) == _lib.XR_ERR_INVALID


def test_product_fails_loudly_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from xroute_env_amd.batch import RegionBatch
    from xroute_env_amd.build_3Dgrid import build_3Dgrid
    from xroute_env_amd.regions import generate_region
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        RegionBatch([generate_region(1, dims=(3, 3, 2), k_range=(1, 1))])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        build_3Dgrid([[1, 1, 1], [], [0, 0, 0], []], set())


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under xroute_env_amd/ may reference it."""
    pkg = os.path.join(ROOT, "xroute_env_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "xr_oracle" not in src and "xro_" not in src and "from oracle" not in src, f


def test_oracle_header_states_its_role_and_parity_status():
    """oracle/ is test infrastructure and says so; the router half is declared parity-unpinned."""
    for f in ("xr_oracle.h", "xr_oracle.c", "xr_oracle.py"):
        text = open(os.path.join(ROOT, "oracle", f)).read()
        assert "TEST INFRASTRUCTURE ONLY" in text, f
    h = open(os.path.join(ROOT, "oracle", "xr_oracle.h")).read()
    assert "PARITY UNPINNED" in h and "PINNED against fixtures" in h
    assert "parity unpinned" in open(os.path.join(ROOT, "DESIGN.md")).read().lower()


def test_no_reference_source_under_tests_or_repo():
    """Fixtures are data: no file of the reference tree (by name) is kept in the repository."""
    names = {"baseline_utils.py", "net_ordering_pb2.py", "train_DQN.py", "train_PPO.py", "launch_training.py"}
    for dirpath, dirs, files in os.walk(ROOT):
        if ".git" in dirpath or "gpurun_out" in dirpath:
            continue
        assert not (names & set(files)), (dirpath, names & set(files))


def test_integration_md_stub_is_generated_from_the_binding():
    """INTEGRATION.md §3.3 (the ctypes stub a maintainer would start from) is generated from xroute_env_amd/_lib.py by
    tools/gen_integration_stub.py: the ABI version, the config struct and every entry point's argument types cannot go stale."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_integration_stub", os.path.join(root, "tools", "gen_integration_stub.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    block = doc[doc.index(gen.BEGIN) + len(gen.BEGIN): doc.index(gen.END)].strip()
    assert block == gen.stub_text().strip(), "run `python tools/gen_integration_stub.py`"
    from xroute_env_amd import _lib
    L = _lib.lib()
    no_args = {"xr_abi_version", "xr_last_error", "xr_agent_obstacle_tower_weights", "xr_agent_actor_weights", "xr_agent_net_tower_weights", "xr_agent_matrix_mode"}
    assert all(getattr(L, n).argtypes for n in _lib.SYMBOLS if n not in no_args), "every exported symbol is bound with its argument types"
    # ... and the header declares exactly the symbols the binding lists
    import re
    hdr = open(os.path.join(root, "include", "xroute_hip.h")).read()
    declared = set(re.findall(r"^(?:int32_t|void|const char\*)\s+(xr_[a-z0-9_]+)\s*\(", hdr, re.M))
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
