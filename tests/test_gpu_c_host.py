"""The C ABI driven by a plain C host (examples/c_host/step_demo.c) — no Python, no torch in that process — gives
the oracle's numbers: the boundary is a real C-ABI shared library, not a torch extension."""
import os
import subprocess

import numpy as np
import pytest

from xroute_env_amd.regions import generate_region

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_host_matches_oracle(tmp_path):
    from oracle import xr_oracle as orc
    exe = tmp_path / "step_demo"
    src = os.path.join(ROOT, "examples", "c_host", "step_demo.c")
    libdir = os.path.join(ROOT, "xroute_env_amd")
    # plain gcc: the host is C; the HIP runtime API header only needs the platform macro on the command line
    r = subprocess.run(["gcc", "-std=c11", "-O2", "-D__HIP_PLATFORM_AMD__", src, "-I", os.path.join(ROOT, "include"),
                        "-I", "/opt/rocm/include", "-L", libdir, "-lxroute_hip", "-L", "/opt/rocm/lib", "-lamdhip64",
                        f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    reg = generate_region(9800, dims=(24, 40, 9), k_range=(7, 7))
    X, Y, Z = reg.dims
    with open(tmp_path / "region.bin", "wb") as f:
        f.write(np.array([X, Y, Z, reg.n_nets, *reg.metrics0.tolist()], np.int32).tobytes())
        f.write(reg.xs.astype(np.int32).tobytes()); f.write(reg.ys.astype(np.int32).tobytes())
        ld = np.zeros((Z + 3) & ~3, np.uint8); ld[:Z] = reg.layer_dir
        f.write(ld.tobytes()); f.write(reg.nodes.astype(np.uint32).tobytes())
    out = subprocess.run([str(exe), str(tmp_path / "region.bin")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = [[int(v) for v in line.split()] for line in out.stdout.strip().splitlines()]
    env = orc.OracleEnv(reg)
    assert len(rows) == reg.n_nets
    def fnv(buf: bytes) -> int:            # position-weighted sum of the fp32 words, mod 2^64 (step_demo.c)
        w = np.frombuffer(buf, np.uint32).astype(np.uint64)
        with np.errstate(over="ignore"):
            return int((w * (np.arange(w.size, dtype=np.uint64) + np.uint64(1))).sum(dtype=np.uint64))

    for a, dv, dw, dvia, done, plen, h, nleg, obs_fnv in rows:
        ref = env.step(a)
        assert [dv, dw, dvia] == ref["delta"].tolist() and bool(done) == ref["done"] and plen == ref["path_len"]
        assert h == env.hash()
        # the observation crossed the plain-C boundary too: xr_batch_observation (odd actions) / xr_batch_step_observe (even ones)
        assert nleg == env.nlegal()
        assert obs_fnv == fnv(np.ascontiguousarray(env.observation()).tobytes()), a
