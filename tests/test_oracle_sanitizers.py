"""The CPU oracle under AddressSanitizer + UBSan (sanitizers run on the CPU build only): one full episode and the
observation builder on odd shapes, in a child process with the sanitizer runtime preloaded."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes as C, sys
sys.path.insert(0, %r)
import numpy as np
from oracle import xr_oracle as orc
orc._LIB = None
orc.build = lambda force=False: %r
from xroute_env_amd.regions import generate_region
for seed, dims in ((1, (7, 5, 3)), (2, (24, 40, 9)), (3, (1, 1, 1)), (4, (3, 9, 2))):
    reg = generate_region(9500 + seed, dims=dims, k_range=(1, 5), net_span=4)
    env = orc.OracleEnv(reg)
    obs0 = env.observation()
    for a in env.legal().tolist():
        r = env.step(a, path_cap=3)           # tiny path buffer: truncation path
        env.observation()
    assert env.nlegal() == 0
    env.reset(); assert np.array_equal(env.observation(), obs0)
    assert env.step(0)["status"] == 1
print("ASAN_OK")
'''


def test_oracle_under_asan_ubsan():
    so = os.path.join(ROOT, "oracle", "libxr_oracle_asan.so")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "libxr_oracle_asan.so"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer build unavailable: " + r.stderr[-200:])
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not found")
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    out = subprocess.run([sys.executable, "-c", CHILD % (ROOT, so)], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0 and "ASAN_OK" in out.stdout, (out.stdout[-500:], out.stderr[-3000:])
