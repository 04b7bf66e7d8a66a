"""Randomised parity fuzz of every router form against the CPU oracle: random region shapes / densities / net shapes, random costs, random
XR-Maze v2 knobs (guide cost + margin, rip-up attempts, random guide boxes), random form (round-3 LDS form, round-2 LDS form, HBM-scratch
form, LDS-window form in front of it, line-segment sweeps) — whole episodes, every step: status, deltas, cumulative metrics, done, path,
owner row, reward; hash chains at the end (tests/test_gpu_route.py::_run_episode_parity).
    python tools/fuzz_router.py [trials=60] [seed=1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_route import _run_episode_parity, _random_guides
from xroute_env_amd.regions import generate_region



def run(T=60, seed=1):
  rng = np.random.default_rng(seed)
  t0 = time.time()
  steps = 0
  tally = {}
  for trial in range(T):
      X, Y, Z = int(rng.integers(6, 31)), int(rng.integers(6, 45)), int(rng.integers(2, 10))
      kmax = int(rng.integers(2, 14))
      gen = dict(dims=(X, Y, Z), k_range=(max(1, kmax // 2), kmax), blockage=(0.05, float(rng.uniform(0.1, 0.4))),
                 prerouted=(0.0, float(rng.uniform(0.02, 0.2))), pins=(2, int(rng.integers(2, 7))), aps=(1, int(rng.integers(1, 5))),
                 net_span=int(rng.integers(3, max(4, min(X, Y)))), used_ap_prob=float(rng.uniform(0.0, 0.15)))
      regions = [generate_region(100000 + 100 * trial + i, **gen) for i in range(10)]
      kw = dict(via_cost=int(rng.choice([1, 400, 800, 3000])), drc_cost=int(rng.choice([0, 1, 8, 40])), drc_unit=int(rng.choice([1, 400, 1000])))
      form = str(rng.choice(["lds", "lds", "lds", "lds-r2", "scratch", "window", "window-small", "sweeps"]))
      v2 = {}
      if form != "sweeps" and not form.startswith("window") and rng.random() < 0.6:
          v2 = dict(guide_cost=int(rng.choice([0, 300, 800, 2500])), guide_margin=int(rng.choice([0, 1, 3])), maze_end_iter=int(rng.choice([1, 2, 3, 4])))
          if (kw["drc_cost"] * kw["drc_unit"]) << (v2["maze_end_iter"] - 1) >= (1 << 22):
              v2["maze_end_iter"] = 1
          if v2["guide_cost"] and rng.random() < 0.5:
              regions = [_random_guides(r, 7 + trial + i) for i, r in enumerate(regions)]
      if form == "lds-r2":
          kw["router"] = 3
      elif form == "scratch":
          kw.update(force_scratch_field=True, window=-1)
      elif form == "window":
          kw.update(force_scratch_field=True, window=1000)
      elif form == "window-small":
          kw.update(force_scratch_field=True, window=int(rng.integers(8, 20)))
      elif form == "sweeps":
          kw["router"] = 1
      policy = str(rng.choice(["random", "min", "max"]))
      try:
          n = _run_episode_parity(regions, policy=policy, **kw, **v2)
      except AssertionError as ex:
          print(f"MISMATCH trial {trial}: dims {(X, Y, Z)} gen {gen} kw {kw} v2 {v2} form {form} policy {policy}: {str(ex)[:300]}")
          raise
      steps += n
      tally[form + ("+v2" if v2 else "")] = tally.get(form + ("+v2" if v2 else ""), 0) + n
  msg = f"fuzz: {T} trials, {steps} env-steps bit-exact against the oracle in {time.time() - t0:.0f}s; per form: {tally}"
  print(msg)
  return steps, tally


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
