"""BASELINE config 1 (single region, batch = 1) through the reference-shaped API: Game.reset / Game.step with the
observation returned as a CPU tensor (PCIe-inclusive), and with the observation left on the device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xroute_env_amd import Game
from xroute_env_amd.regions import config_regions
regions = config_regions(1, 8)
for ret_dev in (False, True):
    game = Game(regions=regions, return_device=ret_dev)
    game.reset()
    n = 0; t_step = 0.0; t_reset = 0.0
    for ep in range(10):
        t0 = time.perf_counter(); obs, _ = game.reset(); torch.cuda.synchronize(); t_reset += time.perf_counter() - t0
        done = False
        while not done:
            a = min(game.legal_action_set)
            t0 = time.perf_counter()
            obs, done, dv, dw, dvia = game.step(a)
            torch.cuda.synchronize()
            t_step += time.perf_counter() - t0; n += 1
    print(f"Game(return_device={ret_dev}): step {t_step / n * 1e3:.3f} ms ({n / t_step:.0f} env-steps/s), reset {t_reset / 10 * 1e3:.3f} ms, obs {tuple(obs.shape)} on {obs.device}")
