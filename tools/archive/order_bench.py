"""Whole-order re-route throughput: B ispd18_test1-sized regions, every env routes its complete net
list in one xr_batch_route_order launch.  Prints launches/s, routed nets/s (= env-steps/s equivalent) and episodes/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xroute_env_amd.envs.order_contracts import OrderSimulator, OrderVectorEnv
from xroute_env_amd.regions import config_regions

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
R = int(sys.argv[2]) if len(sys.argv) > 2 else 256
regs = config_regions(3, R)
sim = OrderSimulator(regs, n_envs=B)
orders = sim.default_orders()
nets = int((orders > 0).sum())
for with_stats in (False, True):
    for _ in range(2):
        sim.route(orders, with_stats)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        sim.route(orders, with_stats)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"route_order B={B} stats={with_stats}: {dt*1e3:.2f} ms/launch, {nets/dt/1e6:.2f} M nets routed/s, {B/dt:.0f} whole-region orders/s")
venv = OrderVectorEnv(regs, n_envs=B)
feats, legal = venv.reset()
torch.cuda.synchronize(); t0 = time.perf_counter()
steps = 6
for s in range(steps):
    a = torch.where(legal.any(1), legal.float().argmax(1), torch.full((B,), -1, device=legal.device))
    feats, reward, done, legal = venv.step(a)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
print(f"OrderVectorEnv.step B={B}: {dt*1e3:.2f} ms/step (one whole-region re-route per env per step), mean reward {reward.mean().item():.4f}")
