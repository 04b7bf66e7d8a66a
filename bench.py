#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched env.step() hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over the rank's batch of ispd18_test1-sized regions
(BASELINE config 3, 4096 env slots per GPU, random net-order policy chosen on the device):

    xr_batch_random_actions -> xr_batch_step (grid build + XR-Maze v1 route + metrics/reward,
    finished envs are re-initialised) -> xr_batch_observation (reference-layout fp32 [2+7K,Z,Y,X] of
    every env) -> (N > 1) RCCL all_gather of the compact per-env result record.

All inputs are resident in HBM before the timed region.  `value` counts REAL env-steps (a slot
that spends the step re-initialising a finished episode is not counted) over all ranks / max-rank time.
The JSON line also carries `roofline` (dominant kernel, HIP-event timed live) and, on rank 0 at N=1,
`cpu_baseline` (the C oracle with OpenMP on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--envs", type=int, default=4096, help="env slots per GPU")
    ap.add_argument("--config", type=int, default=3, help="BASELINE config id (region generator)")
    ap.add_argument("--block-threads", type=int, default=0)
    ap.add_argument("--obs-mode", type=int, default=0, help="xr_config.obs_mode: 0 default (queue form where it applies), 1 fused launch, 2 split, 3 queue")
    ap.add_argument("--writer-blocks", type=int, default=0)
    ap.add_argument("--region-pack", default=None,
                    help="npz of design-derived regions (tools/extract_regions.py), cycled over the env slots, instead of the "
                         "synthetic generator; NOT the headline workload")
    ap.add_argument("--no-observation", action="store_true", help="skip xr_batch_observation (NOT the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fuse", action="store_true",
                    help="two launches (xr_batch_step, xr_batch_observation) instead of the fused xr_batch_step_observe")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--pmc-calibrate", action="store_true",
                    help="run 1 GiB fill/add kernels first (known HBM byte counts for rocprofv3 --pmc passes)")
    return ap.parse_args()


def cpu_baseline(regions, seconds, with_obs=True):
    """Oracle (`port`) on the host cores: same workload, bounded sample."""
    import numpy as np
    from oracle import xr_oracle as orc
    n = min(len(regions), 256)
    ob = orc.OracleBatch(regions[:n])
    threads = ob.max_threads()
    stride = max((2 + 7 * r.n_nets) * r.n_nodes for r in regions[:n])
    obs = np.empty((n, stride), np.float32) if with_obs else None
    t0 = time.perf_counter()
    real = 0
    it = 0
    while True:
        acts = ob.random_actions(99)
        real += ob.step(acts, threads=threads, auto_reset=True)["real_steps"]
        if with_obs:
            ob.observation(obs, stride, threads=threads)
        it += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or it >= 400:
            break
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": real / dt, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": f"{n} envs x {it} batched steps (route + fp32 observation) in {dt:.1f}s, "
                      f"oracle/xr_oracle.c OpenMP over envs, host cpu '{model}' ({os.cpu_count()} logical)"}


def parity_check(regions, seeds, gpu_hash, gpu_cum, n_check=256):
    """Checker leg (oracle as the CHECKER, never the thing measured): replays the bench's own action sequence — the
    device policy is a counter-based hash of (seed, env, step count), bit-identical in the oracle — on the first
    `n_check` envs and compares every env's hash chain (all path nodes, metrics and actions of every step) and
    cumulative metrics with what the GPU produced during the timed run."""
    import numpy as np
    from oracle import xr_oracle as orc
    n = min(len(regions), n_check)
    ob = orc.OracleBatch(regions[:n])
    threads = ob.max_threads()
    steps = 0
    for sd in seeds:
        steps += ob.step(ob.random_actions(sd), threads=threads, auto_reset=True)["real_steps"]
    ref_hash = np.array([e.hash() for e in ob.envs], dtype=np.uint64)
    ref_cum = np.stack([e.cum() for e in ob.envs])
    return {"envs": n, "env_steps": int(steps), "hash_chains_equal": bool(np.array_equal(ref_hash, gpu_hash[:n])),
            "cumulative_metrics_equal": bool(np.array_equal(ref_cum, gpu_cum[:n])),
            "what": "CPU oracle replay of the same actions on the first envs of rank 0, all warm-up + timed steps"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # XR_BENCH_BACKEND=gloo + XR_BENCH_SAME_DEVICE=1: run the N > 1 control flow (barriers, max-over-ranks timing, the
    # batched-env gather) with every rank on cuda:0 of a single-GPU box — a functional check of this path, not a
    # measurement; the driver's multi-GPU runs use the default (RCCL, one rank per GPU)
    backend = os.environ.get("XR_BENCH_BACKEND", "nccl")
    if os.environ.get("XR_BENCH_SAME_DEVICE") == "1":
        local_rank = 0
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    if args.pmc_calibrate:          # known traffic for FETCH_SIZE / WRITE_SIZE calibration (tools/pmc_parse.py)
        ca = torch.empty(1 << 28, dtype=torch.float32, device=dev)
        cb = torch.empty(1 << 28, dtype=torch.float32, device=dev)
        for _ in range(2):
            ca.fill_(1.0)
            torch.add(ca, 1.0, out=cb)
        torch.cuda.synchronize(dev)
        del ca, cb

    from xroute_env_amd.batch import RegionBatch
    from xroute_env_amd.dist import RECORD_WIDTH, gather_records_fixed, pack_records
    from xroute_env_amd.regions import config_regions

    B = args.envs
    if args.region_pack:
        from xroute_env_amd.lefdef import load_region_pack
        regions = load_region_pack(args.region_pack)
    else:
        regions = config_regions(args.config, B, first_env=rank * B)
    batch = RegionBatch(regions, n_envs=B, device=dev, auto_reset=True, block_threads=args.block_threads,
                        obs_mode=args.obs_mode, obs_writer_blocks=args.writer_blocks)
    batch.reset(rotate=True)
    acts = torch.empty(B, dtype=torch.int32, device=dev)
    obs = None if args.no_observation else batch.alloc_observation()
    # compact per-env result record gathered across ranks (xroute_env_amd/dist.py): 6 x f64 per env
    rec_local = torch.empty((B, RECORD_WIDTH), dtype=torch.float64, device=dev)
    rec_all = torch.empty((world * B, RECORD_WIDTH), dtype=torch.float64, device=dev) if world > 1 else None
    reward = torch.empty(B, dtype=torch.float64, device=dev)
    delta = torch.empty((B, 3), dtype=torch.int32, device=dev)
    done = torch.empty(B, dtype=torch.uint8, device=dev)
    nsteps_total = args.warmup + args.steps
    nlegal_log = torch.zeros((max(nsteps_total, 1), B), dtype=torch.int32, device=dev)
    n_nodes = torch.tensor([regions[e % len(regions)].n_nodes for e in range(B)], dtype=torch.float64, device=dev)

    fused = (obs is not None) and not args.no_fuse

    def one_step(i, ev=None):
        batch.random_actions(args.seed + rank * 7919 + i, acts)
        if ev:
            ev[0].record()
        if fused:
            batch.step(acts, obs)                 # one launch: route + observation of every env
            if ev:
                ev[1].record()
        else:
            batch.step(acts)
            if ev:
                ev[1].record()
            if obs is not None:
                batch.observation(obs)
        if ev:
            ev[2].record()
        batch.fetch("nlegal", nlegal_log[i])
        batch.fetch("reward", reward)
        batch.fetch("delta", delta)
        batch.fetch("done", done)
        if world > 1:
            pack_records(reward, delta, done, nlegal_log[i], rec_local)
            gather_records_fixed(rec_local, rec_all)             # RCCL over xGMI: the batched-env gather

    for i in range(args.warmup):
        one_step(i)

    events = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    steps0 = batch.total_steps()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(args.warmup + i, events[i])
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    real_steps = batch.total_steps() - steps0

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    s = torch.tensor([float(real_steps)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
    elapsed_max, total_real = float(t.item()), float(s.item())

    # ---- per-kernel live timing (HIP events on the launch stream) and algorithmic bytes -----------
    route_ms = sum(e[0].elapsed_time(e[1]) for e in events) / max(args.steps, 1)
    obs_ms = sum(e[1].elapsed_time(e[2]) for e in events) / max(args.steps, 1)
    k_after = nlegal_log[args.warmup:args.warmup + args.steps].to(torch.float64)          # K written by each obs launch
    obs_bytes = float(((4.0 * (2.0 + 7.0 * k_after) + 4.0) * n_nodes[None, :]).sum().item()) / max(args.steps, 1)
    # route launch: every stepped env loads its compact state (node_net i16 + owner i16 = 4 B/node); path writes are noise
    route_bytes = float((4.0 * n_nodes).sum().item())
    kernels = []
    if fused:
        form = batch.observe_timing()[0]            # 1 fused launch, 2 split, 3 queue (the default where it applies)
        if form == 3:
            kname = "xr_step_queue_kernel"
            note = ("step kernel, queue form (xr_batch_step_observe): one persistent launch whose workgroups drain two task "
                    "queues — envs to route (LDS-resident field, latency-bound) + their planes 0..1, and net-plane units of "
                    "the fp32 observation (HBM-write-bound); the HIP-event time also covers the planning kernel "
                    "(xr_plan_kernel, ~10 us) that precedes it on the same stream; bytes = state load + observation")
        else:
            kname = "xr_route_kernel"
            note = ("fused step kernel (xr_batch_step_observe): per env one workgroup routes the net (LDS-resident, "
                    "latency-bound) and then streams the fp32 observation (HBM-write-bound); bytes = state load + observation"
                    + ("; split form: the net planes come from xr_netplane_kernel on an internal stream" if form == 2 else ""))
        kernels.append({"kernel": kname, "bound": "hbm", "ms": route_ms, "bytes": obs_bytes + route_bytes,
                        "achieved": (obs_bytes + route_bytes) / (route_ms * 1e-3) / 1e9 if route_ms > 0 else 0.0,
                        "note": note})
    else:
        if obs is not None:
            kernels.append({"kernel": "xr_obs_kernel", "bound": "hbm", "ms": obs_ms, "bytes": obs_bytes,
                            "achieved": obs_bytes / (obs_ms * 1e-3) / 1e9 if obs_ms > 0 else 0.0})
        kernels.append({"kernel": "xr_route_kernel", "bound": "hbm", "ms": route_ms, "bytes": route_bytes,
                        "achieved": route_bytes / (route_ms * 1e-3) / 1e9 if route_ms > 0 else 0.0,
                        "note": "distance field LDS-resident: LDS/latency-bound by construction, HBM bytes are the state load only"})
    dom = max(kernels, key=lambda k: k["ms"])
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")       # measured offline with rocprofv3 --pmc (see profiles/README.md)
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get(dom["kernel"])
            if isinstance(traffic, dict):
                traffic = traffic.get("hbm_total_bytes")
        except Exception:
            traffic = None
    roofline = {"kernel": dom["kernel"], "bound": "hbm", "achieved": round(dom["achieved"], 2), "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": round(dom["achieved"] / HBM_PEAK_GBPS, 4), "traffic": traffic,
                "avg_launch_ms": round(dom["ms"], 4), "algorithmic_bytes_per_launch": int(dom["bytes"])}

    out = None
    if rank == 0:
        mean_k = float(k_after.mean().item())
        out = {
            "metric": "env-steps/sec (batched regions), ispd18_test1-sized regions",
            "value": round(total_real / elapsed_max, 1),
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed_max / max(args.steps, 1) * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 distances / i16 state / fp32 observation",
            "data": "synthetic",
            "config": {"workload": (f"{B} env slots over the design-derived ispd18_test1 region pack {os.path.basename(args.region_pack)}, "
                                    if args.region_pack else
                                    f"BASELINE config {args.config}: {B} ispd18_test1-sized regions (24x40x9, K~U[4,36]) per GPU, ")
                                   +
                                   "full step = random net-order action + XR-Maze v1 route + metrics/reward"
                                   + ("" if obs is None else " + reference-layout fp32 observation of every env")
                                   + (" (one persistent launch after a planning kernel)" if fused and batch.observe_timing()[0] == 3 else " (fused launch)" if fused else "")
                                   + (", RCCL all_gather of per-env results" if world > 1 else ""),
                       "envs_per_gpu": B, "global_envs": B * world, "parallelism": f"env-shard x{world}",
                       "mean_nets_left": round(mean_k, 2), "slots_stepped_per_batch_step": round(total_real / (args.steps * B * world), 4)},
            "roofline": roofline,
            "kernels": [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in kk.items()} for kk in kernels],
        }
        if world == 1 and not args.no_cpu_baseline and not args.region_pack and len(regions) >= B:
            try:        # regions == env slots: rotation keeps every slot on its region, the oracle subset can follow
                seeds = [args.seed + rank * 7919 + i for i in range(args.warmup + args.steps)]
                out["parity"] = parity_check(regions, seeds, batch.fetch("hash").cpu().numpy().view("uint64"),
                                             batch.fetch("cum").cpu().numpy())
            except Exception as ex:
                out["parity"] = {"error": str(ex)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(regions, args.cpu_seconds, with_obs=obs is not None)
            except Exception as ex:          # the oracle is optional for the GPU number itself
                out["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {ex}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
