#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched env.step() hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over the rank's batch of ispd18_test1-sized regions (north_star target: a 4096-env
batch; 4096 env slots per GPU by default = weak scaling, `--global-envs 4096` = the same batch split over the ranks =
strong scaling, BASELINE config 4 at N = 8):

    xr_batch_random_actions (random net-order policy, chosen on the device)
    -> xr_batch_step_observe = xr_plan_kernel + ONE persistent launch (xr_step_queue_kernel) that routes the chosen net of
       every env (grid build + XR-Maze v1 + claim + metrics/reward; finished envs are re-initialised) and writes the
       reference-layout fp32 [2+7K,Z,Y,X] observation of every env
    -> (N > 1) RCCL all_gather of the 48-byte per-env result records.

All inputs are resident in HBM before the timed region.  Before anything is timed the episodes are STAGGERED: env e
is advanced by hash(e) mod (K0_e + 1) untimed route-only steps, so that every env sits at a uniformly random phase of
its (periodic) episode cycle — the nets-left distribution is then stationary and `value` does not depend on --warmup.
`value` counts REAL env-steps (a slot that spends the step re-initialising a finished episode is not counted) over all
ranks / max-rank time.

The JSON line also carries `roofline` (the step kernel, HIP-event timed live on its launch stream), `kernels` (per kernel:
the step kernel, the route-only kernel xr_batch_step, and — rank 0, N = 1 — the BASELINE config 5 route kernel on
256x256x12 regions, each with its own bytes / ms / fraction), `parity` (oracle replay of the run's own actions) and
`cpu_baseline` (the C oracle on the host cores: one thread and all cores, bounded samples).
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# this process mixes two OpenMP users — the oracle (checker / cpu_baseline legs, all host cores) and the framework's CPU operators (the agents' weight
# snapshots are packed by a few small CPU convolutions): idle OpenMP threads must sleep, not spin, or each side's parallel regions start among the other's
# spinning threads (measured: a 8 ms pack took ~100 ms right after an oracle replay).  Before any OpenMP runtime is loaded.
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

PACK_PATH = os.path.join(ROOT, "tests", "golden", "ispd18_test1_regions.npz")    # 256 regions extracted from the reference's ispd18_test1 LEF/DEF/guide
FP32_MATRIX_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / v_pk_fma_f32, 256 FLOP per clock and CU
# How the towers multiply (csrc/xr_agent.hip, `xr_agent_matrix_mode`; XR_TOWER_FP32=1 selects 0).  Both are priced against the fp32 peak above: `achieved` counts the
# ALGORITHMIC fp32 multiply-adds; mode 1 spends three bf16 matrix instructions (hi.hi + hi.lo + lo.hi of operands split into two bf16 halves, fp32 accumulate) on
# each of them and agrees with the framework's fp32 convolutions to ~1e-5 (tests/test_agents.py hold 2e-4), mode 0 is a k-ordered fp32 fma chain.
MATRIX_MODES = {0: "7 -> 7 convolutions on v_mfma_f32_16x16x4_f32",
                1: "7 -> 7 convolutions and the aligning convolution on v_mfma_f32_16x16x32_bf16, split-bf16 operands x3, fp32 accumulate"}


def matrix_mode():
    from xroute_env_amd import _lib
    return int(_lib.lib().xr_agent_matrix_mode())
HBM_PEAK_GBPS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
STAGGER_SEED = 0x5EED5EED


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--envs", type=int, default=4096, help="env slots per GPU (weak scaling)")
    ap.add_argument("--global-envs", type=int, default=0,
                    help="total env slots over all ranks (strong scaling: each rank takes its contiguous share); overrides --envs")
    ap.add_argument("--config", type=int, default=3, help="BASELINE config id (region generator)")
    ap.add_argument("--regions", type=int, default=0,
                    help="distinct synthetic regions generated (the same on every rank), cycled over the env slots: global env g plays region g %% R "
                         "(0 = one region per env slot; BASELINE config 5's multi-GPU form: --config 5 --envs 1024 --regions 128 --no-observation)")
    ap.add_argument("--block-threads", type=int, default=0)
    ap.add_argument("--router", type=int, default=0, help="xr_config.router: 0 default, 1 line-segment sweeps, 2 bucketed frontier")
    ap.add_argument("--dial-mult", type=int, default=0)
    ap.add_argument("--obs-mode", type=int, default=0, help="xr_config.obs_mode: 0 default (queue form where it applies), 1 fused launch, 2 split, 3 queue")
    ap.add_argument("--writer-blocks", type=int, default=0)
    ap.add_argument("--helper-blocks", type=int, default=0, help="xr_config.obs_helper_blocks (0 none = default)")
    ap.add_argument("--launch-order", type=int, default=0, help="xr_config.launch_order (route-only launches: 0 auto, 1 slot order, 2 longest predicted route first)")
    ap.add_argument("--quota", type=int, default=0, help="xr_config.obs_split_permille (queue form: units per route, per mille of the average)")
    ap.add_argument("--region-pack", default=None,
                    help="npz of design-derived regions (tools/extract_regions.py), cycled over the env slots, instead of the "
                         "synthetic generator; NOT the headline workload")
    ap.add_argument("--maze-v2", action="store_true",
                    help="the main batch routes with XR-Maze v2 (the reference's TCL knobs: maze_end_iter 3, guide cost 800, margin 1 with a "
                         "region pack's own guide rectangles / 2 with the default guides) — NOT the headline; used with --region-pack")
    ap.add_argument("--no-observation", action="store_true", help="skip the observation (NOT the headline)")
    ap.add_argument("--no-stagger", action="store_true", help="start every episode at step 0 (round-1 behaviour: NOT stationary)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the extra per-kernel legs (route-only, config 5)")
    ap.add_argument("--no-fuse", action="store_true",
                    help="two launches (xr_batch_step, xr_batch_observation) instead of xr_batch_step_observe")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the `extras` of the default N = 1 line (batch-1 Game.step latency = BASELINE config 1; DQN counterpart attached on 1024 envs = config 3)")
    ap.add_argument("--c5-envs", type=int, default=1024, help="env slots of the BASELINE config 5 leg (256x256x12 regions)")
    ap.add_argument("--c5-regions", type=int, default=128, help="distinct config 5 regions generated (cycled over the env slots)")
    ap.add_argument("--pack-envs", type=int, default=4096,
                    help="env slots of the design-derived leg: the 256 regions extracted from the reference's ispd18_test1.input.{lef,def,guide} "
                         "(tests/golden/ispd18_test1_regions.npz) cycled over this many slots, full step in the queue form (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--learner", action="store_true",
                    help="N > 1: BASELINE config 4's learner flow — records + legal bitmasks gathered, the (random net-order) policy runs on "
                         "rank 0 for ALL envs, actions travel back as one i32 broadcast (default: every rank runs the policy for its own envs)")
    ap.add_argument("--agent", choices=["dqn", "ppo"], default=None,
                    help="agent-attached line (BASELINE configs 3 / 4) INSTEAD of the env-only headline: actions from the batched DQN / PPO "
                         "counterpart (random-init weights of the reference architecture), env in compact-consumer mode")
    ap.add_argument("--agent-lib-tower", action="store_true", help="--agent: the obstacle tower through the framework's convolutions instead of the fused HIP kernel (A/B)")
    ap.add_argument("--agent-full-obs", action="store_true", help="--agent: feed the full fp32 observation (xr_batch_step_observe) instead of the compact mode")
    ap.add_argument("--force-collectives", action="store_true",
                    help="N = 1 only: take every N > 1 branch anyway — process group on the real backend (RCCL: init with a device id), the async "
                         "all_gather buffer pairs + Work.wait(), verify_gather, broadcast, all_reduce, the learner's gather — with one rank.  A functional "
                         "run of the multi-GPU code path on one GPU (also: XR_FORCE_COLLECTIVES=1); the line says so in config.forced_collectives")
    ap.add_argument("--pmc-calibrate", action="store_true",
                    help="run 1 GiB fill/add kernels first (known HBM byte counts for rocprofv3 --pmc passes)")
    return ap.parse_args()


def source_sha():
    """Identity of the kernels being measured (profiles/pmc_traffic.json is only quoted for the same build)."""
    h = hashlib.sha256()
    for f in ("xr_kernels.hip", "xr_dial.h", "xr_dial3.h", "xr_device.h", "xr_batch.cpp"):
        with open(os.path.join(ROOT, "xroute_env_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _gen_chunk(args):
    config, lo, hi = args
    from xroute_env_amd.regions import config_regions
    return config_regions(config, hi - lo, first_env=lo)


def gen_regions(config, n, first_env=0):
    """Synthetic regions of a BASELINE config, generated on the host cores (fork pool: call BEFORE touching the GPU).
    Under a profiler (rocprofv3 preloads its tool into every child and a pool worker that is torn down can hang in the
    tool's signal handler) the regions are generated in this process."""
    workers = min(os.cpu_count() or 1, 32, max(1, n // 16))
    profiled = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)
    if workers <= 1 or profiled or os.environ.get("XR_BENCH_NO_FORK") == "1":
        return _gen_chunk((config, first_env, first_env + n))
    import multiprocessing as mp
    step = (n + workers - 1) // workers
    chunks = [(config, first_env + lo, first_env + min(lo + step, n)) for lo in range(0, n, step)]
    pool = mp.get_context("fork").Pool(workers)
    try:
        parts = pool.map(_gen_chunk, chunks)
        pool.close()                      # workers exit on their own (no SIGTERM)
    except BaseException:
        pool.terminate()
        raise
    finally:
        pool.join()
    return [r for p in parts for r in p]


def splitmix64(x):
    import numpy as np
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def stagger_offsets(nlegal0, first_env):
    """Untimed steps env e is advanced by before the run: hash(global env id) mod (K0 + 1) = a uniform phase of its cycle."""
    import numpy as np
    with np.errstate(over="ignore"):
        ids = np.arange(first_env, first_env + len(nlegal0), dtype=np.uint64)
        r = splitmix64(ids ^ np.uint64(STAGGER_SEED))
    return (r % (nlegal0.astype(np.uint64) + np.uint64(1))).astype(np.int64)


def cpu_baseline(regions, seconds, with_obs=True):
    """Oracle (`port`) on the host cores, same workload, bounded samples: all cores (>= 8 envs per thread, OpenMP dynamic
    schedule) and a single thread."""
    import numpy as np
    from oracle import xr_oracle as orc
    probe = orc.OracleBatch(regions[:1])
    threads = probe.max_threads()
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass

    def run(n, thr, secs, max_it):
        ob = orc.OracleBatch(regions[:n])
        stride = max((2 + 7 * r.n_nets) * r.n_nodes for r in regions[:n])
        obs = np.empty((n, stride), np.float32) if with_obs else None
        # same stagger as the GPU run: the K distribution is the stationary one
        nl0 = np.array([e.nlegal() for e in ob.envs])
        off = stagger_offsets(nl0, 0)
        for i in range(int(off.max()) if len(off) else 0):
            a = ob.random_actions(7 + i)
            a[off <= i] = 0
            ob.step(a, threads=thr, auto_reset=True)
        t0 = time.perf_counter()
        real = it = 0
        while True:
            acts = ob.random_actions(99 + it)
            real += ob.step(acts, threads=thr, auto_reset=True)["real_steps"]
            if with_obs:
                ob.observation(obs, stride, threads=thr)
            it += 1
            dt = time.perf_counter() - t0
            if dt >= secs or it >= max_it:
                break
        return real / dt, n, it, dt

    n_all = min(len(regions), max(256, 8 * threads))
    v_all, n_a, it_a, dt_a = run(n_all, threads, seconds * 0.6, 400)
    v_one, n_1, it_1, dt_1 = run(min(len(regions), 32), 1, seconds * 0.4, 400)
    what = "route + fp32 observation" if with_obs else "route only"
    return {"value": v_all, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "single_thread": {"value": v_one, "unit": "env-steps/s", "cores": 1,
                              "sample": f"{n_1} envs x {it_1} batched steps ({what}) in {dt_1:.1f}s"},
            "sample": f"north_star target workload (4096-env batch of ispd18_test1-sized regions), bounded sample: {n_a} envs "
                      f"({n_a / threads:.1f} per thread, OpenMP dynamic schedule over envs) x {it_a} batched steps ({what}, stationary "
                      f"nets-left distribution) in {dt_a:.1f}s, oracle/xr_oracle.c, host cpu '{model}' ({os.cpu_count()} logical)"}


def obs_sample_sha(obs, nlegal, slot_regions, n_check=256, n_sample=32, head_only=False):
    """sha256 of the fp32 observation rows (reference layout, (2+7K)*N floats) of `n_sample` env slots among the first `n_check`,
    spread over the nets-left distribution (sorted by K, evenly spaced).  Called right after the timed region, BEFORE any leg
    touches the buffer again: these are bytes the timed launches wrote.  {env: (K, sha)}"""
    import numpy as np
    n = min(n_check, int(obs.shape[0]), len(slot_regions))
    nl = nlegal[:n].cpu().numpy()
    order = np.argsort(nl, kind="stable")
    pick = sorted({int(order[int(round(j))]) for j in np.linspace(0, n - 1, min(n_sample, n))})
    out = {}
    for e in pick:
        size = (2 if head_only else 2 + 7 * int(nl[e])) * slot_regions[e].n_nodes          # head_only: the compact-consumer step's planes 0..1
        out[e] = (int(nl[e]), hashlib.sha256(obs[e, :size].cpu().numpy().tobytes()).hexdigest())
    return out


def parity_check(regions, seeds, stagger, gpu_hash, gpu_cum, n_check=256, obs_sha=None, v2=None, actions_log=None, head_only=False):
    """Checker leg (oracle as the CHECKER, never the thing measured): replays the bench's own action sequence — the
    device policy is a counter-based hash of (seed, env, step count), bit-identical in the oracle; `actions_log` (learner flow:
    the actions rank 0 broadcast) replaces it — on the first `n_check` envs (stagger pre-roll, warm-up and timed steps) and
    compares every env's hash chain (all path nodes, metrics and actions of every step) and cumulative metrics with what the GPU
    produced; `obs_sha` ({env: (K, sha256)} of observation rows fetched right after the timed region, `obs_sample_sha`): the
    oracle's build_3Dgrid restatement of the same envs must give the same bytes.  `v2`: XR-Maze v2 knobs of the batch."""
    import numpy as np
    from oracle import xr_oracle as orc
    n = min(len(regions), n_check)
    ob = orc.OracleBatch(regions[:n], **(v2 or {}))
    threads = ob.max_threads()
    steps = 0
    if stagger is not None:
        off, pre_seeds = stagger
        for i, sd in enumerate(pre_seeds):
            a = ob.random_actions(sd)
            a[off[:n] <= i] = 0
            steps += ob.step(a, threads=threads, auto_reset=True)["real_steps"]
    for i, sd in enumerate(seeds):
        a = ob.random_actions(sd) if actions_log is None else np.ascontiguousarray(actions_log[i][:n], np.int32)
        steps += ob.step(a, threads=threads, auto_reset=True)["real_steps"]
    ref_hash = np.array([e.hash() for e in ob.envs], dtype=np.uint64)
    ref_cum = np.stack([e.cum() for e in ob.envs])
    res = {"envs": n, "env_steps": int(steps), "hash_chains_equal": bool(np.array_equal(ref_hash, gpu_hash[:n])),
           "cumulative_metrics_equal": bool(np.array_equal(ref_cum, gpu_cum[:n])),
           "what": "CPU oracle replay of the same actions on the first envs of the rank: stagger pre-roll + warm-up + timed steps"}
    if obs_sha is not None:
        bad = []
        for e, (k, sha) in obs_sha.items():
            if e >= n:
                continue
            o = ob.envs[e].observation()
            if head_only:
                o = np.ascontiguousarray(o).ravel()[:2 * regions[e].n_nodes]
            if ob.envs[e].nlegal() != k or hashlib.sha256(np.ascontiguousarray(o).tobytes()).hexdigest() != sha:
                bad.append(int(e))
        checked = [e for e in obs_sha if e < n]
        res["observations_checked"] = len(checked)
        res["observations_equal"] = len(checked) > 0 and not bad
        res["observation_nets_left_range"] = [min(k for k, _ in obs_sha.values()), max(k for k, _ in obs_sha.values())] if obs_sha else None
        if bad:
            res["observation_mismatch_envs"] = bad[:8]
        res["what"] += "; observations: sha256 of the fp32 rows the last timed launch wrote for envs spread over K vs the oracle's build_3Dgrid restatement"
    res["ok"] = bool(res["hash_chains_equal"] and res["cumulative_metrics_equal"] and res.get("observations_equal", True))
    return res


def kernel_entry(name, ms, nbytes, env_steps, bound, note):
    ach = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    return {"kernel": name, "bound": bound, "ms": ms, "bytes": nbytes, "achieved": ach, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBPS, "env_steps_per_s": env_steps / (ms * 1e-3) if ms > 0 else 0.0, "note": note}


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) WITHOUT a launcher: start the N ranks ourselves — a child `python -m
    torch.distributed.run --nproc-per-node N bench.py <same flags>` — and forward rank 0's JSON line.  Decided before this
    process has touched the GPU (nothing here imports torch); the children are fresh processes (no exec / re-exec).  Exit
    code = the launcher's: non-zero when any rank failed."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"bench.py: --gpus {args.gpus} without WORLD_SIZE: launching {args.gpus} ranks ({' '.join(cmd[1:8])} ...)", file=sys.stderr)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in proc.stdout:                  # rank 0 prints ONE JSON line; anything else the ranks write to stdout goes to stderr
        if ln.startswith("{") and '"metric"' in ln:
            line = ln.rstrip("\n")
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        print("bench.py: the ranks exited 0 but printed no result line", file=sys.stderr)
        rc = 1
    return rc


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.force_collectives or os.environ.get("XR_FORCE_COLLECTIVES") == "1":
        if world != 1:
            sys.exit("bench.py: --force-collectives is for one rank (with more ranks the collectives run anyway)")
        os.environ["XR_FORCE_COLLECTIVES"] = "1"            # xroute_env_amd.dist.collectives_on() reads it
        args.force_collectives = True
        for k_, v_ in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", str(29500 + os.getpid() % 20000)), ("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
            os.environ.setdefault(k_, v_)
    multi = world > 1 or args.force_collectives             # the collective code paths run
    if world != args.gpus:
        # never benchmark a different number of GPUs than the one asked for (a flat, wrong scaling curve)
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE {world}: refusing to run", file=sys.stderr)
        sys.exit(2)

    # ---- env slots of this rank; regions are generated on the host cores BEFORE the GPU is touched (fork pool) -------
    strong = args.global_envs > 0
    if strong:
        from xroute_env_amd.dist import shard_range
        lo, hi = shard_range(args.global_envs, world, rank)
        B, first_env = hi - lo, lo
    else:
        B, first_env = args.envs, rank * args.envs
    if args.region_pack:
        from xroute_env_amd.lefdef import load_region_pack
        regions = load_region_pack(args.region_pack)
    elif args.regions > 0:
        # `--regions R`: R distinct regions of the config, the SAME on every rank, cycled over the env slots (global env g plays region
        # g % R) — BASELINE config 5's multi-GPU form: 1024 slots of 256x256x12 per GPU over 128 distinct regions, like its one-GPU leg
        regions = gen_regions(args.config, min(args.regions, max(B, 1)), 0) if not (multi or strong) else gen_regions(args.config, args.regions, 0)
    else:
        regions = gen_regions(args.config, B, first_env)
    learner_regions = None
    if args.agent and args.learner and multi and rank == 0 and not args.region_pack and args.regions == 0:
        learner_regions = gen_regions(args.config, args.global_envs if strong else args.envs * world, 0)      # the central learner's region table: every env's region
    do_legs = (not multi) and rank == 0 and not args.no_legs and not args.region_pack
    c5_regions = gen_regions(5, min(args.c5_regions, args.c5_envs)) if do_legs and args.c5_envs > 0 else None
    pack_regions = None
    if do_legs and args.pack_envs > 0 and os.path.exists(PACK_PATH):
        from xroute_env_amd.lefdef import load_region_pack
        pack_regions = load_region_pack(PACK_PATH)

    import numpy as np
    import torch
    import torch.distributed as dist

    # XR_BENCH_BACKEND=gloo + XR_BENCH_SAME_DEVICE=1: run the N > 1 control flow (barriers, max-over-ranks timing, the
    # batched-env gather) with every rank on cuda:0 of a single-GPU box — a functional check of this path, not a
    # measurement; the driver's multi-GPU runs use the default (RCCL, one rank per GPU)
    backend = os.environ.get("XR_BENCH_BACKEND", "nccl")
    if os.environ.get("XR_BENCH_SAME_DEVICE") == "1":
        local_rank = 0
    if multi:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
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
    from xroute_env_amd.dist import RECORD_BYTES, gather_records_fixed

    if args.agent and (multi or strong or args.learner):
        # BASELINE config 4 as stated ("4096 regions sharded 8 x MI355X, PPO baseline, RCCL env gather"): every rank evaluates the policy
        # counterpart on ITS shard (or, --learner, rank 0 for all envs from gathered compact state) — one self-certifying line
        rc = agent_sharded(args, regions, dev, world, rank, first_env, B, strong, learner_regions)
        if multi:
            dist.barrier()
            dist.destroy_process_group()
        sys.exit(rc)
    if args.agent:
        print(json.dumps(agent_leg(args, regions, dev, 1)), flush=True)
        return

    main_v2 = dict(V2_KNOBS, guide_margin=1 if args.region_pack else 2) if args.maze_v2 else {}
    batch = RegionBatch(regions, n_envs=B, device=dev, auto_reset=True, block_threads=args.block_threads,
                        obs_mode=args.obs_mode, obs_writer_blocks=args.writer_blocks, router=args.router,
                        dial_mult=args.dial_mult, obs_helper_blocks=args.helper_blocks, obs_split_permille=args.quota,
                        launch_order=args.launch_order, **(dict(max_route_count=1 << 30) if args.regions > 0 else {}), **main_v2)
    # slot -> region: one region per slot (the default generator), the pack cycled in slot order, or — `--regions R` — GLOBAL env g on
    # region g % R whatever the sharding (slots then keep their region: no rotation, the oracle subset can follow)
    if args.regions > 0:
        batch.assign([(first_env + e) % len(regions) for e in range(B)])
        slot_regions = [regions[(first_env + e) % len(regions)] for e in range(B)]
    else:
        slot_regions = [regions[e % len(regions)] for e in range(B)]
    batch.reset(rotate=True)
    acts = torch.empty(B, dtype=torch.int32, device=dev)
    obs = None if args.no_observation else batch.alloc_observation()
    # compact per-env result record gathered across ranks: the 48-byte xr_step_record the kernels write themselves
    rec_local = torch.empty((B, RECORD_BYTES), dtype=torch.uint8, device=dev)
    rec_all = torch.empty((world * B, RECORD_BYTES), dtype=torch.uint8, device=dev) if multi and not strong else None
    if multi and strong:
        rec_all = torch.empty((args.global_envs, RECORD_BYTES), dtype=torch.uint8, device=dev) \
            if args.global_envs % world == 0 else None
    nsteps_total = args.warmup + args.steps
    nlegal_log = torch.zeros((max(nsteps_total, 1), B), dtype=torch.int32, device=dev)
    n_par = min(B, 256 if (not multi) else 32)          # envs of this rank the oracle replays after the run (`parity`)
    if regions[0].n_nodes > 100000:                    # (BASELINE config 5: a Dijkstra over 786 k nodes per search)
        n_par = min(n_par, 16)
    acts_log = torch.zeros((max(nsteps_total, 1), n_par), dtype=torch.int32, device=dev)
    last_gather = [None]
    # N > 1, weak scaling: the gather of step i runs on RCCL's stream WHILE step i + 1 computes — two record / result buffer pairs take turns,
    # a pair is reused only after its collective has completed (Work.wait() = a stream dependency, no host stall)
    rec_pairs = [(rec_local, rec_all), (torch.empty_like(rec_local), torch.empty_like(rec_all))] if rec_all is not None else None
    pending = [None, None]
    n_nodes = torch.tensor([r.n_nodes for r in slot_regions], dtype=torch.float64, device=dev)

    # ---- stagger: every env to a uniform phase of its episode cycle (untimed, route-only steps) ----------------------
    stagger = None
    if not args.no_stagger:
        nl0 = batch.fetch("nlegal").cpu().numpy()
        off = stagger_offsets(nl0, first_env)
        off_d = torch.from_numpy(off).to(dev)
        pre_seeds = [args.seed ^ 0x51A6 ^ (rank * 7919 + i) for i in range(int(off.max()) if B else 0)]
        zero = torch.zeros_like(acts)
        for i, sd in enumerate(pre_seeds):
            batch.random_actions(sd, acts)
            torch.where(off_d > i, acts, zero, out=acts)
            batch.step(acts)
        stagger = (off, pre_seeds)

    with_obs = obs is not None                      # (the buffer itself is released before the late legs)
    fused = with_obs and not args.no_fuse
    learner = args.learner and multi and rec_all is not None
    if learner:
        from xroute_env_amd.dist import random_legal_policy, unpack_records
        Bg_all = rec_all.shape[0]
        legal_local = torch.empty((B, batch.legal_words), dtype=torch.int64, device=dev)
        legal_all = torch.empty((Bg_all, batch.legal_words), dtype=torch.int64, device=dev)
        acts_all = torch.zeros(Bg_all, dtype=torch.int32, device=dev)
        batch.fetch("record", rec_local)
        batch.fetch("legal", legal_local)
        dist.all_gather_into_tensor(rec_all, rec_local)
        dist.all_gather_into_tensor(legal_all, legal_local)

    def one_step(i, ev=None):
        if learner:
            # rank 0 chooses for every env from the gathered state; one broadcast carries the actions back
            if rank == 0:
                acts_all.copy_(random_legal_policy(unpack_records(rec_all), legal_all, args.seed + i))
            dist.broadcast(acts_all, src=0)
            acts.copy_(acts_all[first_env:first_env + B] if strong else acts_all[rank * B:(rank + 1) * B])
        else:
            batch.random_actions(args.seed + rank * 7919 + i, acts)
        if learner:
            acts_log[i].copy_(acts[:n_par])
        if ev:
            ev[0].record()
        if fused:
            batch.step(acts, obs)                 # plan + one persistent launch: route + observation of every env
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
        if multi and rec_pairs is not None and not learner:
            slot = i & 1
            if pending[slot] is not None:
                pending[slot].wait()
            loc, glob = rec_pairs[slot]
            batch.fetch("record", loc)
            pending[slot] = dist.all_gather_into_tensor(glob, loc, async_op=True)      # RCCL over xGMI: the batched-env gather, overlapped with the next step
            last_gather[0] = slot
            return
        batch.fetch("record", rec_local)
        if multi:
            if learner:
                batch.fetch("legal", legal_local)
                dist.all_gather_into_tensor(legal_all, legal_local)
            if rec_all is not None:
                last_gather[0] = gather_records_fixed(rec_local, rec_all)         # (the learner needs the records before it can choose: not overlapped)
            else:
                from xroute_env_amd.dist import gather_records
                last_gather[0] = gather_records(rec_local)

    for i in range(args.warmup):
        one_step(i)
    for w_ in pending:                              # (the warm-up's gathers are not the timed region's)
        if w_ is not None:
            w_.wait()
    pending[:] = [None, None]

    events = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    if multi:
        dist.barrier()
    torch.cuda.synchronize(dev)
    steps0 = batch.total_steps()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(args.warmup + i, events[i])
    for w_ in pending:                              # every gather of the timed steps completes inside the timed region
        if w_ is not None:
            w_.wait()
    pending[:] = [None, None]
    torch.cuda.synchronize(dev)
    if multi:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    real_steps = batch.total_steps() - steps0
    gpu_hash = batch.fetch("hash").cpu().numpy().view("uint64")          # state right after the timed region
    gpu_cum = batch.fetch("cum").cpu().numpy()
    # bytes of the observation the LAST timed launch wrote, for envs spread over K (compared with the oracle in `parity`)
    obs_sha = None
    if obs is not None and nsteps_total > 0:
        obs_sha = obs_sample_sha(obs, nlegal_log[nsteps_total - 1], slot_regions[:n_par], n_check=n_par)

    # ---- N > 1: the run certifies itself — the gather delivered every rank's records, and every rank's envs replay on the oracle
    certify = None
    if multi:
        from xroute_env_amd.dist import verify_gather
        gathered, sent = last_gather[0], rec_local
        if isinstance(gathered, int):               # the overlapped gather: the buffer pair of the last step
            sent, gathered = rec_pairs[gathered]
        if gathered is None:
            certify = {"ranks_seen": 0, "gather_verified": False, "rows": 0}
        else:
            if os.environ.get("XR_BENCH_TEST_CORRUPT_GATHER") == "1" and rank == 0:       # test hook: a slice that is NOT what its owner sent
                gathered[-1, 0] ^= 0xFF
            certify = verify_gather(sent, gathered, first_env)

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    s = torch.tensor([float(real_steps)], dtype=torch.float64, device=dev)
    if multi:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
    elapsed_max, total_real = float(t.item()), float(s.item())

    # ---- per-kernel live timing (HIP events on the launch stream) and algorithmic bytes -----------
    nst = max(args.steps, 1)
    route_ms = sum(e[0].elapsed_time(e[1]) for e in events) / nst
    obs_ms = sum(e[1].elapsed_time(e[2]) for e in events) / nst
    k_after = nlegal_log[args.warmup:args.warmup + args.steps].to(torch.float64)          # K written by each obs launch
    # a slot loads its compact state (node_net i16 + owner i16 = 4 B/node; a slot that re-initialises reads and
    # writes 2 B/node each = 4 B/node too) and the observation writes 4·N·(2+7K); path / result writes are noise
    state_bytes = float((4.0 * n_nodes).sum().item())
    obs_bytes = float((4.0 * (2.0 + 7.0 * k_after) * n_nodes[None, :]).sum().item()) / nst
    real_per_step = real_steps / nst
    kernels = []
    headline_form = batch.observe_timing()[0] if fused else 0      # (asked NOW: the legs below run other forms on the same batch)
    if fused:
        form = headline_form                        # 1 fused launch, 2 split, 3 queue (the default where it applies)
        kname = "xr_step_queue_kernel" if form == 3 else "xr_route_kernel"
        note = ("the step kernel (xr_batch_step_observe, queue form): one persistent launch draining route tasks (LDS-resident "
                "router: latency-bound) and net-plane units of the fp32 observation (HBM-write-bound) — a MIXED kernel, priced "
                "against the HBM roofline because >99 % of its bytes are the observation stream; the HIP-event time also covers "
                "the planning kernel (~8 us) before it; bytes = state load + observation" if form == 3 else
                "fused step kernel (xr_batch_step_observe): per env one workgroup routes, then streams the fp32 observation; "
                "bytes = state load + observation" + ("; split form: net planes from xr_netplane_kernel" if form == 2 else ""))
        kernels.append(kernel_entry(kname, route_ms, obs_bytes + state_bytes, real_per_step, "hbm-write (routing phase: lds-latency)", note))
    else:
        if obs is not None:
            kernels.append(kernel_entry("xr_obs_kernel", obs_ms, obs_bytes + state_bytes, real_per_step, "hbm-write", "stand-alone observation"))
        kernels.append(kernel_entry("xr_route_kernel", route_ms, state_bytes, real_per_step, "lds-latency",
                                    "route-only step (xr_batch_step): distance field LDS-resident, HBM bytes are the state load only"))
    dom = max(kernels, key=lambda k: k["ms"])

    # ---- extra legs (rank 0, N = 1): route-only kernel and BASELINE config 5, each with its own numbers ---------------
    sustained = None
    if do_legs and fused and not args.no_extras:
        try:        # the timed region is tens of milliseconds: the same step for ~1 s more, so that clocks / thermals show
            n_s = 500
            marks = [torch.cuda.Event(enable_timing=True) for _ in range(n_s // 100 + 1)]
            torch.cuda.synchronize(dev)
            s0, t_s = batch.total_steps(), time.perf_counter()
            for i in range(n_s):
                if i % 100 == 0:
                    marks[i // 100].record()
                batch.random_actions(args.seed + 200000 + i, acts)
                batch.step(acts, obs)
            marks[-1].record()
            torch.cuda.synchronize(dev)
            el = time.perf_counter() - t_s
            sustained = {"steps": n_s, "seconds": round(el, 4), "value": round((batch.total_steps() - s0) / el, 1), "unit": "env-steps/s",
                         "ms_per_step": round(el / n_s * 1e3, 4),
                         "ms_per_step_by_100": [round(marks[j].elapsed_time(marks[j + 1]) / 100, 4) for j in range(n_s // 100)],
                         "what": "the headline step (random actions + xr_batch_step_observe, full rewrite) for 500 more batched steps right after "
                                 "the timed region, host wall clock; never part of `value`"}
        except Exception as ex:
            sustained = {"error": str(ex)}
    if do_legs:
        try:
            n_leg = max(args.steps, 5)
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_leg)]
            s0 = batch.total_steps()
            for i, (e0, e1) in enumerate(evs):
                batch.random_actions(args.seed + 100000 + i, acts)
                e0.record()
                batch.step(acts)
                e1.record()
            torch.cuda.synchronize(dev)
            ms = sum(a.elapsed_time(bb) for a, bb in evs) / n_leg
            real = (batch.total_steps() - s0) / n_leg
            if fused:
                kernels.append(kernel_entry("xr_route_kernel", ms, state_bytes, real, "lds-latency",
                                            f"route-only step (xr_batch_step) on the same {B} envs at the same stationary nets-left distribution: "
                                            "the distance field never leaves LDS, so the HBM fraction is ~1 % by construction; "
                                            "the time covers the ordering kernel (longest predicted route first, xr_config.launch_order) "
                                            "ahead of the route launch; env_steps_per_s is the figure of merit"))
        except Exception as ex:          # a leg must never take the headline down
            kernels.append({"kernel": "xr_route_kernel", "error": str(ex)})
        if fused:
            try:        # in-place form: the same step into the same (persistent) buffer, only the planes that change are written
                n_leg = max(args.steps, 5)
                evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_leg)]
                batch.observation(obs)
                units = torch.zeros(n_leg, dtype=torch.int32, device=dev)
                s0 = batch.total_steps()
                for i, (e0, e1) in enumerate(evs):
                    batch.random_actions(args.seed + 200000 + i, acts)
                    e0.record()
                    batch.step(acts, obs, inplace=True)
                    e1.record()
                    batch.fetch("units", units[i:i + 1])
                torch.cuda.synchronize(dev)
                ms = sum(a.elapsed_time(bb) for a, bb in evs) / n_leg
                real = (batch.total_steps() - s0) / n_leg
                N0 = float(regions[0].n_nodes)
                # bytes that MUST change: state load 4·N + planes 0..1 (8·N) per slot + 28·N per planned unit (nets above the routed one;
                # all nets of a slot that re-initialises)
                nb = 12.0 * float(n_nodes.sum().item()) + 28.0 * N0 * float(units.double().mean().item())
                ent = kernel_entry("xr_step_queue_kernel (in-place: xr_batch_step_observe_inplace)", ms, nb, real, "hbm-write (routing phase: lds-latency)",
                                   "the full step into the caller's persistent observation buffer: net planes are static and the channel order is "
                                   "'nets ascending', so only planes 0..1 and the planes of the remaining nets ABOVE the routed one change — the buffer "
                                   "ends up byte-identical to the full write (tests/test_gpu_obs.py); bytes = the planes that change, NOT 4·N·(2+7K)")
                ent["mean_units_per_env_step"] = float(units.double().mean().item()) / B
                ent["full_rewrite_bytes"] = obs_bytes + state_bytes
                kernels.append(ent)
            except Exception as ex:
                kernels.append({"kernel": "xr_step_queue_kernel (in-place)", "error": str(ex)})
        # the headline's observation buffer (36 GB) and batch are not needed by the legs below, which bring their own: released here, so that every
        # leg runs beside nothing but itself, like the stand-alone commands the profiles were taken with (bench.py --region-pack ...)
        obs = None
        torch.cuda.empty_cache()
        try:
            kernels.append(config2_leg(args, dev))
        except Exception as ex:
            kernels.append({"kernel": "xr_ingest_state_kernel + xr_obs_kernel (BASELINE config 2)", "error": str(ex)})
        if c5_regions:
            try:
                kernels.append(config5_leg(args, c5_regions, dev))
            except Exception as ex:
                kernels.append({"kernel": "config5 route", "error": str(ex)})
        try:
            kernels.append(v2_leg(args, regions, dev, first_env))
        except Exception as ex:
            kernels.append({"kernel": "xr_route_kernel (XR-Maze v2: the reference's TCL knobs)", "error": str(ex)})
        if pack_regions:
            try:        # (its observation buffer, 68 GB at K = 77, sits beside the headline's 36 GB: sized for 288 GB of HBM)
                kernels.append(pack_leg(args, pack_regions, dev))
            except Exception as ex:
                kernels.append({"kernel": "xr_step_queue_kernel (design-derived ispd18_test1 region pack)", "error": str(ex)})
            if all(getattr(r, "guide_off", None) is not None for r in pack_regions):
                try:
                    kernels.append(v2_leg(args, None, dev, 0, pack=pack_regions))
                except Exception as ex:
                    kernels.append({"kernel": "xr_route_kernel (XR-Maze v2 + the design's guide rectangles, ispd18_test1 region pack)", "error": str(ex)})
                try:        # the reference's own configuration as a FULL step: v2 knobs + the design's guides + observation, oracle replay incl. observation bytes
                    kernels.append(pack_leg(args, pack_regions, dev, v2=dict(V2_KNOBS, guide_margin=1)))
                except Exception as ex:
                    kernels.append({"kernel": "xr_step_queue_kernel (design-derived ispd18_test1 region pack, XR-Maze v2)", "error": str(ex)})

    traffic = None
    traffic_scaled = False
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")       # rocprofv3 --pmc passes of THIS command (tools/profile_round.sh)
    if os.path.exists(pmc):
        try:
            pj = json.load(open(pmc))
            # one entry per measured command (`runs`); the file's top level is the first of them
            for run in [pj] + list(pj.get("runs", [])):
                # same build, same workload (launches of the stationary distribution: --steps / --warmup only choose how many of them are averaged)
                wl = lambda d: {k: v for k, v in (d or {}).items() if k not in ("steps", "warmup")}
                same = run.get("source_sha") == source_sha() and wl(run.get("bench_args")) == wl(bench_args_key(args, world))
                ent = run.get(dom["kernel"])
                if same and isinstance(ent, dict) and ent.get("hbm_total_bytes"):
                    if run.get("bench_args") == bench_args_key(args, world) or not ent.get("traffic_over_algorithmic"):
                        traffic = ent.get("hbm_total_bytes")
                    else:       # other --steps / --warmup: the measured traffic-to-algorithmic ratio on this run's mean launch
                        traffic = float(ent["traffic_over_algorithmic"]) * float(dom["bytes"])
                        traffic_scaled = True
                    break
        except Exception:
            traffic = None
    roofline = {"kernel": dom["kernel"], "bound": "hbm", "achieved": round(dom["achieved"], 2), "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": round(dom["achieved"] / HBM_PEAK_GBPS, 4), "traffic": traffic,
                "traffic_note": (("HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this build and workload with other --steps / --warmup "
                                  "(profiles/pmc_traffic.json): its measured traffic / algorithmic ratio x this run's algorithmic bytes per launch" if traffic_scaled else
                                  "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this exact command and build "
                                  "(profiles/pmc_traffic.json)") if traffic is not None else
                                 "null: no PMC pass recorded for this build + command (profiles/pmc_traffic.json carries the last one with its source hash)"),
                "avg_launch_ms": round(dom["ms"], 4), "algorithmic_bytes_per_launch": int(dom["bytes"])}

    parity = None
    can_replay = not args.region_pack and (len(regions) >= B or args.regions > 0)      # regions == env slots: rotation keeps every slot on its region (--regions: no rotation), the oracle subset can follow
    if can_replay and (multi or not args.no_cpu_baseline):
        try:
            seeds = [args.seed + rank * 7919 + i for i in range(args.warmup + args.steps)]
            parity = parity_check(slot_regions, seeds, stagger, gpu_hash, gpu_cum, n_check=n_par, obs_sha=obs_sha,
                                  actions_log=acts_log.cpu().numpy() if learner else None, v2=main_v2 or None)
        except Exception as ex:
            parity = {"error": str(ex), "ok": False}
    if multi:
        flag = torch.tensor([1 if (parity or {}).get("ok") else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        parity = dict(parity or {}, all_ranks_ok=bool(flag.item() == 1), envs_per_rank=n_par)

    out = None
    if rank == 0:
        mean_k = float(k_after.mean().item())
        Bg = args.global_envs if strong else B * world
        out = {
            "metric": "env-steps/sec (batched regions), " + ("synthetic 256x256x12 regions (BASELINE config 5)" if args.config == 5 and not args.region_pack else "ispd18_test1-sized regions"),
            "value": round(total_real / elapsed_max, 1),
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed_max / nst * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "u32 distances / i16 state / fp32 observation",
            "data": "synthetic",
            "config": {"workload": (f"{B} env slots over the design-derived ispd18_test1 region pack {os.path.basename(args.region_pack)}, "
                                    if args.region_pack else
                                    f"BASELINE config 5: {Bg} env slots of synthetic 256x256x12 dense-congestion regions (K = 32"
                                    + (f", {len(regions)} distinct regions cycled over the slots" if args.regions > 0 else "") + f"), {B} per GPU, "
                                    if args.config == 5 else
                                    f"north_star target: a {Bg}-env batch of ispd18_test1-sized regions (24x40x9, K~U[4,36]; generator of "
                                    f"BASELINE configs 2-4), {B} per GPU, ")
                                   +
                                   f"full step = random net-order action + XR-Maze {'v2 (maze_end_iter 3, guide cost 800)' if args.maze_v2 else 'v1'} route + metrics/reward"
                                   + ("" if not with_obs else " + reference-layout fp32 observation of every env")
                                   + (" (queue form: one persistent launch after a planning kernel)" if headline_form == 3 else
                                      " (split form: route kernel + concurrent net-plane writer)" if headline_form == 2 else
                                      " (fused launch: one workgroup per env)" if fused else "")
                                   + (", RCCL all_gather of per-env results" + ("" if learner else " overlapped with the next step (two buffer pairs, async)") if multi else "")
                                   + (" + learner flow (policy on rank 0, i32 action broadcast)" if learner else "")
                                   + ("" if args.no_stagger else "; episodes staggered to the stationary nets-left distribution before timing"),
                       "envs_per_gpu": B, "global_envs": Bg, "parallelism": f"env-shard x{world}",
                       "mean_nets_left": round(mean_k, 2), "slots_stepped_per_batch_step": round(total_real / (nst * B * world), 4),
                       "router": {0: "default", 1: "sweep", 2: "dial"}[args.router], "source_sha": source_sha(),
                       "bench_args": bench_args_key(args, world),
                       **({"forced_collectives": "one rank taking every N > 1 branch on the real backend (--force-collectives): a functional run of the multi-GPU code path, not a scaling point"}
                          if args.force_collectives else {})},
            "roofline": roofline,
            "kernels": [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in kk.items()} for kk in kernels],
        }
        if parity is not None:
            out["parity"] = parity
        if certify is not None:
            out["gather_verified"] = certify["gather_verified"]
            out["ranks_seen"] = certify["ranks_seen"]
            out["gathered_rows"] = certify["rows"]
        if do_legs and not args.no_extras:
            out["extras"] = extras_leg(args, regions, dev, batch, obs)
            if sustained is not None:
                out["extras"]["sustained"] = sustained
        if (not multi) and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(regions, args.cpu_seconds, with_obs=with_obs)
            except Exception as ex:          # the oracle is optional for the GPU number itself
                out["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {ex}"}
        if parity is not None:       # mirrored into keys the driver's parser keeps
            out["config"]["parity_ok"] = bool(parity.get("ok"))
            out["roofline"]["parity_ok"] = bool(parity.get("ok"))
        out["legs"] = legs_summary(out)         # LAST: the tail of the line holds every leg
        failed = multi and not (certify["gather_verified"] and certify["ranks_seen"] == args.gpus and parity.get("all_ranks_ok"))
        if failed:          # an N-GPU label is only printed for a run that proved it was one
            out["n_gpus"] = None
            out["error"] = (f"N > 1 self-certification failed: ranks_seen {certify['ranks_seen']} of {args.gpus}, gather_verified "
                            f"{certify['gather_verified']}, parity on every rank {parity.get('all_ranks_ok')}")
            print(out["error"], file=sys.stderr)
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()
        if not (certify["gather_verified"] and certify["ranks_seen"] == args.gpus and parity.get("all_ranks_ok")):
            sys.exit(3)


def bench_args_key(args, world):
    return {"gpus": world, "steps": args.steps, "warmup": args.warmup, "envs": args.envs, "global_envs": args.global_envs,
            "config": args.config, "regions": args.regions, "seed": args.seed, "router": args.router, "obs_mode": args.obs_mode,
            "no_stagger": bool(args.no_stagger), "region_pack": os.path.basename(args.region_pack) if args.region_pack else None,
            "maze_v2": bool(args.maze_v2)}


def config1_wire_leg(seconds=4.0):
    """A `cpu_baseline` leg (kind "port": the oracle is the thing timed here, as in `cpu_baseline`, never the product).
    BASELINE config 1 as SURVEY §8d states it — "B = 1, 24x40x9, K = 10: CPU oracle only, single thread (+ proto encode/decode loopback)" — the
    reference's own plumbing (examples/launch_training.py:57-86 starts ONE simulator; every step is a ZMQ round trip of the whole region,
    baseline/baseline_utils.py:409-423) restated on the host cores of this box, one thread, no GPU anywhere:

        simulator side   oracle/xr_oracle.c routes the chosen net (`OracleEnv.step`), the new state is serialised as the ~200 KB
                         `Message{request}` (serve.SimState.encode -> xr_proto_encode_request)
        agent side       xr_proto_decode (C ABI), `handle_messange`'s `data` lists (proto.request_to_data), the fp32 grid by the oracle's
                         build_3Dgrid restatement, the `Message{response}` going back (xr_proto_encode_response -> xr_proto_decode)

    Sockets are not timed (loopback in one process): this is a LOWER bound of the reference path's per-step cost on these cores, with compiled
    stand-ins for its Python.  Per-component milliseconds; the in-process GPU `Game.step` of the same region stands beside it in `extras`."""
    import numpy as np
    from oracle import xr_oracle as orc
    from xroute_env_amd import proto
    from xroute_env_amd.regions import config_regions, pack_records, unpack_records
    from xroute_env_amd.serve import SimState
    regs = config_regions(1, 8)
    proto.decode_message(proto.encode_response(0))          # (library load outside the timed loop)
    comp = {k: 0.0 for k in ("sim_route", "encode_request", "decode_request", "data_lists", "build_3Dgrid", "response_round_trip")}
    n_steps = n_bytes = 0
    t_end = time.perf_counter() + seconds
    ep = 0
    while time.perf_counter() < t_end or n_steps == 0:
        reg = regs[ep % len(regs)]
        ep += 1
        env = orc.OracleEnv(reg)
        ntype, _, net, pin = unpack_records(reg.nodes)
        while env.nlegal() > 0:
            a = int(env.legal()[0])
            t0 = time.perf_counter()
            raw_resp = proto.encode_response(a - 1)                     # the agent's answer (baseline_utils.py:409-411) ...
            a_sim = proto.decode_message(raw_resp).net_index + 1        # ... parsed by the simulator
            t1 = time.perf_counter()
            env.step(a_sim)
            t2 = time.perf_counter()
            owner = env.owner()[: reg.n_nodes]
            legal = env.legal()
            st = SimState(tuple(reg.dims), proto.region_wire_fields(reg, pack_records(ntype, (owner != 0).astype(np.int64), net, pin)),
                          tuple(int(v) for v in env.cum()), np.asarray(legal, np.uint32) - 1, len(legal) == 0)
            raw = st.encode()
            t3 = time.perf_counter()
            msg = proto.decode_message(raw)
            t4 = time.perf_counter()
            data = proto.request_to_data(msg)
            t5 = time.perf_counter()
            obs = env.observation()
            t6 = time.perf_counter()
            comp["response_round_trip"] += t1 - t0; comp["sim_route"] += t2 - t1; comp["encode_request"] += t3 - t2
            comp["decode_request"] += t4 - t3; comp["data_lists"] += t5 - t4; comp["build_3Dgrid"] += t6 - t5
            n_steps += 1
            n_bytes += len(raw)
            assert len(data[1]) == reg.n_nodes and obs.size == (2 + 7 * len(legal)) * reg.n_nodes
    per = {k: round(v / n_steps * 1e3, 4) for k, v in comp.items()}
    total = sum(comp.values()) / n_steps
    return {"ms_per_step": round(total * 1e3, 4), "value": round(1.0 / total, 2), "unit": "env-steps/s", "cores": 1, "kind": "port", "steps": n_steps,
            "request_bytes_mean": int(n_bytes / n_steps), "ms_per_component": per,
            "what": "BASELINE config 1 (SURVEY 8d): one ispd18_test1-sized region (24x40x9, K = 10), ONE thread of this box's host: oracle route + the real ~200 KB "
                    "Request encoded and decoded through the C-ABI codec + handle_messange's data lists + the oracle's build_3Dgrid + the Response, loopback "
                    "(no sockets): a lower bound of the reference's per-step path with compiled stand-ins for its Python (the reference's own build_3Dgrid alone: "
                    "21 ms per call, BASELINE.md)"}


def config2_leg(args, dev):
    """BASELINE config 2: "256 parallel regions on 1 x MI355X, random net-order policy, grid-build + reward only" — the env driven by an EXTERNAL
    simulator's states: per step every env ingests a new state (xr_batch_ingest_state: occupancy of every node, nets left, cumulative metrics ->
    metric deltas, f64 reward, done, the 48-byte record) and builds the reference-layout fp32 observation of it (xr_batch_observation); NO route in
    the timed region.  The states are produced beforehand by a twin batch that routes (random net-order policy, untimed), so the whole thing is
    checked like every other leg: the oracle replays the twin's actions and rebuilds the observations the timed launches wrote."""
    import numpy as np
    import torch
    from xroute_env_amd.batch import RegionBatch
    B, S, n_t = 256, 6, max(args.steps, 10)
    regs = gen_regions_inline(2, B)
    twin = RegionBatch(regs, n_envs=B, device=dev, auto_reset=False)
    main = RegionBatch(regs, n_envs=B, device=dev, auto_reset=False)
    twin.reset(); main.reset()
    a = torch.empty(B, dtype=torch.int32, device=dev)
    owners, legals, cums, acts_log = [], [], [], []
    for i in range(S):                                     # the scripted states: S random-order steps of every env (K0 >= 4 > S - 2: nobody finishes early... see `done`)
        twin.random_actions(args.seed + 0xC2 + i, a)
        acts_log.append(a.clone())
        twin.step(a)
        owners.append(twin.fetch("owner").clone()); legals.append(twin.fetch("legal").clone()); cums.append(twin.fetch("cum").clone())
    obs = main.alloc_observation()
    rec = torch.empty((B, 48), dtype=torch.uint8, device=dev)
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n_t)]
    for w in range(2):                                     # warm-up (first touch of the 9 GB buffer)
        main.ingest_state(owners[w % S], legals[w % S], cums[w % S]); main.observation(obs)
    main.reset()
    k_sum = 0.0
    for i, ev in enumerate(evs):
        j = i % S
        if j == 0 and i:
            main.reset()                                   # (untimed) back to the initial state: the deltas of state 0 are relative to it
        ev[0].record()
        main.ingest_state(owners[j], legals[j], cums[j])
        ev[1].record()
        main.observation(obs)
        main.fetch("record", rec)
        ev[2].record()
        k_sum += float(main.fetch("nlegal").double().sum().item())
    torch.cuda.synchronize(dev)
    ing_ms = sum(e[0].elapsed_time(e[1]) for e in evs) / n_t
    obs_ms = sum(e[1].elapsed_time(e[2]) for e in evs) / n_t
    N = float(regs[0].n_nodes)
    k_mean = k_sum / (n_t * B)
    # ingest reads the new occupancy and writes the state (2·N + 2·N), the observation reads the compact state (node_net + owner: 4·N) and
    # writes 4·N·(2 + 7K)
    ing_bytes = 4.0 * N * B
    obs_bytes = (4.0 * N + 4.0 * N * (2.0 + 7.0 * k_mean)) * B
    # ---- checks: the last timed step ingested state j_last; its records and observation against the twin (same state) and the oracle
    j_last = (n_t - 1) % S
    r = main.records()
    want_delta = (cums[j_last] - (cums[j_last - 1] if j_last else torch.from_numpy(np.array([rg.metrics0 for rg in regs], np.int32)).to(dev))).cpu().numpy()
    ok_delta = bool(np.array_equal(np.asarray(r["delta"]), want_delta))
    w_rew = -1.0 * (500.0 * want_delta[:, 0].astype(np.float64) + 4.0 * want_delta[:, 2] + 0.5 * want_delta[:, 1])
    ok_rew = bool(np.array_equal(np.asarray(r["reward"], np.float64), w_rew))
    from oracle import xr_oracle as orc
    n_chk, ok_obs = 16, True
    al = torch.stack(acts_log).cpu().numpy()
    for e in range(0, B, B // n_chk):
        env = orc.OracleEnv(regs[e])
        for i in range(j_last + 1):
            if al[i, e]:
                env.step(int(al[i, e]))
        ro = env.observation()
        ok_obs = ok_obs and bool(np.array_equal(ro.ravel(), obs[e, : ro.size].cpu().numpy()))
    ent = kernel_entry("xr_ingest_state_kernel + xr_obs_kernel (BASELINE config 2: 256 envs, grid-build + reward only)", ing_ms + obs_ms, ing_bytes + obs_bytes,
                       float(B), "hbm-write",
                       "256 env slots of ispd18_test1-sized regions (24x40x9, K~U[4,36]); per step every env ingests a new externally produced state "
                       "(xr_batch_ingest_state: owner row copy, nets-left mask, metric deltas, f64 reward, done, record) and writes its reference-layout fp32 "
                       "observation (xr_batch_observation) — no route; bytes = 4·N ingest + 4·N state read + 4·N·(2+7K) observation per env. 256 envs are "
                       "one wave of workgroups on a quarter of the chip's CUs x 4: the launch is too short to reach the 4096-env write rate")
    ent.update(envs=B, ingest_ms=round(ing_ms, 4), observation_ms=round(obs_ms, 4), mean_nets_left=round(k_mean, 2),
               ingest_frac=round(ing_bytes / (ing_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), observation_frac=round(obs_bytes / (obs_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
               parity={"deltas_equal": ok_delta, "rewards_bit_equal": ok_rew, "observations_equal": ok_obs, "observations_checked": n_chk,
                       "ok": bool(ok_delta and ok_rew and ok_obs),
                       "what": "records of the last timed ingest vs the twin batch's cumulative metrics (deltas) and the reference's reward expression in f64; "
                               "observation bytes of 16 envs vs the oracle replaying the twin's actions"})
    twin.close(); main.close()
    return ent


def gen_regions_inline(config, n):
    from xroute_env_amd.regions import config_regions
    return config_regions(config, n)


def legs_summary(out):
    """Every leg of the line in <= 1.5 KB, LAST in the JSON (the driver keeps the tail of the line): ms per launch, fraction of the HBM peak (or
    M env-steps/s where bandwidth is not the bound), and the leg's own oracle check."""
    short = [("xr_step_queue_kernel (in-place", "step_inplace"), ("BASELINE config 5", "c5_route_1024"), ("BASELINE config 2", "c2_obs_reward_256"),
             ("region pack, XR-Maze v2", "pack_v2_step"), ("XR-Maze v2 + the design's guide", "pack_v2_route"), ("design-derived ispd18_test1 region pack)", "pack_step"),
             ("XR-Maze v2", "v2_route"), ("xr_step_queue_kernel", "step"), ("xr_route_kernel", "route_only"), ("xr_obs_kernel", "obs")]
    legs = {}
    for k in out.get("kernels", []):
        name = next((s_ for pat, s_ in short if pat in k.get("kernel", "")), k.get("kernel", "?")[:24])
        while name in legs:
            name += "'"
        if "error" in k:
            legs[name] = {"error": str(k["error"])[:60]}
            continue
        e = {"ms": round(k["ms"], 4), "frac": round(k["frac"], 4), "Meps": round(k["env_steps_per_s"] / 1e6, 3)}
        if isinstance(k.get("parity"), dict):
            e["ok"] = bool(k["parity"].get("ok"))
        if "launch_utilisation" in k:
            e["util"] = k["launch_utilisation"]["utilisation"]
        if "at_4x_slots" in k:
            e["ms_4096"] = k["at_4x_slots"]["ms"]
        legs[name] = e
    if "step" in legs and isinstance(out.get("parity"), dict):
        legs["step"]["ok"] = bool(out["parity"].get("ok"))
    ex = out.get("extras") or {}
    g = ex.get("config1_game_step") or {}
    if "ms_median" in g:
        legs["c1_game_step"] = {"ms": g["ms_median"]}
    w = ex.get("config1_wire_loopback") or {}
    if "ms_per_step" in w:
        legs["c1_wire_loopback_cpu"] = {"ms": w["ms_per_step"], "eps": w["value"]}
    d3 = ex.get("config3_dqn_attached") or {}
    if "value" in d3:
        legs["c3_dqn_1024"] = {"ms": d3["ms_per_step"], "Meps": round(d3["value"] / 1e6, 3), "env_share": d3.get("env_share_of_step_time")}
    c4 = ex.get("config4_ppo_attached_per_gpu_share") or {}
    for key, nm in (("policy_per_rank", "c4_ppo_512_rank"), ("central_learner_from_compact_state", "c4_ppo_512_learner")):
        if isinstance(c4.get(key), dict) and "value" in c4[key]:
            legs[nm] = {"ms": c4[key]["ms_per_step"], "Meps": round(c4[key]["value"] / 1e6, 3), "ok": bool(c4[key].get("parity_ok"))}
    tr = ex.get("ppo_training_cadence") or {}
    if "refill_ms" in tr:
        legs["ppo_cadence_100"] = {"refill_ms": tr["refill_ms"], "window_ms": tr.get("window_ms"), "refill_share": tr.get("refill_share_of_window")}
    su = ex.get("sustained") or {}
    if "ms_per_step" in su:
        legs["sustained_500"] = {"ms": su["ms_per_step"]}
    return legs


def extras_leg(args, regions, dev, batch, obs):
    """Driver-visible side numbers of the default run (never part of `value`): BASELINE config 1 (batch-1 `Game.step` latency through
    the reference-shaped API, host observation) and config 3 (DQN counterpart attached, 1024 envs, compact-consumer mode)."""
    import copy
    import torch
    ex = {}
    try:
        from xroute_env_amd.game import Game
        del obs
        batch.close()
        torch.cuda.empty_cache()
        from xroute_env_amd.regions import config_regions
        g = Game(regions=config_regions(1, 8), device=dev)       # BASELINE config 1: 24x40x9, K = 10 — not the headline's K ~ U[4,36] regions
        g.reset()
        ts = []
        for ep in range(5):
            g.reset()
            done = False
            while not done:
                a = min(g.legal_action_set)
                t0 = time.perf_counter()
                o, done, *_ = g.step(a)
                ts.append(time.perf_counter() - t0)
        ts.sort()
        ex["config1_game_step"] = {"ms_median": round(ts[len(ts) // 2] * 1e3, 4), "ms_p10": round(ts[len(ts) // 10] * 1e3, 4),
                                   "ms_p90": round(ts[9 * len(ts) // 10] * 1e3, 4), "steps": len(ts),
                                   "what": "BASELINE config 1: Game.step on one ispd18_test1-sized region (24x40x9, K = 10: the config's generator) through the reference-shaped API, "
                                           "in-process simulator, observation returned as a CPU tensor like the reference's (PCIe-inclusive)"}
        del g
    except Exception as exn:
        ex["config1_game_step"] = {"error": str(exn)}
    try:
        ex["config1_wire_loopback"] = config1_wire_leg(min(4.0, max(1.0, args.cpu_seconds / 3)))
    except Exception as exn:
        ex["config1_wire_loopback"] = {"error": str(exn)}
    try:
        a2 = copy.copy(args)
        a2.agent, a2.agent_full_obs, a2.steps, a2.warmup = "dqn", False, 20, 3
        r = agent_leg(a2, regions[:1024], dev, 1)
        ex["config3_dqn_attached"] = {k: r[k] for k in ("value", "unit", "steps", "ms_per_step", "env_share_of_step_time", "agent_ms_per_step", "env_ms_per_step", "tower_roofline",
                                                        "net_cache_refill_ms", "net_vectors_by", "net_tower_roofline")}
        ex["config3_dqn_attached"]["what"] = r["config"]["workload"]
    except Exception as exn:
        ex["config3_dqn_attached"] = {"error": str(exn)}
    try:        # the PPO counterpart on the WHOLE 4096-env batch, frozen weights and at the reference's training cadence (an update every 100 steps)
        a5 = copy.copy(args)
        a5.agent, a5.agent_full_obs, a5.steps, a5.warmup = "ppo", False, 20, 3
        r = agent_leg(a5, regions, dev, 1)
        tc = r.get("training_cadence") or {}
        ex["ppo_training_cadence"] = dict({k: r[k] for k in ("value", "unit", "ms_per_step", "agent_ms_per_step", "env_ms_per_step", "net_vectors_by")}, envs=len(regions), **tc)
    except Exception as exn:
        ex["ppo_training_cadence"] = {"error": str(exn)}
    try:        # BASELINE config 4's per-GPU share (512 envs) with the PPO baseline attached, both placements of the policy, on this one GPU
        a4 = copy.copy(args)
        n4 = min(512, len(regions))
        a4.agent, a4.steps, a4.warmup, a4.global_envs, a4.region_pack, a4.regions, a4.maze_v2, a4.no_stagger = "ppo", 20, 3, n4, None, 0, False, False
        keep = ("value", "unit", "steps", "ms_per_step", "env_share_of_step_time", "agent_ms_per_step", "env_ms_per_step", "step_split_ms_rank0", "actions_sha")
        both = {}
        for name, lrn in (("policy_per_rank", False), ("central_learner_from_compact_state", True)):
            a4.learner = lrn
            r = agent_sharded(a4, regions[:n4], dev, 1, 0, 0, n4, True, emit=False)
            both[name] = {k: r[k] for k in keep}
            both[name]["parity_ok"] = bool(r["parity"].get("ok"))
            if lrn:
                both[name]["compact_state"] = r["compact_state"]
        both["same_actions_in_both_placements"] = both["policy_per_rank"]["actions_sha"] == both["central_learner_from_compact_state"]["actions_sha"]
        both["envs"] = n4
        both["what"] = ("BASELINE config 4 (4096 regions over 8 GPUs, PPO baseline): one GPU's share, 512 envs, full maze route per step, the PPO counterpart choosing every action "
                        "— evaluated on the shard itself (what config 4 should use) and from gathered compact state (SURVEY 8e's central learner; at N = 1 the gather is a no-op, "
                        "pack + expand are real); `python bench.py --gpus 8 --global-envs 4096 --agent ppo [--learner]` is the 8-GPU command")
        ex["config4_ppo_attached_per_gpu_share"] = both
    except Exception as exn:
        ex["config4_ppo_attached_per_gpu_share"] = {"error": str(exn)}
    return ex


def tower_macs(dims):
    """Multiply-adds of the obstacle tower on one (D, H, W) grid as the reference computes it on the cells the data can influence
    (baseline/baseline_utils.py:231-379): ResidualBlock(1) 2 x 27 per cell, the aligning 5x5x5 convolution 125 x 7 per output cell,
    ResidualBlock(7) 2 x 27 x 49 per cell of its two activations, the (3, 64, 3) convolution 21 per cell."""
    from xroute_env_amd import agents
    D, H, W = dims
    od, oh, ow = [(s + 2 - 5) // k + 1 for s, k in zip((D, H, W), agents.align_stride((D, H, W)))]
    return 2 * 27 * D * H * W + 875 * od * oh * ow + 1323 * 3 * ((oh + 1) * (ow + 1) + (oh + 2) * (ow + 2)) + 21 * 3 * (oh + 2) * (ow + 2)


def agent_leg(args, regions, dev, world):
    """BASELINE config 3 ("1024 regions, full maze-route per step, DQN baseline attached") / config 4's PPO: the env batch
    stepped with actions from the batched policy counterpart.  Compact-consumer mode: per step the env writes planes 0..1
    only (xr_batch_step_compact); the 7 static planes of a net go through the net tower once per (region, net)
    (agents.NetVectorCache over xr_batch_net_planes) — the same logits as the reference's per-step re-encoding."""
    import torch
    from xroute_env_amd import agents
    from xroute_env_amd.batch import RegionBatch
    mixed = len({tuple(int(v) for v in r.dims) for r in regions}) > 1          # a region pack: several grid shapes in one batch
    B = args.envs if mixed else len(regions)
    torch.manual_seed(0)
    if os.environ.get("XR_CUDNN_BENCHMARK"):        # experiments: let MIOpen search its convolution algorithms (fixed conv batch shapes)
        torch.backends.cudnn.benchmark = True
    model = (agents.RepActor() if args.agent == "dqn" else agents.ActorCritic(64)).to(dev).eval()
    batch = RegionBatch(regions, n_envs=B, device=dev, auto_reset=True, router=args.router, dial_mult=args.dial_mult,
                        launch_order=args.launch_order, **(dict(max_route_count=1 << 30) if mixed else {}),
                        **(dict(V2_KNOBS, guide_margin=1) if (mixed and args.maze_v2) else {}))
    batch.reset(rotate=True)
    dims = regions[0].dims
    full = args.agent_full_obs and not mixed
    buf = batch.alloc_observation() if full else batch.alloc_head()
    if full:
        batch.observation(buf)
    else:
        _head_of(batch, buf)
    grouped = agents.GroupedFusedPolicy(model, batch, dev) if mixed else None      # (per grid shape: its own fused tower; one actor head, one cache)
    cache = grouped.cache if mixed else agents.NetVectorCache(len(regions), batch.k_max, dev)
    # the obstacle tower as one fused HIP kernel (csrc/xr_agent.hip); --agent-lib-tower keeps the framework's convolutions (A/B)
    tower = None if (args.agent_lib_tower or mixed) else agents.FusedObstacleTower(model.representation_network, (dims[2], dims[1], dims[0]), dev)
    head_k = None if (args.agent_lib_tower or full or mixed) else agents.FusedActorHead(model.actor, dev)     # (+ the actor MLP and the arg-max as one kernel)
    if head_k is not None:      # every (region, net) through the net tower once, up front: no cache-miss check (= no host round trip) per step
        cache.prefill(model.representation_network, [r.n_nets for r in regions], batch.net_planes, dims)
    nl = torch.empty(B, dtype=torch.int32, device=dev)
    reg = torch.empty(B, dtype=torch.int32, device=dev)

    def act():
        batch.fetch("nlegal", nl)
        batch.fetch("region", reg)
        if mixed:
            return grouped.actions(buf, nl, reg, sample=(args.agent == "ppo"))
        kw = dict(cache=cache, region=reg, ob_tower=tower, actor_head=head_k)
        if not full:
            kw["planes_fn"] = batch.net_planes
        if args.agent == "dqn":
            return agents.dqn_actions(model, buf, nl, dims, **kw)
        return agents.ppo_actions(model, buf, nl, dims, **kw)[0]

    def env_step(a):
        if full:
            batch.step(a, buf)
        else:
            batch.step_compact(a, buf)

    for _ in range(max(args.warmup, 2)):            # MIOpen kernel selection + the net-vector cache fills
        env_step(act())
    n = max(args.steps, 1)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n)]
    torch.cuda.synchronize(dev)
    s0 = batch.total_steps()
    t0 = time.perf_counter()
    for i in range(n):
        ev[i][0].record()
        a = act()
        ev[i][1].record()
        env_step(a)
        ev[i][2].record()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    real = batch.total_steps() - s0
    agent_ms = sum(e[0].elapsed_time(e[1]) for e in ev) / n
    env_ms = sum(e[1].elapsed_time(e[2]) for e in ev) / n
    N = float(sum(regions[e % len(regions)].n_nodes for e in range(B))) / B
    kfloat = float(batch.fetch("nlegal").double().mean().item())
    env_bytes = B * (4.0 * N + (4.0 * N * (2 + 7 * kfloat) if full else 8.0 * N))
    tower_roof = None
    if tower is not None and tower.supported:       # the agent's dominant kernel alone, priced against the fp32 matrix (= vector) peak
        tev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for _ in range(3):
            tower(buf)
        tev[0].record()
        for _ in range(10):
            tower(buf)
        tev[1].record()
        torch.cuda.synchronize(dev)
        tms = tev[0].elapsed_time(tev[1]) / 10
        fl = 2.0 * B * tower_macs((dims[2], dims[1], dims[0]))
        tower_roof = {"kernel": "xr_ob_tower_kernel (%s)" % MATRIX_MODES[matrix_mode()], "matrix_mode": matrix_mode(), "bound": "mfma", "achieved": round(fl / (tms * 1e-3) / 1e12, 2),
                      "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(fl / (tms * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TFLOPS, 4), "traffic": None,
                      "avg_launch_ms": round(tms, 4), "ms_per_1024_envs": round(tms * 1024 / B, 4), "algorithmic_flops_per_launch": int(fl)}
    net_roof = None
    nt = next(iter(cache._net_towers.values()), None) if (not mixed and getattr(cache, "_net_towers", None)) else None
    if nt is not None and nt.supported:            # the net tower's kernel alone: every (region, net) pair of the batch in one launch
        try:
            reg_all = torch.cat([torch.full((int(r.n_nets),), i, dtype=torch.int64) for i, r in enumerate(regions)]).to(dev)
            net_all = torch.cat([torch.arange(1, int(r.n_nets) + 1, dtype=torch.int32) for r in regions]).to(dev)
            nt(batch, reg_all, net_all)
            tev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            tev[0].record()
            for _ in range(3):
                nt(batch, reg_all, net_all)
            tev[1].record()
            torch.cuda.synchronize(dev)
            nms = tev[0].elapsed_time(tev[1]) / 3
            D_, H_, W_ = dims[2], dims[1], dims[0]
            od, oh, ow = [(s_ + 2 - 5) // k_ + 1 for s_, k_ in zip((D_, H_, W_), agents.align_stride((D_, H_, W_)))]
            back = 1323 * 3 * ((oh + 1) * (ow + 1) + (oh + 2) * (ow + 2)) + 21 * 3 * (oh + 2) * (ow + 2)          # the matrix back end (what the kernel really multiplies)
            dense = 2 * 27 * 49 * D_ * H_ * W_ + 875 * 7 * od * oh * ow + back                                    # the framework path's multiply-adds per net
            npairs = int(reg_all.numel())
            net_roof = {"kernel": "xr_ob_tower_kernel<NET> (sparse front end; %s)" % MATRIX_MODES[matrix_mode()], "matrix_mode": matrix_mode(), "bound": "mfma",
                        "achieved": round(2.0 * npairs * back / (nms * 1e-3) / 1e12, 2), "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(2.0 * npairs * back / (nms * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TFLOPS, 4), "traffic": None, "avg_launch_ms": round(nms, 4),
                        "net_pairs": npairs, "ms_per_1024_nets": round(nms * 1024 / max(npairs, 1), 4), "nets_per_s": round(npairs / (nms * 1e-3), 1),
                        "dense_equivalent_tflops": round(2.0 * npairs * dense / (nms * 1e-3) / 1e12, 2),
                        "note": "achieved / frac count ONLY the multiply-adds the kernel performs on the matrix pipe (the 7-channel block and the last convolution); "
                                "`dense_equivalent_tflops` is what the framework path's dense convolutions of the same nets would have to sustain for this time — the sparse "
                                "front end skips ~85 % of them (the first block and the aligning convolution touch only the neighbourhoods of the access points)"}
        except Exception as exc:
            net_roof = {"error": str(exc)}
    # ---- the price of a weight update (VERDICT r5 #3a): the frozen-weights numbers above never pay the net tower — its vectors are cached per (region, net) —
    # but the reference's callers train: PPO updates every `update_timestep = 100` env steps (baseline/PPO/train_PPO.py:18,80), DQN after every step
    # (baseline/DQN/train_DQN.py:127).  Here: one "optimiser step" that keeps the values (every parameter written in place: all snapshots and the cache are
    # stale), then 100 steps — the first of them re-folds the towers / the actor head and refills the cache of net vectors.
    cadence = None
    if (head_k is not None or mixed) and not getattr(args, "no_cadence", False):
        try:
            with torch.no_grad():
                for p_ in model.parameters():
                    p_.mul_(1.0)
            c0, f0, w0, r0 = cache.computed, cache.fused_fills, cache.framework_fills, cache.refill_ms_total
            torch.cuda.synchronize(dev)
            tw = time.perf_counter()
            for _ in range(100):
                env_step(act())
            torch.cuda.synchronize(dev)
            window_ms = (time.perf_counter() - tw) * 1e3
            refill_ms = cache.refill_ms_total - r0
            cadence = {"update_every_steps": 100, "window_ms": round(window_ms, 3), "frozen_window_ms": round(dt / n * 1e5, 3), "refill_ms": round(refill_ms, 3),
                       "refill_share_of_window": round(refill_ms / max(window_ms, 1e-9), 4), "net_pairs_refilled": cache.computed - c0,
                       "net_vectors_by": {"fused_kernel": cache.fused_fills - f0, "framework_convolutions": cache.framework_fills - w0},
                       "env_steps_per_s_at_this_cadence": round(100.0 * real / n / (window_ms * 1e-3), 1),
                       "what": "PPO's training cadence (baseline/PPO/train_PPO.py:18,80: an update every 100 env steps): all weights written in place, then 100 steps; the first "
                               "re-folds the fused kernels' weight snapshots and refills the per-(region, net) cache of net vectors (xr_batch_net_vectors). DQN updates after "
                               "EVERY step (baseline/DQN/train_DQN.py:127): there the refill is paid per step and the cache buys nothing"}
        except Exception as exc:
            cadence = {"error": str(exc)}
    return {"metric": f"env-steps/sec, {args.agent.upper()} counterpart attached (batched regions), ispd18_test1-sized regions",
            "value": round(real / dt, 1), "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "net_cache_refill_ms": None if cadence is None else cadence.get("refill_ms"), "training_cadence": cadence,
            "net_vectors_by": {"fused_kernel": cache.fused_fills, "framework_convolutions": cache.framework_fills},
            "ms_per_step": round(dt / n * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 distances / i16 state / fp32 observation + fp32 policy (towers: split-bf16 x3 matrix products, fp32 accumulate; XR_TOWER_FP32=1 for fp32 ones)", "data": "synthetic",
            "config": {"workload": (f"{B} env slots over the design-derived ispd18_test1 region pack ({len(grouped.shapes)} grid shapes, fused tower for "
                                    f"{sum(t is not None for t in grouped.towers)} of them; XR-Maze {'v2 (the reference TCL knob values, this build semantics)' if args.maze_v2 else 'v1'}), full maze route per step, "
                                    if mixed else f"BASELINE config 3/4 shape: {B} ispd18_test1-sized regions, full maze route per step, ")
                                   +
                                   f"{args.agent.upper()} counterpart (random-init weights of the reference architecture, eval mode) choosing every action; "
                                   + ("full fp32 observation (xr_batch_step_observe)" if full else
                                      "compact-consumer mode (xr_batch_step_compact: planes 0..1 per step; net planes once per (region, net) via xr_batch_net_planes + NetVectorCache)"),
                       "envs_per_gpu": B, "global_envs": B, "parallelism": "env-shard x1", "mean_nets_left": round(kfloat, 2)},
            "env_share_of_step_time": round(env_ms / max(env_ms + agent_ms, 1e-9), 4),
            "obstacle_tower": ("fused HIP kernel per grid shape (agents.GroupedFusedPolicy)" if mixed else
                               "framework convolutions" if tower is None or not tower.supported else "fused HIP kernel (xr_agent_obstacle_tower)"),
            "agent_ms_per_step": round(agent_ms, 4), "env_ms_per_step": round(env_ms, 4),
            "net_grids_through_the_tower": cache.computed, "tower_roofline": tower_roof, "net_tower_roofline": net_roof,
            "roofline": {"kernel": "xr_route_kernel (+ planes 0..1)" if not full else "xr_step_queue_kernel", "bound": "hbm",
                         "achieved": round(env_bytes / (env_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(env_bytes / (env_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "traffic": None,
                         "avg_launch_ms": round(env_ms, 4), "algorithmic_bytes_per_launch": int(env_bytes)}}


def agent_sharded(args, regions, dev, world, rank, first_env, B, strong, learner_regions=None, emit=True):
    """BASELINE config 4 as stated — "4096 regions sharded 8 x MI355X, PPO baseline, RCCL env gather" — as ONE self-certifying line
    (`--gpus N [--global-envs 4096] --agent ppo`; also N = 1 and `--agent dqn`).  The reference's caller loop is
    `action = ppo_agent.select_action(state); state, done, ... = game.step(action)` (baseline/PPO/train_PPO.py:96-99; the sampling:
    baseline/PPO/PPO.py:205-217).  Two placements of the policy:

      default     every rank evaluates the counterpart on ITS shard (weights replicated from a fixed seed, fused tower + actor head, net
                  vectors of its own regions cached), steps its slice in compact-consumer mode, and the 48-byte result records are
                  all-gathered (overlapped with the next step) — north_star's "RCCL only for the batched-env gather".
      --learner   SURVEY §8e's central learner: every rank packs the compact state of its envs (xr_batch_pack_state: one bit per node +
                  the legal bitmask), ONE gather to rank 0 carries it, rank 0 expands it to head rows (xr_batch_expand_state), evaluates the
                  policy for ALL envs and broadcasts the actions (i32); the ranks step route-only.

    PPO samples with counter-based uniforms of (seed, step, GLOBAL env id, net rank) (agents.counter_uniform), so both placements and
    every sharding choose the same action for the same env: `actions_sha` / `hash_chains_sha` of an N-rank line equal the one-rank line's.
    Self-certification as in the env-only N > 1 line: `gather_verified`, `ranks_seen`, `parity.all_ranks_ok` (oracle replay of the actions
    the policy actually chose + the head planes the last step wrote)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from xroute_env_amd import agents
    from xroute_env_amd.batch import RegionBatch
    from xroute_env_amd.dist import RECORD_BYTES, gather_rows, verify_gather
    learner = bool(args.learner)
    multi = world > 1 or os.environ.get("XR_FORCE_COLLECTIVES") == "1"          # the collective code paths run (--force-collectives: with one rank)
    cycled = bool(args.region_pack) or args.regions > 0            # global env g plays region g % len(regions); else one region per env slot
    Bg = args.global_envs if strong else B * world
    mixed = len({tuple(int(v) for v in r.dims) for r in regions}) > 1
    v2 = dict(V2_KNOBS, guide_margin=1) if (args.maze_v2 and args.region_pack) else {}
    torch.manual_seed(0)                                            # the SAME random-init weights on every rank
    model = (agents.RepActor() if args.agent == "dqn" else agents.ActorCritic(64)).to(dev).eval()
    batch = RegionBatch(regions, n_envs=B, device=dev, auto_reset=True, router=args.router, dial_mult=args.dial_mult,
                        launch_order=args.launch_order, max_route_count=1 << 30, **v2)
    if cycled:
        batch.assign([(first_env + e) % len(regions) for e in range(B)])
        slot_regions = [regions[(first_env + e) % len(regions)] for e in range(B)]
    else:
        slot_regions = list(regions[:B])
    batch.reset()
    acts = torch.empty(B, dtype=torch.int32, device=dev)
    env_ids = torch.arange(first_env, first_env + B, dtype=torch.int64, device=dev)

    # ---- the policy's side: who evaluates it, on which rows ---------------------------------------------------------------------
    # default: this rank, on its own head buffer.  --learner: rank 0, on the rows expanded from everybody's packed state; its region
    # table must hold every region of the job (the pack / the --regions set as they are; one region per env: all Bg of them)
    pol_batch, pol_regions, region_base = batch, regions, 0
    if learner and not cycled:
        region_base = first_env
        if rank == 0 and multi:
            pol_regions = learner_regions                # (every region of the job, generated by main() before the GPU was touched)
            pol_batch = RegionBatch(pol_regions, n_envs=1, device=dev)
    evaluates = (not learner) or rank == 0
    n_rows = Bg if learner else B
    xch = None
    if learner:
        from xroute_env_amd.dist import CompactStateExchange
        xch = CompactStateExchange(batch, Bg, first_env, region_base=region_base, learner_batch=pol_batch)
    rb = xch.row_bytes if learner else 0
    head = nl = reg = None
    grouped = cache = tower = head_k = None
    if evaluates:
        head = torch.empty((n_rows, 2 * pol_batch.n_max), dtype=torch.float32, device=dev)
        nl = torch.empty(n_rows, dtype=torch.int32, device=dev)
        reg = torch.empty(n_rows, dtype=torch.int32, device=dev)
        if mixed:
            grouped = agents.GroupedFusedPolicy(model, pol_batch, dev)
            cache = grouped.cache
        else:
            dims = pol_regions[0].dims
            cache = agents.NetVectorCache(len(pol_regions), pol_batch.k_max, dev)
            tower = agents.FusedObstacleTower(model.representation_network, (dims[2], dims[1], dims[0]), dev)
            head_k = agents.FusedActorHead(model.actor, dev)
            cache.prefill(model.representation_network, [r.n_nets for r in pol_regions], pol_batch.net_planes, dims)
    row_ids = torch.arange(Bg, dtype=torch.int64, device=dev) if learner else env_ids
    acts_all = torch.zeros(Bg, dtype=torch.int32, device=dev) if learner else None

    def policy(i):
        """actions of the rows this rank evaluates (head / nl / reg hold their current state)"""
        uni = agents.counter_uniform(args.seed, i, row_ids) if args.agent == "ppo" else None
        if mixed:
            return grouped.actions(head, nl, reg, sample=(args.agent == "ppo"), uniform=uni)
        kw = dict(cache=cache, region=reg, ob_tower=tower, actor_head=head_k, planes_fn=pol_batch.net_planes)
        if args.agent == "dqn":
            return agents.dqn_actions(model, head, nl, dims, **kw)
        return agents.ppo_actions(model, head, nl, dims, uniform=uni, **kw)[0]

    if not learner:
        _head_of(batch, head)

    # ---- stagger: every env to a uniform phase of its episode (untimed, random actions, route-only).  The random net-order policy is keyed
    # by the GLOBAL env id (dist.random_legal_policy), so the state every env starts the measured steps from does not depend on the sharding;
    # the pre-roll's actions are part of the log the oracle replays
    n_pre = 0
    pre_log = None
    if not args.no_stagger:
        from xroute_env_amd.dist import random_legal_policy, unpack_records
        off = stagger_offsets(batch.fetch("nlegal").cpu().numpy(), first_env)
        off_d = torch.from_numpy(off).to(dev)
        mx = torch.tensor([int(off.max()) if B else 0], dtype=torch.int64, device=dev)
        if multi:
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        n_pre = int(mx.item())
        pre_log = torch.zeros((max(n_pre, 1), B), dtype=torch.int32, device=dev)
        zero = torch.zeros_like(acts)
        for i in range(n_pre):
            a_ = random_legal_policy(unpack_records(batch.fetch("record")), batch.fetch("legal"), args.seed ^ 0xA6E7 ^ i, env_ids=env_ids)
            torch.where(off_d > i, a_, zero, out=acts)
            pre_log[i].copy_(acts)
            batch.step(acts)
        if not learner:
            _head_of(batch, head)

    nsteps_total = max(args.warmup, 2) + args.steps
    acts_log = torch.zeros((nsteps_total, B), dtype=torch.int32, device=dev)
    rec_pairs = [(torch.empty((B, RECORD_BYTES), dtype=torch.uint8, device=dev),
                  torch.empty((Bg, RECORD_BYTES), dtype=torch.uint8, device=dev) if multi and Bg % world == 0 else None) for _ in range(2)]
    pending = [None, None]
    last_slot = [0]
    split = {"pack_gather": 0.0, "expand": 0.0, "policy": 0.0, "broadcast": 0.0, "env": 0.0}

    def one_step(i, ev=None):
        mark = (lambda j: ev[j].record()) if ev else (lambda j: None)
        mark(0)
        if learner:
            rows = xch.gather()                                             # pack + the compact-state gather: Bg x row_bytes per step
            mark(1)
            if rank == 0:
                xch.expand(rows, head, nl, reg)
                mark(2)
                acts_all.copy_(policy(i))
            else:
                mark(2)
            mark(3)
            if multi:
                dist.broadcast(acts_all, src=0)                             # the actions travel back as one i32[Bg]
            acts.copy_(acts_all[first_env:first_env + B])
        else:
            mark(1); mark(2)
            batch.fetch("nlegal", nl)
            batch.fetch("region", reg)
            acts.copy_(policy(i))
            mark(3)
        mark(4)
        acts_log[i].copy_(acts)
        if learner:
            batch.step(acts)
        else:
            batch.step_compact(acts, head)
        mark(5)
        slot = i & 1
        if pending[slot] is not None:
            pending[slot].wait()
            pending[slot] = None
        loc, glob = rec_pairs[slot]
        batch.fetch("record", loc)
        if multi and glob is not None:
            pending[slot] = dist.all_gather_into_tensor(glob, loc, async_op=True)          # the batched-env gather, overlapped with the next step
        last_slot[0] = slot

    n_w = max(args.warmup, 2)                       # (MIOpen kernel selection of the framework path, first-touch of the buffers)
    for i in range(n_w):
        one_step(i)
    for j in range(2):
        if pending[j] is not None:
            pending[j].wait(); pending[j] = None
    n = max(args.steps, 1)
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in range(n)]
    if multi:
        dist.barrier()
    torch.cuda.synchronize(dev)
    steps0 = batch.total_steps()
    t0 = time.perf_counter()
    for i in range(n):
        one_step(n_w + i, events[i])
    for j in range(2):
        if pending[j] is not None:
            pending[j].wait(); pending[j] = None
    torch.cuda.synchronize(dev)
    if multi:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    real_steps = batch.total_steps() - steps0
    for k_, (a_, b_) in zip(split, ((0, 1), (1, 2), (2, 3), (3, 4), (4, 5))):
        split[k_] = sum(e[a_].elapsed_time(e[b_]) for e in events) / n
    gpu_hash = batch.fetch("hash").cpu().numpy().view("uint64")
    gpu_cum = batch.fetch("cum").cpu().numpy()
    nl_now = batch.fetch("nlegal")
    n_par = min(B, 64 if (not multi) else 32)
    obs_sha = None if learner else obs_sample_sha(head, nl_now, slot_regions[:n_par], n_check=n_par, head_only=True)

    # ---- certification: the gather, every rank's oracle replay, and the sharding-invariant digests ---------------------------------
    certify = None
    if multi:
        sent, gathered = rec_pairs[last_slot[0]]
        if gathered is None:
            gathered = gather_rows(sent)
        if os.environ.get("XR_BENCH_TEST_CORRUPT_GATHER") == "1" and rank == 0:
            gathered[-1, 0] ^= 0xFF
        certify = verify_gather(sent, gathered, first_env)
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    sm = torch.tensor([float(real_steps), split["policy"] + split["expand"], split["env"]], dtype=torch.float64, device=dev)
    if multi:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        tsum = sm.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dist.all_reduce(sm, op=dist.ReduceOp.MAX)
        total_real = float(tsum[0].item())
    else:
        total_real = float(real_steps)
    elapsed_max = float(t.item())
    # actions of every env at every step and every env's final hash chain, in GLOBAL env order (tiny gathers, outside the timed region)
    al = acts_log.t().contiguous()                                   # [B, steps]
    hs = torch.from_numpy(gpu_hash.view("int64").copy()).to(dev).reshape(B, 1)
    if multi:
        al, hs = gather_rows(al), gather_rows(hs)
    actions_sha = hashlib.sha256(al.cpu().numpy().tobytes()).hexdigest()
    chains_sha = hashlib.sha256(hs.cpu().numpy().tobytes()).hexdigest()
    try:
        full_log = torch.cat([pre_log[:n_pre], acts_log]) if n_pre else acts_log
        parity = parity_check(slot_regions, list(range(full_log.shape[0])), None, gpu_hash, gpu_cum, n_check=n_par, obs_sha=obs_sha,
                              actions_log=full_log.cpu().numpy(), v2=v2 or None, head_only=True)
        parity["what"] = ("CPU oracle replay of the actions the POLICY chose (stagger pre-roll + warm-up + timed steps) on the first envs of the rank: "
                          "hash chains, cumulative metrics" + ("" if learner else ", sha256 of planes 0..1 the last compact step wrote vs the oracle's build_3Dgrid restatement"))
    except Exception as ex:
        parity = {"error": str(ex), "ok": False}
    if multi:
        flag = torch.tensor([1 if parity.get("ok") else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        parity = dict(parity, all_ranks_ok=bool(flag.item() == 1), envs_per_rank=n_par)
    good = (not multi) or (certify["gather_verified"] and certify["ranks_seen"] == args.gpus and parity.get("all_ranks_ok"))
    good = bool(good and parity.get("ok"))
    if rank == 0:
        agent_ms = split["pack_gather"] + split["expand"] + split["policy"] + split["broadcast"]
        N_mean = float(sum(r.n_nodes for r in slot_regions)) / max(B, 1)
        out = {"metric": f"env-steps/sec, {args.agent.upper()} counterpart attached (batched regions), ispd18_test1-sized regions",
               "value": round(total_real / elapsed_max, 1), "unit": "env-steps/s", "n_gpus": world if good else None, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(elapsed_max / n * 1e3, 4), "higher_is_better": True,
               "scaling": "strong" if strong else "weak", "vs_baseline": None,
               "dtype": "u32 distances / i16 state / fp32 observation + fp32 policy (towers: split-bf16 x3 matrix products, fp32 accumulate; XR_TOWER_FP32=1 for fp32 ones)", "data": "synthetic",
               "config": {"workload": (f"BASELINE config 4 shape: a {Bg}-env batch, {B} per GPU, "
                                       + (f"over the design-derived ispd18_test1 region pack {os.path.basename(args.region_pack)} (XR-Maze {'v2: the reference TCL knob values, this build semantics' if v2 else 'v1'}), "
                                          if args.region_pack else "of ispd18_test1-sized regions (24x40x9, K~U[4,36]), ")
                                       + f"full maze route per step, {args.agent.upper()} counterpart (random-init weights of the reference architecture, replicated from one seed, eval mode"
                                       + (", actions sampled with counter-based uniforms of (seed, step, global env, net rank)" if args.agent == "ppo" else "") + ") choosing every action; "
                                       + ("policy on rank 0 for ALL envs from the gathered compact state (xr_batch_pack_state -> gather to rank 0 -> xr_batch_expand_state), i32 action broadcast, ranks step route-only"
                                          if learner else
                                          "policy evaluated by every rank on ITS shard, compact-consumer step (xr_batch_step_compact: planes 0..1 per step; net vectors cached per (region, net))")
                                       + ("; RCCL all_gather of the 48-byte per-env records overlapped with the next step" if multi else "")
                                       + ("" if args.no_stagger else "; episodes staggered before timing")),
                          "envs_per_gpu": B, "global_envs": Bg, "parallelism": f"env-shard x{world}", "policy_placement": "rank 0 (central learner)" if learner else "every rank (its shard)",
                          "mean_nets_left": round(float(nl_now.double().mean().item()), 2), "source_sha": source_sha(),
                          **({"forced_collectives": "one rank taking every N > 1 branch on the real backend (--force-collectives): a functional run of the multi-GPU code path, not a scaling point"}
                             if (multi and world == 1) else {})},
               "agent_ms_per_step": round(agent_ms, 4), "env_ms_per_step": round(split["env"], 4),
               "env_share_of_step_time": round(split["env"] / max(split["env"] + agent_ms, 1e-9), 4),
               "step_split_ms_rank0": {k_: round(v_, 4) for k_, v_ in split.items()},
               "slowest_rank_ms": {"policy": round(float(sm[1].item()), 4), "env": round(float(sm[2].item()), 4)} if multi else None,
               "actions_sha": actions_sha, "hash_chains_sha": chains_sha, "parity": parity,
               "roofline": {"kernel": "xr_route_kernel" + ("" if learner else " (+ planes 0..1)"), "bound": "hbm",
                            "achieved": round(B * ((4.0 if learner else 12.0) * N_mean) / (max(split["env"], 1e-9) * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": round(B * ((4.0 if learner else 12.0) * N_mean) / (max(split["env"], 1e-9) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "traffic": None,
                            "avg_launch_ms": round(split["env"], 4),
                            "note": "the env step of this line is the LDS-resident router (latency-bound): its HBM bytes are the state load" + ("" if learner else " + the two head planes")}}
        if learner:
            out["compact_state"] = {"row_bytes": rb, "bytes_gathered_per_step": Bg * rb, "fp32_head_bytes_per_step": int(Bg * 8 * N_mean),
                                    "collective": "gather to rank 0 (every other rank sends its rows once over its own link; nobody else receives them)",
                                    "bytes_per_link_per_step": xch.bytes_per_link,
                                    "all_gather_bytes_per_link_per_step": int(Bg * rb * (world - 1) / max(world, 1)),
                                    "ratio_to_fp32_planes": round(rb / (8.0 * N_mean), 5),
                                    "what": "per env: region, nets left, legal-net bitmask, one occupancy bit per node (plane 0); plane 1 is the bitmask — "
                                            "what SURVEY §8e's central learner gathers instead of observations"}
        if certify is not None:
            out["gather_verified"], out["ranks_seen"], out["gathered_rows"] = certify["gather_verified"], certify["ranks_seen"], certify["rows"]
        if not good:
            out["error"] = "self-certification failed: " + json.dumps({"certify": certify, "parity_ok": parity.get("ok"), "all_ranks_ok": parity.get("all_ranks_ok")})
            print(out["error"], file=sys.stderr)
        if not emit:
            return out                     # (called from the default line's `extras`: the caller embeds it)
        print(json.dumps(out), flush=True)
    return 0 if good else 3


def _head_of(batch, head):
    """planes 0..1 of the current state of every env into a head buffer (reset-time fill of the compact mode)."""
    import torch
    full = batch.alloc_observation()
    batch.observation(full)
    head.copy_(full[:, :head.shape[1]])
    del full
    return head


V2_KNOBS = dict(guide_cost=800, maze_end_iter=3)      # + guide_margin: 2 with the default guides (bounding box of the access points), 1 with the design's rectangles


def v2_leg(args, regions, dev, first_env, pack=None):
    """The simulator knobs the reference actually runs (ispd/ispd18_test1/run-net-ordering-training.tcl:3: `-maze_end_iter 3 -drc_cost 8
    -follow_guide 1 -ripup_mode 1`) on the driver-visible line: the same envs, route-only, with XR-Maze v2's rip-up-and-reroute
    (maze_end_iter 3) and guide cost switched on, with an oracle replay of the leg's own actions (`parity`).  Build-defined semantics
    (DESIGN.md §3.1), parity unpinned against TritonRoute like all of a11."""
    import torch
    from xroute_env_amd.batch import RegionBatch
    if pack is not None:      # the design-derived regions with the guide rectangles of ispd18_test1.input.guide (Region.guide_box)
        B = args.pack_envs
        v2 = dict(V2_KNOBS, guide_margin=1)
        b = RegionBatch(pack, n_envs=B, device=dev, auto_reset=True, router=args.router, dial_mult=args.dial_mult,
                        launch_order=args.launch_order, max_route_count=1 << 30, **v2)
        regions = [pack[e % len(pack)] for e in range(B)]
    else:
        B = len(regions)
        v2 = dict(V2_KNOBS, guide_margin=2)
        b = RegionBatch(regions, n_envs=B, device=dev, auto_reset=True, router=args.router, dial_mult=args.dial_mult,
                        launch_order=args.launch_order, **v2)
    b.reset(rotate=True)
    acts = torch.empty(B, dtype=torch.int32, device=dev)
    off = stagger_offsets(b.fetch("nlegal").cpu().numpy(), first_env)
    off_d = torch.from_numpy(off).to(dev)
    zero = torch.zeros_like(acts)
    pre_seeds = [args.seed ^ 0x7C1 ^ i for i in range(int(off.max()) if B else 0)]
    for i, sd in enumerate(pre_seeds):
        b.random_actions(sd, acts)
        torch.where(off_d > i, acts, zero, out=acts)
        b.step(acts)
    n_w, n_t = 3, max(args.steps, 5)
    seeds = [args.seed + 400000 + i for i in range(n_w)] + [args.seed + 400100 + i for i in range(n_t)]
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_t)]
    for i in range(n_w):
        b.random_actions(seeds[i], acts)
        b.step(acts)
    s0 = b.total_steps()
    vio_d = torch.zeros((), dtype=torch.float64, device=dev)         # (summed on the device: no host round trip inside the timed loop)
    for i, (e0, e1) in enumerate(evs):
        b.random_actions(seeds[n_w + i], acts)
        e0.record()
        b.step(acts)
        e1.record()
        vio_d += b.fetch("delta")[:, 0].double().sum()
    torch.cuda.synchronize(dev)
    vio = float(vio_d.item())
    ms = sum(a.elapsed_time(bb) for a, bb in evs) / n_t
    real = (b.total_steps() - s0) / n_t
    nbytes = float(sum(4.0 * r.n_nodes for r in regions))
    gpu_hash = b.fetch("hash").cpu().numpy().view("uint64")
    gpu_cum = b.fetch("cum").cpu().numpy()
    b.close()
    if pack is not None:
        ent = kernel_entry("xr_route_kernel (XR-Maze v2 + the design's guide rectangles, ispd18_test1 region pack)", ms, nbytes, real, "lds-latency",
                           f"route-only step on {B} env slots over the {len(pack)} design-derived regions with maze_end_iter 3 and guide cost 800 / "
                           "margin 1 where a net's guide = the rectangles ispd18_test1.input.guide lists for it, clipped to the region (1-8 boxes "
                           "per net, xr_batch_load_guides) — the closest this build gets to `-follow_guide 1 -maze_end_iter 3`; build-defined "
                           "semantics, parity unpinned against TritonRoute; env_steps_per_s is the figure of merit")
        ent["data"] = "ispd18_test1 (design-derived regions + guides)"
    else:
        ent = kernel_entry("xr_route_kernel (XR-Maze v2: the reference's TCL knobs)", ms, nbytes, real, "lds-latency",
                           f"route-only step on the same {B} envs with maze_end_iter 3 (rip-up and reroute, penalty doubled per attempt) and guide "
                           "cost 800 / margin 2 — the knobs of ispd/ispd18_test1/run-net-ordering-training.tcl:3 that XR-Maze v1 leaves out; "
                           "build-defined semantics, parity unpinned against TritonRoute; env_steps_per_s is the figure of merit")
    ent["violations_per_env_step"] = vio / max(real * n_t, 1.0)
    try:
        ent["parity"] = parity_check(regions, seeds, (off, pre_seeds), gpu_hash, gpu_cum, v2=v2)
    except Exception as ex:
        ent["parity"] = {"error": str(ex), "ok": False}
    return ent


def pack_leg(args, pack, dev, v2=None):
    """The REAL ispd18_test1 regions on the record: 4096 env slots over the 256 regions `xroute_env_amd.lefdef` extracts from the
    reference's own ispd/ispd18_test1/ispd18_test1.input.{lef,def,guide} (per-GCell routeBox + 2000 DBU ring: 22-25 x 27-34 x 9 tracks, K up to 28),
    full step in the default queue form, with its own oracle replay (hash chains, cumulative metrics and the observation bytes of
    32 slots spread over K).  Their N is rarely a multiple of 4, so channel planes are not 16-byte aligned: the unit writer is
    xr_unit_stream.  Slots keep their region (max_route_count = 2^30: the replay needs no rotation bookkeeping; rotation itself is
    covered by the tests).  `v2`: the same full step with the reference's own simulator configuration (run-net-ordering-training.tcl:3:
    maze_end_iter 3, follow_guide with the design's guide rectangles) — the step a user of the reference actually takes."""
    import numpy as np
    import torch
    from xroute_env_amd.batch import RegionBatch
    Bp = args.pack_envs
    slot_regions = [pack[e % len(pack)] for e in range(Bp)]
    b = RegionBatch(pack, n_envs=Bp, device=dev, auto_reset=True, router=args.router, dial_mult=args.dial_mult,
                    launch_order=args.launch_order, max_route_count=1 << 30, **(v2 or {}))
    b.reset(rotate=True)
    acts = torch.empty(Bp, dtype=torch.int32, device=dev)
    obs = b.alloc_observation()
    n_nodes = torch.tensor([r.n_nodes for r in slot_regions], dtype=torch.float64, device=dev)
    nl0 = b.fetch("nlegal").cpu().numpy()
    off = stagger_offsets(nl0, 0)
    off_d = torch.from_numpy(off).to(dev)
    pre_seeds = [args.seed ^ 0x9ACC ^ i for i in range(int(off.max()))]
    zero = torch.zeros_like(acts)
    for i, sd in enumerate(pre_seeds):
        b.random_actions(sd, acts)
        torch.where(off_d > i, acts, zero, out=acts)
        b.step(acts)
    n_w, n_t = 3, max(args.steps, 5)
    seeds = [args.seed + 300000 + i for i in range(n_w + n_t)]
    klog = torch.zeros((n_t, Bp), dtype=torch.int32, device=dev)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_t)]
    for i in range(n_w):
        b.random_actions(seeds[i], acts)
        b.step(acts, obs)
    s0 = b.total_steps()
    vio_d = torch.zeros((), dtype=torch.float64, device=dev)         # (summed on the device: no host round trip inside the timed loop)
    for i, (e0, e1) in enumerate(evs):
        b.random_actions(seeds[n_w + i], acts)
        e0.record()
        b.step(acts, obs)
        e1.record()
        b.fetch("nlegal", klog[i])
        if v2:
            vio_d += b.fetch("delta")[:, 0].double().sum()
    torch.cuda.synchronize(dev)
    vio = float(vio_d.item())
    ms = sum(a.elapsed_time(bb) for a, bb in evs) / n_t
    real = (b.total_steps() - s0) / n_t
    info = b.observe_info()
    k_after = klog.to(torch.float64)
    nbytes = float((4.0 * n_nodes).sum().item()) + float((4.0 * (2.0 + 7.0 * k_after) * n_nodes[None, :]).sum().item()) / n_t
    name = ("xr_step_queue_kernel (design-derived ispd18_test1 region pack, XR-Maze v2 + the design's guide rectangles: the reference's TCL knob VALUES under this build's semantics — rip-up == 4x penalty, DESIGN 3.1)"
            if v2 else "xr_step_queue_kernel (design-derived ispd18_test1 region pack)")
    ent = kernel_entry(name, ms, nbytes, real,
                       "hbm-write (routing phase: lds-latency)",
                       f"{Bp} env slots over the {len(pack)} regions extracted from the reference's ispd18_test1.input.lef/def/guide "
                       "(tests/golden/ispd18_test1_regions.npz, xroute_env_amd/lefdef.py): full step (random net-order action + route + "
                       "fp32 observation of every env), queue form, stationary nets-left distribution; planes are not 16-byte aligned "
                       "(N % 4 != 0 for 98 % of the regions): unit writer xr_unit_stream; bytes = state load + 4·N·(2+7K) per slot"
                       + ("; router = XR-Maze v2 with the knobs of ispd/ispd18_test1/run-net-ordering-training.tcl:3 (maze_end_iter 3, guide cost 800 "
                          "over the rectangles ispd18_test1.input.guide lists per net, margin 1)" if v2 else ""))
    ent["data"] = "ispd18_test1 (design-derived regions" + (" + guides)" if v2 else ")")
    ent["form"] = info
    ent["mean_nets_left"] = float(k_after.mean().item())
    ent["mean_nodes"] = float(n_nodes.mean().item())
    if v2:
        ent["violations_per_env_step"] = vio / max(real * n_t, 1.0)
    gpu_hash = b.fetch("hash").cpu().numpy().view("uint64")
    gpu_cum = b.fetch("cum").cpu().numpy()
    obs_sha = obs_sample_sha(obs, klog[n_t - 1], slot_regions)
    b.close()
    del obs
    try:
        ent["parity"] = parity_check(slot_regions, seeds, (off, pre_seeds), gpu_hash, gpu_cum, obs_sha=obs_sha, v2=v2)
    except Exception as ex:
        ent["parity"] = {"error": str(ex), "ok": False}
    return ent


def config5_leg(args, c5_regions, dev):
    """BASELINE config 5 (synthetic 256x256x12 dense-congestion regions), route-only step with the compact state (the fp32
    observation of one such env would be 711 MB): its own kernel entry with SURVEY §8(d)'s algorithmic bytes."""
    import torch
    from xroute_env_amd.batch import RegionBatch
    Bc = args.c5_envs
    b5 = RegionBatch(c5_regions, n_envs=Bc, device=dev, auto_reset=True, router=args.router, dial_mult=args.dial_mult,
                     launch_order=args.launch_order)
    b5.reset()
    a5 = torch.empty(Bc, dtype=torch.int32, device=dev)
    n_t = 5
    # (round 5) like every other leg: episodes staggered to the stationary nets-left distribution first (rounds 1-4 timed the FIRST steps of 1024
    # fresh episodes: no congestion yet, no measured launch order yet); the first-steps figure is kept as `first_steps_ms`
    first_ms = None
    if not args.no_stagger:
        b0 = RegionBatch(c5_regions, n_envs=Bc, device=dev, auto_reset=True, router=args.router, dial_mult=args.dial_mult, launch_order=args.launch_order)
        b0.reset()
        for i in range(2):
            b0.random_actions(555 + i, a5)
            b0.step(a5)
        ev0 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_t)]
        for i, (e0, e1) in enumerate(ev0):
            b0.random_actions(600 + i, a5)
            e0.record(); b0.step(a5); e1.record()
        torch.cuda.synchronize(dev)
        first_ms = sum(x.elapsed_time(y) for x, y in ev0) / n_t
        b0.close()
    c5_stagger = None
    if not args.no_stagger:
        off5 = stagger_offsets(b5.fetch("nlegal").cpu().numpy(), 0)
        off5_d = torch.from_numpy(off5).to(dev)
        pre5 = [args.seed ^ 0xC5C5 ^ i for i in range(int(off5.max()) if Bc else 0)]
        zero5 = torch.zeros_like(a5)
        for i, sd in enumerate(pre5):
            b5.random_actions(sd, a5)
            torch.where(off5_d > i, a5, zero5, out=a5)
            b5.step(a5)
        c5_stagger = (off5, pre5)
    for i in range(2):
        b5.random_actions(555 + i, a5)
        b5.step(a5)
    c5_seeds = [555, 556] + [600 + i for i in range(n_t)]
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_t)]
    s0 = b5.total_steps()
    sweeps = 0.0
    plen = 0.0
    touched = 0.0
    for i, (e0, e1) in enumerate(evs):
        b5.random_actions(600 + i, a5)
        e0.record()
        b5.step(a5)
        e1.record()
        sweeps += float(b5.fetch("sweeps").double().sum().item())
        plen += float(b5.fetch("path_len").double().sum().item())
        touched += float(b5.fetch("touched").double().sum().item())
    torch.cuda.synchronize(dev)
    ms = sum(a.elapsed_time(bb) for a, bb in evs) / n_t
    real = (b5.total_steps() - s0) / n_t
    # how busy the chip is inside such a launch: cycles of every env's route (XR_FETCH_PHASES slot 7) of the LAST timed launch — a launch lasts
    # as long as its longest route, so mean / max is the fraction of the workgroup slots doing anything
    util = None
    try:
        cyc = b5.fetch("phases").reshape(Bc, 8)[:, 7].double()
        cyc = cyc[cyc > 0]
        if cyc.numel():
            util = {"mean_route_cycles": round(float(cyc.mean().item())), "p50_route_cycles": round(float(cyc.median().item())),
                    "max_route_cycles": round(float(cyc.max().item())), "utilisation": round(float((cyc.mean() / cyc.max()).item()), 4)}
    except Exception:
        util = None
    N = float(c5_regions[0].n_nodes)
    # SURVEY §8(d) prices an HBM-resident maze route as FULL sweeps: 4·N state load + 9·N·S + 10·L per env-step.  The frontier
    # router does no sweep at all: it creates the field word of a node when a neighbour first relaxes it and resets exactly
    # those words afterwards.  Its algorithmic bytes per touched node: node_net + owner read (4), field word created, read for
    # classification, lowered by an atomic (read + write) and reset (4 x 4 + 4) = 24 B; per path node 10 B as in §8(d).
    formula_8d = (4.0 * N * Bc * n_t + 9.0 * N * sweeps + 10.0 * plen) / n_t
    nbytes = (24.0 * touched + 10.0 * plen) / n_t if touched > 0 else formula_8d
    ent = kernel_entry("xr_route_kernel (BASELINE config 5: 256x256x12)", ms, nbytes, real, "l2-latency",
                       f"{Bc} env slots over {len(c5_regions)} distinct regions, K = 32, route-only (compact state). BASELINE calls this config an "
                       "'HBM-roofline stress'; as built it is an L2-ATOMIC LATENCY CHAIN: a route is ~10^3 dependent passes over a frontier held in L2, the "
                       "launch ends with its longest route (`launch_utilisation`), HBM sees 0.6 % of its peak. Bound by the latency of "
                       "dependent L2 atomics, not by bandwidth: bytes = 24 B per node the route touched + 10 B per path node (the frontier "
                       "router's algorithmic bytes); `bytes_8d_full_sweep_formula` = what SURVEY §8(d)'s full-sweep form (4·N + 9·N·S + 10·L, "
                       "S = rounds) would move for the same routes; env_steps_per_s is the figure of merit")
    ent["bytes_8d_full_sweep_formula"] = formula_8d
    # the roofline that bounds this kernel: L2 atomics.  Ceilings from tools/micro/atomic_rate.hip on this pool; the per-launch
    # atomic count from the TCC counters of a rocprofv3 --pmc pass — quoted only for the same build and batch size
    ent["l2_atomic_roofline"] = {"bound": "l2-atomic", "peak": 18.0, "peak_l2_resident": 27.0, "unit": "G atomics/s", "achieved": None, "frac": None,
                                 "dependent_atomic_latency_ns": [309, 436],
                                 "note": "peak: sustained agent-scope returning atomics on random words over a 4 GiB footprint (this leg's scratch is 13 GB; <= 256 MiB: 27 G/s); a route is a "
                                         "chain of DEPENDENT atomics (309-436 ns each), so the launch is as long as its longest chain, not atomics / peak"}
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", "config5_atomics.json")))
        if pj.get("source_sha") == source_sha() and pj.get("envs") == Bc and pj.get("tcc_atomic_per_launch"):
            ach = pj["tcc_atomic_per_launch"] / (ms * 1e-3) / 1e9
            ent["l2_atomic_roofline"].update(achieved=round(ach, 3), frac=round(ach / 18.0, 4), atomics_per_launch=pj["tcc_atomic_per_launch"],
                                             l2_requests_per_launch=pj.get("tcc_req_per_launch"))
    except Exception:
        pass
    if first_ms is not None:
        ent["first_steps_ms"] = round(first_ms, 4)
        ent["note"] += "; stationary nets-left distribution (episodes staggered first, round 5); `first_steps_ms`: the same launches on 1024 FRESH episodes, what rounds 1-4 reported here"
    if util is not None:
        ent["launch_utilisation"] = dict(util, what="cycles of every route of the last timed launch; utilisation = mean / max: the launch is as long as its longest "
                                                    "route, the workgroup slots of the short ones idle (closed in DESIGN.md §9: four rounds of bit-exact A/Bs inside the pass)")
    ent["mean_rounds"] = sweeps / (n_t * Bc)
    ent["mean_path_nodes"] = plen / (n_t * Bc)
    ent["mean_touched_nodes"] = touched / (n_t * Bc)
    ent["envs"] = Bc
    gpu_hash = b5.fetch("hash").cpu().numpy().view("uint64")
    gpu_cum = b5.fetch("cum").cpu().numpy()
    b5.close()
    try:        # the same kernel on 4x the slots (BASELINE's 1024 per GPU is the chain-bound shape: few routes per CU, the launch = its longest route)
        if not args.no_stagger and Bc >= 256 and 4 * Bc <= 4096:
            B4 = 4 * Bc
            b4 = RegionBatch(c5_regions, n_envs=B4, device=dev, auto_reset=True, router=args.router, dial_mult=args.dial_mult, launch_order=args.launch_order)
            b4.reset()
            a4 = torch.empty(B4, dtype=torch.int32, device=dev)
            off4 = torch.from_numpy(stagger_offsets(b4.fetch("nlegal").cpu().numpy(), 0)).to(dev)
            z4 = torch.zeros_like(a4)
            for i in range(int(off4.max().item())):
                b4.random_actions(args.seed ^ 0xC5C5 ^ i, a4)
                torch.where(off4 > i, a4, z4, out=a4)
                b4.step(a4)
            for i in range(2):
                b4.random_actions(555 + i, a4); b4.step(a4)
            ev4 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
            s4 = b4.total_steps()
            for i, (e0, e1) in enumerate(ev4):
                b4.random_actions(600 + i, a4)
                e0.record(); b4.step(a4); e1.record()
            torch.cuda.synchronize(dev)
            ms4 = sum(x.elapsed_time(y) for x, y in ev4) / 3
            cyc = b4.fetch("phases").reshape(B4, 8)[:, 7].double(); cyc = cyc[cyc > 0]
            ent["at_4x_slots"] = {"envs": B4, "ms": round(ms4, 4), "env_steps_per_s": round((b4.total_steps() - s4) / 3 / (ms4 * 1e-3), 1),
                                  "utilisation": round(float((cyc.mean() / cyc.max()).item()), 4) if cyc.numel() else None,
                                  "what": "the same route kernel on 4096 slots of the same regions (timing only; parity is the 1024-slot leg's): more routes per CU hide the long chains"}
            b4.close()
    except Exception as ex4:
        ent["at_4x_slots"] = {"error": str(ex4)}
    try:        # the same launches with the LDS-window form in front (xr_config.window: off by default — this is the measurement behind that)
        bw = RegionBatch(c5_regions, n_envs=Bc, device=dev, auto_reset=True, router=args.router, dial_mult=args.dial_mult,
                         launch_order=args.launch_order, window=1000)
        bw.reset()
        if c5_stagger is not None:
            for i, sd in enumerate(c5_stagger[1]):
                bw.random_actions(sd, a5)
                torch.where(off5_d > i, a5, zero5, out=a5)
                bw.step(a5)
        for sd in c5_seeds[:2]:
            bw.random_actions(sd, a5)
            bw.step(a5)
        evw = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_t)]
        fb = routed = 0.0
        for i, (e0, e1) in enumerate(evw):
            bw.random_actions(c5_seeds[2 + i], a5)
            e0.record()
            bw.step(a5)
            e1.record()
            st = bw.fetch("status")
            real_m = (a5 > 0) & ((st & 9) == 0)
            routed += float(real_m.sum().item())
            fb += float(((bw.fetch("touched") > 0) & real_m).sum().item())
        torch.cuda.synchronize(dev)
        same = bool((bw.fetch("hash").cpu().numpy().view("uint64") == gpu_hash).all())
        ent["window_form"] = {"ms": round(sum(a.elapsed_time(bb) for a, bb in evw) / n_t, 4), "window_tracks": 52,
                              "window_bytes_per_attempt": int(4 * 52 * 52 * 12), "fallback_share": round(fb / max(routed, 1.0), 4),
                              "same_hash_chains_as_the_default": same,
                              "note": "xr_config.window = 1000: the LDS router inside a 52 x 52 x 12 window around the net, accepted with an exactness "
                                      "certificate over the window faces, HBM-scratch form otherwise; same results, not faster (the launch is as long "
                                      "as its heaviest routes, which do not fit a window): off by default"}
        bw.close()
    except Exception as ex:
        ent["window_form"] = {"error": str(ex)}
    try:        # oracle replay of the leg's own actions on its first slots (a Dijkstra over 786 k nodes per search: 16 slots x 7 steps)
        n_chk = min(16, len(c5_regions), Bc)
        ent["parity"] = parity_check([c5_regions[e % len(c5_regions)] for e in range(n_chk)], c5_seeds,
                                     None if c5_stagger is None else (c5_stagger[0][:n_chk], c5_stagger[1]), gpu_hash, gpu_cum, n_check=n_chk)
    except Exception as ex:
        ent["parity"] = {"error": str(ex), "ok": False}
    return ent


if __name__ == "__main__":
    main()
