"""bench.py contract on the GPU box (tiny run): one JSON line with the driver's keys, roofline and cpu_baseline."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                          "--envs", "256", "--cpu-seconds", "2", "--c5-envs", "8", "--c5-regions", "2", "--pack-envs", "512"] + extra, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


@pytest.mark.parametrize("extra", [[], ["--no-fuse"]])
def test_bench_json_contract(extra):
    d = _run(extra)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["unit"] == "env-steps/s" and d["value"] > 0 and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] > 0          # never a pasted constant: null unless a PMC pass of this build + command exists
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "env-steps/s"
    assert c["single_thread"]["cores"] == 1 and c["single_thread"]["value"] > 0
    names = [k["kernel"] for k in d["kernels"]]
    assert not any("error" in k for k in d["kernels"]), d["kernels"]
    if not extra:
        # per-kernel lines: the step kernel, the route-only kernel and the BASELINE config 5 route, each with its own numbers
        assert names[0] == "xr_step_queue_kernel" and "xr_route_kernel" in names and any("config 5" in n for n in names)
        assert any("in-place" in n for n in names)
        # the REAL ispd18_test1 regions (extracted from the reference's LEF/DEF/guide) with their own oracle replay
        pk = [k for k in d["kernels"] if "design-derived" in k["kernel"]][0]
        assert any("guide rectangles" in n for n in names)          # ... and XR-Maze v2 on them with the design's own guides
        assert pk["parity"]["hash_chains_equal"] is True and pk["parity"]["cumulative_metrics_equal"] is True and pk["form"]["form"] == 3
        assert pk["parity"]["observations_equal"] is True and pk["parity"]["observations_checked"] >= 16
        # the reference's own configuration (maze_end_iter 3, follow_guide over the design's rectangles) as a FULL step, and every
        # XR-Maze v2 / config 5 leg: each with its own oracle replay
        v2full = [k for k in d["kernels"] if "the reference's TCL knob VALUES" in k["kernel"]]
        assert len(v2full) == 1 and v2full[0]["parity"]["ok"] is True and v2full[0]["parity"]["observations_equal"] is True
        for k in d["kernels"]:
            if "XR-Maze v2" in k["kernel"] or "config 5" in k["kernel"]:
                assert k["parity"]["ok"] is True, (k["kernel"], k["parity"])
        assert "queue form" in d["config"]["workload"]
        for k in d["kernels"]:
            assert k["ms"] > 0 and k["bytes"] > 0 and abs(k["frac"] - k["achieved"] / 8000.0) < 1e-3 and k["env_steps_per_s"] > 0
        # driver-visible side numbers: BASELINE config 1 latency and config 3 with the DQN counterpart attached
        ex = d["extras"]
        assert ex["config1_game_step"]["ms_median"] > 0 and ex["config3_dqn_attached"]["value"] > 0, ex
        # ... and BASELINE config 4's per-GPU share with the PPO baseline attached, in both placements of the policy (same actions env for env)
        c4 = ex["config4_ppo_attached_per_gpu_share"]
        assert c4["policy_per_rank"]["value"] > 0 and c4["central_learner_from_compact_state"]["value"] > 0, c4
        assert c4["policy_per_rank"]["parity_ok"] and c4["central_learner_from_compact_state"]["parity_ok"] and c4["same_actions_in_both_placements"]
        assert c4["central_learner_from_compact_state"]["compact_state"]["ratio_to_fp32_planes"] < 0.02
        # round 6: BASELINE config 1 as SURVEY 8d states it (one host thread: oracle + the real Request through the codec), config 2 (ingest + reward +
        # observation of 256 envs, no route), config 5's launch utilisation, the PPO counterpart at the reference's training cadence — and ALL legs
        # once more, compact, as the LAST key of the line (the driver keeps the tail of it)
        w1 = ex["config1_wire_loopback"]
        assert w1["cores"] == 1 and w1["ms_per_step"] > 0 and w1["request_bytes_mean"] > 100000 and set(w1["ms_per_component"]) >= {"sim_route", "encode_request", "decode_request", "build_3Dgrid"}
        c2 = [k for k in d["kernels"] if "BASELINE config 2" in k["kernel"]][0]
        assert c2["envs"] == 256 and c2["parity"]["ok"] is True and c2["parity"]["rewards_bit_equal"] is True and c2["ingest_ms"] > 0 and c2["observation_ms"] > 0
        c5 = [k for k in d["kernels"] if "config 5" in k["kernel"]][0]
        assert 0 < c5["launch_utilisation"]["utilisation"] <= 1 and "LATENCY CHAIN" in c5["note"]
        tc = ex["ppo_training_cadence"]
        assert tc["update_every_steps"] == 100 and tc["refill_ms"] > 0 and tc["window_ms"] > tc["refill_ms"] and tc["net_vectors_by"]["framework_convolutions"] == 0
        assert list(d.keys())[-1] == "legs" and len(json.dumps(d["legs"])) <= 1600
        lg = d["legs"]
        assert lg["step"]["ok"] is True and lg["c2_obs_reward_256"]["ok"] is True and lg["c5_route_1024"]["ok"] is True and "c1_wire_loopback_cpu" in lg and "ppo_cadence_100" in lg
        assert d["config"]["parity_ok"] is True and d["roofline"]["parity_ok"] is True
        # ~1 s of the same step after the timed region (clocks / thermals visible); never part of `value`
        su = ex["sustained"]
        assert su["steps"] == 500 and su["value"] > 0 and len(su["ms_per_step_by_100"]) == 5 and min(su["ms_per_step_by_100"]) > 0
    p = d["parity"]                                  # the checker leg: oracle replay of the run's own actions
    assert p["hash_chains_equal"] is True and p["cumulative_metrics_equal"] is True and p["env_steps"] > 0 and p["envs"] == 256
    # ... and the bytes of the observation the last timed launch wrote, for 32 envs spread over K (fused and two-launch form alike)
    assert p["observations_equal"] is True and p["observations_checked"] == 32 and p["ok"] is True
    # value is consistent with the reported step time: real env-steps <= slots
    assert d["value"] <= 256 * 1 / (d["ms_per_step"] * 1e-3) * 1.001


@pytest.mark.gpu
def test_step_kernel_keeps_four_workgroups_per_cu_on_ispd_sized_regions():
    """The LDS budget of the step kernel at 24x40x9 is laid out for 4 workgroups per CU; 3 would cost a
    quarter of the throughput without failing any parity test."""
    from xroute_env_amd.batch import RegionBatch
    from xroute_env_amd.regions import config_regions
    batch = RegionBatch(config_regions(3, 8), device="cuda:0")
    wgs, lds = batch.route_occupancy()
    assert wgs == 4 and lds <= 160 * 1024 // 4, (wgs, lds)
    # xr_config.block_threads is honoured: 8 waves per workgroup halve the resident workgroups (register-bound)
    wide = RegionBatch(config_regions(3, 8), device="cuda:0", block_threads=512)
    assert wide.route_occupancy()[0] < wgs
    # regions too large for LDS run 1024-thread workgroups by default
    from xroute_env_amd.regions import generate_region
    big = RegionBatch([generate_region(77, dims=(64, 64, 12), k_range=(3, 3), net_span=20)], device="cuda:0")
    assert big.route_occupancy()[0] >= 1


@pytest.mark.gpu
def test_bench_two_rank_control_flow_on_one_gpu():
    """The N > 1 path of bench.py (rendezvous, barriers, max-over-ranks timing, per-step gather of the env records, one
    JSON line from rank 0) with two ranks sharing cuda:0 over gloo: a functional check on a single-GPU box."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, XR_BENCH_BACKEND="gloo", XR_BENCH_SAME_DEVICE="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "3", "--warmup", "1", "--envs", "128"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["global_envs"] == 256 and "cpu_baseline" not in d and "RCCL all_gather" in d["config"]["workload"]
    # the run certifies itself: every rank's records arrived in the gather, two distinct ranks answered, and every rank's first
    # 32 envs replay on the oracle (hash chains, metrics, observation bytes), AND-reduced over the ranks
    assert d["gather_verified"] is True and d["ranks_seen"] == 2 and d["gathered_rows"] == 256
    assert d["parity"]["all_ranks_ok"] is True and d["parity"]["ok"] is True and d["parity"]["envs_per_rank"] == 32
    assert d["parity"]["observations_equal"] is True
    # strong scaling (BASELINE config 4 shape): the same global batch split over the ranks
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "3", "--warmup", "1", "--global-envs", "256"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["scaling"] == "strong" and d["config"]["global_envs"] == 256 and d["config"]["envs_per_gpu"] == 128
    assert d["gather_verified"] is True and d["ranks_seen"] == 2 and d["parity"]["all_ranks_ok"] is True
    # BASELINE config 4's learner flow: gather records + legal sets, policy on rank 0, i32 action broadcast
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "3", "--warmup", "1", "--global-envs", "256", "--learner"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert "learner flow" in d["config"]["workload"] and d["value"] > 0 and d["config"]["slots_stepped_per_batch_step"] > 0.5
    assert d["gather_verified"] is True and d["ranks_seen"] == 2 and d["parity"]["all_ranks_ok"] is True      # (replay of the BROADCAST actions)
    # a slice of the gathered records that is not what its owner sent fails the run: rc != 0, no N-GPU label
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "2", "--warmup", "1", "--envs", "64"],
                         capture_output=True, text=True, timeout=900, env=dict(env, XR_BENCH_TEST_CORRUPT_GATHER="1"))
    assert out.returncode != 0
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert lines and json.loads(lines[0])["gather_verified"] is False and json.loads(lines[0])["n_gpus"] is None


def _torchrun(nproc, bench_args, extra_env=None, expect_rc=0):
    import socket
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    env = dict(os.environ, XR_BENCH_BACKEND="gloo", XR_BENCH_SAME_DEVICE="1", **(extra_env or {}))
    if nproc == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + bench_args
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + bench_args
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    if expect_rc is not None:
        assert out.returncode == expect_rc, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stderr[-3000:]
    return json.loads(lines[0])


def test_bench_config4_ppo_attached_two_ranks_on_one_gpu():
    """BASELINE config 4 as stated ("4096 regions sharded 8 x MI355X, PPO baseline, RCCL env gather") as ONE runnable, self-certifying
    command, exercised with two gloo ranks sharing cuda:0:
      (a) `--gpus N --global-envs G --agent ppo`: every rank evaluates the PPO counterpart on its shard; the line carries agent-attached
          env-steps/s, env_share_of_step_time, gather_verified / ranks_seen / parity.all_ranks_ok (oracle replay of the actions the policy
          chose + the head planes) — and chooses the SAME action for every env at every step as the one-rank line (`actions_sha`), with
          identical final hash chains;
      (b) `--learner`: rank 0 evaluates the policy for all envs from the gathered compact state (pack -> all_gather -> expand), i32
          action broadcast — again the same actions env for env, plus the bytes it gathers;
      (c) BASELINE config 5's multi-GPU form (`--config 5 --envs E --regions R --no-observation --gpus N`)."""
    common = ["--global-envs", "192", "--agent", "ppo", "--steps", "3", "--warmup", "2"]
    one = _torchrun(1, common)
    assert one["n_gpus"] == 1 and one["scaling"] == "strong" and one["value"] > 0 and one["parity"]["ok"] is True
    assert one["parity"]["observations_equal"] is True and one["parity"]["observations_checked"] >= 16
    assert 0 < one["env_share_of_step_time"] < 1 and one["config"]["policy_placement"].startswith("every rank")
    # (a) two ranks, policy per rank
    two = _torchrun(2, common)
    assert two["n_gpus"] == 2 and two["config"]["global_envs"] == 192 and two["config"]["envs_per_gpu"] == 96 and two["value"] > 0
    assert two["gather_verified"] is True and two["ranks_seen"] == 2 and two["gathered_rows"] == 192
    assert two["parity"]["all_ranks_ok"] is True and two["parity"]["ok"] is True and two["parity"]["observations_equal"] is True
    assert 0 < two["env_share_of_step_time"] < 1 and two["agent_ms_per_step"] > 0 and two["env_ms_per_step"] > 0
    assert two["actions_sha"] == one["actions_sha"] and two["hash_chains_sha"] == one["hash_chains_sha"]
    assert "PPO counterpart" in two["metric"] and "RCCL all_gather" in two["config"]["workload"]
    # (b) the central learner: compact state gathered, expanded and evaluated on rank 0
    ltwo = _torchrun(2, common + ["--learner"])
    for d in (ltwo,):
        assert d["config"]["policy_placement"].startswith("rank 0") and d["parity"]["ok"] is True
        assert d["actions_sha"] == one["actions_sha"] and d["hash_chains_sha"] == one["hash_chains_sha"]
        cs = d["compact_state"]
        assert cs["bytes_gathered_per_step"] == 192 * cs["row_bytes"] and cs["ratio_to_fp32_planes"] < 0.02
        assert d["step_split_ms_rank0"]["expand"] > 0 and d["step_split_ms_rank0"]["pack_gather"] > 0
    assert ltwo["gather_verified"] is True and ltwo["ranks_seen"] == 2 and ltwo["parity"]["all_ranks_ok"] is True
    # a corrupted gather fails this line too: rc 3, no N-GPU label
    bad = _torchrun(2, ["--envs", "32", "--agent", "ppo", "--steps", "2", "--warmup", "2"], {"XR_BENCH_TEST_CORRUPT_GATHER": "1"}, expect_rc=None)
    assert bad["gather_verified"] is False and bad["n_gpus"] is None
    # DQN (greedy) and ragged shards (193 = 97 + 96) go through the same flow
    dq2 = _torchrun(2, ["--global-envs", "193", "--agent", "dqn", "--steps", "2", "--warmup", "2"])
    assert dq2["gather_verified"] is True and dq2["gathered_rows"] == 193 and dq2["parity"]["all_ranks_ok"] is True
    assert dq2["config"]["envs_per_gpu"] == 97
    # (c) BASELINE config 5, two ranks: 256x256x12 regions, route-only, compact state
    c5 = _torchrun(2, ["--config", "5", "--envs", "8", "--regions", "4", "--no-observation", "--steps", "2", "--warmup", "1"])
    assert c5["n_gpus"] == 2 and c5["config"]["global_envs"] == 16 and "config 5" in c5["config"]["workload"] and c5["value"] > 0
    assert c5["gather_verified"] is True and c5["ranks_seen"] == 2 and c5["parity"]["all_ranks_ok"] is True and c5["parity"]["ok"] is True


def test_bench_gpus_n_without_a_launcher_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no WORLD_SIZE must not benchmark ONE GPU under an N = 2 label: it starts the two ranks
    itself (a child torch.distributed.run) and forwards rank 0's line; a WORLD_SIZE that contradicts --gpus is refused."""
    env = dict(os.environ, XR_BENCH_BACKEND="gloo", XR_BENCH_SAME_DEVICE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--envs", "128"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_envs"] == 256 and d["value"] > 0
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--envs", "16"],
                         capture_output=True, text=True, timeout=300, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert bad.returncode == 2 and "refusing" in bad.stderr and not [l for l in bad.stdout.splitlines() if l.startswith("{")]


def test_bench_value_is_stationary_in_warmup():
    """The episodes are staggered before timing: mean nets left (and so the bytes per step) does not drift with --warmup."""
    a = _run(["--no-legs", "--no-cpu-baseline", "--envs", "1024", "--steps", "10"])
    b = _run(["--no-legs", "--no-cpu-baseline", "--envs", "1024", "--steps", "10", "--warmup", "25"])
    ka, kb = a["config"]["mean_nets_left"], b["config"]["mean_nets_left"]
    assert abs(ka - kb) / kb < 0.04, (ka, kb)


def test_bench_region_pack_with_the_reference_configuration():
    """`bench.py --region-pack <the design-derived pack> --maze-v2`: the main batch itself routes with the reference's simulator
    configuration (XR-Maze v2: maze_end_iter 3, guide cost over the design's guide rectangles) — the command `tools/final_round5.sh`
    profiles under rocprofv3.  One JSON line, the queue form, the workload text says what ran."""
    pack = os.path.join(ROOT, "tests", "golden", "ispd18_test1_regions.npz")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--envs", "512",
                          "--no-cpu-baseline", "--no-legs", "--region-pack", pack, "--maze-v2"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert "XR-Maze v2" in d["config"]["workload"] and "region pack" in d["config"]["workload"] and d["value"] > 0
    assert d["kernels"][0]["kernel"] == "xr_step_queue_kernel" and d["roofline"]["frac"] > 0


def test_config4_rollout_example_same_episode_in_every_placement():
    """examples/config4_rollout.py (the library-level form of BASELINE config 4: RegionBatch shards + agents.ppo_actions + dist.CompactStateExchange):
    one rank or two, policy per rank or central learner from gathered compact state — the same rewards, done counts and actions step by step."""
    import re
    import socket

    def run(nproc, extra):
        env = dict(os.environ, XR_BENCH_BACKEND="gloo", XR_BENCH_SAME_DEVICE="1")
        script = os.path.join(ROOT, "examples", "config4_rollout.py")
        if nproc == 1:
            for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
                env.pop(k, None)
            cmd = [sys.executable, script, "96"] + extra
        else:
            s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
                   "--master-port", str(port), script, "96"] + extra
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [re.sub(r"^(step \d+): .*?rank\(s\): ", r"\1: ", l) for l in out.stdout.splitlines() if l.startswith("step ")]
        assert len(lines) == 6
        return lines
    ref = run(1, [])
    assert run(1, ["--central"]) == ref
    assert run(2, []) == ref
    assert run(2, ["--central"]) == ref


def _forced_nccl(bench_args):
    """bench.py --gpus 1 --force-collectives on the REAL backend (no XR_BENCH_BACKEND override: "nccl" = RCCL)."""
    env = {k: v for k, v in os.environ.items() if k not in ("XR_BENCH_BACKEND", "XR_BENCH_SAME_DEVICE", "WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-collectives"] + bench_args,
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stderr[-3000:]
    return json.loads(lines[0])


def test_rccl_code_path_runs_with_one_rank():
    """VERDICT r5 missing #1: every distributed test forces gloo, so `init_process_group("nccl", device_id=..)`, the two-buffer async
    `all_gather_into_tensor` of uint8 rows + `Work.wait()` on RCCL's stream, `verify_gather`, `broadcast`, `all_reduce(MAX/SUM/MIN)` and the
    central learner's exchange (row-size `all_reduce`, `gather` to rank 0) had never executed — the first 8-GPU run would also have been the
    first RCCL call.  `--force-collectives` makes ONE rank take every N > 1 branch on the real backend: env-only, `--agent ppo` per rank,
    `--learner`, and BASELINE config 5's multi-GPU form.  Same actions and hash chains as the plain one-rank lines."""
    # env-only headline shape (weak scaling form: the async gather pairs) and its learner form (records + legal masks gathered, broadcast)
    d = _forced_nccl(["--steps", "4", "--warmup", "2", "--envs", "256", "--no-cpu-baseline"])
    assert d["n_gpus"] == 1 and d["gather_verified"] is True and d["ranks_seen"] == 1 and d["gathered_rows"] == 256
    assert d["parity"]["all_ranks_ok"] is True and d["parity"]["ok"] is True and "forced_collectives" in d["config"]
    assert "RCCL all_gather" in d["config"]["workload"] and d["value"] > 0
    dl = _forced_nccl(["--steps", "3", "--warmup", "1", "--envs", "128", "--no-cpu-baseline", "--learner"])
    assert dl["gather_verified"] is True and dl["ranks_seen"] == 1 and dl["parity"]["all_ranks_ok"] is True
    # strong-scaling form (--global-envs): the same gather through the other buffer set-up
    ds = _forced_nccl(["--steps", "3", "--warmup", "1", "--global-envs", "192", "--no-cpu-baseline"])
    assert ds["scaling"] == "strong" and ds["gather_verified"] is True and ds["gathered_rows"] == 192 and ds["parity"]["all_ranks_ok"] is True
    # BASELINE config 4: PPO per rank, then the central learner — same actions env for env as the plain one-rank line
    common = ["--global-envs", "192", "--agent", "ppo", "--steps", "3", "--warmup", "2"]
    one = _torchrun(1, common)
    per_rank = _forced_nccl(common)
    assert per_rank["gather_verified"] is True and per_rank["ranks_seen"] == 1 and per_rank["parity"]["all_ranks_ok"] is True
    assert per_rank["actions_sha"] == one["actions_sha"] and per_rank["hash_chains_sha"] == one["hash_chains_sha"]
    assert "forced_collectives" in per_rank["config"] and "RCCL all_gather" in per_rank["config"]["workload"]
    central = _forced_nccl(common + ["--learner"])
    assert central["gather_verified"] is True and central["ranks_seen"] == 1 and central["parity"]["all_ranks_ok"] is True
    assert central["actions_sha"] == one["actions_sha"] and central["hash_chains_sha"] == one["hash_chains_sha"]
    cs = central["compact_state"]
    assert cs["collective"].startswith("gather to rank 0") and cs["bytes_per_link_per_step"] == 192 * cs["row_bytes"]
    assert central["step_split_ms_rank0"]["pack_gather"] > 0 and central["step_split_ms_rank0"]["expand"] > 0
    # BASELINE config 5's multi-GPU form
    c5 = _forced_nccl(["--config", "5", "--envs", "8", "--regions", "4", "--no-observation", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert c5["gather_verified"] is True and c5["ranks_seen"] == 1 and c5["parity"]["all_ranks_ok"] is True and "config 5" in c5["config"]["workload"]
