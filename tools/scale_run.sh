#!/usr/bin/env bash
# The three 8-GPU commands of BASELINE configs 4 and 5 (one node, one rank per GPU, RCCL over xGMI), beside the headline's scaling curve.
# Each prints ONE self-certifying JSON line (gather_verified / ranks_seen / parity.all_ranks_ok; rc 3 when the certification fails).
#   tools/scale_run.sh [N_GPUS=8] [OUT_DIR=gpurun_out/scale]
# The driver's SCALE_rNN.json comes from `bench.py --gpus N` itself; this script adds the config-4 / config-5 shapes the verdicts ask for.
set -u
N=${1:-8}
OUT=${2:-gpurun_out/scale}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {   # name, bench args...
    local name=$1; shift
    local port=$((29600 + RANDOM % 2000))
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$port" \
        "$ROOT/bench.py" --gpus "$N" "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"
    echo "$name rc=$? $(head -c 300 "$OUT/$name.json")"
}
# headline, weak scaling (4096 env slots per GPU) and strong scaling (the 4096-env batch split over the ranks)
run headline_weak   --steps 20 --warmup 3
run headline_strong --steps 20 --warmup 3 --global-envs 4096
# BASELINE config 4: 4096 regions sharded over the node, PPO counterpart attached — policy on every rank / central learner on rank 0
run config4_ppo_per_rank        --global-envs 4096 --agent ppo --steps 20 --warmup 3
run config4_ppo_central_learner --global-envs 4096 --agent ppo --learner --steps 20 --warmup 3
# BASELINE config 5: 256x256x12 regions, 1024 slots per GPU, route-only with compact state
run config5 --config 5 --envs 1024 --regions 128 --no-observation --steps 10 --warmup 2
