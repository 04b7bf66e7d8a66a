#!/bin/bash
# round 3, call AB: MIOpen find modes for the agent-attached step (no code change: env vars / torch flag via XR_CUDNN_BENCHMARK)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_ab; mkdir -p $OUT; cd $R
run() { echo "== $1"; shift; env "$@" timeout 600 python bench.py --agent dqn --envs 1024 --steps 10 --warmup 3 2>$OUT/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('agent_ms_per_step'), d['config'].get('env_ms_per_step'))"; tail -2 $OUT/err.txt | cut -c1-200; }
run default XR_DUMMY=1
run find_mode_normal MIOPEN_FIND_MODE=NORMAL
run find_enforce MIOPEN_FIND_MODE=NORMAL MIOPEN_FIND_ENFORCE=SEARCH
run benchmark_flag XR_CUDNN_BENCHMARK=1
