#!/bin/bash
TAG=${1:-r04_q}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_game.py tests/test_gpu_route.py -x -q -m gpu -k "examples or fuzz" 2>&1 | tail -5
timeout 300 python examples/ispd18_rollout.py 4096 30 2>&1 | grep -v amdgpu | tail -8
