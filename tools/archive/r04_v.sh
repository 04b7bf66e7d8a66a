#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_game.py -x -q -m gpu 2>&1 | tail -5
