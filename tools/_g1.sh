cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_obs.py -x -q -m gpu -k "helper_writers" 2>&1 | tail -8
