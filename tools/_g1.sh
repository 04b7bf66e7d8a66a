cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_config5.py -x -q -m gpu -k overflow 2>&1 | tail -8
