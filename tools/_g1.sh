cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_serve.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -8
