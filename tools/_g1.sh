cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_game.py tests/test_gpu_bench_contract.py tests/test_gpu_edges.py -x -q 2>&1 | tail -8
python tools/config1_probe.py 2>&1 | tail -8
