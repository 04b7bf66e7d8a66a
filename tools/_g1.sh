cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_route.py tests/test_gpu_config5.py tests/test_gpu_edges.py tests/test_order_contracts.py -x -q -m gpu 2>&1 | tail -8
python tools/phase_probe5.py 256 0 | head -6
python tools/config5_probe.py 1024 64 2>&1 | tail -2
for M in 2 4 8; do python tools/phase_probe.py 1024 0 0 $M | grep -E "router|scan|expand|barrier|trace|build"; done
for E in 512 4096; do python bench.py --envs $E --steps 20 --warmup 5 --no-cpu-baseline --no-observation --no-legs | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['envs_per_gpu'], d['value'], d['ms_per_step'])"; done
