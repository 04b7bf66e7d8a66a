cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_obs.py tests/test_lefdef.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3
python bench.py --region-pack tests/golden/ispd18_test1_regions.npz --steps 10 --warmup 3 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], [(k['kernel'][:30], round(k['ms'],4)) for k in d['kernels']])"
XR_LIB=libxroute_hip_prev.so python bench.py --region-pack tests/golden/ispd18_test1_regions.npz --steps 10 --warmup 3 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], [(k['kernel'][:30], round(k['ms'],4)) for k in d['kernels']])"
