cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_bench_contract.py -x -q -m gpu 2>&1 | tail -5
python bench.py --steps 20 --warmup 5 --no-cpu-baseline | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for k in d['kernels']: print({kk: k.get(kk) for kk in ('kernel','ms','bytes','frac','env_steps_per_s','mean_rounds','mean_touched_nodes','bytes_8d_full_sweep_formula','error')})"
