cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lo in 0 2 1; do for r in 0 2; do
python bench.py --envs 4096 --steps 20 --warmup 5 --no-cpu-baseline --c5-envs 0 --no-extras --launch-order $lo --router $r | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('launch_order $lo router $r', d['value'], d['roofline']['frac'], [round(k['ms'],4) for k in d['kernels']])"
done; done; done
