cd $GRAFT_REPO_ROOT
SECONDS=0
python bench.py > gpurun_out/default_bench.json 2> gpurun_out/default_bench.err; echo "default bench wall seconds: $SECONDS"
python -c "
import json; d=json.loads(open('gpurun_out/default_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])
print(json.dumps(d['extras'])[:1200])"
