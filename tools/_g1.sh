cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --router 2 > gpurun_out/b_dial.json 2> gpurun_out/b_dial.err; tail -3 gpurun_out/b_dial.err
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --router 1 --no-legs > gpurun_out/b_sweep.json 2> gpurun_out/b_sweep.err
python bench.py --steps 20 --warmup 40 --no-cpu-baseline --router 2 --no-legs > gpurun_out/b_dial_w40.json 2>/dev/null
for f in b_dial b_sweep b_dial_w40; do python - <<PY
import json
d=json.loads(open("gpurun_out/$f.json").read().strip().splitlines()[-1])
print("$f", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"]["mean_nets_left"], d.get("parity",{}).get("hash_chains_equal"))
for k in d["kernels"]: print("   ", k.get("kernel"), k.get("ms"), k.get("frac"), k.get("env_steps_per_s"), k.get("error"), k.get("mean_rounds"))
PY
done
