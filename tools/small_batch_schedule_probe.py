"""Round 6, VERDICT r5 #5 (time-boxed): the SCHEDULE of the queue-form step kernel at small batches (512 / 1024 envs per GPU = the 8- / 4-GPU shares of the
4096-env batch).  Sweeps what can be varied without touching the router: which workgroups start with units (XR_QUEUE_SKIP_SHIFT: bit of the workgroup index;
3 / 5: whole CUs route or write, 8: two of each kind per CU, -1: every workgroup starts with a route), LDS-free helper writers beside the step kernel
(--helper-blocks), units per route (--quota).  One bench.py child per point (the switch is read once per process), ms per step kernel from its HIP events."""
import itertools, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = []
envs_list = [int(v) for v in sys.argv[1:]] or [512, 1024]
for envs in envs_list:
    for shift, helpers, quota in [(5, 0, 0), (8, 0, 0), (3, 0, 0), (-1, 0, 0), (5, 256, 0), (5, 512, 0), (5, 1024, 0), (8, 512, 0), (5, 0, 500), (5, 0, 1000), (8, 0, 1000), (5, 512, 1000)]:
        env = dict(os.environ, XR_QUEUE_SKIP_SHIFT=str(shift))
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--envs", str(envs), "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--no-legs", "--no-extras",
               "--helper-blocks", str(helpers), "--quota", str(quota)]
        best = None
        for rep in range(2):
            out = subprocess.run(cmd, capture_output=True, text=True, env=env)
            d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
            ms = d["kernels"][0]["ms"]
            best = ms if best is None else min(best, ms)
        rows.append({"envs": envs, "unit_first_bit": shift, "helper_blocks": helpers, "quota_permille": quota or 750, "step_kernel_ms": round(best, 4), "value": d["value"]})
        print(json.dumps(rows[-1]), flush=True)
