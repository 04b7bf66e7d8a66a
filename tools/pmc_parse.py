"""Parse the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_driver.py into profiles/pmc_traffic.json.
Corrections (MI355X_MICROARCH.md §HBM): counters are in KiB; on gfx950 FETCH_SIZE reports half of the
bytes of a wide coalesced read -> x2, verified here on the 1 GiB copy; WRITE_SIZE is calibrated on the
1 GiB fill of the same run."""
import csv, json, re, sys, collections

def load(path, counter):
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            rows[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return rows

fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
bench = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])     # the bench JSON line of one of the passes
steps = int(bench["steps"])
GiB = float(1 << 30)
def pick(rows, pat):
    for k, v in rows.items():
        if re.search(pat, k):
            return v
    return []
def biggest(rows, pat):
    best = 0.0
    for k, v in rows.items():
        if re.search(pat, k):
            best = max([best] + v)
    return best
fill_w = biggest(write, "FillFunctor<float>")
add_r = biggest(fetch, "CUDAFunctorOnSelf_add<float>")
add_w = biggest(write, "CUDAFunctorOnSelf_add<float>")
cal = {"fill_1GiB_WRITE_SIZE_KiB": fill_w, "add_1GiB_FETCH_SIZE_KiB": add_r, "add_1GiB_WRITE_SIZE_KiB": add_w}
wfac = (GiB / (fill_w * 1024.0)) if fill_w > 1e5 else 1.0
rfac = (GiB / (add_r * 1024.0)) if add_r > 1e5 else 2.0
# identity of the measurement: bench.py quotes this file only for the same kernel sources and the same command
bench_args = bench["config"].get("bench_args") or {"gpus": 1, "steps": bench["steps"], "warmup": bench["warmup"], "envs": bench["config"]["envs_per_gpu"], "global_envs": 0,
                                                  "config": 3, "seed": 2024, "router": 0, "obs_mode": 0, "no_stagger": False}
out = {"source_sha": bench["config"].get("source_sha"), "bench_args": bench_args,
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) around `python3 bench.py --pmc-calibrate`",
       "units": "HBM bytes per launch, mean over the timed launches (KiB counters x 1024 x calibration factor)",
       "calibration": cal, "read_factor": rfac, "write_factor": wfac, "timed_launches": steps}
for k in bench["kernels"]:
    name = k["kernel"]
    f = pick(fetch, name)[-steps:]
    w = pick(write, name)[-steps:]
    rd = sum(v * 1024.0 * rfac for v in f) / max(len(f), 1)
    wr = sum(v * 1024.0 * wfac for v in w) / max(len(w), 1)
    out[name] = {"hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_total_bytes": rd + wr,
                 "algorithmic_bytes": k["bytes"], "traffic_over_algorithmic": (rd + wr) / k["bytes"]}
json.dump(out, sys.stdout, indent=1)
print()
