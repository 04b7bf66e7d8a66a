#!/bin/bash
# Run ON THE GPU BOX: HBM / L2 counters of the BASELINE config 5 route kernel (separate passes, --kernel-trace only).
TAG=${1:-c5pmc}; ENVS=${2:-1024}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU"; do
  N=$(echo $P | tr ' ' '_' | cut -c1-24)
  timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/$N -o p -- python3 $R/tools/config5_probe.py $ENVS 64 > $OUT/$N.log 2>&1
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
for p in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True) + glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        if "xr_route_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
print("# xr_route_kernel<false,-1> (frontier router, HBM-scratch form), config 5, $ENVS envs; per launch (mean over", max(cnt.values()) if cnt else 0, "launches)")
for k in sorted(tot): print(f"{k:24s} {tot[k] / max(cnt[k], 1):16.1f}")
if tot.get("TCC_HIT_sum"): print("L2 hit rate", tot["TCC_HIT_sum"] / (tot["TCC_HIT_sum"] + tot["TCC_MISS_sum"]))
PY
