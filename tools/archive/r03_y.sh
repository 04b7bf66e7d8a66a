#!/bin/bash
# round 3, call Y: where a 512-env step's time goes on the GPU timeline (kernel start/end timestamps of consecutive steps)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03_y; mkdir -p $OUT; cd /tmp
export TMPDIR=/tmp XR_BENCH_NO_FORK=1
for e in 512 1024; do
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t$e -o t -- python3 $R/bench.py --envs $e --steps 20 --warmup 5 --no-cpu-baseline --no-legs > $OUT/t$e.log 2>&1
python3 - <<PY | tee -a $OUT/gaps.txt
import csv, glob
f = glob.glob("$OUT/t$e/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
# the last 20 step kernels and what surrounds them
idx = [i for i, r in enumerate(rows) if "xr_step_queue_kernel" in r[2]][-20:]
import statistics as st
dur = [rows[i][1] - rows[i][0] for i in idx]
per = [rows[idx[k + 1]][0] - rows[idx[k]][0] for k in range(len(idx) - 1)]
print("envs $e: step kernel duration mean %.1f us (min %.1f max %.1f); start-to-start period mean %.1f us" % (st.mean(dur) / 1e3, min(dur) / 1e3, max(dur) / 1e3, st.mean(per) / 1e3))
i = idx[-3]
for j in range(i - 6, i + 2):
    r = rows[j]; print("   %-40s start +%8.1f us  dur %7.1f us  gap before %6.1f us" % (r[2], (r[0] - rows[i - 6][0]) / 1e3, (r[1] - r[0]) / 1e3, (r[0] - rows[j - 1][1]) / 1e3))
PY
done
