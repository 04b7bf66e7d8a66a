#!/bin/bash
# Run ON THE GPU BOX (through gpurun): tests, bench, rocprofv3 kernel trace and the two PMC traffic passes of the
# SAME bench command.  Everything lands under gpurun_out/<tag>/ ; copy the summaries into profiles/ afterwards.
#   gpurun -- 'bash tools/profile_round.sh r01'
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json
cd /tmp
export XR_BENCH_NO_FORK=1      # no worker pool under the profiler (its tool is preloaded into every child)
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmcF -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs --pmc-calibrate > $OUT/pmcF.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmcW -o p -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs --pmc-calibrate > $OUT/pmcW.log 2>&1
# summaries: kernel stats from the rocpd database, HBM traffic of the timed launches
python3 $R/tools/rocpd_summary.py $OUT/trace > $OUT/kernel_stats.csv 2> $OUT/kernel_stats.err
python3 $R/tools/pmc_parse.py $(ls $OUT/pmcF/*/*counter_collection.csv $OUT/pmcF/*counter_collection.csv 2>/dev/null | head -1) \
        $(ls $OUT/pmcW/*/*counter_collection.csv $OUT/pmcW/*counter_collection.csv 2>/dev/null | head -1) $OUT/pmcF.log > $OUT/pmc_traffic.json 2> $OUT/pmc_parse.err
head -c 1500 $OUT/pmc_traffic.json
ls $OUT $OUT/trace | head -30
