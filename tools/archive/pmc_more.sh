#!/bin/bash
# Run ON THE GPU BOX: FETCH_SIZE / WRITE_SIZE passes of bench.py for another (steps, warmup) command; merge with tools/pmc_merge.py
TAG=${1:-pmcmore}; STEPS=${2:-20}; WARM=${3:-3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp XR_BENCH_NO_FORK=1
cd /tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmcF -o p -- python3 $R/bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-legs --pmc-calibrate > $OUT/pmcF.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmcW -o p -- python3 $R/bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-legs --pmc-calibrate > $OUT/pmcW.log 2>&1
python3 $R/tools/pmc_parse.py $(ls $OUT/pmcF/*counter_collection.csv | head -1) $(ls $OUT/pmcW/*counter_collection.csv | head -1) $OUT/pmcF.log > $OUT/pmc_traffic.json 2> $OUT/pmc_parse.err
head -c 600 $OUT/pmc_traffic.json
