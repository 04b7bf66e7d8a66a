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
python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmcF -o p -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --pmc-calibrate > $OUT/pmcF.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmcW -o p -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --pmc-calibrate > $OUT/pmcW.log 2>&1
ls $OUT $OUT/trace | head -30
