#!/bin/bash
# round 3, call K: the profile round of the final build — GPU suite, default bench, rocprofv3 kernel stats + PMC traffic of the same
# command, SQ counters of the route kernel, config 5 counters (incl. L2 atomics), the L2-atomic ceilings, the write-pattern probe
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=r03_k; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "suite rc=$?"; tail -4 $OUT/pytest_gpu.log
timeout 120 tools/micro/atomic_rate > $OUT/atomic_rate.txt 2>&1; cat $OUT/atomic_rate.txt
timeout 60 rocprofv3 --list-avail 2>/dev/null | grep -io "TCC[A-Z0-9_]*ATOMIC[A-Za-z0-9_]*" | sort -u > $OUT/tcc_atomic_counters.txt; cat $OUT/tcc_atomic_counters.txt | head -20
bash tools/profile_round.sh $TAG > $OUT/profile_round.log 2>&1; tail -25 $OUT/profile_round.log
bash tools/pmc_sq.sh ${TAG}_sq 4096 6 > $OUT/sq.log 2>&1; tail -30 $OUT/sq.log
bash tools/config5_pmc.sh ${TAG}_c5 1024 > $OUT/c5.log 2>&1; tail -20 $OUT/c5.log
