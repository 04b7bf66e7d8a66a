#!/bin/bash
# round-2 baseline on the round-1 kernels: SQ counters of the route kernel (final r01 build) + 512/1024/4096-env steps
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02a
mkdir -p $OUT
cd $R
for E in 512 1024 4096; do
  python bench.py --envs $E --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_$E.json 2> $OUT/bench_$E.err
  python bench.py --envs $E --steps 20 --warmup 5 --no-cpu-baseline --no-observation > $OUT/bench_route_$E.json 2>> $OUT/bench_$E.err
done
cat $OUT/bench_*.json | cut -c1-400
bash tools/pmc_sq.sh r02a_sq 4096 6
