#!/bin/bash
# round 4, first GPU call: the new parity tests + the bench with the new legs + XR-Maze v2 phase baseline (before the v2 rework)
TAG=${1:-r04_a}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "suite rc=$?"; tail -15 $OUT/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-300 $OUT/bench.json; tail -5 $OUT/bench.err
timeout 300 python tools/phase_probe_v2.py 4096 1 1 2>&1 | grep -v amdgpu > $OUT/v2_phase_cycles_pack.txt; cat $OUT/v2_phase_cycles_pack.txt
timeout 300 python tools/phase_probe_v2.py 4096 0 1 2>&1 | grep -v amdgpu > $OUT/v1_phase_cycles_pack.txt; cat $OUT/v1_phase_cycles_pack.txt
timeout 300 python tools/phase_probe_v2.py 4096 1 0 2>&1 | grep -v amdgpu > $OUT/v2_phase_cycles_synth.txt; cat $OUT/v2_phase_cycles_synth.txt
