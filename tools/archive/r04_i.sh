#!/bin/bash
TAG=${1:-r04_i}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
for lib in libxroute_hip.so libxroute_hip_biglists.so libxroute_hip.so libxroute_hip_biglists.so; do
  echo "== $lib"; XR_LIB=$lib timeout 300 python tools/config5_probe.py 1024 64 2>&1 | grep "^step"
done > $OUT/ab_config5_list_capacities.txt 2>&1; cat $OUT/ab_config5_list_capacities.txt
for m in 8 12 16; do echo "== dial_mult $m"; timeout 300 python tools/config5_probe.py 1024 64 0 $m 2>&1 | grep "^step"; done > $OUT/ab_config5_bucket_width.txt 2>&1; cat $OUT/ab_config5_bucket_width.txt
for t in 512 1024; do echo "== block_threads $t"; timeout 300 python tools/config5_probe.py 1024 64 $t 2>&1 | grep "^step"; done > $OUT/ab_config5_threads.txt 2>&1; cat $OUT/ab_config5_threads.txt
