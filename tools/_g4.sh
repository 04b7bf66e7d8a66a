cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_config5.py -x -q -m gpu 2>&1 | tail -3
for lib in libxroute_hip_old.so libxroute_hip_run2.so libxroute_hip_run3.so libxroute_hip_run4.so libxroute_hip.so libxroute_hip_old.so libxroute_hip_run3.so libxroute_hip.so; do
  echo "== $lib"; XR_LIB=$lib timeout 300 python tools/ab_launch_order.py 5 1024 2>&1 | grep "launch_order=0"
  XR_LIB=$lib timeout 300 python tools/ab_launch_order.py 5 4096 2>&1 | grep "launch_order=0"
done
