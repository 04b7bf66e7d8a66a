cd $GRAFT_REPO_ROOT
for lib in libxroute_hip.so libxroute_hip_g1.so libxroute_hip_g2.so libxroute_hip_g3.so libxroute_hip.so libxroute_hip_g1.so libxroute_hip_g2.so libxroute_hip_g3.so; do
  echo "== $lib"; XR_LIB=$lib timeout 200 python tools/ab_launch_order.py 3 4096 2>&1 | grep "launch_order=0"
  XR_LIB=$lib timeout 200 python tools/ab_launch_order.py 5 1024 2>&1 | grep "launch_order=0"
done
python tools/ab_lib.py libxroute_hip.so libxroute_hip_g1.so libxroute_hip_g2.so libxroute_hip_g3.so 512
