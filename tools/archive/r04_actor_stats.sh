cd /tmp; export TMPDIR=/tmp XR_BENCH_NO_FORK=1
for E in 1024 4096; do
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/agent_trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --agent dqn --envs $E --steps 20 --warmup 3 > /tmp/agent_trace.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/agent_trace 2>/dev/null | grep "xr_actor\|xr_ob_tower" | sed 's/(anonymous namespace):://g' | awk -F'",' '{print substr($1,1,40), $2}'; rm -rf /tmp/agent_trace
done
