cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python tools/config5_probe.py 1024 64 2>&1 | tail -2
