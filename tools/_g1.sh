cd $GRAFT_REPO_ROOT
python tools/ab_step.py 4096
