cd $GRAFT_REPO_ROOT
python tools/ab_step.py 4096
python tools/ab_step.py 512 | grep -v "quota=  6\|quota=  9"
