cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02soak
python tools/soak.py 4096 300 3 obs 2>&1 | tail -2 | tee gpurun_out/r02soak/soak3.txt
python tools/soak.py 4096 300 3 2>&1 | tail -1 | tee -a gpurun_out/r02soak/soak3.txt
python tools/soak.py 256 40 5 2>&1 | tail -1 | tee gpurun_out/r02soak/soak5.txt
