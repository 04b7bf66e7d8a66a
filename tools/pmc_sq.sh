#!/bin/bash
# Run ON THE GPU BOX: SQ counter passes on a route-only run (tools/route_only.py) + summary.
#   bash tools/pmc_sq.sh <tag> [envs] [launches] [extra args of route_only.py]
TAG=${1:-sq}; ENVS=${2:-4096}; N=${3:-6}; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
P1="SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM"
P2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"
P3="SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INST_LEVEL_LDS SQ_INSTS_LDS_ATOMIC SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAVES"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -o p -- python3 $R/tools/route_only.py $ENVS $N "$@" > $OUT/p$i.log 2>&1
done
python3 $R/tools/pmc_sq_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
