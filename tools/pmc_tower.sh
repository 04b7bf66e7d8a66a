#!/bin/bash
# Run ON THE GPU BOX: SQ counter passes on the obstacle tower alone (tools/tower_probe.py --child) + summary.   bash tools/pmc_tower.sh <tag>
TAG=${1:-pmc_tower}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQ_[A-Z0-9_]*MFMA[A-Z0-9_]*" | sort -u > $OUT/mfma_counters_listed.txt
cat $OUT/mfma_counters_listed.txt | tr '\n' ' '; echo
P1="SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM"
P2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"
P3="$(grep -E '^SQ_(INSTS_MFMA|VALU_MFMA_BUSY_CYCLES|INSTS_VALU_MFMA_MOPS_F32|INSTS_VALU_MFMA_F32|INSTS_VALU_MFMA_MOPS_BF16|INSTS_VALU_MFMA_BF16)$' $OUT/mfma_counters_listed.txt | tr '\n' ' ') SQ_WAVES SQ_BUSY_CU_CYCLES"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -o p -- python3 $R/tools/tower_probe.py --child 1024 9 40 24 /tmp/tower_pmc.pt > $OUT/p$i.log 2>&1
done
python3 $R/tools/pmc_sq_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
rm -rf $OUT/p1 $OUT/p2 $OUT/p3          # (the raw traces exceed what gpurun copies back)
