#!/bin/bash
# K1 under the SQ / TCP / TCC counters, one rocprofv3 --pmc pass per group (never combined with the tracing domains), summarised per kernel.
# usage (GPU box): bash tools/voxel_pmc_passes.sh   -> gpurun_out/pmc_voxel_detail.txt
cd "$(dirname "$0")/.." && R=$PWD && cd /tmp && export TMPDIR=/tmp
i=0
dirs=""
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i + 1))
  d=$R/gpurun_out/pmc_vx_$i
  rocprofv3 --pmc $group --kernel-trace -d $d -o p --output-format csv -- python3 $R/tools/voxel_pmc.py > $R/gpurun_out/pmc_vx_$i.log 2>&1 || echo "pass $i ($group) failed" >> $R/gpurun_out/pmc_vx_fail.log
  dirs="$dirs $d"
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY
SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC
SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES GRBM_GUI_ACTIVE
TCP_TOTAL_READ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_UTCL1_TRANSLATION_MISS_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum
GROUPS
cd $R
python3 tools/pmc_kernels.py $dirs --json gpurun_out/pmc_voxel_detail.json > gpurun_out/pmc_voxel_detail.txt 2>&1
for d in $dirs; do find $d -name "*.csv" -delete; done
