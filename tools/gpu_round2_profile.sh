#!/bin/bash
# One GPU-box session: kernel stats + PMC passes of bench.py (run from the repo root on the GPU box; outputs under gpurun_out/).
# PMC passes are separate runs with --kernel-trace only (gpurun refuses --pmc combined with the tracing domains).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --no-cpu-baseline --no-kernel-timing"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r2_stats -o r2 --output-format csv -- python3 bench.py --steps 10 --warmup 4 --no-cpu-baseline > gpurun_out/prof_r2_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch -o f --output-format csv -- $B --steps 3 --warmup 2 > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_write -o w --output-format csv -- $B --steps 3 --warmup 2 > gpurun_out/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/pmc_sq -o s --output-format csv -- $B --steps 3 --warmup 2 > gpurun_out/pmc_sq.log 2>&1
for d in pmc_fetch pmc_write pmc_sq; do python3 tools/pmc_sum.py gpurun_out/$d > gpurun_out/$d.summary.txt 2>&1; done
# keep only the summaries (the raw counter CSVs are large)
find gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq -name "*.csv" -size +2M -delete
tail -3 gpurun_out/pmc_sq.summary.txt
