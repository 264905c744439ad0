#!/bin/bash
# One GPU-box session: kernel stats + PMC passes of bench.py (run from the repo root on the GPU box; outputs under gpurun_out/).
# PMC passes are separate runs with --kernel-trace only (gpurun refuses --pmc combined with the tracing domains).
# The stats run uses --no-kernel-timing: its averages are then those of the launches INSIDE the replayed step (64 replays + 3 eager
# warm-up steps), which is what bench.py's `roofline` (in-kernel stamps) must agree with -- tools/check_roofline.py compares them.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --no-cpu-baseline --no-kernel-timing"
rocprofv3 --kernel-trace --memory-copy-trace --stats -d gpurun_out/prof_r4_stats -o r4 --output-format csv -- $B --steps 10 --warmup 4 > gpurun_out/prof_r4_stats.log 2>&1
python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/prof_r4_bench.log 2>&1
python3 tools/check_roofline.py gpurun_out/prof_r4_bench.log gpurun_out/prof_r4_stats/r4_kernel_stats.csv > gpurun_out/prof_r4_check.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch -o f --output-format csv -- $B --steps 3 --warmup 2 > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_write -o w --output-format csv -- $B --steps 3 --warmup 2 > gpurun_out/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/pmc_sq -o s --output-format csv -- $B --steps 3 --warmup 2 > gpurun_out/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_vfetch -o vf --output-format csv -- python3 tools/voxel_pmc.py > gpurun_out/pmc_vfetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_vwrite -o vw --output-format csv -- python3 tools/voxel_pmc.py > gpurun_out/pmc_vwrite.log 2>&1
python3 tools/pmc_kernels.py gpurun_out/pmc_fetch gpurun_out/pmc_write --json gpurun_out/pmc_traffic_kernels.json > gpurun_out/pmc_traffic_kernels.txt 2>&1
python3 tools/pmc_kernels.py gpurun_out/pmc_sq --json gpurun_out/pmc_sq_kernels.json > gpurun_out/pmc_sq_kernels.txt 2>&1
python3 tools/pmc_kernels.py gpurun_out/pmc_vfetch gpurun_out/pmc_vwrite --json gpurun_out/pmc_voxel_kernels.json > gpurun_out/pmc_voxel_kernels.txt 2>&1
# the raw counter CSVs are large: keep the per-kernel summaries only
find gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq gpurun_out/pmc_vfetch gpurun_out/pmc_vwrite -name "*.csv" -delete
find gpurun_out/prof_r4_stats -name "*_trace.csv" -size +20M -delete
cat gpurun_out/prof_r4_check.txt
