#!/bin/bash
# rocprofv3 counter passes for the benchmark kernel (run on the GPU box through gpurun):
#   pass 1: FETCH_SIZE   pass 2: WRITE_SIZE   pass 3/4: SQ instruction / cycle counters
# Outputs rocpd databases under gpurun_out/pmc/; profiles/tools/pmc_report.py turns them into profiles/*.txt + traffic.json
set -u
OUT=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/pmc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P="python ${GRAFT_REPO_ROOT:-/root/repo}/profiles/tools/pmc_probe.py"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT" -o fetch -- $P > "$OUT/fetch.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT" -o write -- $P > "$OUT/write.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d "$OUT" -o sq1 -- $P > "$OUT/sq1.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM -d "$OUT" -o sq2 -- $P > "$OUT/sq2.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum -d "$OUT" -o l2 -- $P > "$OUT/l2.log" 2>&1
ls -la "$OUT"
