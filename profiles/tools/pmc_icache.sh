#!/bin/bash
# instruction-fetch counters of the benchmark kernel
OUT=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/pmc_ic
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
P="python ${GRAFT_REPO_ROOT:-/root/repo}/profiles/tools/pmc_probe.py"
timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d "$OUT" -o ic -- $P > "$OUT/ic.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES -d "$OUT" -o mem -- $P > "$OUT/mem.log" 2>&1
