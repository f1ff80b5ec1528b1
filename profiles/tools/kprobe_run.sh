#!/bin/bash
# Runs kprobe builds (profiles/tools/kprobe.sh NAME ... leaves dgpmp2_amd/lib/kprobe_NAME) three times each, interleaved:
#   gpurun -- bash profiles/tools/kprobe_run.sh OUTNAME NAME [NAME ...]      -> gpurun_out/exp/OUTNAME.txt
# (round 3: the kernel-argument, square-root, reciprocal and store-scope experiments of DESIGN.md section 5 "(h)"-"(j)"; outputs in profiles/r03_kernel_variants_late.txt)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/exp
O=$1; shift
for rep in 1 2 3; do for k in "$@"; do timeout 120 dgpmp2_amd/lib/kprobe_$k; done; done 2>&1 | tee gpurun_out/exp/$O.txt
