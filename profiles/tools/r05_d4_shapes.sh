#!/bin/bash
# d = 4 learned-mode kernels in every launch shape that holds n = 64 (per-state and scalar covariances; step, backward, the training-iteration calls), B = 4096 and 512
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r05d
for B in 4096 512; do
for sh in auto 16,4 32,2 32,4 64,1 64,2; do
  if [ "$sh" = auto ]; then unset DGP_FORCE_SHAPE; else export DGP_FORCE_SHAPE=$sh; fi
  for cov in perstate scalar static; do
    python profiles/tools/ubench.py --what step,bwd,bwd_errs --covs $cov --B $B --reps 300 2>/dev/null | grep -a '^{' | python -c "
import sys, json
for l in sys.stdin:
  d = json.loads(l)
  print('B=$B', '$sh', d['covs'], d['shape'], {k: v['kernel_us'] for k, v in d.items() if isinstance(v, dict) and 'kernel_us' in v})"
  done
done
done | tee gpurun_out/r05d/d4_shapes.txt
