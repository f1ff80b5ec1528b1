#!/bin/bash
# d = 6 learned-mode kernels: every launch shape that holds n = 64 states, per-state (Kronecker), q_full (general) and scalar (scaled static) covariances,
# step and backward, B = 4096 -- is the shape dgp_host::choose_shape picks the fastest one?   gpurun -- bash profiles/tools/r05_d6_shapes.sh
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r05d
for sh in auto 16,4 32,2 32,4 64,1 64,2 64,4; do
  if [ "$sh" = auto ]; then unset DGP_FORCE_SHAPE; else export DGP_FORCE_SHAPE=$sh; fi
  for cov in perstate qfull scalar; do
    python profiles/tools/ubench.py --what step,bwd --dof 3 --covs $cov --reps 300 2>/dev/null | grep -a '^{' | python -c "
import sys, json
for l in sys.stdin:
  d = json.loads(l)
  print('$sh', d['covs'], d['shape'], {k: v['kernel_us'] for k, v in d.items() if isinstance(v, dict) and 'kernel_us' in v})"
  done
done | tee gpurun_out/r05d/d6_shapes.txt
