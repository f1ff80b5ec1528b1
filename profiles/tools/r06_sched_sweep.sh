#!/bin/bash
# Round 6: does another machine-scheduler setting of hipcc buy the issue-bound kernels anything?  Libraries libdgpmp2_sw_<variant>.so = devbuild of the units
# 2_f32_g0 3_f32_g0 2_f32_g3 3_f32_g3 (static / Woodbury and per-state Kronecker step kernels, d = 4 and 6) under one -mllvm switch each; steady-clock kernel time of
# dgp_gn_step at B = 4096, n = 64 (profiles/tools/ubench.py), two passes.
cd "${GRAFT_REPO_ROOT:-.}"
for pass in 1 2; do
for f in dgpmp2_amd/lib/libdgpmp2_sw_*.so; do
  v=$(basename $f .so | sed 's/libdgpmp2_sw_//')
  for args in "--what step" "--what step --dof 3" "--what step --covs perstate" "--what step --dof 3 --covs perstate"; do
    r=$(DGP_LIB_PATH=$PWD/$f timeout 120 python profiles/tools/ubench.py $args --reps 1500 2>/dev/null | grep -a '^{' | python -c "
import sys, json
for l in sys.stdin:
  d = json.loads(l); print('kernel_us', d['step']['kernel_us'], 'period_us', d['step']['period_us'], d.get('shape'), d.get('kernel'))" | tr '\n' ' ')
    echo "pass $pass $v [$args]: $r"
  done
done
done
