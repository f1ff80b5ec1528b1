cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do
  echo "=== variant $v (0: no tiled branch, 1: tiled branch in the 16/32-lane kernels)"
  export DGP_LIB_PATH=$PWD/dgpmp2_amd/lib/libdgpmp2_dev_v$v.so
  ( U="python profiles/tools/ubench.py"; $U --what step,step_errs,bwd,bwd_errs --covs perstate; $U --what step,step_errs,bwd,bwd_errs; $U --what step,bwd,bwd_errs --covs scalar ) 2>/dev/null | grep -a '^{' | python -c "
import sys, json
for l in sys.stdin:
  d = json.loads(l)
  print(d['covs'], {k: (v['kernel_us'], v['period_us']) for k, v in d.items() if isinstance(v, dict) and 'kernel_us' in v})"
done
