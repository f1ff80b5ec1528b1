#!/bin/bash
# Does the placement of the kernel-argument block matter?  HIP_FORCE_DEV_KERNARG unset / 0 / 1 on the headline launch (2 000 timed steps)
# and on the driver's 20-step invocation.   gpurun -- bash profiles/tools/exp_kernarg.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/exp; mkdir -p "$O"; cd "$R"
: > "$O/kernarg.txt"
for rep in 1 2; do
for v in unset 0 1; do
  if [ $v = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
  for steps in 2000 20; do
    timeout 300 python bench.py --steps $steps --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | grep -a '^{' | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('HIP_FORCE_DEV_KERNARG=$v steps=$steps value', round(d['value']), 'wall us', round(d['ms_per_step']*1e3,2), 'kernel us', round(d['roofline']['kernel_avg_ms']*1e3,2))" | tee -a "$O/kernarg.txt"
  done
done
done
