#!/bin/bash
# Round 6: tests/stress_random_configs.py over seeds $1..$2 on the shipped library (one gpurun call); per seed: the cases attributed to conditioning + the summary line.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
O=gpurun_out/r06_stress_more_seeds_$1_$2.txt
echo "# tests/stress_random_configs.py, seeds $1-$2 on the final round-6 build (sha256 $(sha256sum dgpmp2_amd/lib/libdgpmp2_hip.so | cut -c1-12); 200 configurations each; every configuration with n <= 256" > $O
echo "# also runs dgp_gn_step_errors against the step + error kernels, and -- n <= 128 -- the tiled-grid twins against the row-major result)" >> $O
for s in $(seq $1 $2); do
  echo "== seed $s" >> $O
  (timeout 600 python tests/stress_random_configs.py --seed $s 2>&1 | grep -v amdgpu.ids | grep -E "cond\(|FAIL|fail|cases:|Error|error" | tail -12) >> $O
done
tail -3 $O
