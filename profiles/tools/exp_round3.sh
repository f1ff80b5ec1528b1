#!/bin/bash
# Round-3 experiments in one gpurun call (run from the repo root on the GPU box; the probes are cross-compiled beforehand with
# profiles/tools/kprobe.sh into dgpmp2_amd/lib/):  kernel variants timed stand-alone, launch-shape checks of the d = 6 variants,
# long-trajectory kernels, two full stress seeds.   gpurun -- bash profiles/tools/exp_round3.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/exp
rm -rf "$O"; mkdir -p "$O"
cd "$R"
{
for p in dgpmp2_amd/lib/kprobe_*; do
  for rep in 1 2; do echo -n "$(basename $p) "; timeout 120 "$p" 4096; done
done
} > "$O/kprobe.txt" 2>&1
cat "$O/kprobe.txt"
U="timeout 300 python profiles/tools/ubench.py"
{
for sh in 16,4 32,2; do
  DGP_FORCE_SHAPE=$sh $U --what step,solve,bwd --dof 3 --covs perstate --tag shape_$sh
  DGP_FORCE_SHAPE=$sh $U --what step,solve,bwd --dof 3 --covs qfull --tag shape_$sh
done
$U --what step,solve,bwd --dof 3 --covs qfull --tag auto_shape
$U --what step,solve,bwd --dof 3 --covs perstate --tag auto_shape
} 2>/dev/null | grep -a '^{' > "$O/ubench.jsonl"
python - <<PY
import json
for l in open("$O/ubench.jsonl"):
  d = json.loads(l)
  print(d['tag'], 'dof', d['dof'], 'n', d['n'], 'B', d['B'], d['covs'], d['flags'], d['shape'], {k: (v.get('us_per_iter') or v.get('kernel_us')) for k, v in d.items() if isinstance(v, dict)})
PY
for seed in 0 1; do (timeout 900 python tests/stress_random_configs.py --seed $seed 2>&1 | grep -v amdgpu.ids | tail -3) >> "$O/stress.txt"; done
cat "$O/stress.txt"
