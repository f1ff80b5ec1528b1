#!/bin/bash
# Round-3 experiments in one gpurun call (run from the repo root on the GPU box; the probes are cross-compiled beforehand with
# profiles/tools/kprobe.sh into dgpmp2_amd/lib/):  kernel variants timed stand-alone, launch-shape checks of the d = 6 variants,
# long-trajectory kernels.   gpurun -- bash profiles/tools/exp_round3.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/exp
rm -rf "$O"; mkdir -p "$O"
cd "$R"
{
for p in dgpmp2_amd/lib/kprobe_*; do
  case "$p" in *wps2*) continue;; esac
  for rep in 1 2; do timeout 120 "$p" 4096; done
done
for p in dgpmp2_amd/lib/kprobe_d4_base dgpmp2_amd/lib/kprobe_d4_wps2 dgpmp2_amd/lib/kprobe_d4_wpb4; do [ -x "$p" ] && timeout 120 "$p" 32768; done
} > "$O/kprobe.txt" 2>&1
cat "$O/kprobe.txt"
U="timeout 300 python profiles/tools/ubench.py"
{
for sh in 16,4 32,2 32,4 64,1; do
  DGP_FORCE_SHAPE=$sh $U --what step --dof 3 --covs perstate --tag shape_$sh
  DGP_FORCE_SHAPE=$sh $U --what step --dof 3 --covs qfull --tag shape_$sh
  DGP_FORCE_SHAPE=$sh $U --what step --dof 3 --flags vel,nonhol --tag shape_$sh
done
$U --what step,solve,eval,bwd --n 512 --B 512 --tag long
$U --what step,bwd --n 512 --B 512 --dof 3 --tag long
$U --what step --n 1024 --B 256 --tag long
$U --what step --n 256 --B 1024 --tag unrolled_64x4
} 2>/dev/null | grep -a '^{' > "$O/ubench.jsonl"
python - <<PY
import json
for l in open("$O/ubench.jsonl"):
  d = json.loads(l)
  print(d['tag'], 'dof', d['dof'], 'n', d['n'], 'B', d['B'], d['covs'], d['flags'], d['shape'], {k: v.get('kernel_us') for k, v in d.items() if isinstance(v, dict)})
PY
