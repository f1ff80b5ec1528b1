#!/bin/bash
# Round 6: build the twin translation units that hold the two wrong step-errors kernels under different compiler settings.
#   bash profiles/tools/r06_twin_repro.sh build          (here, CPU: ~3 min per variant, two at a time)
#   bash profiles/tools/r06_twin_repro.sh run            (on the GPU box: every variant found, both cases)
cd "$(dirname "$0")/../.."
UNITS="3e_f32_g0 2e_f32_g1"      # the twin units that hold the two wrong kernels (compared with the C oracle)
declare -A V
V[O3]=""
V[O2]="--flag=-O2"
V[O1]="--flag=-O1"
V[noalias]="--flag=-fno-strict-aliasing"
V[nocontract]="--flag=-ffp-contract=off"
V[noaa]="--flag=-mllvm --flag=-amdgpu-use-aa-in-codegen=0"
V[nodpp]="--flag=-mllvm --flag=-amdgpu-dpp-combine=false"
V[waitzero]="--flag=-mllvm --flag=-amdgpu-waitcnt-forcezero"
V[nosgpr2vgpr]="--flag=-mllvm --flag=-amdgpu-spill-sgpr-to-vgpr=false"
V[noagpr]="--flag=-mllvm --flag=-amdgpu-spill-vgpr-to-agpr=false"
V[nomachsched]="--flag=-mllvm --flag=-enable-misched=false"
V[nopostra]="--flag=-mllvm --flag=-enable-post-misched=false"
if [ "$1" = build ]; then
  shift
  names=${@:-O3 O2 O1 noalias nocontract noaa nodpp waitzero nosgpr2vgpr noagpr nomachsched nopostra}
  for v in $names; do
    ( python profiles/tools/devbuild.py $UNITS -D DGP_TWIN_REPRO=1 --raw ${V[$v]} -o libdgpmp2_dev_$v.so --show ',16,4,float,0,1' > /tmp/r06_build_$v.log 2>&1; echo "built $v rc $?" ) &
    while [ $(jobs -r | wc -l) -ge 1 ]; do sleep 2; done
  done
  wait
else
  mkdir -p gpurun_out
  for f in dgpmp2_amd/lib/libdgpmp2_dev_*.so; do
    for c in d6static d4general; do
      DGP_LIB_PATH=$PWD/$f timeout 300 python profiles/tools/r06_twin_repro.py $c 3 2>&1 | grep -v "^$" | tail -8
      DGP_TWIN_NO_ERRS=1 DGP_LIB_PATH=$PWD/$f timeout 300 python profiles/tools/r06_twin_repro.py $c 1 2>&1 | grep RESULT | sed 's/RESULT/RESULT(epilogue skipped at run time)/'
    done
  done
fi
