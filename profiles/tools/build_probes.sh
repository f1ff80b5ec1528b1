#!/bin/bash
# Cross-compiles the stand-alone probes of profiles/tools/ for gfx950 into dgpmp2_amd/lib/ (git-ignored; travels to the GPU box with
# the gpurun snapshot).  Run in the build container before `gpurun -- bash profiles/tools/collect_round.sh`.
set -eu
R=$(cd "$(dirname "$0")/../.." && pwd)
H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -I$R/dgpmp2_amd/csrc"
mkdir -p "$R/dgpmp2_amd/lib"
$H "$R/profiles/tools/atomic_probe.hip" -o "$R/dgpmp2_amd/lib/atomic_probe" &
$H "$R/profiles/tools/mfma_probe.hip" -o "$R/dgpmp2_amd/lib/mfma_probe" &
for q in 1 3; do $H -DPROBE_STOP=0 -DPROBE_QK=$q -DDGP_PHASE_STAMPS "$R/profiles/tools/phase_probe.hip" -o "$R/dgpmp2_amd/lib/phase_probe_qk$q" & done
wait
ls -la "$R/dgpmp2_amd/lib"
