#!/bin/bash
# Builds one variant of profiles/tools/kprobe.hip:  bash profiles/tools/kprobe.sh NAME [-DPROBE_...=.. -DDGP_...]   -> dgpmp2_amd/lib/kprobe_NAME
# (travels to the GPU box with the gpurun snapshot) and prints the kernel's ISA statistics.
set -eu
R=$(cd "$(dirname "$0")/../.." && pwd)
N=$1; shift
W=/tmp/kp/$N; rm -rf "$W"; mkdir -p "$W" "$R/dgpmp2_amd/lib"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -I"$R/dgpmp2_amd/csrc" -save-temps=obj "$@" "$R/profiles/tools/kprobe.hip" -o "$W/kprobe_$N"
cp "$W/kprobe_$N" "$R/dgpmp2_amd/lib/kprobe_$N"
python "$R/profiles/tools/isa_stats.py" "$W"/*gfx950.s | python -c "
import json, sys
for k, v in json.load(sys.stdin).items():
  print('$N', k, 'vgpr %(vgpr)d agpr %(agpr)d scratch %(scratch_bytes_per_lane)d vspill %(vgpr_spill)d sspill %(sgpr_spill)d occ %(waves_per_simd)d | valu %(valu)d fma %(fma_f64)d mul %(mul_f64)d add %(add_f64)d dpp %(dpp)d agprmov %(agpr_moves)d lds %(lds)d scratch_ops %(scratch_ops)d total %(total)d' % v)"
