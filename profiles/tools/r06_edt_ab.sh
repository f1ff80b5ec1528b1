#!/bin/bash
# Round 6: the row pass of dgp_sdf_2d, A/B on one box: offsets per side and trip of the padded search (DGP_EDT_UNROLL = 2: rounds 4-5; 4; 8) x the workgroup form
# (blk 0: one row per workgroup, 64 consecutive pixels per wavefront; 4 / 8: tiles of 4 x 16 / 8 x 8 pixels per wavefront -- measured, dropped) x the column pass
# (DGP_EDT_BITS = 0: rounds 4-5, downward values through memory; 1: bit planes in LDS, every word written once).  Libraries holding only csrc/sdf_edt.hip:
#   profiles/tools/r06_edt_ab.sh build      (in the build container: nine libraries of ~100 KB, seconds each)
set -u
VARIANTS=("2 0 0" "4 0 0" "8 0 0" "2 0 1" "8 0 1")      # (the "blk 4 / 8" lines of profiles/r06_sdf_edt_ab.txt: the tile form kept in profiles/tools/r06_edt_rows_blk.inc, no longer in the source)
if [ "${1:-}" = build ]; then
  for V in "${VARIANTS[@]}"; do set -- $V; hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DDGP_EDT_UNROLL=$1 -DDGP_EDT_BITS=$3 -o dgpmp2_amd/lib/libedt_u$1_b$2_p$3.so dgpmp2_amd/csrc/sdf_edt.hip 2>/dev/null || exit 1; done
  ls -la dgpmp2_amd/lib/libedt_*; exit 0
fi
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
O=gpurun_out/r06_edt_ab.txt; : > $O
for pass in 1 2; do
for V in "${VARIANTS[@]}"; do
  set -- $V
  DGP_EDT_LIB=$R/dgpmp2_amd/lib/libedt_u$1_b$2_p$3.so DGP_EDT_TAG="unroll $1 blk $2 bitplanes $3 pass $pass" DGP_EDT_CASES=${CASES:-4096x256,64x512,1x256} timeout 300 python profiles/tools/edt_bench.py 2>&1 | grep -a '^{' >> $O
done; done
cat $O
