cd $GRAFT_REPO_ROOT
for v in 0 1 2 3; do
  [ -f dgpmp2_amd/lib/libdgpmp2_dev_v$v.so ] || continue
  DGP_LIB_PATH=$PWD/dgpmp2_amd/lib/libdgpmp2_dev_v$v.so timeout 600 python profiles/tools/r05_bisect.py ${1:-3} ${2:-f32} ${3:-static_full,qfull,perstate} 2>&1 | tail -1 | cut -c1-600
done
