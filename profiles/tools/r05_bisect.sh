cd $GRAFT_REPO_ROOT
for v in 0 1 2; do
  DGP_LIB_PATH=$PWD/dgpmp2_amd/lib/libdgpmp2_dev_v$v.so timeout 600 python profiles/tools/r05_bisect.py 2 f32 2>&1 | tail -3 | cut -c1-400
done
