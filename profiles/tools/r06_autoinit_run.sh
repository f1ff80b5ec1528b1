cd $GRAFT_REPO_ROOT
for v in autozero autopattern; do
  for c in d6static d4general; do
    DGP_LIB_PATH=$PWD/dgpmp2_amd/lib/libdgpmp2_dev_$v.so timeout 300 python profiles/tools/r06_twin_repro.py $c 3 2>&1 | grep -v "^$" | tail -7
  done
done
echo "== standard units with pattern-initialised locals, every shape vs the C oracle"
DGP_LIB_PATH=$PWD/dgpmp2_amd/lib/libdgpmp2_std_autopattern.so timeout 600 python profiles/tools/r05_bisect.py 3 f32 static 2>&1 | tail -1 | cut -c1-1500
DGP_LIB_PATH=$PWD/dgpmp2_amd/lib/libdgpmp2_std_autopattern.so timeout 600 python profiles/tools/r05_bisect.py 2 f32 static_full,qfull 2>&1 | tail -1 | cut -c1-1500
