cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp
for rep in 1 2 3; do for k in base4 dep4 base6 dep6 base4s dep4s dep4k; do timeout 120 dgpmp2_amd/lib/kprobe_$k; done; done 2>&1 | tee gpurun_out/exp/kernarg_dep.txt
