cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp
for rep in 1 2 3; do for k in wt0 wt6 wt7; do timeout 120 dgpmp2_amd/lib/kprobe_$k; done; done 2>&1 | tee gpurun_out/exp/store_wt_scalar.txt
