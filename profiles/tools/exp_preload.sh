cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp
./dgpmp2_amd/lib/rcp_probe | tee gpurun_out/exp/rcp.txt
for rep in 1 2 3; do for k in sq4 cub4 sq6 cub6; do timeout 120 dgpmp2_amd/lib/kprobe_$k; done; done 2>&1 | tee gpurun_out/exp/rcp_cubic.txt
