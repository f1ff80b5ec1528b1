cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp
for rep in 1 2 3; do for k in k4ns0 k4ns2 k4ns3; do timeout 120 dgpmp2_amd/lib/kprobe_$k; done; done 2>&1 | tee gpurun_out/exp/k4stash.txt
for seed in 0 1; do (timeout 1200 python tests/stress_random_configs.py --seed $seed 2>&1 | grep -v amdgpu.ids | tail -3) ; done | tee gpurun_out/exp/stress.txt
