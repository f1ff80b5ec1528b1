#!/bin/bash
# Round-6 collection (collect_r05.sh + the --strong N = 1 point; the PMC passes now include the SQ issue / wait counters of the d = 6 kernels and the d = 6 learned mode)
# Round-6 collection (one gpurun call, run from the repo root on the GPU box): GPU test log, the driver's bench invocation (x3) and the default run, rocprofv3
# kernel-trace summaries of the driver's command and of the training iteration with the grid gradient, PMC traffic passes (incl. the tiled per-sample grids),
# entry-point microbenchmarks, graph-replay breakdown, tile probe, dgp_sdf_2d timing, the 2-rank one-device smoke of bench.py's N > 1 branch, a stress seed.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
rm -rf "$O" "$R/gpurun_out/pmc"; mkdir -p "$O"
cd "$R"
(timeout 1800 python -m pytest tests -m gpu -q > "$O/pytest_gpu.log" 2>&1; echo "pytest rc $?" >> "$O/pytest_gpu.log")
tail -3 "$O/pytest_gpu.log"
timeout 900 python bench.py --steps 20 --warmup 5 2> "$O/bench.err" | grep -a '^{' > "$O/bench_driver_invocation.json"
for i in 2 3; do timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | grep -a '^{' > "$O/bench_driver_invocation_run$i.json"; done
timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | grep -a '^{' > "$O/bench_default.json"
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | grep -a '^{' > "$O/bench_dist_world1.json"
DGP_BENCH_ONE_DEVICE=1 DGP_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 \
  bench.py --gpus 2 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | grep -a '^{' > "$O/bench_2rank_one_device.json"
timeout 600 python bench.py --strong --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | grep -a '^{' > "$O/bench_strong_n1.json"      # BASELINE configs[4] on ONE GPU: 32768 trajectories
timeout 900 python profiles/tools/train_iteration.py 2>/dev/null | grep -a '^{' > "$O/train_iteration.json"
timeout 600 python profiles/tools/graph_replay_breakdown.py 2>/dev/null > "$O/graph_replay.txt"
./dgpmp2_amd/lib/tile_probe > "$O/tile_probe.txt" 2>&1
timeout 600 python profiles/tools/edt_bench.py 2>/dev/null > "$O/sdf_edt.txt"
U="python profiles/tools/ubench.py"
( $U --what step,solve,eval,bwd,bwd_sdf8w,step_errs,bwd_errs,bwd_errs_sdf; $U --what step,solve,bwd,bwd_sdf8w,step_errs,bwd_errs,bwd_errs_sdf --covs perstate; $U --what step,bwd,bwd_errs --covs scalar;
  $U --what step,solve,bwd --covs qfull; $U --what step,solve,bwd --dof 3; $U --what step,solve --dof 3 --covs perstate; $U --what step,bwd --covs scalar --dof 3; $U --what step --flags vel;
  $U --what step,eval,bwd,bwd_sdf,bwd_sparse,step_errs,bwd_errs --sdf persample --grids 6; $U --what step,eval,bwd,bwd_sdf,step_errs,bwd_errs --sdf persample --grids 6 --layout tiled4;
  $U --what step,bwd --sdf persample --grids 6 --covs perstate; $U --what step,bwd --sdf persample --grids 6 --covs perstate --layout tiled4; $U --what step --layout tiled4;
  $U --what step,bwd --io f64; $U --what traced,chain; $U --what step --B 32768; $U --what step --n 128 --B 2048 ) 2>/dev/null | grep -a '^{' > "$O/ubench.jsonl"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$O/trace" -o trace -- python "$R/bench.py" --steps 20 --warmup 5 --no-extras --no-cpu-baseline > "$O/trace.log" 2>&1
for db in $(find "$O/trace" -name '*_results.db'); do python "$R/profiles/tools/summarize_rocpd.py" "$db" "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline" > "$O/kernel_trace.txt" 2>&1; done
rm -rf "$O/trace"
for w in per_sample shared; do
  timeout 600 rocprofv3 --kernel-trace --stats -d "$O/trace_$w" -o trace -- python "$R/profiles/tools/train_iteration.py" --profile $w > "$O/trace_$w.log" 2>&1
  for db in $(find "$O/trace_$w" -name '*_results.db'); do python "$R/profiles/tools/summarize_rocpd.py" "$db" "rocprofv3 --kernel-trace --stats -- python profiles/tools/train_iteration.py --profile $w" > "$O/train_iteration_trace_$w.txt" 2>&1; done
  rm -rf "$O/trace_$w"
done
cd "$R"
PMC_WORKLOADS="gn_step per_sample_sdf per_sample_sdf_tiled learned_covariances config4_xyh config4_perstate" bash profiles/tools/pmc_traffic.sh > "$O/pmc.log" 2>&1
cp gpurun_out/pmc/report.txt "$O/pmc_counters.txt"; cp gpurun_out/pmc/traffic.json "$O/traffic.json"
(timeout 900 python tests/stress_random_configs.py --seed 0 2>&1 | grep -v amdgpu.ids | tail -4) > "$O/stress.txt"
ls -la "$O"; du -sh "$R/gpurun_out"
