#!/bin/bash
# Copies what `gpurun -- bash profiles/tools/collect_r05.sh` left in gpurun_out/r05 into profiles/ under the r05_ prefix (run in the build container, repo root).
set -eu
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05
for f in bench_driver_invocation.json bench_driver_invocation_run2.json bench_driver_invocation_run3.json bench_default.json bench_dist_world1.json bench_2rank_one_device.json \
         train_iteration.json train_iteration_trace_per_sample.txt train_iteration_trace_shared.txt train_iteration_trace_replay.txt train_iteration_trace_replay_static.txt graph_replay.txt tile_probe.txt kernel_trace.txt kernel_trace_tiled.txt pmc_counters.txt \
         pytest_gpu.log stress.txt; do
  cp "$O/$f" "$R/profiles/r05_$f"
done
cat "$O/ubench.jsonl" "$O/ubench_more.jsonl" > "$R/profiles/r05_microbench_entry_points.txt"      # (+ the d = 6 training-iteration calls, the q_full twin, the tiled sparse gradient)
cp "$O/sdf_edt.txt" "$R/profiles/r05_sdf_edt_final_build.txt"
cp "$O/traffic.json" "$R/profiles/traffic.json"
cp "$R/dgpmp2_amd/lib/kernel_stats.json" "$R/profiles/r05_kernel_resources.json"
ls -la "$R/profiles" | grep "r05_"
