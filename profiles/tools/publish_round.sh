#!/bin/bash
# Copies what one `gpurun -- bash profiles/tools/collect_round.sh` call left in gpurun_out/ into profiles/ under the round's prefix
# (run in the build container, from the repo root):   bash profiles/tools/publish_round.sh r02
set -eu
P=${1:?round prefix, e.g. r02}
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/round
for f in bench_driver_invocation.json bench_driver_invocation_run2.json bench_driver_invocation_run3.json bench_dist_world1.json bench_default.json kernel_trace.txt kernel_trace_full.txt pmc_counters.txt atomic_probe.txt mfma_probe.txt phase_timeline.txt; do
  cp "$O/$f" "$R/profiles/${P}_$f"
done
cp "$O/pytest_gpu.log" "$R/profiles/${P}_pytest_gpu.log"
cp "$O/stress.txt" "$R/profiles/${P}_stress.txt"
cp "$O/shape_sweep.jsonl" "$R/profiles/${P}_shape_sweep.txt"
cat "$O/microbench.jsonl" "$O/ubench.jsonl" > "$R/profiles/${P}_microbench_entry_points.txt"
cp "$O/traffic.json" "$R/profiles/traffic.json"
cp "$R/dgpmp2_amd/lib/kernel_stats.json" "$R/profiles/${P}_kernel_resources.json"
ls -la "$R/profiles" | grep "${P}_"
