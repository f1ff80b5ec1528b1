#!/bin/bash
# Copies what `gpurun -- bash profiles/tools/collect_r06.sh` left in gpurun_out/r06 (and the round's reproducer / sweep outputs in gpurun_out/) into profiles/ under
# the r06_ prefix (run in the build container, repo root).  Files that a collection did not produce are skipped.
set -u
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r06
for f in bench_driver_invocation.json bench_driver_invocation_run2.json bench_driver_invocation_run3.json bench_default.json bench_dist_world1.json bench_2rank_one_device.json \
         bench_strong_n1.json train_iteration.json train_iteration_trace_per_sample.txt train_iteration_trace_shared.txt graph_replay.txt tile_probe.txt kernel_trace.txt pmc_counters.txt \
         pytest_gpu.log stress.txt sdf_edt.txt; do
  [ -s "$O/$f" ] && cp "$O/$f" "$R/profiles/r06_$f"
done
[ -s "$O/ubench.jsonl" ] && cp "$O/ubench.jsonl" "$R/profiles/r06_microbench_entry_points.txt"
[ -s "$O/traffic.json" ] && cp "$O/traffic.json" "$R/profiles/traffic.json"
cp "$R/dgpmp2_amd/lib/kernel_stats.json" "$R/profiles/r06_kernel_resources.json"
cp "$R/dgpmp2_amd/lib/build_info.json" "$R/profiles/r06_build_info.json"
# the compiler-fault investigation (profiles/r06_compiler_fault.md)
cat "$R/gpurun_out/r06_twin_repro.txt" > "$R/profiles/r06_twin_repro.txt" 2>/dev/null
{ echo; echo "== the same two kernels built through the repaired pipeline (__graft_entry__.compile_hip_unit)"; cat "$R/gpurun_out/r06_twin_patched.txt"; } >> "$R/profiles/r06_twin_repro.txt" 2>/dev/null
{ echo; echo "== -ftrivial-auto-var-init variants, and the STANDARD units with pattern-initialised locals against the C oracle (profiles/tools/r06_autoinit_run.sh)"; cat "$R/gpurun_out/r06_autoinit.txt"; } >> "$R/profiles/r06_twin_repro.txt" 2>/dev/null
cp "$R/gpurun_out/r06_twin_rows.txt" "$R/profiles/r06_twin_rows.txt" 2>/dev/null
cp "$R/gpurun_out/r06_twin_residual.txt" "$R/profiles/r06_twin_residual.txt" 2>/dev/null
cp "$R/gpurun_out/r06_sched_sweep_a.txt" "$R/profiles/r06_sched_sweep.txt" 2>/dev/null
sed -i 's#/opt/amdgpu/share/libdrm/amdgpu.ids: No such file or directory##' "$R/profiles/r06_twin_repro.txt" 2>/dev/null
ls -la "$R/profiles" | grep "r06_"
