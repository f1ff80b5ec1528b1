#!/bin/bash
# The short version of collect_round.sh (one gpurun call, ~10 minutes of box time): GPU test log, the driver's bench invocation, its
# rocprofv3 kernel-trace summary, and bench.py's distributed branch with a world of one rank (torch.distributed.run --nproc-per-node 1:
# process group on RCCL, barriers, the all-gather of the final trajectories, the max-over-ranks reduction -- the code an N-GPU SCALE run executes).
#   gpurun -- bash profiles/tools/quick_round.sh [skip-tests]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/quick
rm -rf "$O"; mkdir -p "$O"
cd "$R"
if [ "${1:-}" != "skip-tests" ]; then
  (timeout 1500 python -m pytest tests -m gpu -q > "$O/pytest_gpu.log" 2>&1; echo "pytest rc $?" >> "$O/pytest_gpu.log")
  tail -5 "$O/pytest_gpu.log"
fi
timeout 600 python bench.py --steps 20 --warmup 5 2> "$O/bench.err" | grep -a '^{' > "$O/bench_driver_invocation.json"
python - <<EOF
import json
d = json.load(open("$O/bench_driver_invocation.json"))
print({k: d[k] for k in ('value', 'ms_per_step')}, d['roofline']['kernel_avg_ms'], d['roofline']['frac'])
for k in ('config3_vel_limits', 'config4_xyh', 'learned_covariances', 'per_sample_sdf'):
  if k in d: print(k, round(d[k]['kernel_avg_us'], 2), round(d[k]['roofline']['frac'], 4))
for k in d:
  if k.startswith('planner_'): print(k, {a: b for a, b in d[k].items() if a != 'note'})
EOF
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2> "$O/bench_dist.err" | grep -a '^{' > "$O/bench_dist_world1.json"
python -c "import json; d = json.load(open('$O/bench_dist_world1.json')); print('dist world1:', d['value'], d['rccl_ranks'], d['ms_per_step'])"
timeout 300 python profiles/tools/api_overhead.py 2>/dev/null | tail -1 | tee "$O/api_overhead.json"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$O/trace" -o trace -- python "$R/bench.py" --steps 20 --warmup 5 --no-extras --no-cpu-baseline > "$O/trace.log" 2>&1
for db in $(find "$O/trace" -name '*_results.db'); do python "$R/profiles/tools/summarize_rocpd.py" "$db" "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline" > "$O/kernel_trace.txt" 2>&1; done
rm -rf "$O/trace"
head -12 "$O/kernel_trace.txt"
