#!/bin/bash
# one development iteration on the GPU box: the every-kernel + parity + planner tests, then the training iteration with the grid gradient
#   gpurun -- bash profiles/tools/r05_step.sh [pytest -k expression]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05b
rm -rf "$O"; mkdir -p "$O"
cd "$R"
K="${1:-}"
if [ -n "$K" ]; then
  (timeout 1800 python -m pytest tests -m gpu -q -x -k "$K" > "$O/pytest_gpu.log" 2>&1; echo "pytest rc $?" >> "$O/pytest_gpu.log")
else
  (timeout 1800 python -m pytest tests -m gpu -q > "$O/pytest_gpu.log" 2>&1; echo "pytest rc $?" >> "$O/pytest_gpu.log")
fi
tail -15 "$O/pytest_gpu.log"
timeout 900 python profiles/tools/train_iteration.py 2> "$O/ti.err" | grep -a '^{' > "$O/train_iteration.json"
tail -3 "$O/ti.err"
python - <<PY
import json
d = json.load(open("$O/train_iteration.json"))
for B in ('B4096', 'B32'):
  for k in ('per_sample', 'shared'):
    for t in ('no_sdf_grad', 'sdf_grad'):
      e = d[B][k][t]
      print(B, k, t, 'eager %.1f' % e['eager_us'], 'replay', e.get('hip_graph_replay_us'), e.get('grad_layout', ''), e.get('grad_bytes', ''), e.get('graph_error', ''))
PY
timeout 600 python profiles/tools/graph_replay_breakdown.py 2>/dev/null | tee "$O/graph_replay.txt"
( U="python profiles/tools/ubench.py"; $U --what step,step_errs,bwd,bwd_errs,bwd_errs_noobs; $U --what step,step_errs,bwd,bwd_errs,bwd_errs_noobs --covs perstate; $U --what step,step_errs,bwd_errs,bwd_errs_noobs --covs scalar; $U --what bwd,bwd_sdf16,bwd_sdf16w,bwd_sdf8w --covs perstate; $U --what bwd,bwd_sdf,bwd_sparse --covs perstate --sdf persample; $U --what bwd,bwd_sdf16,bwd_sdf16w; $U --what bwd,bwd_sdf,bwd_sparse --sdf persample; $U --what step,eval,bwd,bwd_sdf --sdf persample --grids 6; $U --what step,eval,bwd,bwd_sdf --sdf persample --grids 6 --layout tiled4; $U --what step --layout tiled4; $U --what step ) 2>/dev/null | grep -a '^{' > "$O/ubench.jsonl"
cat "$O/ubench.jsonl"
cd /tmp && export TMPDIR=/tmp
for w in shared per_sample; do
  timeout 600 rocprofv3 --kernel-trace --stats -d "$O/trace_$w" -o trace -- python "$R/profiles/tools/train_iteration.py" --profile $w > "$O/trace_$w.log" 2>&1
  for db in $(find "$O/trace_$w" -name '*_results.db'); do python "$R/profiles/tools/summarize_rocpd.py" "$db" "rocprofv3 --kernel-trace --stats -- python profiles/tools/train_iteration.py --profile $w" > "$O/train_iteration_trace_$w.txt" 2>&1; done
  rm -rf "$O/trace_$w"
  cut -c1-150 "$O/train_iteration_trace_$w.txt" | awk -F'|' 'NR<=12 {print $1, "|", $2, "|", $4}'
done
cd "$R"
./dgpmp2_amd/lib/tile_probe | tee "$O/tile_probe.txt"
DGP_BENCH_ONE_DEVICE=1 DGP_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 \
  bench.py --gpus 2 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2> "$O/bench_2rank.err" | grep -a '^{' > "$O/bench_2rank_one_device.json"
python -c "
import json; d = json.load(open('$O/bench_2rank_one_device.json')); print('2 ranks on one device (gloo smoke):', {k: d.get(k) for k in ('value', 'n_gpus', 'rccl_ranks', 'value_steps_only', 'steps_per_s_at_5000', 'region_fixed_us')})" || tail -5 "$O/bench_2rank.err"
