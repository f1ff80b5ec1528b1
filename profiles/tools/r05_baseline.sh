#!/bin/bash
# round-5 baseline on the round-4 code: GPU tests, the training iteration with the grid gradient, its rocprofv3 summary
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05a
rm -rf "$O"; mkdir -p "$O"
cd "$R"
(timeout 1500 python -m pytest tests -m gpu -q -x > "$O/pytest_gpu.log" 2>&1; echo "pytest rc $?" >> "$O/pytest_gpu.log")
tail -3 "$O/pytest_gpu.log"
timeout 900 python profiles/tools/train_iteration.py 2> "$O/ti.err" | grep -a '^{' > "$O/train_iteration.json"
cat "$O/train_iteration.json"
cd /tmp && export TMPDIR=/tmp
for w in per_sample shared; do
  timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d "$O/trace_$w" -o trace -- python "$R/profiles/tools/train_iteration.py" --profile $w > "$O/trace_$w.log" 2>&1
  for db in $(find "$O/trace_$w" -name '*_results.db'); do python "$R/profiles/tools/summarize_rocpd.py" "$db" "rocprofv3 --kernel-trace --stats -- python profiles/tools/train_iteration.py --profile $w" > "$O/train_iteration_trace_$w.txt" 2>&1; done
  rm -rf "$O/trace_$w"
  head -30 "$O/train_iteration_trace_$w.txt"
done
