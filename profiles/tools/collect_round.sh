#!/bin/bash
# One gpurun call that produces every artefact profiles/ keeps for a round (run from the repo root on the GPU box):
#   bench JSON, rocprofv3 kernel-trace/stats of the bench command, PMC passes, launch-shape sweep.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/round
mkdir -p "$O"
cd "$R"
timeout 600 python bench.py 2>/dev/null | grep -a '^{' > "$O/bench.json"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$O" -o trace -- python "$R/bench.py" --steps 4000 --warmup 1000 --no-cpu-baseline > "$O/trace.log" 2>&1
cd "$R"
bash profiles/tools/pmc_traffic.sh > "$O/pmc.log" 2>&1
timeout 600 python profiles/tools/shape_sweep.py 2>/dev/null > "$O/shape_sweep.jsonl"
ls -la "$O" "$R/gpurun_out/pmc"
