cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_planner_api.py -q -x -k tiled 2>&1 | grep -a "^>\|^E\|passed\|failed" | head -12
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/edt_trace -o trace -- python $GRAFT_REPO_ROOT/profiles/tools/edt_bench.py > /tmp/edt_trace.log 2>&1
for db in $(find /tmp/edt_trace -name '*_results.db'); do python $GRAFT_REPO_ROOT/profiles/tools/summarize_rocpd.py "$db" "edt" 2>&1 | grep -a "edt_" | cut -c1-200; done
