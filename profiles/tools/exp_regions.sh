#!/bin/bash
# Round 4, VERDICT r3 item 1: is the repeated-region median stable, and where does a slow region lose its time?
#   gpurun -- bash profiles/tools/exp_regions.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/regions
rm -rf "$O"; mkdir -p "$O"
cd "$R"
show='import json,sys; d=json.loads(sys.stdin.read()); r=d["regions"]["ms_per_step"]; print(sys.argv[1], round(d["value"]), "ms/step", {k: round(v*1e3,2) for k,v in r.items()}, "kernel", round(d["roofline"]["kernel_avg_ms"]*1e3,2), "span", round(d["roofline"]["region_span_ms_per_launch"]*1e3,2), "frac", round(d["roofline"]["frac"],4), "fixed", d.get("region_fixed_us"))'
for i in 1 2 3 4; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2> "$O/bench_$i.err" | grep -a '^{' | tee "$O/bench_driver_$i.json" | python -c "$show" driver_$i
done
timeout 600 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | grep -a '^{' | tee "$O/bench_default.json" | python -c "$show" default
for i in 1 2; do
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$i \
  bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2> "$O/bench_dist_$i.err" | grep -a '^{' | tee "$O/bench_dist_world1_$i.json" | python -c "$show" dist_world1_$i
done
cd /tmp && export TMPDIR=/tmp
DGP_BENCH_REGIONS=300 DGP_BENCH_DUMP_REGIONS=1 timeout 900 rocprofv3 --kernel-trace --hip-trace --output-format csv -d "$O/gaps" -o gaps -- \
  python "$R/bench.py" --steps 20 --warmup 5 --no-extras --no-cpu-baseline > "$O/gaps_bench.log" 2>&1
grep -a '^{' "$O/gaps_bench.log" > "$O/gaps_bench.json"
python "$R/profiles/tools/region_gaps.py" "$O/gaps" 20 2>&1 | tee "$O/region_gaps.txt"
find "$O/gaps" -name '*.csv' | xargs ls -la
rm -rf "$O/gaps"
# the same without the profiler: the distribution of 300 regions
DGP_BENCH_REGIONS=300 DGP_BENCH_DUMP_REGIONS=1 timeout 600 python "$R/bench.py" --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | grep -a '^{' > "$O/regions300.json"
python - <<PY
import json, numpy as np
d = json.load(open("$O/regions300.json")); w = np.asarray(d['regions']['wall_us']); s = np.asarray(d['regions']['span_us'])
print('300 regions without profiler: wall us/step  min %.2f p10 %.2f median %.2f p90 %.2f max %.2f' % tuple(np.percentile(w, [0, 10, 50, 90, 100]) / 20))
print('                              span us/step  min %.2f p10 %.2f median %.2f p90 %.2f max %.2f' % tuple(np.percentile(s, [0, 10, 50, 90, 100]) / 20))
print('regions > 12 %% above the median: %d' % int((w > 1.12 * np.median(w)).sum()))
PY
