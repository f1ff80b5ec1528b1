#!/bin/bash
# bench.py's distributed branch with a world of one rank, three times, next to the single-process run on the same box (what the all-gather + closing barrier
# cost a 20-step timed region).   gpurun -- bash profiles/tools/exp_dist.sh
# Round 3: polling the closing barrier (dist.barrier(async_op=True) + is_completed()) instead of blocking in it made the region TWICE as long
# (32-39 us per step against 15-16; single process 14.0 on that box) -- the plain dist.barrier() stays.
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$i bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | grep -a '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dist world1:', round(d['value']), d['rccl_ranks'], round(d['ms_per_step']*1e3,2))"
done
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | grep -a '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('single:', round(d['value']), round(d['ms_per_step']*1e3,2))"
