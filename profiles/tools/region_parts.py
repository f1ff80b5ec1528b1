#!/usr/bin/env python
"""What the fixed cost of bench.py's 20-launch timed region consists of: medians over `reps` repetitions, microseconds.
   gpurun -- python profiles/tools/region_parts.py"""
import ctypes, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench as Bn
from dgpmp2_amd import _capi
from dgpmp2_amd.gpmp2.plan_layer import solver_config

dev = torch.device('cuda', 0); torch.cuda.set_device(0)
B, n = 4096, 64
th0, start, goal, sdf = Bn.make_inputs(B, n, 256, dev)
solver = _capi.Solver(solver_config(num_states=n, dof=2, io_dtype=torch.float32))
dth = torch.empty_like(th0); err = torch.empty(B, device=dev); eex = torch.empty(B, device=dev); info = torch.zeros(B, dtype=torch.int32, device=dev)
pc, hnd = _capi.get_pycall(), solver.h
st = torch.cuda.current_stream(); raw = st.cuda_stream
a = (th0.data_ptr(), start.data_ptr(), goal.data_ptr(), sdf.data_ptr(), dth.data_ptr(), err.data_ptr(), eex.data_ptr(), info.data_ptr())
def step(k=0): pc.gn_step(hnd, B, a[0], a[1], a[2], a[3], 256, 256, 0, 0, 0, None, 0, None, None, None, a[4], a[5], a[6], a[7], raw)
Bn.prewarm(step, 0.6)
sync = torch.cuda.synchronize
reps = int(os.environ.get('REPS', 200)); K = 20
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
for x, y in evs: x.record(); y.record()
sync()

def med(f, refill=200):
  ts = []
  for r in range(reps):
    for _ in range(refill): step()
    sync(); t0 = time.perf_counter(); f(r); ts.append((time.perf_counter() - t0) * 1e6)
  ts = np.asarray(ts)
  return {'median': round(float(np.median(ts)), 2), 'p10': round(float(np.percentile(ts, 10)), 2), 'p90': round(float(np.percentile(ts, 90)), 2)}

out = {}
out['idle_synchronize'] = med(lambda r: sync())
def f_ev(r):
  e = evs[r][0]; e.record()
  while not e.query(): pass
out['event_record_and_spin'] = med(f_ev)
def f_one_q(r):
  step()
  while not st.query(): pass
out['one_launch_spin_stream_query'] = med(f_one_q)
def f_one_s(r):
  step(); sync()
out['one_launch_synchronize'] = med(f_one_s)
def f_cur(r):
  e0, e1 = evs[r]; e0.record()
  for k in range(K): step()
  e1.record()
  while not e1.query(): pass
  sync()
out['region_events_spin_sync (bench.py)'] = med(f_cur)
def f_q(r):
  for k in range(K): step()
  while not st.query(): pass
  sync()
out['region_spin_stream_query_sync'] = med(f_q)
def f_q2(r):
  for k in range(K): step()
  while not st.query(): pass
out['region_spin_stream_query_only'] = med(f_q2)
def f_s(r):
  for k in range(K): step()
  sync()
out['region_synchronize_only'] = med(f_s)
def f_e1(r):
  e1 = evs[r][1]
  for k in range(K): step()
  e1.record()
  while not e1.query(): pass
  sync()
out['region_end_event_spin_sync'] = med(f_e1)
def f_host(r):
  for k in range(K): step()
out['host_time_of_20_launch_calls'] = med(f_host)
out['region_no_refill_events_spin_sync'] = med(f_cur, refill=0)
for k, v in out.items(): print('%-44s %s' % (k, v))
print(json.dumps(out))
