"""Per-kernel durations (dgp_time_next_launch events) of the first 20 launches after a synchronisation, 40 repetitions, with an optional idle time
in front (argv[1], seconds): how much a 20-launch timed region depends on what the GPU did just before it.  Measured on an MI355X: no idle 9.8 us per
kernel (median over repetitions; single repetitions up to 11.6), 1 ms of idle 10.8 us, 50 ms of idle 11.0 us -- the clocks sag within a millisecond of idling,
which is the +-5 % spread of `bench.py --steps 20` from box to box.   usage: python profiles/tools/first_kernels.py [idle_seconds]"""
import sys, time, ctypes, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
from dgpmp2_amd import _capi
from dgpmp2_amd.gpmp2.plan_layer import solver_config
dev = torch.device('cuda:0')
B, n, G = 4096, 64, 256
th0, start, goal, sdf = bench.make_inputs(B, n, G, dev)
s = _capi.Solver(solver_config(num_states=n, dof=2, io_dtype=torch.float32))
pc = _capi.get_pycall()
dth = torch.empty_like(th0); err = torch.empty(B, device=dev); eex = torch.empty(B, device=dev); info = torch.zeros(B, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
def step(): pc.gn_step(s.h, B, th0.data_ptr(), start.data_ptr(), goal.data_ptr(), sdf.data_ptr(), G, G, 0, 0, 0, None, 0, None, None, None, dth.data_ptr(), err.data_ptr(), eex.data_ptr(), info.data_ptr(), st)
bench.prewarm(lambda k: step())
K = 20
timer = _capi.KernelTimer(K)
rows = []; walls = []
for rep in range(40):
  for _ in range(5): step()
  torch.cuda.synchronize(); time.sleep(float(sys.argv[1]) if len(sys.argv) > 1 else 0.0); torch.cuda.synchronize()
  timer.reset()
  t0 = time.perf_counter()
  for k in range(K):
    timer.arm(); step()
  torch.cuda.synchronize()
  walls.append((time.perf_counter() - t0) / K * 1e6)
  rows.append(np.asarray(timer.durations_ms()) * 1e3)
r = np.asarray(rows)
print('idle %s s: per-position median kernel us:' % (sys.argv[1] if len(sys.argv) > 1 else 0), np.round(np.median(r, 0), 2))
print('  per-rep mean (min / median / max):', round(r.mean(1).min(), 2), round(np.median(r.mean(1)), 2), round(r.mean(1).max(), 2))
