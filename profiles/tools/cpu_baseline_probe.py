"""Time the dense PyTorch-CPU restatement (oracle/dense_torch.py) on this host at several thread counts, each in a
subprocess with a timeout, and print what the host really offers (affinity, cgroup quota)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for f in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
  if os.path.exists(f): print(f, open(f).read().strip())
os.system("lscpu | egrep 'Model name|Socket|Core|Thread|^CPU\\(s\\)|NUMA node\\(s\\)'; uptime")
CODE = r'''
import sys, time, torch, numpy as np
sys.path.insert(0, %r)
thr, B = int(sys.argv[1]), int(sys.argv[2])
torch.set_num_threads(thr)
from oracle import dense_torch as DT
from oracle.gpmp2_oracle import OracleParams, circles_sdf, C2_CIRCLES, straight_line_trajb
DT.set_explicit_inverse(sys.argv[3])
p = OracleParams(dof=2, total_time_step=63); P = DT.params_from_oracle(p)
rs = np.random.RandomState(0)
start = np.concatenate([rs.uniform(-4,4,(B,1,2)), np.zeros((B,1,2))], -1); goal = np.concatenate([rs.uniform(-4,4,(B,1,2)), np.zeros((B,1,2))], -1)
th = torch.from_numpy(straight_line_trajb(start[:,:,:2], goal[:,:,:2], 10.0, 63, 2))
sdf = torch.from_numpy(circles_sdf(256, C2_CIRCLES))[None,None].expand(B,1,256,256)
qc, ow, eps = [torch.from_numpy(a) for a in p.static_covs(B)]
ts = []
with torch.no_grad():
  for k in range(3):
    t0 = time.perf_counter(); DT.plan_layer_forward(th, torch.from_numpy(start), torch.from_numpy(goal), sdf, qc, ow, eps, P); ts.append(time.perf_counter() - t0)
print('threads', thr, 'B', B, sys.argv[3], 'times', [round(t, 3) for t in ts])
''' % ROOT
for thr, B, kind in [(16, 64, 'solve_triangular'), (16, 256, 'solve_triangular'), (64, 256, 'solve_triangular'),
                     (len(os.sched_getaffinity(0)), 256, 'solve_triangular'), (16, 64, 'inverse')]:
  t0 = time.time()
  try:
    r = subprocess.run([sys.executable, '-c', CODE, str(thr), str(B), kind], capture_output=True, text=True, timeout=120)
    print((r.stdout.strip() or r.stderr.strip()[-300:]), '| wall', round(time.time() - t0, 1))
  except subprocess.TimeoutExpired:
    print('threads', thr, 'B', B, kind, 'TIMEOUT 120s')
