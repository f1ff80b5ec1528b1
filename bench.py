#!/usr/bin/env python
"""bench.py -- Gauss-Newton steps/sec of the fused HIP solver on BASELINE.json's headline workload.

Workload (BASELINE.json configs[1]): 2-D point robot, batch B = 4096 trajectories per GPU x n = 64 support
states (d = 4), one shared 256x256 signed-distance field (union of three circles), static covariances from
examples/configs/gpmp2_2d_params.yaml, fp32 I/O, fp64 arithmetic.  A "step" is one whole-batch Gauss-Newton step ==
one call of the C-ABI's dgp_gn_step == the reference's PlanLayer.forward (factor evaluation + block-tridiagonal
assembly + solve + err + err_ext).  The inputs of step k are the trajectories after (k mod 10) GN iterations from the
straight-line initialisation ("10 GN iters"), precomputed before the timed region and resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W]
N > 1 is launched by the driver with torch.distributed.run, one rank per GPU (RCCL): every rank owns its own 4096
trajectories (weak scaling, no data-path collective); the only collective is one all-gather of the final trajectories
at the end of the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU, N_STATES, DOF, GRID = 4096, 64, 2, 256
GN_ITERS = 10
HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6     # MI355X vector FP64 (spec)
# fp64 FMA/MUL/ADD instructions one wavefront of gn_kernel<DOF=2,LPT=16,C=4,float,STEP,static> executes (ISA histogram of the
# straight-line kernel); every one is a 64-lane operation, an FMA counting 2 flops
FP64_VALU_INSTS_PER_WAVE_16x4 = {'fma': 2330, 'mul_add': 579}


def algorithmic_bytes_per_trajectory(n, d, nl=1, io_bytes=4):
  """th in + dtheta out + 4 SDF taps per state + start, goal + err, err_ext (static covariances)."""
  return io_bytes * (2 * n * d + 4 * n * nl + 2 * d + 2)


def make_inputs(B, n, G, device, seed=0):
  """Deterministic synthetic inputs of SURVEY 8(d): start/goal ~ U(-4,4)^2 (start first, then goal), zero velocities,
  straight-line initial trajectories (utils/planner_utils.py:47-56), analytic three-circle SDF."""
  from dgpmp2_amd.utils.planner_utils import straight_line_trajb
  from dgpmp2_amd.utils.sdf_utils import circles_sdf, C2_CIRCLES
  g = torch.Generator().manual_seed(seed)
  start = torch.cat([torch.rand(B, 1, 2, generator=g, dtype=torch.float64) * 8 - 4, torch.zeros(B, 1, 2, dtype=torch.float64)], -1)
  goal = torch.cat([torch.rand(B, 1, 2, generator=g, dtype=torch.float64) * 8 - 4, torch.zeros(B, 1, 2, dtype=torch.float64)], -1)
  th0 = straight_line_trajb(start[:, :, :2], goal[:, :, :2], 10.0, n - 1, 2)
  sdf = torch.from_numpy(circles_sdf(G, C2_CIRCLES))[None, None]
  f = lambda t: t.to(torch.float32).contiguous().to(device)
  return f(th0), f(start), f(goal), f(sdf)


def measured_traffic():
  """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC pass (profiles/traffic.json, produced by
  profiles/tools/pmc_traffic.sh with the guide's gfx950 FETCH_SIZE correction); None when no such measurement is committed."""
  f = os.path.join(ROOT, 'profiles', 'traffic.json')
  try:
    return float(json.load(open(f))['hbm_bytes_per_launch'])
  except (OSError, ValueError, KeyError):
    return None


def usable_cores():
  """Host cores this process may really use: min(affinity, cgroup CPU quota).  (The GPU boxes expose 256 logical CPUs but
  cap the container at 16 through cgroup cpu.max; running 256 threads against that quota is ~100x slower.)"""
  n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
  try:
    if os.path.exists('/sys/fs/cgroup/cpu.max'):
      q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
      if q != 'max': n = min(n, max(1, int(int(q) / int(per))))
    elif os.path.exists('/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
      q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read()); per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
      if q > 0: n = min(n, max(1, q // per))
  except (OSError, ValueError):
    pass
  return n


def cpu_baseline(th_hist_cpu, start_cpu, goal_cpu, sdf_cpu, chunk=256, steps=3):
  """The reference's dense PyTorch-CPU op sequence (oracle/dense_torch.py, kind 'port'), fp64, all host cores, on a
  bounded sample: `chunk` of the 4096 trajectories, 1 warm-up + `steps` timed steps; scaled to whole-batch steps/s."""
  from oracle import dense_torch as DT
  from oracle.gpmp2_oracle import OracleParams
  cores = usable_cores()
  torch.set_num_threads(cores)
  p = OracleParams(dof=DOF, total_time_step=N_STATES - 1)
  P = DT.params_from_oracle(p)
  B = chunk
  qc = torch.from_numpy(p.static_covs(B)[0]); ow = torch.from_numpy(p.static_covs(B)[1]); eps = torch.from_numpy(p.static_covs(B)[2])
  sdf = sdf_cpu.double().expand(B, 1, GRID, GRID)
  ts = []
  inverse_impl = 'torch.inverse'
  with torch.no_grad():
    k = 0
    while k < steps + 1:
      th = th_hist_cpu[k % len(th_hist_cpu)][:B].double()
      t0 = time.perf_counter()
      try:
        DT.plan_layer_forward(th, start_cpu[:B].double(), goal_cpu[:B].double(), sdf, qc, ow, eps, P)
      except RuntimeError:
        # some hosts' MKL rejects batched torch.inverse ("Parameter 6 was incorrect on entry to DLASWP" -> "Pivots given to
        # lu_solve ..."): form the two explicit inverses with triangular solves against I instead and start over
        if inverse_impl != 'torch.inverse': raise
        inverse_impl = 'torch.linalg.solve_triangular(u, I) (torch.inverse fails in MKL on this host)'
        DT.set_explicit_inverse('solve_triangular')
        ts, k = [], 0
        continue
      ts.append(time.perf_counter() - t0)
      k += 1
  t_chunk = float(np.median(ts[1:]))
  return {'value': 1.0 / (t_chunk * (B_PER_GPU / B)), 'unit': 'GN steps/s (batch 4096)', 'cores': cores, 'kind': 'port',
          'sample': 'dense PyTorch-CPU fp64 restatement of PlanLayer.forward on %d of the 4096 trajectories, 1 warm-up + %d timed '
                    'steps, median %.3f s per %d-trajectory step, scaled by 4096/%d; explicit inverses via %s' % (B, steps, t_chunk, B, B, inverse_impl),
          'torch_threads': torch.get_num_threads()}


def cpu_blocktri(th_hist_cpu, start_cpu, goal_cpu, sdf_cpu, steps=5):
  """"Best CPU" line: the oracle's block-tridiagonal fp64 C restatement (oracle/gn_blocktri.c, OpenMP over trajectories)
  on the full 4096-trajectory batch, all usable cores."""
  from oracle import blocktri as BT
  from oracle.gpmp2_oracle import OracleParams
  cores = usable_cores()
  p = OracleParams(dof=DOF, total_time_step=N_STATES - 1)
  a = lambda t: t.double().numpy()
  st, go, sdf = a(start_cpu), a(goal_cpu), a(sdf_cpu)
  ts = []
  for k in range(steps + 1):
    th = a(th_hist_cpu[k % len(th_hist_cpu)])
    t0 = time.perf_counter(); BT.gn_step(p, th, st, go, sdf, nthreads=cores); ts.append(time.perf_counter() - t0)
  t = float(np.median(ts[1:]))
  return {'value': 1.0 / t, 'unit': 'GN steps/s (batch 4096)', 'cores': cores, 'kind': 'port',
          'sample': 'block-tridiagonal fp64 C restatement (oracle/gn_blocktri.c), full 4096-trajectory batch, 1 warm-up + %d timed '
                    'steps, median %.4f s' % (steps, t)}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=20000)
  ap.add_argument('--warmup', type=int, default=2000)
  ap.add_argument('--no-cpu-baseline', action='store_true')
  args = ap.parse_args()

  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if args.gpus > 1 and world != args.gpus:
    raise SystemExit('--gpus %d needs WORLD_SIZE=%d (launch with torch.distributed.run)' % (args.gpus, args.gpus))
  assert torch.cuda.is_available(), 'bench.py needs a GPU (the product path has no CPU fallback)'
  torch.cuda.set_device(local_rank)
  device = torch.device('cuda', local_rank)
  dist = None
  if world > 1:
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)

  import __graft_entry__
  if rank == 0: __graft_entry__.build()
  if dist is not None: dist.barrier()
  from dgpmp2_amd import _capi
  from dgpmp2_amd.gpmp2.plan_layer import solver_config

  B, n, d = B_PER_GPU, N_STATES, 2 * DOF
  th0, start, goal, sdf = make_inputs(B, n, GRID, device, seed=rank)
  cfg = solver_config(num_states=n, dof=DOF, io_dtype=torch.float32)
  solver = _capi.Solver(cfg)
  sdf_arg = solver.sdf_arg(sdf.data_ptr(), GRID, GRID, 0)
  stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

  # trajectories after 0..9 GN iterations (inputs of the timed steps)
  th_hist = [th0]
  dth = torch.empty_like(th0); err = torch.empty(B, device=device); eex = torch.empty(B, device=device)
  info = torch.zeros(B, dtype=torch.int32, device=device)
  for k in range(GN_ITERS - 1):
    solver.gn_step(B, th_hist[-1].data_ptr(), start.data_ptr(), goal.data_ptr(), sdf_arg, None, dth.data_ptr(), err.data_ptr(),
                   eex.data_ptr(), info.data_ptr(), stream)
    th_hist.append(th_hist[-1] + dth)
  torch.cuda.synchronize()
  assert int(info.abs().max()) == 0 and bool(torch.isfinite(th_hist[-1]).all())
  th_ptrs = [t.data_ptr() for t in th_hist]
  sp, gp, dp, ep, xp = start.data_ptr(), goal.data_ptr(), dth.data_ptr(), err.data_ptr(), eex.data_ptr()

  def run(k0, k1):
    for k in range(k0, k1):
      solver.gn_step(B, th_ptrs[k % GN_ITERS], sp, gp, sdf_arg, None, dp, ep, xp, None, stream)

  run(0, args.warmup)
  gathered = [torch.empty_like(th0) for _ in range(world)] if world > 1 else None
  ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize()
  if dist is not None: dist.barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  ev0.record()
  run(0, args.steps)
  ev1.record()
  if dist is not None:
    dist.all_gather(gathered, th_hist[-1])          # collect final trajectories (the only collective of the path)
  torch.cuda.synchronize()
  if dist is not None: dist.barrier()
  torch.cuda.synchronize()
  elapsed = time.perf_counter() - t0
  kernel_ms = ev0.elapsed_time(ev1) / args.steps      # average per-launch duration on the launch stream
  if dist is not None:
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

  # Beside the headline (one launch per GN step, the reference's step()): the same 10 GN iterations from th_init as ONE
  # launch of the fused loop (the reference's forward(); dgp_gn_solve, tol 0 so that all 10 run) -- no launch gaps, th
  # stays in registers, err / err_ext of every iteration written.  Reported as an extra field, never as `value`.
  tho = torch.empty_like(th0); its = torch.zeros(B, dtype=torch.int32, device=device)
  eh = torch.empty(B, GN_ITERS, device=device); eeh = torch.empty(B, GN_ITERS, device=device)
  def run_fused(reps):
    for _ in range(reps):
      solver.gn_solve(B, th_ptrs[0], sp, gp, sdf_arg, None, GN_ITERS, 0.0, tho.data_ptr(), its.data_ptr(), eh.data_ptr(), eeh.data_ptr(),
                      None, None, stream)
  fused_reps = max(1, args.steps // GN_ITERS)
  run_fused(max(1, args.warmup // GN_ITERS))
  fe0 = torch.cuda.Event(enable_timing=True); fe1 = torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize(); fe0.record(); run_fused(fused_reps); fe1.record(); torch.cuda.synchronize()
  fused_ms = fe0.elapsed_time(fe1) / fused_reps
  assert int(its.min()) == GN_ITERS

  if rank == 0:
    bytes_per_launch = algorithmic_bytes_per_trajectory(n, d) * B
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
    lpt, cc = solver.launch_shape(B)
    waves = (B + (64 // lpt) - 1) // (64 // lpt)
    fp64_flops = (2 * FP64_VALU_INSTS_PER_WAVE_16x4['fma'] + FP64_VALU_INSTS_PER_WAVE_16x4['mul_add']) * 64 * waves if (lpt, cc) == (16, 4) else None
    out = {
        'metric': 'Gauss-Newton steps/sec (whole node), batch=4096 x 64 states, 2D point robot',
        'value': world * args.steps / elapsed, 'unit': 'GN steps/s (one step = one whole-batch step of 4096 trajectories; per-GPU batches add up)',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'BASELINE configs[1]: 2D point robot, batch=4096 per GPU, 64 states, 256x256 shared SDF, '
                               'static covariances, inputs = trajectories after (k mod 10) GN iterations',
                   'batch_per_gpu': B, 'num_states': n, 'state_dim': d, 'sdf': [GRID, GRID], 'io_dtype': 'f32',
                   'parallelism': 'trajectory batch sharded, %d rank(s)' % world},
        'trajectory_steps_per_s': world * args.steps * B / elapsed,
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                     'traffic': measured_traffic(), 'kernel': 'gn_kernel<DOF=2,LPT=%d,C=%d,float,STEP,static>' % solver.launch_shape(B), 'kernel_avg_ms': kernel_ms,
                     'algorithmic_bytes_per_launch': bytes_per_launch,
                     'note': 'HBM is the bound SURVEY 8(d) prescribes; the measured limiter is fp64 VALU issue (see valu_fp64 and DESIGN.md section 5)'},
        'valu_fp64': None if fp64_flops is None else {'achieved_tflops': fp64_flops / (kernel_ms * 1e-3) / 1e12, 'peak_tflops': FP64_VECTOR_PEAK_TFLOPS,
                                                      'frac': fp64_flops / (kernel_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS, 'flops_per_launch': fp64_flops},
    }
    out['fused_forward'] = {'gn_iterations_per_launch': GN_ITERS, 'ms_per_launch': fused_ms, 'us_per_gn_iteration': 1e3 * fused_ms / GN_ITERS,
                            'gn_steps_per_s_per_gpu': GN_ITERS / (fused_ms * 1e-3),
                            'note': 'dgp_gn_solve: the 10 GN iterations of BASELINE configs[1] in one launch (rank 0, outside the timed region)'}
    if world == 1 and not args.no_cpu_baseline:
      hist_cpu = [t.cpu() for t in th_hist]
      out['cpu_baseline'] = cpu_baseline(hist_cpu, start.cpu(), goal.cpu(), sdf.cpu())
      out['cpu_baseline_blocktri'] = cpu_blocktri(hist_cpu, start.cpu(), goal.cpu(), sdf.cpu())
    print(json.dumps(out))
  if dist is not None:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
