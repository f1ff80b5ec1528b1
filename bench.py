#!/usr/bin/env python
"""bench.py -- Gauss-Newton steps/sec of the fused HIP solver on BASELINE.json's headline workload.

Workload (BASELINE.json configs[1]): 2-D point robot, batch B = 4096 trajectories per GPU x n = 64 support
states (d = 4), one shared 256x256 signed-distance field (union of three circles), static covariances from
examples/configs/gpmp2_2d_params.yaml, fp32 I/O, fp64 arithmetic.  A "step" is one whole-batch Gauss-Newton step ==
one call of the C-ABI's dgp_gn_step == the reference's PlanLayer.forward (factor evaluation + block-tridiagonal
assembly + solve + err + err_ext + the per-trajectory SPD flags the planner always asks for).  The inputs of step k are
the trajectories after (k mod 10) GN iterations from the straight-line initialisation ("10 GN iters"), precomputed
before the timed region and resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W]
Before the W warm-up steps the GPU is brought to its steady clocks by a TIME-based pre-warm (>= 0.6 s of the same launches,
reported as `prewarm_s`): a cold MI355X runs the first milliseconds ~12 % slower, and K = 20 steps last 0.2 ms.
The timed region -- synchronise, EXACTLY K launches, synchronise -- is repeated R times (9 <= R <= 41, about 0.4 s in all; each
repetition is the whole region, nothing skipped) and `value` is K / the MEDIAN region: one 0.2 ms window right after an idling
synchronisation is decided by a single 40-70 us hiccup (`regions` in the JSON holds min / quartiles / max and the first region).
`roofline` is computed from the dominant kernel's average duration in the regime it is benchmarked in (`kernel_avg_ms`: HIP events on
the launch stream around 2000 back-to-back launches / 2000, median of three passes), which is what rocprofv3 --kernel-trace --stats
reports as the kernel's average for this command; the mean over launches that each record their own begin / end events
(`kernel_isolated_avg_ms`) and the event span of a K-launch region per launch (`region_span_ms_per_launch`) are separate fields.
N > 1 is launched by the driver with torch.distributed.run, one rank per GPU (RCCL): every rank owns its own 4096
trajectories (weak scaling, no data-path collective); the only collective is one all-gather of the final trajectories
at the end of each timed region, through the product's helper dgpmp2_amd.parallel.all_gather_trajectories into a
buffer allocated once; a region opens with barrier + synchronise and closes when the (stream-ordered) all-gather has completed --
it cannot before every rank has contributed -- and the slowest rank's clock counts (all-reduce MAX per region).
Rank 0 prints ONE JSON line.  Beside the headline it carries (rank 0, N = 1, outside the timed region): the fused 10-iteration
launch, BASELINE configs[2] and [3], the per-sample-SDF and learned-covariance regimes of configs[1], each with its own
roofline block, the planner-API call rate, the signed-distance-field transform (dgp_sdf_2d), and the CPU baselines.
"""
import argparse
import contextlib
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
STRONG_TOTAL = 32768      # --strong: BASELINE configs[4], one batch of 32768 trajectories for the whole node
GN_ITERS = 10
PREWARM_S = float(os.environ.get("DGP_BENCH_PREWARM_S", 0.6))      # (override: tuning only; 0.6 against 2.5 s makes no difference to the 20-step runs)
HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6     # MI355X vector FP64 (spec)


def algorithmic_bytes_per_trajectory(n, d, nl=1, io_bytes=4, cov_tensors=False, dof=None):
  """th in + dtheta out + 4 SDF taps per state + start, goal + err, err_ext; plus, when the per-state
  covariance tensors are streamed (the learned mode / the reference's API shape), qc_inv (n-1, dof, dof) + obs_w + eps."""
  b = io_bytes * (2 * n * d + 4 * n * nl + 2 * d + 2)
  if cov_tensors:
    dof = d // 2 if dof is None else dof
    b += io_bytes * ((n - 1) * dof * dof + 2 * n * nl)
  return b


def make_inputs(B, n, G, device, seed=0, dof=2):
  """Deterministic synthetic inputs of SURVEY 8(d): start/goal ~ U(-4,4)^2 (start first, then goal), zero velocities,
  straight-line initial trajectories (utils/planner_utils.py:47-56), analytic three-circle SDF.  dof = 3 (BASELINE
  configs[3]): start heading 0, goal heading pi/2 (examples/diff_gpmp2_nonholonomic_example.py:44-46)."""
  from dgpmp2_amd.utils.planner_utils import straight_line_trajb
  from dgpmp2_amd.utils.sdf_utils import circles_sdf, C2_CIRCLES
  g = torch.Generator().manual_seed(seed)
  sp = torch.rand(B, 1, 2, generator=g, dtype=torch.float64) * 8 - 4
  gp = torch.rand(B, 1, 2, generator=g, dtype=torch.float64) * 8 - 4
  if dof == 3:
    sp = torch.cat([sp, torch.zeros(B, 1, 1, dtype=torch.float64)], -1)
    gp = torch.cat([gp, torch.full((B, 1, 1), float(np.pi / 2), dtype=torch.float64)], -1)
  start = torch.cat([sp, torch.zeros(B, 1, dof, dtype=torch.float64)], -1)
  goal = torch.cat([gp, torch.zeros(B, 1, dof, dtype=torch.float64)], -1)
  th0 = straight_line_trajb(start[:, :, :dof], goal[:, :, :dof], 10.0, n - 1, dof)
  sdf = torch.from_numpy(circles_sdf(G, C2_CIRCLES))[None, None]
  f = lambda t: t.to(torch.float32).contiguous().to(device)
  return f(th0), f(start), f(goal), f(sdf)


def make_per_sample_sdfs(B, G, device, seed=1, chunk=256):
  """SURVEY 8(d) per-sample mode: one GxG grid per trajectory, three circles each, centres ~ U(-3.5,3.5)^2, radii ~
  U(0.4,1.0), torch.Generator().manual_seed(1); fp64 arithmetic on the device, stored in fp32 (B,1,G,G)."""
  g = torch.Generator().manual_seed(seed)
  c = (torch.rand(B, 3, 2, generator=g, dtype=torch.float64) * 7.0 - 3.5).to(device)
  r = (torch.rand(B, 3, generator=g, dtype=torch.float64) * 0.6 + 0.4).to(device)
  xs = torch.linspace(-5.0, 5.0, G, dtype=torch.float64, device=device)
  ys = torch.linspace(5.0, -5.0, G, dtype=torch.float64, device=device)
  out = torch.empty(B, 1, G, G, dtype=torch.float32, device=device)
  for lo in range(0, B, chunk):
    cc, rr = c[lo:lo + chunk], r[lo:lo + chunk]
    dx = xs.view(1, 1, 1, G) - cc[:, :, 0].reshape(-1, 3, 1, 1)
    dy = ys.view(1, 1, G, 1) - cc[:, :, 1].reshape(-1, 3, 1, 1)
    out[lo:lo + chunk, 0] = (torch.sqrt(dx * dx + dy * dy) - rr.reshape(-1, 3, 1, 1)).min(1).values.to(torch.float32)
  return out


def measured_traffic(key='gn_step'):
  """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json, written by profiles/tools/pmc_report.py
  with the guide's gfx950 FETCH_SIZE correction); None when no such measurement is committed for this workload."""
  f = os.path.join(ROOT, 'profiles', 'traffic.json')
  try:
    d = json.load(open(f))
    if key == 'gn_step' and 'hbm_bytes_per_launch' in d: return float(d['hbm_bytes_per_launch'])
    return float(d[key]['hbm_bytes_per_launch'])
  except (OSError, ValueError, KeyError, TypeError):
    return None


def kernel_stats():
  """Per-kernel static ISA statistics written by __graft_entry__.build() (profiles/tools/isa_stats.py)."""
  try:
    return json.load(open(os.path.join(ROOT, 'dgpmp2_amd', 'lib', 'kernel_stats.json')))
  except (OSError, ValueError):
    return {}


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


def cpu_model():
  try:
    for line in open('/proc/cpuinfo'):
      if line.lower().startswith('model name'):
        return line.split(':', 1)[1].strip()
  except OSError:
    pass
  return 'unknown'


@contextlib.contextmanager
def quiet_fds():
  """Silence C-level writes to stdout / stderr (MKL prints 'Intel oneMKL ERROR: Parameter 6 was incorrect on entry to DLASWP'
  once per matrix when batched torch.inverse fails on some hosts) so that nothing precedes the JSON line."""
  sys.stdout.flush(); sys.stderr.flush()
  saved = [os.dup(1), os.dup(2)]
  null = os.open(os.devnull, os.O_WRONLY)
  try:
    os.dup2(null, 1); os.dup2(null, 2)
    yield
  finally:
    sys.stdout.flush(); sys.stderr.flush()
    os.dup2(saved[0], 1); os.dup2(saved[1], 2)
    for fd in saved + [null]: os.close(fd)


def mem_available_gb():
  """Host memory this process may really take: min(MemAvailable, what the container's cgroup limit leaves) -- /proc/meminfo shows the HOST's memory inside a
  container, and a process that outgrows its cgroup is killed without warning."""
  avail = 0.0
  try:
    for line in open('/proc/meminfo'):
      if line.startswith('MemAvailable:'): avail = int(line.split()[1]) / (1 << 20)
  except (OSError, ValueError, IndexError):
    return 0.0
  for lim, use in (('/sys/fs/cgroup/memory.max', '/sys/fs/cgroup/memory.current'),
                   ('/sys/fs/cgroup/memory/memory.limit_in_bytes', '/sys/fs/cgroup/memory/memory.usage_in_bytes')):
    try:
      v = open(lim).read().strip()
      if v != 'max' and int(v) < (1 << 60):
        avail = min(avail, (int(v) - int(open(use).read().strip())) / (1 << 30))
    except (OSError, ValueError):
      pass
  return max(avail, 0.0)


def cpu_baseline(th_hist_cpu, start_cpu, goal_cpu, sdf_cpu, chunk=256, steps=3):
  """The reference's dense PyTorch-CPU op sequence (oracle/dense_torch.py, kind 'port'), fp64, all host cores.  SURVEY 8(d): the FULL 4096-trajectory batch in one
  call when the host has >= 40 GB available (the reference itself needs 30.5 GB RSS there; 1 warm-up + 1 timed step: a step takes ~7 s on 16 cores), otherwise a bounded
  sample: `chunk` of the 4096 trajectories, 1 warm-up + `steps` timed steps, scaled to whole-batch steps/s.  `sample` says which.  DGP_BENCH_CPU_FULL=0 / 1 overrides."""
  from oracle import dense_torch as DT
  from oracle.gpmp2_oracle import OracleParams
  cores = usable_cores()
  torch.set_num_threads(cores)
  p = OracleParams(dof=DOF, total_time_step=N_STATES - 1)
  P = DT.params_from_oracle(p)
  avail = mem_available_gb()
  env = os.environ.get('DGP_BENCH_CPU_FULL')
  full = (avail >= 40.0) if env is None else env == '1'
  B = B_PER_GPU if full else chunk
  if full: steps = 1
  qc = torch.from_numpy(p.static_covs(B)[0]); ow = torch.from_numpy(p.static_covs(B)[1]); eps = torch.from_numpy(p.static_covs(B)[2])
  sdf = sdf_cpu.double().expand(B, 1, GRID, GRID)
  # does this host's MKL accept batched torch.inverse (what the reference calls, plan_layer.py:227-228)?  Probe it on a tiny
  # batch with the C-level output silenced; if it does not, form the two explicit inverses with triangular solves against I.
  inverse_impl = 'torch.inverse'
  with quiet_fds():
    try:
      t = torch.eye(N_STATES * 2 * DOF, dtype=torch.float64).expand(2, -1, -1).contiguous().triu() + 0.0
      ok = bool(torch.isfinite(torch.inverse(t)).all())
    except RuntimeError:
      ok = False
    if ok:
      try:
        DT.plan_layer_forward(th_hist_cpu[0][:2].double(), start_cpu[:2].double(), goal_cpu[:2].double(), sdf[:2], qc[:2], ow[:2], eps[:2], P)
      except RuntimeError:
        ok = False
  if not ok:
    inverse_impl = 'torch.linalg.solve_triangular(u, I) (torch.inverse fails in MKL on this host)'
    DT.set_explicit_inverse('solve_triangular')
  ts = []
  with torch.no_grad():
    for k in range(steps + 1):
      th = th_hist_cpu[k % len(th_hist_cpu)][:B].double()
      t0 = time.perf_counter()
      DT.plan_layer_forward(th, start_cpu[:B].double(), goal_cpu[:B].double(), sdf, qc, ow, eps, P)
      ts.append(time.perf_counter() - t0)
  t_chunk = float(np.median(ts[1:]))
  sample = ('dense PyTorch-CPU fp64 restatement of PlanLayer.forward on the FULL batch of 4096 trajectories in one call (MemAvailable %.0f GB >= 40), 1 warm-up + %d timed '
            'step(s), %.3f s per step; explicit inverses via %s' % (avail, steps, t_chunk, inverse_impl)) if full else \
           ('dense PyTorch-CPU fp64 restatement of PlanLayer.forward on %d of the 4096 trajectories (MemAvailable %.0f GB < 40: not the full batch), 1 warm-up + %d timed '
            'steps, median %.3f s per %d-trajectory step, scaled by 4096/%d; explicit inverses via %s' % (B, avail, steps, t_chunk, B, B, inverse_impl))
  return {'value': 1.0 / (t_chunk * (B_PER_GPU / B)), 'unit': 'GN steps/s (batch 4096)', 'cores': cores, 'cpu_model': cpu_model(), 'kind': 'port',
          'sample': sample, 'full_batch': bool(full), 'torch_threads': torch.get_num_threads()}


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
  return {'value': 1.0 / t, 'unit': 'GN steps/s (batch 4096)', 'cores': cores, 'cpu_model': cpu_model(), 'kind': 'port',
          'sample': 'block-tridiagonal fp64 C restatement (oracle/gn_blocktri.c), full 4096-trajectory batch, 1 warm-up + %d timed '
                    'steps, median %.4f s' % (steps, t)}


def prewarm(launch, seconds=PREWARM_S, chunk=400):
  """Run `launch(k)` back to back for at least `seconds` of wall time (device kept busy: one synchronisation per chunk)."""
  t0 = time.perf_counter(); k = 0
  while True:
    for _ in range(chunk):
      launch(k); k += 1
    torch.cuda.synchronize()
    if time.perf_counter() - t0 >= seconds: break
  return time.perf_counter() - t0


def time_launches(launch, reps, warm_s=0.3):
  """Steady-clock average per-launch time in microseconds over `reps` launches, HIP events on the launch stream; the median of three
  such passes (a pass that shares the GPU with some one-off -- an allocator trim, a clock excursion -- used to land in the JSON as is)."""
  prewarm(launch, warm_s, 100)
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  us = []
  for _ in range(3):
    torch.cuda.synchronize(); e0.record()
    for k in range(reps): launch(k)
    e1.record(); torch.cuda.synchronize()
    us.append(e0.elapsed_time(e1) / reps * 1e3)
  return sorted(us)[1]


def traffic_provenance():
  """Where `roofline.traffic` comes from: NOT measured by this run (PMC counters need rocprofv3 around the process) but read from the committed counter passes."""
  try:
    d = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))
    c = d.get('collected', {})
    return ('HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/traffic.json; FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 corrections of '
            'the guide; collected %s at commit %s, profiles/tools/pmc_traffic.sh) -- not re-measured in this run' % (c.get('date', 'in round 5'), c.get('commit', '33f5ca1')))
  except (OSError, ValueError):
    return None


def roofline_block(bytes_per_launch, us, kernel, traffic_key=None, note=None):
  achieved = bytes_per_launch / (us * 1e-6) / 1e9
  r = {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
       'traffic': measured_traffic(traffic_key) if traffic_key else None, 'kernel': kernel, 'kernel_avg_ms': us * 1e-3,
       'algorithmic_bytes_per_launch': bytes_per_launch}
  if r['traffic'] is not None: r['traffic_source'] = traffic_provenance()
  if note: r['note'] = note
  return r


def sdf_fields_rate(device, batch=256):
  """dgp_sdf_2d (csrc/sdf_edt.hip): the signed distance fields of `batch` occupancy images of the benchmark's grid size in one call, next to the reference's host path
  (utils/sdf_utils.py:6-21, two scipy distance transforms per image) timed on one image on this host."""
  from dgpmp2_amd.utils import sdf_utils
  rs = np.random.RandomState(5)
  yy, xx = np.ogrid[:GRID, :GRID]
  ims = np.ones((batch, GRID, GRID), dtype=np.float32)
  for b in range(batch):
    for _ in range(3 + b % 5):
      cy, cx, r = rs.randint(0, GRID), rs.randint(0, GRID), rs.randint(5, GRID // 8)
      ims[b][(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 0.0
  d = torch.as_tensor(ims).to(device)
  res = 10.0 / GRID
  for _ in range(3): out = sdf_utils.sdf_2d_batch(d, padlen=1, res=res)
  torch.cuda.synchronize(device)
  t0 = time.perf_counter()
  for _ in range(10): out = sdf_utils.sdf_2d_batch(d, padlen=1, res=res)
  torch.cuda.synchronize(device)
  us = (time.perf_counter() - t0) / 10 * 1e6
  t0 = time.perf_counter(); ref = sdf_utils.sdf_2d(ims[0], padlen=1, res=res); host_ms = (time.perf_counter() - t0) * 1e3
  alg = batch * (GRID * GRID * 4 + (GRID + 2) ** 2 * 8)
  return {'workload': '%d occupancy images of %dx%d (3-7 random discs), padlen 1, float32 in, float64 out' % (batch, GRID, GRID), 'us_per_call': us, 'us_per_image': us / batch,
          'host_scipy_ms_per_image': host_ms, 'bit_identical_to_host': bool(np.array_equal(out[0].cpu().numpy(), ref)),
          'roofline': {'bound': 'hbm', 'achieved': alg / us * 1e-3, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': alg / us * 1e-3 / HBM_PEAK_GBS, 'traffic': None},
          'note': 'algorithmic bytes = image in + field out per padded pixel; the call is bound by the row pass, whose search length is the distance itself (DESIGN.md sections 3 and 9; round 6: '
                  'bit-plane column pass + eight offsets per trip of the row search, profiles/r06_sdf_edt_ab.txt)'}


def extra_workloads(device, stream, reps=1500):
  """BASELINE configs[2] (velocity limits) and configs[3] (non-holonomic x,y,theta robot, 512x512 SDF), and configs[1] in its
  two other input regimes (one SDF per trajectory; per-state covariance tensors streamed): steady-clock kernel time of
  dgp_gn_step at B = 4096 x 64 states, each with its own SURVEY 8(d) algorithmic byte count."""
  from dgpmp2_amd import _capi
  from dgpmp2_amd.gpmp2.plan_layer import solver_config
  B, n = B_PER_GPU, N_STATES
  out = {}

  def run(tag, dof, G, cfg_kw, sdf=None, sdf_stride=0, covs=False, note=None, traffic_key=None, layout=0):
    d = 2 * dof
    th0, start, goal, sdf_shared = make_inputs(B, n, G, device, seed=0, dof=dof)
    s = _capi.Solver(solver_config(num_states=n, dof=dof, io_dtype=torch.float32, **cfg_kw))
    grids = [sdf_shared] if sdf is None else sdf
    sas = [s.sdf_arg(g_.data_ptr(), G, G, sdf_stride, layout=layout) for g_ in grids]
    sa = sas[0]
    dth = torch.empty_like(th0); err = torch.empty(B, device=device); eex = torch.empty(B, device=device)
    info = torch.zeros(B, dtype=torch.int32, device=device)
    cv, keep = None, []
    if covs:
      qc = torch.eye(dof, device=device).expand(B, n - 1, dof, dof).contiguous(); ow = torch.full((B, n), 1e4, device=device)
      ep = torch.full((B, n), float(cfg_kw.get('epsilon_dist', 0.4)), device=device); keep = [qc, ow, ep]
      cv = s.covs_arg(_capi.DGP_QC_PERSTATE, qc.data_ptr(), ow.data_ptr(), ep.data_ptr())
    # a few GN iterations so that the timed inputs are not the straight line (hinge active on a realistic share of states)
    ths = [th0]
    for _ in range(3):
      s.gn_step(B, ths[-1].data_ptr(), start.data_ptr(), goal.data_ptr(), sa, cv, dth.data_ptr(), err.data_ptr(), eex.data_ptr(), info.data_ptr(), stream)
      ths.append(ths[-1] + dth)
    torch.cuda.synchronize()
    assert int(info.abs().max()) == 0 and bool(torch.isfinite(ths[-1]).all()), tag
    ptrs = [t.data_ptr() for t in ths]
    sp, gp, dp, ep_, xp, ip = start.data_ptr(), goal.data_ptr(), dth.data_ptr(), err.data_ptr(), eex.data_ptr(), info.data_ptr()
    us = time_launches(lambda k: s.gn_step(B, ptrs[k % 4], sp, gp, sas[k % len(sas)], cv, dp, ep_, xp, ip, stream), reps)
    by = algorithmic_bytes_per_trajectory(n, d, cov_tensors=covs) * B
    lpt, c = s.launch_shape(B)
    kname = 'gn_kernel<%d,%d,%d,float,0,%d>' % (dof, lpt, c, 2 if covs else s.step_kernel_variant(B))      # <dof, LPT, C, io, MODE_STEP, QK: 1 static, 2 per-state Kronecker, 3 Woodbury>
    out[tag] = {'workload': note, 'kernel_avg_us': us, 'gn_steps_per_s': 1e6 / us,
                'roofline': roofline_block(by, us, kname, traffic_key=traffic_key)}
    del keep

  run('config3_vel_limits', 2, GRID, dict(use_vel_limits=True, K_v=0.01, v_x=1.0, v_y=1.0),
      note='BASELINE configs[2]: 2D point robot + velocity-limit factors, batch=4096, 64 states, 256x256 shared SDF')
  run('config4_xyh', 3, 512, dict(non_holonomic=True, K_d=0.01, epsilon_dist=0.2, reg=0.0), traffic_key='config4_xyh',
      note='BASELINE configs[3]: non-holonomic (x,y,theta) robot, 6-dim state, batch=4096, 64 states, 512x512 shared SDF')
  run('learned_covariances', 2, GRID, {}, covs=True, traffic_key='learned_covariances',
      note='configs[1] with per-state qc_inv (B,n-1,2,2), obs_w, eps tensors streamed (the learned mode; generic kernels)')
  ps = [make_per_sample_sdfs(B, GRID, device, seed=1 + i) for i in range(6)]
  run('per_sample_sdf', 2, GRID, {}, sdf=ps, sdf_stride=GRID * GRID, traffic_key='per_sample_sdf',
      note='configs[1] with one 256x256 SDF PER trajectory (the reference API shape sdfb (B,1,H,W)): 1 GiB of grids per batch, six '
           'different batches of grids cycled so that the tap lines come from HBM, not from the 256 MiB Infinity Cache')
  # the same six batches of grids stored as 4 x 4 tiles (DgpSdf::layout = DGP_SDF_TILED4; utils.sdf_utils.tile_sdf, or written directly by dgp_sdf_2d): the 2 x 2 footprint of a
  # lookup in one 64-byte tile instead of two rows a kilobyte apart -- 29 instead of 70 distinct lines per trajectory
  from dgpmp2_amd.utils.sdf_utils import tile_sdf
  pt = [tile_sdf(g_) for g_ in ps]
  run('per_sample_sdf_tiled', 2, GRID, {}, sdf=pt, sdf_stride=GRID * GRID, layout=_capi.DGP_SDF_TILED4, traffic_key='per_sample_sdf_tiled',
      note='per_sample_sdf with the grids stored as 4 x 4 tiles (an API extension: utils.sdf_utils.tile_sdf(sdfb) / sdf_2d_batch(layout="tiled4") in place of sdfb)')
  del pt
  # the regime a GN loop is in: step k+1 reads (nearly) the lines step k read -- ONE batch of grids, the trajectory inputs cycling
  run('per_sample_sdf_same_grids', 2, GRID, {}, sdf=ps[:1], sdf_stride=GRID * GRID,
      note='configs[1] with one 256x256 SDF per trajectory, the SAME 4096 grids every step (GN iteration k+1 on the grids of iteration k): '
           'the tap lines of the previous step are still in the Infinity Cache / L2')
  # ... and the 10 GN iterations as ONE launch of the fused loop on 4096 distinct grids (DiffGPMP2Planner.forward on per-sample grids)
  th0, start, goal, _ = make_inputs(B, n, GRID, device, seed=0, dof=2)
  s = _capi.Solver(solver_config(num_states=n, dof=2, io_dtype=torch.float32))
  tho = torch.empty_like(th0); its = torch.zeros(B, dtype=torch.int32, device=device); info = torch.zeros(B, dtype=torch.int32, device=device)
  eh = torch.empty(B, GN_ITERS, device=device); eeh = torch.empty(B, GN_ITERS, device=device)
  sas = [s.sdf_arg(g_.data_ptr(), GRID, GRID, GRID * GRID) for g_ in ps]
  tp, sp, gp = th0.data_ptr(), start.data_ptr(), goal.data_ptr()
  fus = time_launches(lambda k: s.gn_solve(B, tp, sp, gp, sas[k % len(sas)], None, GN_ITERS, 0.0, tho.data_ptr(), its.data_ptr(), eh.data_ptr(),
                                           eeh.data_ptr(), None, info.data_ptr(), stream), 120, warm_s=0.2)
  assert int(its.min()) == GN_ITERS and int(info.abs().max()) == 0
  by = algorithmic_bytes_per_trajectory(n, 4) * B       # per GN iteration, as if every iteration were a step() (the fused loop moves LESS: th stays on chip)
  out['per_sample_sdf_fused_forward'] = {
      'workload': 'configs[1] with one 256x256 SDF per trajectory, the 10 GN iterations as ONE dgp_gn_solve launch, six batches of grids cycled (cold '
                  'first iteration, iterations 2..10 re-touch the lines of the first)',
      'ms_per_launch': fus * 1e-3, 'us_per_gn_iteration': fus / GN_ITERS, 'gn_steps_per_s': GN_ITERS / (fus * 1e-6),
      'roofline': roofline_block(by, fus / GN_ITERS, 'gn_kernel<2,16,4,float,1,3>', note='algorithmic bytes of ONE step() per GN iteration')}
  del ps
  return out


def two_stream_rate(solver, B, th_ptrs, sp, gp, sdf_arg, device, reps=2000):
  """Two INDEPENDENT whole-batch problems in flight (two planners, e.g. two environment sets): the headline launches alternating
  between two HIP streams, each with its own output buffers.  A single GN loop is sequential (step k+1 needs step k), so this is NOT
  the headline; it shows how much of the ~2 us between dependent launches (end-of-kernel release, dispatch) two streams hide."""
  streams = [torch.cuda.Stream(device=device) for _ in range(2)]
  outs = []
  for _ in range(2):
    outs.append((torch.empty(B, N_STATES, 2 * DOF, device=device), torch.empty(B, device=device), torch.empty(B, device=device),
                 torch.zeros(B, dtype=torch.int32, device=device)))
  raw = [ctypes.c_void_p(s_.cuda_stream) for s_ in streams]
  ptrs = [tuple(t.data_ptr() for t in o) for o in outs]

  def launch(k):
    i = k & 1
    solver.gn_step(B, th_ptrs[k % GN_ITERS], sp, gp, sdf_arg, None, ptrs[i][0], ptrs[i][1], ptrs[i][2], ptrs[i][3], raw[i])

  torch.cuda.synchronize()
  for k in range(2000): launch(k)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for k in range(reps): launch(k)
  torch.cuda.synchronize()
  us = (time.perf_counter() - t0) / reps * 1e6
  return {'us_per_step': us, 'gn_steps_per_s': 1e6 / us, 'streams': 2,
          'note': 'two independent 4096-trajectory batches, launches alternating between two HIP streams (wall time over %d launches); not the headline: '
                  'one GN loop is a chain of dependent launches' % reps}


def planner_api_rate(device, reps=300):
  """DiffGPMP2Planner.step() through the Python mirror (reference param dicts, autograd Function, info buffer), no_grad:
  wall microseconds per call at B = 4096 -- what a caller of the reference API sees, next to the C-ABI kernel rate."""
  from dgpmp2_amd.robot_models import PointRobot2D
  from dgpmp2_amd.gpmp2 import DiffGPMP2Planner
  B, n = B_PER_GPU, N_STATES
  t = lambda v: torch.tensor(v, dtype=torch.float64)
  gp = {'Q_c_inv': torch.eye(2, dtype=torch.float64), 'K_s': t(0.01), 'K_g': t(0.01)}
  ob = {'cost_sigma': t(0.01), 'epsilon_dist': t(0.4)}
  pp = {'dof': 2, 'state_dim': 4, 'total_time_sec': 10.0, 'total_time_step': n - 1}
  op = {'method': 'gauss_newton', 'reg': 0.1, 'max_iters': GN_ITERS, 'tol_err': 1e-3, 'tol_delta': 1e-4}
  planner = DiffGPMP2Planner(gp, ob, pp, op, {'x_lims': [-5.0, 5.0], 'y_lims': [-5.0, 5.0]}, PointRobot2D(t(0.4), B, n, use_cuda=True),
                             batch_size=B, use_cuda=True)
  th0, start, goal, sdf = make_inputs(B, n, GRID, device)
  sdfb = sdf.expand(B, 1, GRID, GRID)
  best = float('inf')
  with torch.no_grad():
    for _ in range(200): planner.step(th0, start, goal, None, sdfb)
    for _ in range(3):                               # best of three batches: a single batch picks up host jitter (GC, CPU clock ramp)
      torch.cuda.synchronize(); t0 = time.perf_counter()
      for _ in range(reps): planner.step(th0, start, goal, None, sdfb)
      torch.cuda.synchronize()
      best = min(best, (time.perf_counter() - t0) / reps * 1e6)
  us = best
  return {'us_per_call': us, 'gn_steps_per_s': 1e6 / us, 'note': 'DiffGPMP2Planner.step() under torch.no_grad(), B=4096, wall time per call '
          '(host-side Python + ctypes + one kernel launch); the headline `value` is the C-ABI launch rate'}


def planner_api_backward_rate(device, reps=200):
  """DiffGPMP2Planner.step() + the backward pass through it, wall microseconds per (forward + backward) at B = 4096, through the
  Python mirror and torch autograd -- (a) static covariances, gradient w.r.t. the trajectory (a TBPTT link); (b) the learning loop's
  shape: per-state qc_inv / obscov_inv / eps tensors that require grad (what learn modules emit, handed to PlanLayer.forward as
  diff_gpmp2_planner.py:200 does), gradients w.r.t. all of them and the trajectory."""
  from dgpmp2_amd.robot_models import PointRobot2D
  from dgpmp2_amd.gpmp2 import DiffGPMP2Planner
  B, n = B_PER_GPU, N_STATES
  t = lambda v: torch.tensor(v, dtype=torch.float64)
  gp = {'Q_c_inv': torch.eye(2, dtype=torch.float64), 'K_s': t(0.01), 'K_g': t(0.01)}
  ob = {'cost_sigma': t(0.01), 'epsilon_dist': t(0.4)}
  pp = {'dof': 2, 'state_dim': 4, 'total_time_sec': 10.0, 'total_time_step': n - 1}
  op = {'method': 'gauss_newton', 'reg': 0.1, 'max_iters': GN_ITERS, 'tol_err': 1e-3, 'tol_delta': 1e-4}
  planner = DiffGPMP2Planner(gp, ob, pp, op, {'x_lims': [-5.0, 5.0], 'y_lims': [-5.0, 5.0]}, PointRobot2D(t(0.4), B, n, use_cuda=True),
                             batch_size=B, use_cuda=True)
  th0, start, goal, sdf = make_inputs(B, n, GRID, device)
  sdfb = sdf.expand(B, 1, GRID, GRID)
  thr = th0.clone().requires_grad_(True)
  g = torch.randn_like(th0)
  qc = torch.eye(2, device=device).expand(B, n - 1, 2, 2).contiguous().requires_grad_(True)
  ow = torch.full((B, n, 1, 1), 1e4, device=device, requires_grad=True)
  ep = torch.full((B, n, 1, 1), 0.4, device=device, requires_grad=True)

  def static_fb():
    dth = planner.step(thr, start, goal, None, sdfb)[0]
    torch.autograd.grad(dth, thr, g)

  def learned_fb():
    dth = planner.plan_layer(thr, start, goal, None, sdfb, qc, ow, ep)[0]
    torch.autograd.grad(dth, (thr, qc, ow, ep), g)

  def wall(f):
    best = float('inf')
    for _ in range(30): f()
    for _ in range(3):
      torch.cuda.synchronize(); t0 = time.perf_counter()
      for _ in range(reps): f()
      torch.cuda.synchronize()
      best = min(best, (time.perf_counter() - t0) / reps * 1e6)
    return best

  def tbptt_fb(K=10):      # a truncated-BPTT window as learning/train_planner.py:297-374 builds it: K chained steps, one backward
    x = thr
    for _ in range(K): x = x + planner.step(x, start, goal, None, sdfb)[0]
    torch.autograd.grad(x, thr, g)

  # one iteration of the reference's training loop (learning/train_planner.py:311-327, 366): step, the unweighted errors at th + dtheta, backward of
  # both -- as the two calls of the reference API (two autograd nodes) and as planner.step_with_errors (one node; dgp_gn_step_errors[_backward])
  cw = torch.randn(B, 1, 1, device=device)
  cws = cw.view(B, 1).contiguous()
  leaves = (thr, qc, ow, ep)

  def train_iteration_two_calls():
    dth = planner.plan_layer(thr, start, goal, None, sdfb, qc, ow, ep)[0]
    sg, gp_, ob = planner.unweighted_errors_batch(thr + dth, sdfb)
    torch.autograd.grad((dth, sg, gp_, ob), leaves, (g, cws, cw, cw))      # (cotangents handed over directly: no loss arithmetic in the measurement)

  def train_iteration_fused():
    dth, _, _, sg, gp_, ob = planner.plan_layer.forward_with_errors(thr, start, goal, None, sdfb, qc, ow, ep)
    torch.autograd.grad((dth, sg, gp_, ob), leaves, (g, cws, cw, cw))

  # the same iteration in the reference's DEFAULT learned mode, dynamics_mode 'diag_identity' (diff_gpmp2_planner.py:255-258): the learn module's output vector ->
  # get_covariances -> q_k^2 I blocks (tagged with their scalars: DGP_QC_SCALAR, the static kernels with scaled lane masks) + obstacle weights; gradients w.r.t.
  # the trajectory and the module output
  lm_out = torch.cat([torch.ones(B, 1, n - 1, device=device), torch.full((B, 1, n), 100.0, device=device)], dim=2).requires_grad_(True)

  with torch.no_grad():
    qc_t, ow_t = planner.get_covariances(lm_out, 'diag_identity')
  tag = qc_t.__dict__['_dgp_scalar'][0]
  qc_l = qc_t.detach().clone().requires_grad_(True); ow_l = ow_t.detach().clone().requires_grad_(True)
  qc_l.__dict__['_dgp_scalar'] = (tag, qc_l._version)      # (what get_covariances puts on the tensor it returns)

  def train_iteration_diag_identity_layer_only():      # the layer's share: the tagged blocks as leaves, no covariance construction in the measurement
    dth, _, _, sg, gp_, ob = planner.plan_layer.forward_with_errors(thr, start, goal, None, sdfb, qc_l, ow_l, None)
    torch.autograd.grad((dth, sg, gp_, ob), (thr, qc_l, ow_l), (g, cws, cw, cw))

  def train_iteration_diag_identity():
    qc_s, ow_s = planner.get_covariances(lm_out, 'diag_identity')
    dth, _, _, sg, gp_, ob = planner.plan_layer.forward_with_errors(thr, start, goal, None, sdfb, qc_s, ow_s, None)
    torch.autograd.grad((dth, sg, gp_, ob), (thr, lm_out), (g, cws, cw, cw))

  def train_iteration_diag_identity_raw():      # round 5: the module output goes to the kernels as it is (DGP_COVS_SQUARED) -- what planner.step_with_errors() does in this mode
    raw = planner.plan_layer.raw_covs(lm_out, 'diag_identity', False)
    dth, _, _, sg, gp_, ob = planner.plan_layer.forward_raw(thr, start, goal, None, sdfb, raw, with_errors=True)[:6]
    torch.autograd.grad((dth, sg, gp_, ob), (thr, lm_out), (g, cws, cw, cw))

  # planner.forward with the graph kept (examples/diff_gpmp2_2d_example.py:77): 10 GN iterations + the backward pass through all of them, two launches
  sdf_leaf = sdf.clone().requires_grad_(True)
  planner.optim_params['tol_delta'] = 0.0            # all 10 iterations, as the fused_forward block

  def forward_backward():
    thf = planner.forward(thr, start, goal, None, sdf_leaf.expand(B, 1, GRID, GRID))[0]
    torch.autograd.grad(thf, (thr, sdf_leaf), g)

  # ... and the two launches behind it on their own (C-ABI, HIP events): the traced fused loop and the chain backward
  from dgpmp2_amd import _capi
  sv = planner.plan_layer._solver(torch.float32)
  hist = torch.empty((GN_ITERS, B, n, 4), dtype=torch.float64, device=device)
  tho = torch.empty_like(th0); its = torch.zeros(B, dtype=torch.int32, device=device); inf = torch.zeros(B, dtype=torch.int32, device=device)
  gth = torch.empty_like(th0); gst = torch.empty_like(start); ggo = torch.empty_like(goal)
  sarg = sv.sdf_arg(sdf.data_ptr(), GRID, GRID, 0)
  raw = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
  eh = torch.empty(B, GN_ITERS, dtype=torch.float32, device=device); eeh = torch.empty_like(eh); ef = torch.empty(B, dtype=torch.float32, device=device)
  k_fwd = time_launches(lambda k: sv.gn_solve_traced(B, th0.data_ptr(), start.data_ptr(), goal.data_ptr(), sarg, None, GN_ITERS, 0.0, tho.data_ptr(), its.data_ptr(),
                                                     eh.data_ptr(), eeh.data_ptr(), ef.data_ptr(), inf.data_ptr(), hist.data_ptr(), raw), 100, warm_s=0.1)
  k_bwd = time_launches(lambda k: sv.gn_solve_backward(B, start.data_ptr(), goal.data_ptr(), sarg, GN_ITERS, hist.data_ptr(), tho.data_ptr(), its.data_ptr(),
                                                       g.data_ptr(), gth.data_ptr(), gst.data_ptr(), ggo.data_ptr(), None, 0, raw), 100, warm_s=0.1)
  # ... with the gradient of the shared grid, as forward_backward() asks for it (what PlanLayer launches: eight float64 partial grids, zero-filled, accumulated by the chain
  # kernel, summed and cast by dgp_sum_partial_grids) -- the figure us_per_call is to be held against
  copies = 8                                       # (plan_layer._SDF_GRAD_COPIES)
  gpart = torch.empty((copies, 1, GRID, GRID), dtype=torch.float64, device=device); gsum = torch.empty((1, 1, GRID, GRID), dtype=torch.float32, device=device)
  sarg64 = sv.sdf_arg(sdf.data_ptr(), GRID, GRID, 0, grad_mode=_capi.DGP_GSDF_DENSE_F64)
  pc = _capi.get_pycall()

  def bwd_with_grid(k):
    gpart.zero_()
    sv.gn_solve_backward(B, start.data_ptr(), goal.data_ptr(), sarg64, GN_ITERS, hist.data_ptr(), tho.data_ptr(), its.data_ptr(), g.data_ptr(), gth.data_ptr(),
                         gst.data_ptr(), ggo.data_ptr(), gpart.data_ptr(), 0, raw, g_sdf_copies=copies)
    pc.sum_partial_grids(gpart.data_ptr(), _capi.DGP_F64, copies, GRID * GRID, 1.0, gsum.data_ptr(), _capi.DGP_F32, raw.value)
  k_bwd_grid = time_launches(bwd_with_grid, 100, warm_s=0.1)

  def graphed(f):
    """f (forward + torch.autograd.grad) captured once in a HIP graph (torch.cuda.CUDAGraph, torch's whole-iteration capture recipe: warm-up on a side
    stream, then capture) -> the replay callable.  Replay re-issues the recorded launches without Python, Function.apply or the autograd engine."""
    side = torch.cuda.Stream(device)
    side.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(side):
      for _ in range(3): f()
    torch.cuda.current_stream(device).wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
      f()
    return gr.replay

  a, b = wall(static_fb), wall(learned_fb)
  t2, t1 = wall(train_iteration_two_calls), wall(train_iteration_fused)
  tdi = wall(train_iteration_diag_identity)
  tdr = wall(train_iteration_diag_identity_raw)
  gdr = None
  try:
    ga, gt1, gk = wall(graphed(static_fb)), wall(graphed(train_iteration_fused)), wall(graphed(tbptt_fb)) / 10.0
    gdi = wall(graphed(train_iteration_diag_identity)); gdl = wall(graphed(train_iteration_diag_identity_layer_only))
    gdr = wall(graphed(train_iteration_diag_identity_raw))
  except Exception as e:      # noqa: BLE001  (measurement extra: a torch build without graph capture must not cost the bench line)
    print('bench: HIP-graph capture of the training iteration failed (%s: %s)' % (type(e).__name__, e), file=sys.stderr)
    ga = gt1 = gk = gdi = gdl = None
  # round 6: the SAME iteration called eagerly through planner.graphed_iteration -- a loop that hands in fresh (th, qc_inv, obscov_inv, eps) every call, as a training
  # loop does: the helper copies them into the graph's static tensors and replays (utils/graph_utils.py); held against the hand-written replay figure above
  helper_us = None
  try:
    def iteration_fn(th_, qc_, ow_, ep_):
      dth, _, _, sg, gp_, ob = planner.plan_layer.forward_with_errors(th_, start, goal, None, sdfb, qc_, ow_, ep_)
      return torch.autograd.grad((dth, sg, gp_, ob), (th_, qc_, ow_, ep_), (g, cws, cw, cw))
    it = planner.graphed_iteration(iteration_fn)
    fresh = [(thr.detach().clone().requires_grad_(True), qc.detach().clone().requires_grad_(True), ow.detach().clone().requires_grad_(True),
              ep.detach().clone().requires_grad_(True)) for _ in range(2)]
    kk = [0]

    def through_helper():
      kk[0] ^= 1
      it(*fresh[kk[0]])
    helper_us = wall(through_helper)
  except Exception as e:      # noqa: BLE001
    print('bench: graphed_iteration failed (%s: %s)' % (type(e).__name__, e), file=sys.stderr)
  reps = max(20, reps // 10)
  c = wall(tbptt_fb) / 10.0
  fb = wall(forward_backward)
  return {'us_per_call': a, 'learned_covariances_us_per_call': b, 'tbptt_window10_us_per_step': c,
          'hip_graph_replay': {'us_per_call': ga, 'train_iteration_us': gt1, 'tbptt_window10_us_per_step': gk,
                               'note': 'the same forward + backward callables captured ONCE in a HIP graph (torch.cuda.CUDAGraph) and replayed: the recorded launches '
                                       'without Python, Function.apply or the autograd engine -- what is left is the kernels (results bit-identical to the eager calls, '
                                       'tests/test_planner_api.py)'},
          'train_iteration_api': {'two_calls_us': t2, 'step_with_errors_us': t1, 'step_with_errors_hip_graph_replay_us': gt1,
                                  'step_with_errors_through_graphed_iteration_us': helper_us,      # eager-style calls with fresh inputs: 4 input copies + one replay
                                  'diag_identity_mode': {'step_with_errors_us': tdi, 'hip_graph_replay_us': gdi, 'layer_only_hip_graph_replay_us': gdl,
                                                         'raw_module_output_us': tdr, 'raw_module_output_hip_graph_replay_us': gdr,
                                                         'note': "the reference's default learned mode: get_covariances(out, 'diag_identity') -> one scalar per GP factor -> DGP_QC_SCALAR "
                                                                 '(scaled-mask static kernels, forward and backward).  get_covariances and its backward -- a dozen small torch kernels: slices, q q^T, x I -- are inside the first two '
                                                                 'figures and cost more than the solver; layer_only: the tagged blocks as leaves; raw_module_output (round 5, what planner.step_with_errors() runs in this mode): the output vector '
                                                                 'itself is the kernels\' covariance input (DGP_COVS_SQUARED: squared in-kernel, d/d out written by the backward kernel) -- covariance construction included'},
                                  'note': 'learned covariances (per-state qc_inv / obscov_inv / eps that require grad): step + unweighted errors at th + dtheta + '
                                          'backward of a loss on all four outputs w.r.t. all four inputs, wall per iteration -- PlanLayer.forward + '
                                          'unweighted_errors_batch (2 + 2 launches, two autograd nodes) against PlanLayer.forward_with_errors (one node, one '
                                          'C-ABI call each way: dgp_gn_step_errors / dgp_gn_step_errors_backward, ONE launch each since round 5 -- the step kernels with an errors epilogue, the errors\' '
                                          'backward as a prologue of the step\'s backward kernel)'},
          'forward_backward_fused': {'us_per_call': fb, 'us_per_gn_iteration': fb / GN_ITERS, 'gn_iterations': GN_ITERS,
                                     'kernel_us': {'dgp_gn_solve_traced': k_fwd, 'dgp_gn_solve_backward': k_bwd, 'per_gn_iteration': (k_fwd + k_bwd) / GN_ITERS,
                                                   'dgp_gn_solve_backward_with_grid_gradient': k_bwd_grid, 'sum_as_called': k_fwd + k_bwd_grid},
                                     'wall_over_kernels': fb / (k_fwd + k_bwd_grid),
                                     'note': 'DiffGPMP2Planner.forward with requires_grad inputs + torch.autograd.grad through all 10 iterations w.r.t. the '
                                             'initial trajectory and the grid: dgp_gn_solve_traced + dgp_gn_solve_backward, one launch each (us_per_call: wall, with '
                                             'the host side of forward() -- one device-to-host copy of the per-sample errors, which waits for the forward launch, and the '
                                             'reference API\'s python lists; kernel_us: the launches by HIP events -- the forward with its error outputs; the backward without and, _with_grid_gradient, with '
                                             'the shared grid\'s gradient as this call asks for it: zero fill of the eight float64 partial grids + the chain kernel + '
                                             'dgp_sum_partial_grids; wall_over_kernels = us_per_call / sum_as_called)'},
          'note': 'wall time of DiffGPMP2Planner.step() + torch.autograd.grad through it, B=4096: static covariances with the gradient w.r.t. the '
                  'trajectory (us_per_call), and per-state qc_inv / obscov_inv / eps tensors with gradients w.r.t. all four (learned_covariances_us_per_call); '
                  'two kernel launches (dgp_gn_step, dgp_gn_step_backward) + the autograd engine.  tbptt_window10_us_per_step: ten chained steps '
                  '(th <- th + dtheta) and ONE backward through all of them, per step -- the fixed cost of entering the autograd engine (~40 us for a '
                  'trivial custom Function on this host) is paid once per window, as in the training loop'}


def _bench_planner(B, n):
  from dgpmp2_amd.robot_models import PointRobot2D
  from dgpmp2_amd.gpmp2 import DiffGPMP2Planner
  t = lambda v: torch.tensor(v, dtype=torch.float64)
  gp = {'Q_c_inv': torch.eye(2, dtype=torch.float64), 'K_s': t(0.01), 'K_g': t(0.01)}
  ob = {'cost_sigma': t(0.01), 'epsilon_dist': t(0.4)}
  pp = {'dof': 2, 'state_dim': 4, 'total_time_sec': 10.0, 'total_time_step': n - 1}
  op = {'method': 'gauss_newton', 'reg': 0.1, 'max_iters': GN_ITERS, 'tol_err': 1e-3, 'tol_delta': 1e-4}
  return DiffGPMP2Planner(gp, ob, pp, op, {'x_lims': [-5.0, 5.0], 'y_lims': [-5.0, 5.0]}, PointRobot2D(t(0.4), B, n, use_cuda=True),
                          batch_size=B, use_cuda=True)


def train_iteration_sdf_grad(device, batches=(B_PER_GPU, 32), reps=100):
  """One iteration of the reference's training loop WITH the gradient set that loop really asks for (learning/train_planner.py:267-268,
  311-327,366-374): `sdf_b.requires_grad_(True)` on PER-SAMPLE grids (B,1,H,W) and `th.requires_grad_(True)`, learned per-state
  covariance tensors -- planner.step_with_errors + backward of (dtheta, err_sg, err_gp, err_obs) w.r.t. th, qc_inv, obscov_inv, eps
  AND sdf.  Wall microseconds per iteration, eager and replayed from a HIP graph, at B = 4096 and B = 32, for per-sample grids (the
  gradient is a (B,1,H,W) tensor in the reference: 1 GiB at B = 4096 -- `sdf_grad` says in which form the layer returns it) and for a
  shared grid; `no_sdf_grad` is the same iteration without the grid gradient."""
  n = N_STATES
  res = {}
  for B in batches:
    planner = _bench_planner(B, n)
    pl = planner.plan_layer
    th0, start, goal, sdf = make_inputs(B, n, GRID, device)
    g = torch.randn_like(th0); cw = torch.randn(B, 1, 1, device=device); cws = cw.view(B, 1).contiguous()
    thr = th0.clone().requires_grad_(True)
    qc = torch.eye(2, device=device).expand(B, n - 1, 2, 2).contiguous().requires_grad_(True)
    ow = torch.full((B, n, 1, 1), 1e4, device=device, requires_grad=True)
    ep = torch.full((B, n, 1, 1), 0.4, device=device, requires_grad=True)
    grids = {'per_sample': make_per_sample_sdfs(B, GRID, device, seed=1).requires_grad_(True),
             'shared': sdf.clone().requires_grad_(True)}
    if B == B_PER_GPU:      # the same per-sample grids stored as 4 x 4 tiles (API extension; the gradient: a sparse tensor of the tiled tensor's shape)
      from dgpmp2_amd.utils.sdf_utils import tile_sdf
      grids['per_sample_tiled'] = tile_sdf(grids['per_sample'].detach()).requires_grad_(True)
    # the layer's DEFAULT gradient layout for per-sample grids since round 6 is the reference's dense (B,1,H,W) tensor; the rows above opt into sdf_grad = 'auto'
    # (sparse taps for large leaf grids).  One row with the default, so that the price of the reference layout stays visible (1 GiB zero fill at B = 4096)
    grids['per_sample_dense_default'] = grids['per_sample']

    def iteration(sdf_in, with_sdf):
      sdfb = sdf_in if sdf_in.shape[0] == B else sdf_in.expand(B, 1, GRID, GRID)
      if not with_sdf: sdfb = sdfb.detach()
      dth, _, _, sg, gp_, ob = pl.forward_with_errors(thr, start, goal, None, sdfb, qc, ow, ep)
      leaves = (thr, qc, ow, ep) + ((sdf_in,) if with_sdf else ())
      return torch.autograd.grad((dth, sg, gp_, ob), leaves, (g, cws, cw, cw))

    def wall(f):
      best = float('inf')
      for _ in range(10): f()
      for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): f()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps * 1e6)
      return best

    def graphed(f):
      side = torch.cuda.Stream(device)
      side.wait_stream(torch.cuda.current_stream(device))
      with torch.cuda.stream(side):
        for _ in range(3): f()
      torch.cuda.current_stream(device).wait_stream(side)
      gr = torch.cuda.CUDAGraph()
      with torch.cuda.graph(gr):
        f()
      return gr.replay

    blk = {}
    for name, leaf in grids.items():
      row = {}
      pl.sdf_grad = 'dense' if name.endswith('_dense_default') else 'auto'
      for tag, with_sdf in (('no_sdf_grad', False), ('sdf_grad', True)):
        f = lambda leaf=leaf, with_sdf=with_sdf: iteration(leaf, with_sdf)
        out = f()
        e = {'eager_us': wall(f)}
        if with_sdf:
          gs = out[-1]
          e['grad_layout'] = str(gs.layout).replace('torch.', '')
          e['grad_bytes'] = int(gs._values().numel() * gs._values().element_size() + gs._indices().numel() * 8) if gs.is_sparse else int(gs.numel() * gs.element_size())
        try:
          e['hip_graph_replay_us'] = wall(graphed(f))
        except Exception as ex:      # noqa: BLE001
          e['hip_graph_replay_us'] = None; e['graph_error'] = '%s: %s' % (type(ex).__name__, str(ex)[:200])
          torch.cuda.synchronize()
        row[tag] = e
      blk[name] = row
    res['B%d' % B] = blk
    del grids, planner
    torch.cuda.empty_cache()
  res['note'] = ("plan_layer.sdf_grad = 'auto' (opt-in: sparse tap gradients for large leaf grids) except in the *_dense_default rows (the layer's default, the reference's dense layout); "
                 'planner.plan_layer.forward_with_errors + torch.autograd.grad of (dtheta, err_sg, err_gp, err_obs) w.r.t. (th, qc_inv, obscov_inv, eps[, sdf]); '
                 'per_sample: sdfb (B,1,256,256) leaf as learning/train_planner.py:267; shared: one (1,1,256,256) leaf expand()ed over the batch; wall us per iteration '
                 '(best of three batches of %d), eager and replayed from a HIP graph' % reps)
  return res


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=5000)
  ap.add_argument('--warmup', type=int, default=500)
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-extras', action='store_true', help='skip the extra workload blocks (profiling runs)')
  ap.add_argument('--strong', action='store_true', help='BASELINE configs[4] literally: ONE batch of 32768 trajectories split over the N ranks (strong scaling; N = 1: all of it '
                                                        'on one GPU); the default is 4096 trajectories per rank (weak scaling, the metric BASELINE.json quotes)')
  args = ap.parse_args()

  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if args.gpus > 1 and world != args.gpus:
    raise SystemExit('--gpus %d needs WORLD_SIZE=%d (launch with torch.distributed.run)' % (args.gpus, args.gpus))
  assert torch.cuda.is_available(), 'bench.py needs a GPU (the product path has no CPU fallback)'
  # DGP_BENCH_ONE_DEVICE=1 (with DGP_BENCH_BACKEND=gloo: RCCL refuses two ranks on one device): every rank on cuda:0 -- a world of N > 1 ranks on a ONE-GPU
  # box, to execute the N > 1 branch (barriers, all-gather, max-over-ranks) end to end; a smoke configuration, its numbers mean nothing
  one_device = os.environ.get('DGP_BENCH_ONE_DEVICE') == '1'
  dev_index = 0 if one_device else local_rank
  torch.cuda.set_device(dev_index)
  device = torch.device('cuda', dev_index)
  dist = None
  # one process per GPU under torch.distributed.run (RANK / WORLD_SIZE / MASTER_* in the environment).  A world of ONE rank launched
  # that way (torchrun --nproc-per-node 1, or DGP_BENCH_FORCE_DIST=1) takes the same RCCL branch -- process group, barriers, the
  # all-gather of the final trajectories, the max-over-ranks reduction -- so that the code an N-GPU scaling run executes can be
  # exercised on a one-GPU box (profiles/r03_bench_dist_world1.json)
  under_launcher = 'RANK' in os.environ and 'WORLD_SIZE' in os.environ and 'MASTER_PORT' in os.environ
  if world > 1 or under_launcher or os.environ.get('DGP_BENCH_FORCE_DIST') == '1':
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    backend = os.environ.get('DGP_BENCH_BACKEND', 'nccl')      # ('nccl' IS RCCL on ROCm; 'gloo' only for the one-device smoke configuration above)
    if backend == 'nccl': dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
    else: dist.init_process_group(backend, rank=rank, world_size=world)

  import __graft_entry__
  if rank == 0: __graft_entry__.build()
  if dist is not None: dist.barrier()
  from dgpmp2_amd import _capi, parallel
  from dgpmp2_amd.gpmp2.plan_layer import solver_config

  B, n, d = B_PER_GPU, N_STATES, 2 * DOF
  if args.strong:
    # configs[4]: the 32768 trajectories of the node as contiguous shards (dgpmp2_amd.parallel.shard_range: the first B mod N ranks hold one more)
    lo, hi = parallel.shard_range(STRONG_TOTAL, rank, world)
    B = hi - lo
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
  sp, gp, dp, ep, xp, ip = start.data_ptr(), goal.data_ptr(), dth.data_ptr(), err.data_ptr(), eex.data_ptr(), info.data_ptr()

  # exactly what PlanLayer.forward launches (plan_layer.py: _GNStep.launch), info buffer included, through the same binding: the METH_FASTCALL
  # trampoline onto the C-ABI's dgp_gn_step (csrc/dgp_pycall.c; the ctypes binding of the same entry point costs 4 us more host time per call, which a
  # 20-launch region sees once, as the delay of its first launch)
  pc, hnd, raw_stream = _capi.get_pycall(), solver.h, stream.value or 0
  sdf_ptr = sdf.data_ptr()

  def step(k):
    rc = pc.gn_step(hnd, B, th_ptrs[k % GN_ITERS], sp, gp, sdf_ptr, GRID, GRID, 0, 0, 0, None, 0, None, None, None, dp, ep, xp, ip, raw_stream)
    if rc: solver.api.check(rc)

  prewarm_s = prewarm(step)
  gather_out = None
  total_B = STRONG_TOTAL if args.strong else world * B
  # whole-batch steps per region launch: weak scaling -- every rank steps its own 4096-trajectory batch, the batches add up; strong -- ONE 32768-trajectory batch per step
  batches_per_launch = 1 if args.strong else world
  if dist is not None:
    # untimed: the first call of a collective sets up RCCL's channels / loads its kernels (milliseconds); the 4 MB x world output buffer is
    # allocated ONCE here and handed to the product's helper as `out=` (a GN loop that gathers every outer iteration does the same)
    gather_out = parallel.gather_buffer(th_hist[-1], total_B)
    for _ in range(2): parallel.all_gather_trajectories(th_hist[-1], total_B, out=gather_out)
    torch.cuda.synchronize()
  for k in range(args.warmup): step(k)
  torch.cuda.synchronize()

  # ---- the timed regions -------------------------------------------------------------------------------------------------------------
  # ONE region is the contract's measurement: synchronised (N > 1: barrier + synchronise) -> EXACTLY K launches [-> the one all-gather of
  # the final trajectories] -> synchronised, wall clock around it, max over ranks.  With the driver's K = 20 a region lasts 0.2 ms on a GPU
  # the opening synchronisation has just idled, and a single 40-70 us hiccup (BENCH_r03: -24 %) decides it.  So the SAME region is repeated
  # R times, each with its own synchronisations and nothing skipped, and `value` is the MEDIAN region; min / quartiles / max and the first
  # region (what a single-shot measurement would have reported) are in the JSON.  Regions follow each other directly: the GPU idles for
  # microseconds between them and keeps its clocks.
  K = args.steps
  R = int(os.environ.get('DGP_BENCH_REGIONS', 0)) or max(9, min(41, int(0.4 / max(K * 10e-6, 1e-6))))
  cur_stream = torch.cuda.current_stream()
  gathered = None

  def region(ev=None, steps=None, gather=True):
    """One timed region; `ev` = (begin, end) HIP events recorded around the launches (only in the extra span regions below: two event
    records cost a 20-launch region ~13 us, profiles/r04_region_parts.txt).  steps / gather: the diagnostic regions of an N > 1 run (below)."""
    nonlocal gathered
    if dist is not None:
      torch.cuda.synchronize(); dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if ev: ev[0].record()
    for k in range(K if steps is None else steps): step(k)
    if dist is not None and gather:      # collect the final trajectories -- the only collective of the path -- through the product's helper; it is stream-ordered
      gathered = parallel.all_gather_trajectories(th_hist[-1], total_B, out=gather_out)      # behind the K launches and cannot complete before every rank has contributed
    if ev: ev[1].record()
    while not cur_stream.query(): pass              # spin until the stream has drained: synchronize() then returns at once, not after an interrupt wake-up
    torch.cuda.synchronize()
    return time.perf_counter() - t0

  walls = np.asarray([region() for _ in range(R)])
  # a few more regions WITH events around the launches: the device-side span of a region (start-up of the first launch and gaps included)
  evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(9)]
  for a_, b_ in evs: a_.record(); b_.record()      # (a first record() of an event costs ~40 us of host time: not inside a region)
  torch.cuda.synchronize()
  for e in evs: region(e)
  spans_ms = np.asarray([a_.elapsed_time(b_) for a_, b_ in evs])
  if dist is not None:
    assert tuple(gathered.shape) == (total_B, n, d)
    t = torch.tensor(walls, dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)        # every region: the slowest rank's clock
    walls = t.cpu().numpy()
  elapsed = float(np.median(walls))
  # N > 1 diagnostics (so that a --steps 20 scaling run can be read: the all-gather + its wait are a constant of 25-55 us, 20 % of such a region and 0.1 % of a
  # 5000-step one): the same regions WITHOUT the gather, and ONE long region (5000 steps + the gather), both max over ranks
  steps_only = long_region = None
  if dist is not None:
    w0 = torch.tensor([region(gather=False) for _ in range(R)], dtype=torch.float64, device=device)
    dist.all_reduce(w0, op=dist.ReduceOp.MAX)
    steps_only = float(np.median(w0.cpu().numpy()))
    w1 = torch.tensor([region(steps=5000)], dtype=torch.float64, device=device)
    dist.all_reduce(w1, op=dist.ReduceOp.MAX)
    long_region = float(w1.item())
  q = lambda v, f: float(np.percentile(v, f))
  regions = {'count': R, 'steps_per_region': K,
             'ms_per_step': {'min': float(walls.min()) * 1e3 / K, 'p25': q(walls, 25) * 1e3 / K, 'median': elapsed * 1e3 / K, 'p75': q(walls, 75) * 1e3 / K,
                             'max': float(walls.max()) * 1e3 / K, 'first_region': float(walls[0]) * 1e3 / K},
             'note': 'each region = synchronise [+ barrier] -> K launches [-> all-gather] -> synchronise, wall clock, max over ranks; value = K / median region'}
  if os.environ.get('DGP_BENCH_DUMP_REGIONS') == '1':      # (profiles/tools/region_gaps.py: which regions were slow, next to a rocprofv3 trace of the same run)
    regions['wall_us'] = [float(w) * 1e6 for w in walls]; regions['span_us'] = [float(x) * 1e3 for x in spans_ms]
  region_span_ms = float(np.median(spans_ms)) / K   # HIP events around a region's launches on the launch stream: includes the start-up of the first launch and the gaps
  # The dominant kernel's own duration: every launch records its OWN begin / end (dgp_time_next_launch == hipExtLaunchKernelGGL events, rocprofv3's
  # definition of a kernel's duration).  A separate pass: such launches dispatch ~5 us slower, so they are kept out of the timed regions.
  ktimer = _capi.KernelTimer(1000)
  for k in range(len(ktimer.pairs)):
    ktimer.arm(); step(k)
  torch.cuda.synchronize()
  kdur = np.asarray(ktimer.durations_ms())
  kernel_ms = float(kdur.mean())
  # ... and the steady-state launch PERIOD (kernel + the gap to the next dependent launch): events around 3 x 2000 back-to-back launches, median
  period_ms = time_launches(step, 2000, warm_s=0.05) * 1e-3
  # fixed cost of an N > 1 region that is not the path's: the all-gather + the wait for it, measured on their own
  region_fixed_us = None
  if dist is not None:
    ts = []
    for _ in range(20):
      torch.cuda.synchronize(); t0 = time.perf_counter()
      parallel.all_gather_trajectories(th_hist[-1], total_B, out=gather_out)
      torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    region_fixed_us = float(np.median(ts)) * 1e6

  # Beside the headline (one launch per GN step, the reference's step()): the same 10 GN iterations from th_init as ONE
  # launch of the fused loop (the reference's forward(); dgp_gn_solve, tol 0 so that all 10 run) -- no launch gaps, th
  # stays in registers, err / err_ext of every iteration written.  Reported as an extra field, never as `value`.
  tho = torch.empty_like(th0); its = torch.zeros(B, dtype=torch.int32, device=device)
  eh = torch.empty(B, GN_ITERS, device=device); eeh = torch.empty(B, GN_ITERS, device=device)
  fused_us = time_launches(lambda k: solver.gn_solve(B, th_ptrs[0], sp, gp, sdf_arg, None, GN_ITERS, 0.0, tho.data_ptr(), its.data_ptr(),
                                                     eh.data_ptr(), eeh.data_ptr(), None, ip, stream), max(20, min(400, args.steps // GN_ITERS)), warm_s=0.1)
  assert int(its.min()) == GN_ITERS

  if rank == 0:
    bytes_per_launch = algorithmic_bytes_per_trajectory(n, d) * B
    lpt, cc = solver.launch_shape(B)
    waves = (B + (64 // lpt) - 1) // (64 // lpt)
    kname = 'gn_kernel<%d,%d,%d,float,0,%d>' % (DOF, lpt, cc, solver.step_kernel_variant(B))      # <dof, LPT, C, io, MODE_STEP, QK: 1 block elimination, 3 Woodbury>
    ks = kernel_stats().get(kname)
    out = {
        'metric': ('Gauss-Newton steps/sec (whole node), batch=32768 x 64 states sharded over the GPUs, 2D point robot (BASELINE configs[4])' if args.strong
                   else 'Gauss-Newton steps/sec (whole node), batch=4096 x 64 states, 2D point robot'),
        'value': batches_per_launch * args.steps / elapsed,
        'unit': ('GN steps/s (one step = one step of the whole 32768-trajectory batch, every rank stepping its shard)' if args.strong
                 else 'GN steps/s (one step = one whole-batch step of 4096 trajectories; per-GPU batches add up)'),
        'value_first_region': batches_per_launch * args.steps / float(walls[0]),
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'prewarm_s': prewarm_s, 'ms_per_step': 1e3 * elapsed / args.steps,
        'higher_is_better': True, 'scaling': 'strong' if args.strong else 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': ('BASELINE configs[4]: 2D point robot, ONE batch of 32768 trajectories split into contiguous shards over %d rank(s), ' % world if args.strong
                                else 'BASELINE configs[1]: 2D point robot, batch=4096 per GPU, ') +
                               '64 states, 256x256 shared SDF, static covariances, inputs = trajectories after (k mod 10) GN iterations; C-ABI dgp_gn_step launch rate '
                               '(err, err_ext and the SPD info flags written every step, as PlanLayer.forward does)',
                   'batch_per_gpu': B, 'batch_total': total_B, 'num_states': n, 'state_dim': d, 'sdf': [GRID, GRID], 'io_dtype': 'f32',
                   'parallelism': 'trajectory batch sharded, %d rank(s)' % world},
        'build': __graft_entry__.build_provenance(),
        'rccl_ranks': world if dist is not None else 0,
        'trajectory_steps_per_s': args.steps * total_B / elapsed,
        'regions': regions,
        'kernel_events': {'launches': int(kdur.size), 'mean_ms': float(kdur.mean()), 'median_ms': float(np.median(kdur)),
                          'note': 'per-launch begin/end events (dgp_time_next_launch), a separate pass after the timed regions'},
        'roofline': roofline_block(bytes_per_launch, period_ms * 1e3, kname, traffic_key='gn_step',
                                   note='kernel_avg_ms = the dominant kernel\'s average duration in the regime it is benchmarked in -- HIP events on the launch stream around 2000 '
                                        'back-to-back launches / 2000, median of three passes -- which is what rocprofv3 --kernel-trace --stats reports as its average for this '
                                        'command (profiles/r05_kernel_trace.txt: within 2 % on every box so far); kernel_isolated_avg_ms = mean over 1000 launches that each record '
                                        'their OWN begin / end events (dgp_time_next_launch): such launches dispatch ~5 us apart, every kernel starts on an idle GPU without '
                                        'overlapping its predecessor\'s tail, and runs 2-5 % longer.  HBM is the bound SURVEY 8(d) prescribes; the measured limiter is fp64 VALU '
                                        'issue (see valu_fp64 and DESIGN.md section 5)'),
    }
    out['roofline']['kernel_isolated_avg_ms'] = kernel_ms                # per-launch begin / end events, launches dispatched one by one
    out['roofline']['region_span_ms_per_launch'] = region_span_ms      # events around one K-launch region / K (median region): start-up and gaps included
    if steps_only is not None:
      out['value_steps_only'] = batches_per_launch * args.steps / steps_only      # the same K-launch regions without the all-gather (median, max over ranks)
      out['steps_per_s_at_5000'] = batches_per_launch * 5000 / long_region         # one region of 5000 launches + the all-gather: the fixed cost amortised
      out['scaling_note'] = ('value = K / median region INCLUDING one all-gather of the final trajectories and the wait for it (region_fixed_us, a constant); '
                             'value_steps_only = the same regions without it; steps_per_s_at_5000 = one 5000-launch region with it -- at --steps 20 the constant is '
                             '~20 %% of a region, so per-N efficiency is better read from the last two')
    if region_fixed_us is not None:
      out['region_fixed_us'] = region_fixed_us
      out['region_fixed_note'] = ('one all-gather of the final trajectories + the wait for it, timed on its own (median of 20): the part of an N > 1 region '
                                  'that is not GN steps; at --steps 20 it is a visible share of the region, at the default 5000 steps it is noise')
    if ks:
      flops = (2 * ks['fma_f64'] + ks['mul_f64'] + ks['add_f64']) * 64 * waves
      tf = flops / (period_ms * 1e-3) / 1e12
      out['valu_fp64'] = {'achieved_tflops': tf, 'peak_tflops': FP64_VECTOR_PEAK_TFLOPS, 'frac': tf / FP64_VECTOR_PEAK_TFLOPS, 'flops_per_launch': flops,
                          'insts_per_wave': {k: ks[k] for k in ('valu', 'fma_f64', 'mul_f64', 'add_f64', 'rcp_f64', 'dpp', 'agpr_moves')},
                          'registers': {'vgpr': ks.get('vgpr'), 'agpr': ks.get('agpr'), 'scratch_bytes_per_lane': ks.get('scratch_bytes_per_lane')},
                          'source': 'static ISA counts of %s from the build (dgpmp2_amd/lib/kernel_stats.json)' % kname}
    out['fused_forward'] = {'gn_iterations_per_launch': GN_ITERS, 'ms_per_launch': fused_us * 1e-3, 'us_per_gn_iteration': fused_us / GN_ITERS,
                            'gn_steps_per_s_per_gpu': GN_ITERS / (fused_us * 1e-6),
                            'note': 'dgp_gn_solve: the 10 GN iterations of BASELINE configs[1] in one launch (rank 0, outside the timed region)'}
    if world == 1 and not args.no_extras and not args.strong:
      out['two_streams'] = two_stream_rate(solver, B, th_ptrs, sp, gp, sdf_arg, device)
      out.update(extra_workloads(device, stream))
      out['sdf_fields'] = sdf_fields_rate(device)
      out['planner_step_api'] = planner_api_rate(device)
      out['planner_step_backward_api'] = planner_api_backward_rate(device)
      out['train_iteration_sdf_grad'] = train_iteration_sdf_grad(device)
    if world == 1 and not args.no_cpu_baseline:      # (strong mode: the CPU baseline stays the 4096-trajectory batch-step the metric is quoted on)
      hist_cpu = [t[:B_PER_GPU].cpu() for t in th_hist]
      out['cpu_baseline'] = cpu_baseline(hist_cpu, start[:B_PER_GPU].cpu(), goal[:B_PER_GPU].cpu(), sdf.cpu())
      out['cpu_baseline_blocktri'] = cpu_blocktri(hist_cpu, start[:B_PER_GPU].cpu(), goal[:B_PER_GPU].cpu(), sdf.cpu())
    print(json.dumps(out))
  if dist is not None:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
