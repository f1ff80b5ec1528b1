#!/usr/bin/env python
"""Steady-clock kernel timing of one C-ABI entry point on a chosen workload (tuning loop; honours DGP_LIB_PATH for a
profiles/tools/devbuild.py library and DGP_FORCE_SHAPE).  Two numbers per case: `period_us` = HIP events around `reps`
back-to-back launches / reps (kernel + inter-kernel gap), `kernel_us` = mean of per-kernel begin/end events
(dgp_time_next_launch; what rocprofv3 reports as the kernel's duration).

  python profiles/tools/ubench.py --what step,solve,bwd,bwd_sdf8 [--covs static|perstate|qfull|scalar] [--dof 2|3] [--sdf shared|persample]
                                  [--B 4096] [--n 64] [--G 256] [--io f32|f64] [--flags vel|nonhol] [--reps 1000] [--tag text]
"""
import argparse, ctypes, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import make_inputs, make_per_sample_sdfs, algorithmic_bytes_per_trajectory, prewarm
from dgpmp2_amd import _capi
from dgpmp2_amd.gpmp2.plan_layer import solver_config


def measure(launch, reps, timer, warm=0.3):
  prewarm(launch, warm, 100)
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize(); e0.record()
  for k in range(reps): launch(k)
  e1.record(); torch.cuda.synchronize()
  period = e0.elapsed_time(e1) / reps * 1e3
  timer.reset()
  nk = min(reps, len(timer.pairs))
  for k in range(nk):
    timer.arm(); launch(k)
  torch.cuda.synchronize()
  d = np.asarray(timer.durations_ms()) * 1e3
  return round(period, 2), round(float(d.mean()), 2), round(float(np.median(d)), 2)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--what', default='step')
  ap.add_argument('--covs', default='static'); ap.add_argument('--dof', type=int, default=2); ap.add_argument('--sdf', default='shared')
  ap.add_argument('--B', type=int, default=4096); ap.add_argument('--n', type=int, default=64); ap.add_argument('--G', type=int, default=0)
  ap.add_argument('--io', default='f32'); ap.add_argument('--flags', default=''); ap.add_argument('--reps', type=int, default=1000)
  ap.add_argument('--iters', type=int, default=10); ap.add_argument('--tag', default=''); ap.add_argument('--info', type=int, default=1); ap.add_argument('--grids', type=int, default=1, help='per-sample SDF: number of distinct 1 GiB grid sets cycled through (>= 5: the touched lines no longer fit the 256 MiB Infinity Cache)'); ap.add_argument('--th', type=int, default=3, help='GN iterations behind the timed trajectory (0: straight-line init)'); ap.add_argument('--layout', default='rowmajor', help="rowmajor | tiled4 (the grids as 4 x 4 tiles, DgpSdf::layout)")
  a = ap.parse_args()
  dev = torch.device('cuda:0')
  dt = torch.float32 if a.io == 'f32' else torch.float64
  B, n, dof = a.B, a.n, a.dof
  d = 2 * dof
  G = a.G or (512 if dof == 3 else 256)
  kw = {}
  if 'vel' in a.flags: kw.update(use_vel_limits=True, K_v=0.01, v_x=1.0, v_y=1.0)
  if 'nonhol' in a.flags or (dof == 3 and a.flags == ''): kw.update(non_holonomic=True, K_d=0.01, epsilon_dist=0.2, reg=0.0)
  th0, start, goal, sdf = [t.to(dt) for t in make_inputs(B, n, G, dev, dof=dof)]
  sdfs = [sdf]
  if a.sdf == 'persample':
    sdfs = [make_per_sample_sdfs(B, G, dev, seed=1 + i).to(dt) for i in range(a.grids)]; sdf = sdfs[0]; stride = G * G
  else:
    stride = 0
  s = _capi.Solver(solver_config(num_states=n, dof=dof, io_dtype=dt, **kw))
  lay = _capi.DGP_SDF_ROWMAJOR
  if a.layout == 'tiled4':
    from dgpmp2_amd.utils.sdf_utils import tile_sdf
    sdfs = [tile_sdf(t if t.dim() == 4 else t.reshape(-1, 1, G, G)) for t in sdfs]; sdf = sdfs[0]; lay = _capi.DGP_SDF_TILED4
  sa = s.sdf_arg(sdf.data_ptr(), G, G, stride, layout=lay)
  sas = [s.sdf_arg(t.data_ptr(), G, G, stride, layout=lay) for t in sdfs]
  st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
  P = lambda t: None if t is None else t.data_ptr()
  covs, keep = None, []
  if a.covs != 'static':
    if a.covs == 'perstate':
      qc = torch.eye(dof, device=dev, dtype=dt).expand(B, n - 1, dof, dof).contiguous(); mode = _capi.DGP_QC_PERSTATE
    elif a.covs == 'scalar':      # one scalar per GP factor (dynamics_mode diag_identity): DGP_QC_SCALAR, the static kernels with scaled lane masks (step only)
      qc = torch.ones(B, n - 1, device=dev, dtype=dt); mode = _capi.DGP_QC_SCALAR
    else:
      from oracle import gpmp2_oracle as O
      p = O.OracleParams(dof=dof, total_time_step=n - 1)
      Q = O.calc_Q_inv_batch(np.broadcast_to(np.eye(dof), (1, n - 1, dof, dof)).copy(), p.dt)
      qc = torch.from_numpy(Q).to(dt).to(dev).expand(B, n - 1, d, d).contiguous(); mode = _capi.DGP_QC_QFULL
    ow = torch.full((B, n), 1e4, device=dev, dtype=dt); ep = torch.full((B, n), float(kw.get('epsilon_dist', 0.4)), device=dev, dtype=dt)
    keep = [qc, ow, ep]
    covs = s.covs_arg(mode, qc.data_ptr(), ow.data_ptr(), ep.data_ptr())
  dth = torch.empty_like(th0); err = torch.empty(B, device=dev, dtype=dt); eex = torch.empty_like(err)
  info = torch.zeros(B, dtype=torch.int32, device=dev) if a.info else None
  ths = [th0]
  for _ in range(3):
    s.gn_step(B, P(ths[-1]), P(start), P(goal), sa, covs, P(dth), P(err), P(eex), P(info), st); ths.append(ths[-1] + dth)
  torch.cuda.synchronize()
  assert bool(torch.isfinite(ths[-1]).all())
  tp = [t.data_ptr() for t in ths]
  timer = _capi.KernelTimer(min(a.reps, 1000))
  out = dict(layout=a.layout, grids=a.grids, tag=a.tag, lib=os.path.basename(_capi.LIB_PATH), B=B, n=n, dof=dof, io=a.io, covs=a.covs, sdf=a.sdf, flags=a.flags, shape=list(s.launch_shape(B)))
  for what in a.what.split(','):
    if what == 'step':
      f = lambda k: s.gn_step(B, tp[k % 4], P(start), P(goal), sas[k % len(sas)], covs, P(dth), P(err), P(eex), P(info), st)
      by = algorithmic_bytes_per_trajectory(n, d, io_bytes=4 if a.io == 'f32' else 8, cov_tensors=(a.covs in ('perstate', 'scalar'))) * B
    elif what == 'solve':
      tho = torch.empty_like(th0); it = torch.empty(B, dtype=torch.int32, device=dev); eh = torch.empty(B, a.iters, device=dev, dtype=dt)
      eeh = torch.empty_like(eh)
      f = lambda k: s.gn_solve(B, tp[0], P(start), P(goal), sa, covs, a.iters, 0.0, P(tho), P(it), P(eh), P(eeh), None, P(info), st)
    elif what in ('traced', 'chain'):
      # dgp_gn_solve_traced (the fused loop + its fp64 history) / dgp_gn_solve_backward (the whole loop's backward, one launch); static covariances
      tho = torch.empty_like(th0); it = torch.empty(B, dtype=torch.int32, device=dev)
      hist = torch.empty((a.iters, B, n, d), dtype=torch.float64, device=dev)
      gfin = torch.randn_like(th0); gth = torch.empty_like(th0); gst = torch.empty_like(start); ggo = torch.empty_like(goal)
      trace = lambda k: s.gn_solve_traced(B, tp[0], P(start), P(goal), sa, None, a.iters, 0.0, P(tho), P(it), None, None, None, P(info), P(hist), st)
      trace(0)
      if what == 'traced': f = trace
      else: f = lambda k: s.gn_solve_backward(B, P(start), P(goal), sa, a.iters, P(hist), P(tho), P(it), P(gfin), P(gth), P(gst), P(ggo), None, 0, st)
    elif what == 'step_errs':      # dgp_gn_step_errors: the step + the unweighted errors at th + dtheta (two launches; `period_us` is the pair, `kernel_us` the first of them)
      us_, ug_, uo_ = (torch.empty(B, device=dev, dtype=dt) for _ in range(3))
      f = lambda k: s.gn_step_errors(B, tp[k % 4], P(start), P(goal), sas[k % len(sas)], covs, P(dth), P(err), P(eex), P(info), P(us_), P(ug_), P(uo_), st)
    elif what in ('bwd_errs', 'bwd_errs_sdf', 'bwd_errs_noobs'):      # dgp_gn_step_errors_backward with every cotangent: ONE launch (the errors' backward as the kernel's prologue;
      # _noobs: without the obs_error cotangent -- the prologue then reads no grid: what its gather + hinge arithmetic cost)
      g = torch.randn_like(th0); gth = torch.empty_like(th0); gst = torch.empty_like(start); ggo = torch.empty_like(goal)
      ge = torch.ones(B, device=dev, dtype=dt); c1, c2, c3 = (torch.randn(B, device=dev, dtype=dt) for _ in range(3))
      gq = (torch.empty(B, n - 1, dof, dof, device=dev, dtype=dt) if a.covs == 'scalar' else torch.empty_like(keep[0])) if covs else None; gw = torch.empty(B, n, device=dev, dtype=dt) if covs else None
      gp = torch.empty(B, n, device=dev, dtype=dt) if covs else None
      gs, sab, copies = None, sa, 1
      if what == 'bwd_errs_sdf':
        copies = 8 if stride == 0 else 1
        gs = torch.zeros((copies if stride == 0 else B, 1, G, G), device=dev, dtype=torch.float64 if stride == 0 else dt)
        if stride == 0: sab = s.sdf_arg(sdf.data_ptr(), G, G, stride, layout=lay, grad_mode=_capi.DGP_GSDF_DENSE_F64)
      s.gn_step(B, tp[a.th], P(start), P(goal), sa, covs, P(dth), P(err), P(eex), P(info), st)
      if what == 'bwd_errs_noobs': c3 = None
      ws = torch.empty_like(th0) if dof == 3 else None      # (the two-launch form of the (x, y, theta) robot hands dL/d(th + dtheta) over in a workspace)
      f = lambda k, c3=c3: s.gn_step_errors_backward(B, tp[a.th], P(start), P(goal), sab, covs, P(dth), P(g), P(ge), P(c1), P(c2), P(c3), P(gth), P(gst), P(ggo), P(gs), stride,
                                                     P(gq), P(gw), P(gp), P(ws), st, g_sdf_copies=copies)
    elif what == 'eval':
      f = lambda k: s.eval_errors(B, tp[k % 4], P(start), P(goal), sa, covs, P(err), P(eex), None, None, None, st)
    elif what.startswith('bwd'):
      g = torch.randn_like(th0); gth = torch.empty_like(th0); gst = torch.empty_like(start); ggo = torch.empty_like(goal)
      ge = torch.ones(B, device=dev, dtype=dt)
      gq = (torch.empty(B, n - 1, dof, dof, device=dev, dtype=dt) if a.covs == 'scalar' else torch.empty_like(keep[0])) if covs else None; gw = torch.empty(B, n, device=dev, dtype=dt) if covs else None
      gp = torch.empty(B, n, device=dev, dtype=dt) if covs else None
      # bwd_sdf[N]: dense grid gradient (N partial copies of a shared grid); bwd_sdf[N]w: the partial copies in float64 (DGP_GSDF_DENSE_F64); bwd_sparse: the taps as
      # COO values + indices (DGP_GSDF_SPARSE, per-sample grids)
      wide = what.startswith('bwd_sdf') and what.endswith('w')
      num = what[7:-1] if wide else what[7:]
      copies = int(num) if what.startswith('bwd_sdf') and num else 1
      gs, sab = None, sa
      if what == 'bwd_sparse':
        gs = torch.empty(B * n * 4, device=dev, dtype=dt); gi = torch.empty((6 if a.layout == 'tiled4' else 4, B * n * 4), device=dev, dtype=torch.int64)      # (a tiled grid tensor has six index rows)
        sab = s.sdf_arg(sdf.data_ptr(), G, G, stride, layout=lay, grad_mode=_capi.DGP_GSDF_SPARSE, grad_indices=gi.data_ptr())
      elif what != 'bwd':
        gs = torch.zeros((copies if stride == 0 else B, 1, G, G), device=dev, dtype=torch.float64 if wide else dt)
        if wide: sab = s.sdf_arg(sdf.data_ptr(), G, G, stride, layout=lay, grad_mode=_capi.DGP_GSDF_DENSE_F64)
      s.gn_step(B, tp[a.th], P(start), P(goal), sa, covs, P(dth), P(err), P(eex), P(info), st)
      f = lambda k: s.gn_step_backward(B, tp[a.th], P(start), P(goal), sab, covs, P(dth), P(g), P(ge), P(gth), P(gst), P(ggo), P(gs), stride,
                                       P(gq), P(gw), P(gp), st, g_sdf_copies=copies)
    else:
      raise SystemExit('unknown --what ' + what)
    reps = a.reps if what not in ('solve', 'traced', 'chain') else max(20, a.reps // a.iters)
    period, kmean, kmed = measure(f, reps, timer)
    out[what] = dict(period_us=period, kernel_us=kmean, kernel_med_us=kmed)
    if what == 'step': out[what]['alg_GBs'] = round(by / (kmean * 1e-6) / 1e9, 1)
    if what in ('solve', 'traced', 'chain'): out[what]['us_per_iter'] = round(kmean / a.iters, 2)
  print(json.dumps(out), flush=True)


if __name__ == '__main__':
  main()
