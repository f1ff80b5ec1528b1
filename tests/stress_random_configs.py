#!/usr/bin/env python
"""Randomised stress run (not collected by pytest): random trajectory lengths, batch sizes, robots, factor flags, SDF shapes
(non-square, trajectories leaving the grid), I/O dtypes and covariance modes through the C-ABI on the GPU, every
trajectory compared with oracle/gn_blocktri.c.   usage (GPU box): python tests/stress_random_configs.py [--cases 200]"""
import argparse, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import harness
import parity_cases as PC
from oracle import gpmp2_oracle as O, blocktri as BT

ap = argparse.ArgumentParser(); ap.add_argument('--cases', type=int, default=200); ap.add_argument('--seed', type=int, default=0); ap.add_argument('--backend', default='hip'); ap.add_argument('--maxB', type=int, default=100000)
args = ap.parse_args()
be = harness.Backend(args.backend)
rs = np.random.RandomState(args.seed)
worst = 0.0
conditioned = []      # cases beyond the fp64 tolerance that the extended-precision arbiter attributed to conditioning
for case in range(args.cases):
  dof = int(rs.choice([2, 2, 3]))
  n = int(rs.choice([2, 3, 5, 8, 16, 17, 31, 32, 33, 48, 63, 64, 65, 101, 128, 129, 200, 256, 257, 300, 384]))      # > 256: the loop kernels of gn_long.h
  B = min(args.maxB, int(rs.choice([1, 2, 3, 4, 5, 7, 8, 63, 64, 65, 257, 1000])))
  if n > 256: B = min(B, 5)                          # (the dense gradient oracle is O(n^3) per trajectory)
  io = str(rs.choice(['f64', 'f32']))
  shapes = [(l, c) for l in (16, 32, 64) for c in (1, 2, 4) if l * c >= n]
  forced = shapes[int(rs.randint(len(shapes)))] if (shapes and rs.rand() < 0.5) else None      # half of the cases pin a random launch shape that covers n
  if forced: os.environ['DGP_FORCE_SHAPE'] = '%d,%d' % forced
  else: os.environ.pop('DGP_FORCE_SHAPE', None)
  kw = {}
  if dof == 3 and rs.rand() < 0.6: kw.update(non_holonomic=True, K_d=float(rs.choice([0.01, 0.1])))
  if rs.rand() < 0.4: kw.update(use_vel_limits=True, K_v=0.01, v_x=float(rs.uniform(0.2, 1.5)), v_y=float(rs.uniform(0.2, 1.5)))
  qmode = rs.choice(['identity', 'diag', 'full'])
  if qmode == 'diag': kw['Q_c_inv'] = np.diag(rs.uniform(0.5, 2.0, dof))
  if qmode == 'full':
    A = rs.randn(dof, dof) * 0.3
    kw['Q_c_inv'] = np.eye(dof) + A @ A.T
  p = O.OracleParams(dof=dof, total_time_step=n - 1, reg=float(rs.choice([0.1, 0.1, 1e-3])), epsilon_dist=float(rs.uniform(0.1, 0.6)), **kw)
  d = 2 * dof
  H, W = int(rs.choice([2, 5, 33, 64, 100, 256])), int(rs.choice([2, 3, 33, 64, 100, 256]))
  yy, xx = np.meshgrid(np.linspace(5, -5, H), np.linspace(-5, 5, W), indexing='ij')
  per_sample = rs.rand() < 0.3 and B <= 65
  nsd = B if per_sample else 1
  cs = rs.uniform(-3, 3, (nsd, 3, 2)); rr = rs.uniform(0.4, 1.2, (nsd, 3))
  sdf = np.min(np.sqrt((xx[None, None] - cs[:, :, 0, None, None]) ** 2 + (yy[None, None] - cs[:, :, 1, None, None]) ** 2) - rr[:, :, None, None], axis=1)[:, None]
  span = float(rs.choice([4.0, 4.0, 5.5]))          # 5.5: some trajectories leave the grid
  start = np.zeros((B, 1, d)); goal = np.zeros((B, 1, d))
  start[:, 0, :2] = rs.uniform(-span, span, (B, 2)); goal[:, 0, :2] = rs.uniform(-span, span, (B, 2))
  if dof == 3: goal[:, 0, 2] = rs.uniform(-np.pi, np.pi, B)
  th = O.straight_line_trajb(start[:, :, :dof], goal[:, :, :dof], 10.0, n - 1, dof) + rs.randn(B, n, d) * float(rs.choice([0.0, 0.05, 0.3]))
  qc = ow = eps = None; q_full = False
  cov = rs.choice(['static', 'static', 'perstate', 'qfull'])
  if cov != 'static':
    ow = rs.uniform(50, 2e4, (B, n)); eps = rs.uniform(0.1, 0.6, (B, n))
    if cov == 'perstate':
      A = rs.randn(B, n - 1, dof, dof) * 0.2; qc = np.eye(dof) + A @ np.swapaxes(A, -1, -2)
    else:
      A = rs.randn(B, n - 1, d, d) * 0.2; qc = (np.eye(d) + A @ np.swapaxes(A, -1, -2)) * rs.uniform(0.5, 3.0); q_full = True
  r = lambda a: None if a is None else PC.rnd(a, io)
  # every other per-state case with a diagonal Q_c_inv becomes a ONE-SCALAR-PER-FACTOR case (dynamics_mode diag_identity): the step receives the (B, n-1) scalars
  # (DGP_QC_SCALAR: the static kernels with scaled lane masks), everything else -- oracles, fused loop, backward -- the dense tensors s_k Q_c_inv (no extra random draws)
  qc_step = None
  if cov == 'perstate' and qmode != 'full' and n <= 256 and case % 2 == 0:
    s2 = r(1.0 + 4.0 * A[..., 0, 0] ** 2)
    qc = s2[:, :, None, None] * p.Q_c_inv
    qc_step = s2
    cov = 'scalar'
  th, start, goal, sdf, qc, ow, eps = r(th), r(start), r(goal), r(sdf), r(qc), r(ow), r(eps)
  dth, err, eex, info = be.step(p, th, start, goal, sdf, qc=qc if qc_step is None else qc_step, ow=ow, eps=eps, q_full=q_full, io=io)
  sh = (B, n, 1, 1)
  c_dth, c_err, c_eex, c_info = BT.gn_step(p, th, start, goal, sdf, qc=qc, ow=None if ow is None else ow.reshape(sh), eps=None if eps is None else eps.reshape(sh),
                                           q_full=q_full, nthreads=4)
  ok = (c_info == 0) & (info == 0)
  assert np.array_equal(c_info != 0, info != 0), ('SPD flags differ', case)
  if ok.any():
    scale = np.abs(c_dth).reshape(B, -1).max(1) + 1e-300
    e = (np.abs(dth - c_dth).reshape(B, -1).max(1) / scale)[ok].max()
    ee = np.abs(err - c_err)[ok].max() / (np.abs(c_err)[ok].max() + 1e-300)
    tol = PC.TOL[io] * (30 if p.reg < 0.01 else 1)          # weakly regularised systems: cond(Lambda) up to 1e7
    worst = max(worst, e / tol)
    status = 'ok' if (e < tol and ee < 10 * PC.TOL_ERR[io]) else 'FAIL'
    if status == 'FAIL' and ee < 10 * PC.TOL_ERR[io]:
      # conditioning or a wrong result?  An INDEPENDENT arbiter decides: oracle/gn_blocktri.c built with the assembly and the block
      # solve in 80-bit extended precision (libgn_blocktri_ld.so; it shares nothing with the kernels or their host emulator).  On a
      # weakly regularised system (cond(Lambda) up to 1e7) the fp64 C oracle itself is cond * 2^-53 away from that solution; the GPU
      # result is accepted -- and REPORTED as 'cond', never as 'ok' -- only when it is no further from the extended-precision solution
      # than 3 x the fp64 C oracle is.  Anything else fails.
      x_dth, _, _, x_info = BT.gn_step(p, th, start, goal, sdf, qc=qc, ow=None if ow is None else ow.reshape(sh), eps=None if eps is None else eps.reshape(sh),
                                       q_full=q_full, nthreads=4, extended=True)
      xs = np.abs(x_dth).reshape(B, -1).max(1) + 1e-300
      okx = ok & (x_info == 0)
      e_gpu = (np.abs(dth - x_dth).reshape(B, -1).max(1) / xs)
      e_c = (np.abs(c_dth - x_dth).reshape(B, -1).max(1) / xs)
      accept = (e_gpu <= tol) | (e_gpu <= 3.0 * e_c)
      note = ''
      if not np.all(accept[okx]) and n * d <= 2400:      # (up to the longest stress trajectory, n = 384 at d = 6: a 2 304 x 2 304 dense system, seconds)
        # Block Thomas (the C oracle) is far more accurate than its bound on these systems, so "3 x the C oracle" can reject a block-PCR result
        # that is as good as a backward-stable solver's.  The algorithm-independent criterion: the NORMWISE BACKWARD ERROR of the GPU solution
        # on the DENSE system of the torch restatement (oracle/autograd_torch.py: Lambda, eta assembled as plan_layer.py:152-220 does; residual in extended precision),
        #   beta = |eta - Lambda x| / (|Lambda| |x| + |eta|)   (inf-norms),
        # must be small -- then x is the exact solution of a system within that relative distance of the given one and the forward error is
        # conditioning (cond_2(Lambda) is printed next to it).  The bound is 1e-11, not a few units of round-off: the kernels eliminate with EXPLICIT
        # block inverses (adjugate formulas: short dependency chains), whose backward error is cond(block) x round-off, and this script draws
        # q_full covariances of order 1 next to factor weights of 1e4 inside one 6 x 6 block (cond(block) ~ 1e4; measured beta: 1e-13 .. 4e-13,
        # forward errors up to 1.2e-8 at cond(Lambda) 2.6e5; DESIGN.md section 7 "accuracy").  A wrong kernel does not produce a 1e-11 backward error.
        import torch
        from oracle import autograd_torch as AT
        sq, so, se = p.static_covs(B)
        T_ = lambda a: torch.from_numpy(np.array(a, dtype=np.float64))
        sdfB = np.broadcast_to(sdf, (B,) + sdf.shape[1:])
        for t in np.nonzero(okx & ~accept)[0]:
          sl = slice(t, t + 1)
          LAM, R = AT.normal_equations(T_(th[sl]), T_(start[sl]), T_(goal[sl]), T_(sdfB[sl]), T_((sq if qc is None else qc)[sl]), T_((so if ow is None else ow.reshape(so.shape))[sl]),
                                       T_((se if eps is None else eps.reshape(se.shape))[sl]), p, q_full=q_full)
          LAM, R = LAM.numpy(), R.numpy()
          L_, r_, x_ = LAM[0].astype(np.longdouble), R[0, :, 0].astype(np.longdouble), dth[t].reshape(-1).astype(np.longdouble)
          beta = float(np.abs(r_ - L_ @ x_).max() / (np.abs(L_).sum(1).max() * np.abs(x_).max() + np.abs(r_).max()))
          cond = float(np.linalg.cond(LAM[0]))
          note = ' backward error %.1e, cond %.1e' % (beta, cond)
          # ... or the forward error is within what a solver with a backward error of ONE unit of round-off guarantees, cond_2(Lambda) x 2^-52: the loop kernels
          # (n > 256: six rows per lane, six PCR rounds) reach beta = 4e-12 on a cond 4e7 system (seed 2, case 59: n = 384, d = 6, forward error 1.1e-9 < 9e-9)
          # (1e-11 since seeds 15 and 29: d = 6 q_full draws at cond(Lambda) 2e7 -- 6 x 6 blocks mixing order-1 covariances with 1e4 factor weights -- reach beta = 2.1e-12,
          #  forward error 8e-8; the parity target is 1e-5 relative in fp32, and a miscompiled kernel is wrong by O(1))
          if beta <= 1e-11 or e_gpu[t] <= cond * 2.0 ** -52: accept[t] = True
      if np.all(accept[okx]):
        status = 'cond(gpu %.1e, fp64 C oracle %.1e off the extended-precision solve;%s)' % (e_gpu[okx].max(), e_c[okx].max(), note)
        conditioned.append(case)
      else:
        bw = int(np.argmax(np.where(okx & ~accept, e_gpu, 0.0)))
        status = 'FAIL(trajectory %d: gpu %.1e, fp64 C oracle %.1e off the extended-precision solve;%s)' % (bw, e_gpu[bw], e_c[bw], note)
    print('%3d %s dof=%d n=%3d B=%4d %s shape=%s sdf=%dx%d%s cov=%s Qc=%s flags=%s  dth %.1e err %.1e' % (case, status, dof, n, B, io, forced or 'auto', H, W, '(per-sample)' if per_sample else '', cov, qmode,
          ','.join(k for k in ('non_holonomic', 'use_vel_limits') if k in kw), e, ee), flush=True)
    assert not status.startswith('FAIL'), status
  # round 5: the twin translation units on the same configuration (no extra random draws: the seeds keep their meaning) -- the step kernels with the errors
  # epilogue (dgp_gn_step_errors in one launch where the host offers it; otherwise the call runs its two-launch form, checked all the same) against the step above
  # and the error kernel at th + dtheta, and the tiled-grid twins against the row-major result
  if ok.all() and n <= 256:
    tw = 10 * PC.TOL[io] * (30 if p.reg < 0.01 else 1)      # (ten times the main check's bound, which sends what exceeds it to the extended-precision arbiter: seed 4 case 83 is 2.4 x
                                                            #  over on a cond 1e7 system in another launch shape; a miscompiled twin is wrong by O(1))
    os.environ.pop('DGP_FORCE_SHAPE', None)
    kwc = dict(qc=qc if qc_step is None else qc_step, ow=ow, eps=eps, q_full=q_full, io=io)
    fw = be.step_errors(p, th, start, goal, sdf, **kwc)
    npdt = np.float64 if io == 'f64' else np.float32
    th_new = (th.astype(npdt) + fw[0].astype(npdt)).astype(np.float64)
    _, _, s2, g2, o2 = be.eval_errors(p, th_new, start, goal, sdf, eps=eps, io=io)
    e1 = np.abs(fw[0] - c_dth).reshape(B, -1).max(1) / (np.abs(c_dth).reshape(B, -1).max(1) + 1e-300)
    tu = 1e-9 if io == 'f64' else 2e-4
    e2 = max(PC.rel_err(fw[4], s2), PC.rel_err(fw[5], g2), PC.rel_err(fw[6], o2))
    if not e1.max() < tw:
      # (round 6, seed 92 case 50: 2.5 x over on an fp64 draw.)  Same arbiter as above: against the extended-precision solution the twin may be no further off than 3 x the STANDARD
      # kernel's result of the same system (which the main check has just accepted or arbitrated) or 3 x the fp64 C oracle -- a miscompiled twin is wrong by O(1), not by a factor
      xt, _, _, _ = BT.gn_step(p, th, start, goal, sdf, qc=qc, ow=None if ow is None else ow.reshape(sh), eps=None if eps is None else eps.reshape(sh), q_full=q_full, nthreads=4, extended=True)
      xs_ = np.abs(xt).reshape(B, -1).max(1) + 1e-300
      e_tw, e_std, e_co = (np.abs(a_ - xt).reshape(B, -1).max(1) / xs_ for a_ in (fw[0], dth, c_dth))
      over = e1 >= tw
      assert np.all(e_tw[over] <= 3.0 * np.maximum(e_std[over], e_co[over]) + 1e-300), ('step_errors', case, e1.max(), e_tw[over].max(), e_std[over].max(), e_co[over].max())
      print('%3d twin cond(step-errors twin %.1e, standard kernel %.1e, fp64 C oracle %.1e off the extended-precision solve)' % (case, e_tw[over].max(), e_std[over].max(), e_co[over].max()), flush=True)
    assert e2 < tu and not fw[3].any(), ('step_errors', case, e1.max(), e2)
    if n <= 128 and W >= 4 and H >= 4:
      be.sdf_tiled = True
      try: dt_, et_, xt_, it_ = be.step(p, th, start, goal, sdf, **kwc)
      finally: be.sdf_tiled = False
      e3 = np.abs(dt_ - c_dth).reshape(B, -1).max(1) / (np.abs(c_dth).reshape(B, -1).max(1) + 1e-300)
      if not e3.max() < tw:      # (round 6, seed 173 case 182: 5.0e-7 on a system where the standard kernel itself is 4.3e-7 off the extended-precision solve.)  The same arbiter
        xt, _, _, _ = BT.gn_step(p, th, start, goal, sdf, qc=qc, ow=None if ow is None else ow.reshape(sh), eps=None if eps is None else eps.reshape(sh), q_full=q_full, nthreads=4, extended=True)
        xs_ = np.abs(xt).reshape(B, -1).max(1) + 1e-300
        e_tl, e_std, e_co = (np.abs(a_ - xt).reshape(B, -1).max(1) / xs_ for a_ in (dt_, dth, c_dth))
        over = e3 >= tw
        assert np.all(e_tl[over] <= 3.0 * np.maximum(e_std[over], e_co[over]) + 1e-300), ('tiled grids', case, e3.max(), e_tl[over].max(), e_std[over].max(), e_co[over].max())
        print('%3d twin cond(tiled twin %.1e, standard kernel %.1e, fp64 C oracle %.1e off the extended-precision solve)' % (case, e_tl[over].max(), e_std[over].max(), e_co[over].max()), flush=True)
      assert not it_.any(), ('tiled grids', case, e3.max())
    if forced: os.environ['DGP_FORCE_SHAPE'] = '%d,%d' % forced
  # (not for n > 256 with fp32 I/O: the loop kernels keep the fused loop's state in th_out, i.e. rounded to fp32 between iterations, so neither the f64-I/O
  #  loop nor a host-side chain of f32 steps -- which rounds dtheta AND the sum -- is a reference at better than cond(Lambda) x 6e-8 per iteration;
  #  tests/parity_cases.py::case_long_trajectories pins that path)
  if case % 3 == 0 and ok.all() and not (io == 'f32' and n > 256):      # the fused loop (dgp_gn_solve) on the same configuration
    tho, its, eh, eeh, ef, sinfo = be.solve(p, th, start, goal, sdf, 3, 0.0, qc=qc, ow=ow, eps=eps, q_full=q_full, io=io)
    if io == 'f64':                   # three iterations == three chained steps
      cur = th.copy(); good = True
      for k in range(3):
        d_k, e_k, _, i_k = be.step(p, cur, start, goal, sdf, qc=qc, ow=ow, eps=eps, q_full=q_full, io=io)
        good = good and not i_k.any() and np.all(np.isfinite(d_k))
        if not good: break
        cur = cur + d_k
      ref_name = 'chained steps'
    else:                             # fp32 I/O: chained steps would round the state to fp32 between iterations (the fused loop keeps it in
      cur, _, _, _, _, i2 = be.solve(p, th, start, goal, sdf, 3, 0.0, qc=qc, ow=ow, eps=eps, q_full=q_full, io='f64')   # fp64, and GN amplifies
      good = not i2.any() and np.all(np.isfinite(cur)); ref_name = 'the f64-I/O fused loop on the same fp32-rounded inputs'    # that); compare the two I/O builds
    if good and not sinfo.any():
      es = np.abs(tho - cur).max() / (np.abs(cur).max() + 1e-300)
      bound = (1e-7 if io == 'f64' else 1e-5) * (30 if p.reg < 0.01 else 1)
      if not es < bound:
        # (round 6, seed 196 case 24: 4.5e-6 on a weakly regularised d = 6 system -- the fused loop and the step kernels run in different launch shapes, three Gauss-Newton
        #  iterations amplify the rounding.)  Arbiter: three chained steps of the extended-precision C oracle; the fused loop may be no further from them than 3 x the reference is
        xc = th.copy()
        for k in range(3):
          xd, _, _, _ = BT.gn_step(p, xc, start, goal, sdf, qc=qc, ow=None if ow is None else ow.reshape(sh), eps=None if eps is None else eps.reshape(sh), q_full=q_full, nthreads=4, extended=True)
          xc = xc + xd
        nx = np.abs(xc).max() + 1e-300
        e_f, e_r = np.abs(tho - xc).max() / nx, np.abs(cur - xc).max() / nx
        assert e_f <= 3.0 * max(e_r, bound), ('fused loop differs from ' + ref_name, case, es, e_f, e_r)
        print('%3d fused-loop cond(fused loop %.1e, %s %.1e off three extended-precision steps)' % (case, e_f, ref_name, e_r), flush=True)
  # the backward kernel of the same configuration against an INDEPENDENT gradient oracle: torch autograd over the dense restatement of
  # the reference's step (oracle/autograd_torch.py; pinned to the reference's own autograd fixtures) -- every backward variant, on
  # batches small enough for the dense solve
  if B <= 8 and n <= 65 and ok.all() and be.kind == 'hip':
    from oracle import autograd_torch as AT
    gbar = rs.randn(B, n, d); gext = rs.randn(B)
    shared = sdf.shape[0] == 1
    kwb = dict(qc=qc, ow=ow, eps=eps, q_full=q_full, io=io, sdf_copies=(16 if shared and case % 2 else 1))
    rh = be.backward(p, th, start, goal, sdf, dth, r(gbar), r(gext), **dict(kwb, qc=qc if qc_step is None else qc_step))      # (scalar-mode cases: the scaled-mask backward)
    ro = AT.step_gradients(p, th, start, goal, sdf, r(gbar), r(gext), qc=qc, ow=ow, eps=eps, q_full=q_full)
    for key in ('th', 'start', 'goal', 'sdf', 'qc', 'ow', 'eps'):
      if rh[key] is None: continue
      if key == 'sdf' and io == 'f32': continue      # accumulated in fp32 IN MEMORY by atomics: the order-dependent cancellation noise of ~1e7-sized summands is not a code-generation signal
      a_, b_ = rh[key], ro[key].reshape(rh[key].shape if not (key == 'sdf' and rh[key].shape[0] != sdf.shape[0]) else sdf.shape)
      if key == 'sdf' and a_.shape[0] != sdf.shape[0]: a_ = a_.sum(0, keepdims=True)
      # (the SDF gradient is a sum of signed tap contributions: judged against the size of the summands, for which the trajectory
      #  gradient stands in, when the sum itself cancels)
      # (... and a gradient tensor that is itself below the resolution of the I/O type relative to the largest one -- dL/d obs_w of 1e-10 next to a
      #  trajectory gradient of 1e2 with fp32 I/O -- is judged against that resolution, not against its own size)
      # (fp64 I/O: 1e-11 of the largest gradient -- seed 13, case 194: dL/d obs_w of 5e-12 next to a trajectory gradient of 59 agreed to 4e-17 absolute, 9e-6 of itself)
      # (round 6, seed 109 case 151: dL/d obs_w of 4.2e-10 agreed to 6e-16 absolute, 1.4e-6 of itself, with the trajectory gradient too small for the 1e-11 floor: 1e-10)
      floor = (1e-7 if io == 'f32' else 1e-10) * np.abs(ro['th']).max()
      eb = np.abs(a_ - b_).max() / max(np.abs(b_).max(), np.abs(ro['th']).max() if key == 'sdf' else 0.0, floor, 1e-300)
      # (fp32 I/O: the kernels rebuild rho = e - H dtheta from the fp32-ROUNDED forward output, the oracle from its own fp64 one: cond(Lambda) * 6e-8)
      assert eb < (1e-6 if io == 'f64' else 2e-3) * (30 if p.reg < 0.01 else 1),  ('backward differs from the autograd oracle', case, key, eb, dict(dof=dof, n=n, B=B, io=io, shape=forced, H=H, W=W, per_sample=per_sample, cov=cov, copies=kwb['sdf_copies'], amax=float(np.abs(a_).max()), bmax=float(np.abs(b_).max()), thmax=float(np.abs(ro['th']).max())))
  # round 4: the fused loop's backward (dgp_gn_solve_backward) == the single-step backward kernels chained by hand through the traced loop's history
  if cov == 'static' and qmode != 'full' and n <= 256 and B <= 65 and io == 'f64' and ok.all() and be.kind == 'hip' and case % 2 == 0:
    K = 3
    nrm = np.sqrt((dth.reshape(B, -1) ** 2).sum(1))
    tolc = float(np.median(nrm))                     # about half of the trajectories stop after the first iteration
    tho, its, hist, sinfo = be.solve_traced(p, th, start, goal, sdf, K, tolc, io=io)
    if not sinfo.any() and np.all(np.isfinite(tho)):
      gb = rs.randn(B, n, d)
      rc = be.solve_backward(p, start, goal, sdf, K, hist, tho, its, gb, io=io, want_sdf=False)
      gcur = gb.copy(); a_s = np.zeros_like(start); a_g = np.zeros_like(goal)
      for k in range(K - 1, -1, -1):
        on = its > k
        thk = np.where(on[:, None, None], np.nan_to_num(hist[k]), tho)
        nxt = np.where((its > k + 1)[:, None, None], np.nan_to_num(hist[min(k + 1, K - 1)]), tho)
        one = be.backward(p, thk, start, goal, sdf, nxt - thk, gcur * on[:, None, None], None, io=io)
        gcur = gcur + one['th'] * on[:, None, None]; a_s += one['start'] * on[:, None, None]; a_g += one['goal'] * on[:, None, None]
      ec = max(np.abs(rc['th'] - gcur).max() / (np.abs(gcur).max() + 1e-300), np.abs(rc['start'] - a_s).max() / (np.abs(a_s).max() + 1e-300),
               np.abs(rc['goal'] - a_g).max() / (np.abs(a_g).max() + 1e-300))
      assert ec < 1e-8 * (30 if p.reg < 0.01 else 1), ('chain backward differs from the chained single-step backward', case, ec, dict(dof=dof, n=n, B=B, shape=forced, its=its.tolist()))
print('%d cases: %d within tolerance of the fp64 C oracle, %d beyond it but no further from the extended-precision solve than 3 x the fp64 C oracle is, or with a normwise backward error below 1e-11, or a forward error below cond x 2^-52 (%s), 0 failed; worst dtheta error / tolerance = %.2f'
      % (args.cases, args.cases - len(conditioned), len(conditioned), ','.join(map(str, conditioned)) or '-', worst))
