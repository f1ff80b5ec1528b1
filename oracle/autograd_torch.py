"""Independent GRADIENT oracle (TEST INFRASTRUCTURE, not product code): the reference's dense Gauss-Newton step restated in
differentiable PyTorch-CPU fp64 for EVERY configuration the kernels implement -- both robots, velocity-limit and non-holonomic
factors, static / per-state Q_c^-1 / full Q^-1 ('q_full') covariances -- so that torch autograd over it gives the gradients the
reference's own autograd would give where the reference's batched path cannot run, and the very same ones where
it can (pinned by tests/golden/g5_grads.npz and g7_errors.npz in tests/test_oracle_golden.py::test_autograd_oracle_*).

It shares NO code with the kernels or their host emulator (tests/emul): dense A (B,M,N), b, K (B,M,M) are assembled factor by factor
as plan_layer.py:152-200 does, Lambda = A^T K A + delta I (:217-219) is factorised with torch.linalg.cholesky and solved through the
two explicit triangular inverses of :226-228 (so that d L / d Lambda is symmetrised exactly as in the reference -- this fixes the
convention of the q_full gradient), and the errors are plan_layer.py:273-345 / :374-388.

Factor restatements (file:line in /root/reference/diff_gpmp2/):
  prior gpmp2/gp/prior_factor.py:15-18 | GP gpmp2/gp/gp_factor.py:31-37,65-73,100-110 | obstacle gpmp2/obstacle/obstacle_factor.py:35-40,
  obstacle_cost.py:29-38, utils/sdf_utils.py:38-107 | velocity limit gpmp2/custom_factors/velocity_limit_factor.py:17-29 |
  non-holonomic gpmp2/custom_factors/nonholonomic_factor.py:16-30 (H as the reference writes it; autograd differentiates THROUGH H).
Only tests/ may import this.
"""
import torch

from . import dense_torch as DT


def _zeros(*shape):
  return torch.zeros(*shape, dtype=torch.float64)


def factors(th, start, goal, sdf, eps, p):
  """-> list of (e (B,k,1), H (B,k,N) dense Jacobian rows with the reference's sign convention b = e, A = H) per factor group,
  in any order (the row order of A does not enter Lambda)."""
  B, n, d = th.shape
  dof = d // 2
  N = n * d
  rows = []
  I = torch.eye(d, dtype=torch.float64)
  # priors (prior_factor.py:15-18): e = mu - x, H = I
  for idx, mu in ((0, start), (n - 1, goal)):
    e = (mu.reshape(B, d) - th[:, idx]).reshape(B, d, 1)
    H = _zeros(B, d, N); H[:, :, idx * d:(idx + 1) * d] = I
    rows.append(('prior_s' if idx == 0 else 'prior_g', e, H))
  # GP factors (gp_factor.py:100-110): e = x_{i+1} - Phi x_i, H1 = Phi, H2 = -I
  phi = DT._phi(dof, p.dt, torch.float64)
  e_gp = th[:, 1:] - torch.einsum('ij,bsj->bsi', phi, th[:, :-1])
  H = _zeros(B, (n - 1) * d, N)
  for i in range(n - 1):
    H[:, i * d:(i + 1) * d, i * d:(i + 1) * d] = phi
    H[:, i * d:(i + 1) * d, (i + 1) * d:(i + 2) * d] = -I
  rows.append(('gp', e_gp.reshape(B, (n - 1) * d, 1), H))
  # obstacle (obstacle_cost.py:29-38): hinge on the bilinear distance, H = -J (sphere centre = x[0:2], H_fk = I_d[0:2,:])
  res = (p.x_lims[1] - p.x_lims[0]) / sdf.shape[-1]
  dist, J = DT._bilinear(sdf[:, 0], th[:, :, 0:2].contiguous(), res, p.x_lims, p.y_lims)
  eps_tot = eps.reshape(B, n, 1) + p.radius
  act = dist <= eps_tot
  cost = torch.where(act, eps_tot - dist, _zeros(1))
  He = torch.where(act, -1.0 * J, _zeros(1, 2))
  H = _zeros(B, n, N)
  for i in range(n):
    H[:, i, i * d:i * d + 2] = He[:, i]
  rows.append(('obs', cost.reshape(B, n, 1), H))
  if p.non_holonomic:      # nonholonomic_factor.py:16-30
    t, vx, vy = th[:, :, 2], th[:, :, 3], th[:, :, 4]
    e = vy * torch.cos(t) - vx * torch.sin(t)
    H = _zeros(B, n, N)
    for i in range(n):
      H[:, i, i * d + 2] = -vy[:, i] * torch.sin(t[:, i]) + vx[:, i] * torch.cos(t[:, i])
      H[:, i, i * d + 3] = -torch.sin(t[:, i])
      H[:, i, i * d + 4] = torch.cos(t[:, i])
    rows.append(('dyn', e.reshape(B, n, 1), H))
  if p.use_vel_limits:     # velocity_limit_factor.py:17-29 ('>=', H = -sign(v) on the velocity column)
    es, Hs = [], []
    for a, vmax in enumerate((p.v_x, p.v_y)):
      v = th[:, :, dof + a]
      actv = torch.abs(v) >= vmax
      es.append(torch.where(actv, torch.abs(v) - vmax, _zeros(1)))
      Hv = _zeros(B, n, N)
      hv = torch.where(actv, -torch.sign(v), _zeros(1))
      for i in range(n):
        Hv[:, i, i * d + dof + a] = hv[:, i]
      Hs.append(Hv)
    rows.append(('vel', torch.cat(es, 1).reshape(B, 2 * n, 1), torch.cat(Hs, 1)))
  return rows


def _weights(name, B, n, d, Q_inv, ow, p):
  """K block (B,k,k) of one factor group."""
  I = torch.eye(d, dtype=torch.float64)
  if name == 'prior_s': return ((1.0 / p.K_s ** 2.0) * I).expand(B, d, d)
  if name == 'prior_g': return ((1.0 / p.K_g ** 2.0) * I).expand(B, d, d)
  if name == 'gp': return torch.stack([torch.block_diag(*Q_inv[b]) for b in range(B)], 0)
  if name == 'obs': return torch.diag_embed(ow.reshape(B, n))
  if name == 'dyn': return ((1.0 / p.K_d ** 2.0) * torch.eye(n, dtype=torch.float64)).expand(B, n, n)
  if name == 'vel': return ((1.0 / p.K_v ** 2.0) * torch.eye(2 * n, dtype=torch.float64)).expand(B, 2 * n, 2 * n)
  raise KeyError(name)


def _maha(rows, B, n, d, Q_inv, ow, p):
  err = _zeros(B)
  for name, e, H in rows:
    K = _weights(name, B, n, d, Q_inv, ow, p)
    err = err + 0.5 * torch.einsum('bi,bij,bj->b', e[:, :, 0], K, e[:, :, 0])
  return (err / p.M).reshape(B, 1, 1)


def plan_layer_forward(th, start, goal, sdf, qc, ow, eps, p, q_full=False):
  """plan_layer.py:87-99 on fp64 CPU tensors (any of which may require grad): th (B,n,d), start / goal (B,1,d), sdf (B,1,H,W),
  qc (B,n-1,dof,dof) or (q_full) Q^-1 (B,n-1,d,d), ow / eps (B,n,1,1); p: oracle.gpmp2_oracle.OracleParams.
  -> dtheta (B,n,d), err (B,1,1) [detached, :275], err_ext (B,1,1)."""
  B, n, d = th.shape
  N = n * d
  Q_inv = qc if q_full else DT._q_inv(qc, p.dt)
  rows = factors(th, start, goal, sdf, eps, p)
  A = torch.cat([H for _, _, H in rows], 1)
  b = torch.cat([e for _, e, _ in rows], 1)
  K = torch.stack([torch.block_diag(*[_weights(name, B, n, d, Q_inv, ow, p)[bb] for name, _, _ in rows]) for bb in range(B)], 0)
  # (M counts dof rows per state for the velocity-limit factor, plan_layer.py:45, the factor itself has two -- v_x, v_y,
  #  velocity_limit_factor.py:17-29 -- so for the (x,y,theta) robot M, the normaliser of err, exceeds the row count by n)
  assert A.shape[1] == p.M - ((d // 2 - 2) * n if p.use_vel_limits else 0), (A.shape, p.M)
  AtK = torch.bmm(A.transpose(1, 2), K)                                   # plan_layer.py:217-220
  LAM = torch.bmm(AtK, A) + p.reg * torch.eye(N, dtype=torch.float64)
  R = torch.bmm(AtK, b)
  u = torch.linalg.cholesky(LAM).mH                                       # :226 (upper)
  z = torch.bmm(torch.inverse(u.transpose(1, 2)), R)                      # :227
  dth = torch.bmm(torch.inverse(u), z).view(B, n, d)                      # :228
  with torch.no_grad():
    err = _maha(factors(th, start, goal, sdf, eps, p), B, n, d, Q_inv, ow, p)          # :97, :273-308
  dof = d // 2
  qf = torch.as_tensor(p.Q_c_inv, dtype=torch.float64).expand(B, n - 1, dof, dof)
  err_ext = _maha(factors(th, start, goal, sdf, eps, p), B, n, d, DT._q_inv(qf, p.dt),
                  torch.full((B, n, 1, 1), 1.0 / p.cost_sigma ** 2.0, dtype=torch.float64), p)   # :98, :310-345 (fixed weights, current eps)
  return dth, err, err_ext


def normal_equations(th, start, goal, sdf, qc, ow, eps, p, q_full=False):
  """(Lambda (B,N,N), eta (B,N,1)) = (A^T K A + delta I, A^T K b) of plan_layer.py:217-220 as fp64 torch tensors, every factor row present (velocity-limit AND
  non-holonomic rows side by side; the reference's own mask layout would let them overlap) -- the dense system the kernels solve."""
  B, n, d = th.shape
  Q_inv = qc if q_full else DT._q_inv(qc, p.dt)
  rows = factors(th, start, goal, sdf, eps, p)
  A = torch.cat([H for _, _, H in rows], 1)
  b = torch.cat([e for _, e, _ in rows], 1)
  K = torch.stack([torch.block_diag(*[_weights(name, B, n, d, Q_inv, ow, p)[bb] for name, _, _ in rows]) for bb in range(B)], 0)
  AtK = torch.bmm(A.transpose(1, 2), K)
  return torch.bmm(AtK, A) + p.reg * torch.eye(n * d, dtype=torch.float64), torch.bmm(AtK, b)


def unweighted_errors(th, start, goal, sdf, eps, p):
  """plan_layer.py:374-388 -> (err_sg (B,1), err_gp (B,1,1), err_obs (B,1,1))."""
  B, n, d = th.shape
  r = {name: e for name, e, _ in factors(th, start, goal, sdf, eps, p)}
  sg = (0.5 * (r['prior_s'][:, :, 0] ** 2).sum(1) + 0.5 * (r['prior_g'][:, :, 0] ** 2).sum(1)).reshape(B, 1)
  gp = (0.5 * (r['gp'].reshape(B, n - 1, d) ** 2).sum(-1)).mean(1).reshape(B, 1, 1)
  ob = (0.5 * r['obs'].reshape(B, n) ** 2).mean(1).reshape(B, 1, 1)
  return sg, gp, ob


def step_gradients(p, th, start, goal, sdf, gbar, gext, qc=None, ow=None, eps=None, q_full=False):
  """numpy in / numpy out: gradients of  sum(gbar * dtheta) + sum(gext * err_ext)  w.r.t. every input of one GN step -- the
  contract of dgp_gn_step_backward.  sdf (1,1,H,W) (shared: the gradient is summed over the batch) or (B,1,H,W)."""
  import numpy as np
  B, n, d = th.shape
  T = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64)))
  sq, so, se = p.static_covs(B)
  L = dict(th=T(th), start=T(start), goal=T(goal), sdf=T(sdf), qc=T(sq if qc is None else qc),
           ow=T(so if ow is None else np.reshape(ow, so.shape)), eps=T(se if eps is None else np.reshape(eps, se.shape)))
  for v in L.values(): v.requires_grad_(True)
  sdfB = L['sdf'].expand(B, *L['sdf'].shape[1:]) if L['sdf'].shape[0] == 1 else L['sdf']
  dth, err, eex = plan_layer_forward(L['th'], L['start'], L['goal'], sdfB, L['qc'], L['ow'], L['eps'], p, q_full=q_full)
  loss = (T(gbar) * dth).sum() + (T(gext).reshape(B, 1, 1) * eex).sum()
  names = list(L.keys())
  gr = torch.autograd.grad(loss, [L[k] for k in names], allow_unused=True)
  out = {k: (np.zeros(tuple(L[k].shape)) if g is None else g.numpy()) for k, g in zip(names, gr)}
  out['dtheta'] = dth.detach().numpy()
  return out
