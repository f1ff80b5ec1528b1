"""CPU baseline (TEST/BENCH INFRASTRUCTURE, not product code): the reference's dense op sequence restated in
PyTorch-CPU fp64, so that bench.py can time "the reference PyTorch-CPU path" on the GPU box's host cores (the
reference's own sources never travel there).  Op-for-op what PlanLayer.forward does (plan_layer.py:87-99):
  dense zero-filled A (B,M,N), b (B,M,1), K (B,M,M)  (:153-155) -> factor blocks written in (:169-198)
  -> bmm(A^T,K), bmm(.,A) + delta I, bmm(.,b) (:217-220) -> cholesky(upper) (:226) -> two explicit inverses (:227-228)
  -> error_batch (:273-308) and error_ext_batch (:310-345): two more full factor evaluations.
Only bench.py's cpu_baseline leg and tests may import this.  Validated against the golden fixtures in
tests/test_oracle_golden.py::test_dense_torch_baseline.
"""
import torch


_EXPLICIT_INVERSE = 'inverse'


def set_explicit_inverse(kind):
  """'inverse' (torch.inverse, what the reference calls, plan_layer.py:227-228) or 'solve_triangular' (explicit inverse of
  the triangular factor by a triangular solve against I; used on hosts whose MKL build rejects batched torch.inverse)."""
  global _EXPLICIT_INVERSE
  assert kind in ('inverse', 'solve_triangular')
  _EXPLICIT_INVERSE = kind


def _tri_inverse(t, upper):
  if _EXPLICIT_INVERSE == 'inverse':
    return torch.inverse(t)
  I = torch.eye(t.shape[-1], dtype=t.dtype).expand_as(t)
  return torch.linalg.solve_triangular(t, I, upper=upper)


def _phi(dof, dt, dtype):
  I = torch.eye(dof, dtype=dtype)
  return torch.cat((torch.cat((I, dt * I), 1), torch.cat((torch.zeros(dof, dof, dtype=dtype), I), 1)), 0)


def _q_inv(qc, dt):
  m1 = 12.0 * (dt ** -3.0) * qc; m2 = -6.0 * (dt ** -2.0) * qc; m3 = 4.0 * (dt ** -1.0) * qc
  return torch.cat((torch.cat((m1, m2), -1), torch.cat((m2, m3), -1)), -2)


def _bilinear(imb, pts, res, x_lims, y_lims):
  """utils/sdf_utils.py:38-107 (same quirks as oracle.gpmp2_oracle.bilinear_interpolate)."""
  B, S, _ = pts.shape
  H, W = imb.shape[-2], imb.shape[-1]
  px = ((0. - x_lims[0] / res) + pts[:, :, 0] / res).reshape(-1)
  py = ((0. - y_lims[0] / res) - pts[:, :, 1] / res).reshape(-1)
  px1 = torch.floor(px).long(); px2 = px1 + 1
  py1 = torch.floor(py).long(); py2 = py1 + 1
  px1 = px1.clamp(0, W - 1); px2 = px2.clamp(0, W - 1); py1 = py1.clamp(0, H - 1); py2 = py2.clamp(0, H - 1)
  pz = torch.arange(B).repeat_interleave(S)
  d11 = imb[pz, py1, px1]; d21 = imb[pz, py1, px2]; d12 = imb[pz, py2, px1]; d22 = imb[pz, py2, px2]
  fx1, fx2, fy1, fy2 = px1.double(), px2.double(), py1.double(), py2.double()
  d = (fx2 - px) * (fy2 - py) * d11 + (px - fx1) * (fy2 - py) * d21 + (fx2 - px) * (py - fy1) * d12 + (px - fx1) * (py - fy1) * d22
  Jx = (-1.0 * ((fy2 - py) * (d21 - d11) + (py - fy1) * (d22 - d12)) / res)
  Jy = (((fx2 - px) * (d12 - d11) + (px - fx1) * (d22 - d21)) / res)
  return d.reshape(B, S, 1), torch.stack((Jx, Jy), -1).reshape(B, S, 2)


def _factors(th, start, goal, sdf, eps, P):
  B, n, d = th.shape
  dof = d // 2
  phi = _phi(dof, P['dt'], th.dtype)
  e_s = (start.reshape(B, d) - th[:, 0]).reshape(B, d, 1)
  e_g = (goal.reshape(B, d) - th[:, n - 1]).reshape(B, d, 1)
  e_gp = (th[:, 1:].transpose(1, 2) - torch.bmm(phi.unsqueeze(0).repeat(B, 1, 1), th[:, :-1].transpose(1, 2))).transpose(1, 2).unsqueeze(-1)
  res = (P['x_lims'][1] - P['x_lims'][0]) / sdf.shape[-1]
  dist, J = _bilinear(sdf[:, 0], th[:, :, 0:2].contiguous(), res, P['x_lims'], P['y_lims'])
  eps_tot = eps.reshape(B, n, 1) + P['radius']
  act = dist <= eps_tot
  cost = torch.where(act, eps_tot - dist, torch.zeros((), dtype=th.dtype))
  He = torch.where(act, -1.0 * J, torch.zeros(1, 2, dtype=th.dtype))
  H_o = torch.zeros(B, n, 1, d, dtype=th.dtype); H_o[:, :, 0, 0:2] = He
  return e_s, e_g, e_gp, phi, cost.reshape(B, n, 1, 1), H_o


def _error(e_s, e_g, e_gp, e_o, Q_inv, ow, P):
  err = 0.5 * P['w_s'] * torch.bmm(e_s.transpose(1, 2), e_s) + 0.5 * P['w_g'] * torch.bmm(e_g.transpose(1, 2), e_g)
  err = err + torch.sum(0.5 * torch.einsum('bsij,bsjk->bsik', torch.einsum('bsij,bsjk->bsik', e_gp.transpose(2, 3), Q_inv), e_gp), dim=1)
  err = err + torch.sum(0.5 * torch.einsum('bsij,bsjk->bsik', torch.einsum('bsij,bsjk->bsik', e_o.transpose(2, 3), ow), e_o), dim=1)
  return err / P['M']


def plan_layer_forward(th, start, goal, sdf, qc, ow, eps, P):
  """th (B,n,d) fp64 CPU tensors; qc (B,n-1,dof,dof); ow, eps (B,n,1,1).  P: dict(dt,x_lims,y_lims,radius,w_s,w_g,reg,M,
  qc_fix (dof,dof), ow_fix).  2-D point robot / xyh without the custom factors (the timed C2 workload)."""
  B, n, d = th.shape
  M, N = P['M'], n * d
  A = torch.zeros(B, M, N, dtype=th.dtype); b = torch.zeros(B, M, 1, dtype=th.dtype); K = torch.zeros(B, M, M, dtype=th.dtype)
  e_s, e_g, e_gp, phi, e_o, H_o = _factors(th, start, goal, sdf, eps, P)
  Q_inv = _q_inv(qc, P['dt'])
  I = torch.eye(d, dtype=th.dtype)
  A[:, 0:d, 0:d] = I; b[:, 0:d] = e_s; K[:, 0:d, 0:d] = P['w_s'] * I
  for i in range(n - 1):
    r = slice((i + 1) * d, (i + 2) * d)
    A[:, r, i * d:(i + 1) * d] = phi; A[:, r, (i + 1) * d:(i + 2) * d] = -I
    b[:, r] = e_gp[:, i]; K[:, r, r] = Q_inv[:, i]
  off = d * n
  A[:, off:off + d, N - d:N] = I; b[:, off:off + d] = e_g; K[:, off:off + d, off:off + d] = P['w_g'] * I
  off += d
  idx = torch.arange(n)
  A[:, off + idx, :] = 0
  for i in range(n):
    A[:, off + i, i * d:(i + 1) * d] = H_o[:, i, 0]
  b[:, off:off + n, 0] = e_o[:, :, 0, 0]
  K[:, off + idx, off + idx] = ow[:, :, 0, 0]
  # plan_layer.py:214-228
  Id = torch.eye(N, dtype=th.dtype).unsqueeze(0).repeat(B, 1, 1)
  AtK = torch.bmm(A.transpose(1, 2), K)
  LAM = torch.bmm(AtK, A) + P['reg'] * Id
  R = torch.bmm(AtK, b)
  u = torch.linalg.cholesky(LAM).mH
  z = torch.bmm(_tri_inverse(u.transpose(1, 2), upper=False), R)
  dth = torch.bmm(_tri_inverse(u, upper=True), z).view(B, n, d)
  # plan_layer.py:97-98: two more full factor evaluations
  e_s, e_g, e_gp, _, e_o, _ = _factors(th, start, goal, sdf, eps, P)
  err = _error(e_s, e_g, e_gp, e_o, Q_inv, ow, P)
  e_s, e_g, e_gp, _, e_o, _ = _factors(th, start, goal, sdf, eps, P)
  Qf = _q_inv(P['qc_fix'].expand(B, n - 1, d // 2, d // 2), P['dt'])
  err_ext = _error(e_s, e_g, e_gp, e_o, Qf, torch.full_like(ow, P['ow_fix']), P)
  return dth, err, err_ext


def params_from_oracle(p):
  import numpy as np
  return dict(dt=p.dt, x_lims=p.x_lims, y_lims=p.y_lims, radius=p.radius, w_s=1.0 / p.K_s ** 2.0, w_g=1.0 / p.K_g ** 2.0,
              reg=p.reg, M=p.M, qc_fix=torch.from_numpy(np.asarray(p.Q_c_inv, dtype=np.float64)), ow_fix=1.0 / p.cost_sigma ** 2.0)
