"""CPU oracle (TEST INFRASTRUCTURE, not product code) for the dGPMP2 inner Gauss-Newton step.

A literal numpy/fp64 restatement of the reference's dense algorithm:
dense A (B,M,N), b (B,M,1), block-diagonal K (B,M,M) -> LAM = A^T K A + delta I ->
Cholesky + two explicit inverses, exactly in the reference's op order.  Every function cites the
reference file:line (paths relative to /root/reference/diff_gpmp2/) it follows.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (dgpmp2_amd/) never does.

Parity pin: the reference has no golden vectors of its own (its test/ dir holds py2 plotting
scripts without assertions).  This oracle is pinned against outputs of the reference itself,
generated in the build container by tests/golden/make_golden.py (which imports /root/reference)
and committed as tests/golden/*.npz; tests/test_oracle_golden.py checks every fixture.
"""
import numpy as np


# --------------------------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------------------------
class OracleParams(object):
  """Plain container mirroring the reference's param dicts (examples/configs/*.yaml)."""

  def __init__(self, dof=2, total_time_sec=10.0, total_time_step=63, Q_c_inv=None, K_s=0.01, K_g=0.01,
               cost_sigma=0.01, epsilon_dist=0.4, radius=0.4, reg=0.1, x_lims=(-5.0, 5.0), y_lims=(-5.0, 5.0),
               non_holonomic=False, use_vel_limits=False, K_d=0.01, K_v=0.01, v_x=1.0, v_y=1.0, nlinks=1):
    self.dof = int(dof)
    self.state_dim = 2 * self.dof
    self.total_time_sec = float(total_time_sec)
    self.total_time_step = int(total_time_step)
    self.n = self.total_time_step + 1                                   # plan_layer.py:30
    self.dt = self.total_time_sec * 1.0 / self.total_time_step * 1.0    # plan_layer.py:31
    self.Q_c_inv = np.eye(self.dof) if Q_c_inv is None else np.asarray(Q_c_inv, dtype=np.float64)
    self.K_s, self.K_g = float(K_s), float(K_g)
    self.cost_sigma, self.epsilon_dist, self.radius = float(cost_sigma), float(epsilon_dist), float(radius)
    self.reg = float(reg)
    self.x_lims, self.y_lims = tuple(map(float, x_lims)), tuple(map(float, y_lims))
    self.non_holonomic, self.use_vel_limits = bool(non_holonomic), bool(use_vel_limits)
    self.K_d, self.K_v, self.v_x, self.v_y = float(K_d), float(K_v), float(v_x), float(v_y)
    self.nlinks = int(nlinks)
    # plan_layer.py:39-46
    d, n = self.state_dim, self.n
    self.M = d * ((n - 1) + 2) + n * self.nlinks
    if self.non_holonomic: self.M += n
    if self.use_vel_limits: self.M += self.dof * n
    self.N = d * n

  def static_covs(self, B):
    """diff_gpmp2_planner.py:41-50,202-205 -- static covariances expanded over batch and time."""
    qc = np.broadcast_to(self.Q_c_inv, (B, self.n - 1, self.dof, self.dof)).copy()
    ow = np.full((B, self.n, self.nlinks, 1), 1.0 / self.cost_sigma ** 2)
    eps = np.full((B, self.n, self.nlinks, 1), self.epsilon_dist)
    return qc, ow, eps


# --------------------------------------------------------------------------------------------
# factors
# --------------------------------------------------------------------------------------------
def calc_phi(dof, dt):
  """gpmp2/gp/gp_factor.py:31-37."""
  I = np.eye(dof)
  return np.block([[I, dt * I], [np.zeros((dof, dof)), I]])


def calc_Q_inv_batch(Q_c_inv, dt):
  """gpmp2/gp/gp_factor.py:65-73.  Q_c_inv (B,n-1,dof,dof) -> Q_inv (B,n-1,d,d)."""
  m1 = 12.0 * (dt ** -3.0) * Q_c_inv
  m2 = -6.0 * (dt ** -2.0) * Q_c_inv
  m3 = 4.0 * (dt ** -1.0) * Q_c_inv
  up = np.concatenate((m1, m2), axis=-1)
  lo = np.concatenate((m2, m3), axis=-1)
  return np.concatenate((up, lo), axis=-2)


def gp_factor_error(thb, dof, dt):
  """gpmp2/gp/gp_factor.py:100-110.  err = x_{i+1} - Phi x_i (B,n-1,d,1); H1 = Phi; H2 = -I."""
  B, n, d = thb.shape
  phi = calc_phi(dof, dt)
  s1, s2 = thb[:, :-1, :], thb[:, 1:, :]
  err = s2 - np.einsum('ij,bsj->bsi', phi, s1)
  H1 = np.broadcast_to(phi, (B, n - 1, d, d))
  H2 = np.broadcast_to(-np.eye(d), (B, n - 1, d, d))
  return err[..., None], H1, H2


def prior_error(meanb, stateb):
  """gpmp2/gp/prior_factor.py:15-18.  err = mean - state (B,d,1); H = +I."""
  B = stateb.shape[0]
  d = stateb.shape[-1]
  err = (meanb - stateb).reshape(B, d, 1)
  return err, np.broadcast_to(np.eye(d), (B, d, d))


def bilinear_interpolate(imb, stateb, res, x_lims, y_lims):
  """utils/sdf_utils.py:38-107.  imb (B,H,W) fp64, stateb (B,S,2) -> d_obs (B,S,1), J (B,S,2).

  Quirks kept: px2 = px1+1 is taken before clamping (:65); both are clamped (:69-72);
  weights use clamped ints against unclamped floats (:81-89); the in-limits test (:96-106) is a
  no-op on torch>=1.2 (bool + bool = OR, compared with 1 -> always true), so MAX_D is never used.
  """
  B, S, _ = stateb.shape
  H, W = imb.shape[-2], imb.shape[-1]
  orig_pix_x = (0. - x_lims[0] / res)                       # :57
  orig_pix_y = (0. - y_lims[0] / res)                       # :58
  px = (orig_pix_x + stateb[:, :, 0] / res).reshape(-1)     # :61
  py = (orig_pix_y - stateb[:, :, 1] / res).reshape(-1)     # :62
  px1 = np.floor(px).astype(np.int64); px2 = px1 + 1        # :64-65
  py1 = np.floor(py).astype(np.int64); py2 = py1 + 1        # :66-67
  px1 = np.clip(px1, 0, W - 1); px2 = np.clip(px2, 0, W - 1)   # :69-70
  py1 = np.clip(py1, 0, H - 1); py2 = np.clip(py2, 0, H - 1)   # :71-72
  pz = np.repeat(np.arange(B), S)                           # :73-74
  dx1y1 = imb[pz, py1, px1]; dx2y1 = imb[pz, py1, px2]      # :76-77
  dx1y2 = imb[pz, py2, px1]; dx2y2 = imb[pz, py2, px2]      # :78-79
  fx1, fx2, fy1, fy2 = px1.astype(np.float64), px2.astype(np.float64), py1.astype(np.float64), py2.astype(np.float64)
  wa = (fx2 - px) * (fy2 - py); wb = (px - fx1) * (fy2 - py)   # :81-82
  wc = (fx2 - px) * (py - fy1); wd = (px - fx1) * (py - fy1)   # :83-84
  wja = (fy2 - py); wjb = (py - fy1); wjc = (fx2 - px); wjd = (px - fx1)   # :86-89
  d_obs = wa * dx1y1 + wb * dx2y1 + wc * dx1y2 + wd * dx2y2    # :90 (left-to-right)
  J = np.zeros((B, S, 2))
  J[:, :, 0] = (-1.0 * (wja * (dx2y1 - dx1y1) + wjb * (dx2y2 - dx1y2)) / res).reshape(B, S)   # :93
  J[:, :, 1] = ((wjc * (dx1y2 - dx1y1) + wjd * (dx2y2 - dx2y1)) / res).reshape(B, S)          # :94
  return d_obs.reshape(B, S, 1), J


def hinge_loss_signed_batch(centersb, r, epsb, sdfb, x_lims, y_lims):
  """gpmp2/obstacle/obstacle_cost.py:29-38.  centersb (B,n,nl,2), epsb (B,n,nl,1), sdfb (B,1,H,W)."""
  B, n, nl, _ = centersb.shape
  eps_tot = (epsb + r).reshape(B, n * nl, 1)                 # :30,33
  qpts = centersb.reshape(B, n * nl, -1)                     # :32
  res = (x_lims[1] - x_lims[0]) / (sdfb.shape[-1])           # :34  (SDF width incl. padding, Q3)
  dist, J = bilinear_interpolate(sdfb[:, 0], qpts, res, x_lims, y_lims)
  act = dist <= eps_tot                                      # :36 (<=, Q5)
  cost = np.where(act, eps_tot - dist, 0.0)
  H = np.where(act, -1.0 * J, 0.0)                           # :37
  return cost.reshape(B, n, nl, 1), H.reshape(centersb.shape)


def obstacle_error(thb, sdfb, epsb, p):
  """gpmp2/obstacle/obstacle_factor.py:35-40 + robot_models/point_robot_2d.py:58-63 (and the XYH
  model's H_fk, point_robot_xyh.py:28-36): sphere centre = state[0:2], H_fk = I_d[0:2,:]."""
  B, n, d = thb.shape
  centers = thb[:, :, 0:2].reshape(B, n, 1, 2)
  H_fk = np.zeros((2, d)); H_fk[0, 0] = 1.0; H_fk[1, 1] = 1.0
  err, H_e = hinge_loss_signed_batch(centers, p.radius, epsb, sdfb, p.x_lims, p.y_lims)
  H = np.einsum('bsij,jk->bsik', H_e, H_fk)                  # obstacle_factor.py:39
  return err, H


def vel_limit_error(thb, p):
  """gpmp2/custom_factors/velocity_limit_factor.py:17-29 applied per trajectory (the reference's
  batched call is broken -- SURVEY a9); vx = state[dof+0], vy = state[dof+1] (the reference
  hard-codes columns 2,3 for dof=2).  cost (B,n,dof,1), H (B,n,dof,d).  Note '>=' not '>'."""
  B, n, d = thb.shape
  dof = p.dof
  cost = np.zeros((B, n, dof, 1)); H = np.zeros((B, n, dof, d))
  for a, vmax in enumerate((p.v_x, p.v_y)):
    v = thb[:, :, dof + a]
    act = np.abs(v) >= vmax
    cost[:, :, a, 0] = np.where(act, np.abs(v) - vmax, 0.0)
    H[:, :, a, dof + a] = np.where(act, -np.sign(v), 0.0)
  return cost, H


def nonholonomic_error(thb):
  """gpmp2/custom_factors/nonholonomic_factor.py:16-30 applied per trajectory.
  state [x,y,th,vx,vy,w]; err = vy cos(th) - vx sin(th);
  H = [0,0,(-vy sin th + vx cos th), -sin th, cos th, 0] exactly as the reference writes it."""
  B, n, d = thb.shape
  th, vx, vy = thb[:, :, 2], thb[:, :, 3], thb[:, :, 4]
  err = vy * np.cos(th) - vx * np.sin(th)
  H = np.zeros((B, n, 1, d))
  H[:, :, 0, 2] = -vy * np.sin(th) + vx * np.cos(th)
  H[:, :, 0, 3] = -np.sin(th)
  H[:, :, 0, 4] = np.cos(th)
  return err.reshape(B, n, 1, 1), H


# --------------------------------------------------------------------------------------------
# dense linear system (plan_layer.py)
# --------------------------------------------------------------------------------------------
def construct_linear_system_batch(thb, startb, goalb, sdfb, Q_inv, obs_w, epsb, p):
  """gpmp2/plan_layer.py:152-200 with the row layout of create_factor_masks (:408-451):
  start rows 0:d, GP factor i rows (i+1)d:(i+2)d, goal rows d*n:d*(n+1), obstacle rows after,
  then dyn (1/state), then vel (dof/state).  Q_inv (B,n-1,d,d); obs_w (B,n,nl,1|nl)."""
  B, n, d = thb.shape
  M, N = p.M, p.N
  A = np.zeros((B, M, N)); b = np.zeros((B, M, 1)); K = np.zeros((B, M, M))
  # start prior (:157-158,169-171); weights plan_layer.py:64-68
  e, H = prior_error(startb.reshape(B, d), thb[:, 0])
  A[:, 0:d, 0:d] = H; b[:, 0:d] = e; K[:, 0:d, 0:d] = np.eye(d) * (1.0 / p.K_s ** 2.0)
  # GP factors (:160-161,173-176)
  e_gp, H1, H2 = gp_factor_error(thb, p.dof, p.dt)
  for i in range(n - 1):
    r = slice((i + 1) * d, (i + 2) * d)
    A[:, r, i * d:(i + 1) * d] = H1[:, i]
    A[:, r, (i + 1) * d:(i + 2) * d] = H2[:, i]
    b[:, r] = e_gp[:, i]
    K[:, r, r] = Q_inv[:, i]
  # goal prior (:163-164,178-180)
  off = d * n
  e, H = prior_error(goalb.reshape(B, d), thb[:, n - 1])
  A[:, off:off + d, N - d:N] = H; b[:, off:off + d] = e
  K[:, off:off + d, off:off + d] = np.eye(d) * (1.0 / p.K_g ** 2.0)
  # obstacle factors (:166-167,182-184)
  off += d
  e_o, H_o = obstacle_error(thb, sdfb, epsb, p)
  nl = p.nlinks
  for i in range(n):
    r = slice(off + i * nl, off + (i + 1) * nl)
    A[:, r, i * d:(i + 1) * d] = H_o[:, i]
    b[:, r] = e_o[:, i]
    K[:, r, r] = obs_w[:, i].reshape(B, nl, -1) if obs_w.shape[-1] == nl else obs_w[:, i].reshape(B, nl, 1) * np.eye(nl)
  off += n * nl
  if p.non_holonomic:                                       # :186-191, masks :433-441
    e_d, H_d = nonholonomic_error(thb)
    for i in range(n):
      A[:, off + i, i * d:(i + 1) * d] = H_d[:, i, 0]
      b[:, off + i] = e_d[:, i, 0]
      K[:, off + i, off + i] = 1.0 / p.K_d ** 2.0
    # NB plan_layer.py:442: offset is NOT advanced past the dyn rows (Q10)
  if p.use_vel_limits:                                      # :193-198, masks :443-451
    e_v, H_v = vel_limit_error(thb, p)
    dof = p.dof
    for i in range(n):
      r = slice(off + i * dof, off + (i + 1) * dof)
      A[:, r, i * d:(i + 1) * d] = H_v[:, i]
      b[:, r] = e_v[:, i]
      K[:, r, r] = np.eye(dof) * (1.0 / p.K_v ** 2.0)
  return A, b, K


def normal_equations(A, b, K, delta):
  """plan_layer.py:215-220: LAM = A^T K A + delta I ; R = A^T K b."""
  AtK = np.matmul(A.transpose(0, 2, 1), K)
  LAM = np.matmul(AtK, A) + delta * np.eye(A.shape[2])[None]
  R = np.matmul(AtK, b)
  return LAM, R


def solve_linear_system_batch(A, b, K, delta, n, d):
  """plan_layer.py:214-234: u = chol(LAM, upper); dtheta = inv(u) (inv(u^T) R).
  Raises numpy.linalg.LinAlgError when LAM is not SPD (the reference raises a torch RuntimeError)."""
  LAM, R = normal_equations(A, b, K, delta)
  u = np.linalg.cholesky(LAM).transpose(0, 2, 1)            # upper factor (:226)
  z = np.matmul(np.linalg.inv(u.transpose(0, 2, 1)), R)     # :227
  dtheta = np.matmul(np.linalg.inv(u), z)                   # :228
  return dtheta.reshape(A.shape[0], n, d)                   # :234


def _maha_sum(e, W):
  """0.5 * sum_s e_s^T W_s e_s with e (B,S,k,1), W (B,S,k,k) -> (B,1,1)  (plan_layer.py:283-284)."""
  return (0.5 * np.einsum('bsi,bsij,bsj->b', e[..., 0], W, e[..., 0])).reshape(-1, 1, 1)


def error_batch(thb, startb, goalb, sdfb, Q_inv, obs_w, epsb, p):
  """plan_layer.py:273-308 (error_batch) and, when called with the *fixed* Q_inv / obs_w but the
  current eps, plan_layer.py:310-345 (error_ext_batch).  Returns (B,1,1), normalised by M (:308)."""
  B, n, d = thb.shape
  err = np.zeros((B, 1, 1))
  e, _ = prior_error(startb.reshape(B, d), thb[:, 0])
  err = err + 0.5 * (1.0 / p.K_s ** 2.0) * np.sum(e * e, axis=1, keepdims=True)
  e_gp, _, _ = gp_factor_error(thb, p.dof, p.dt)
  err = err + _maha_sum(e_gp, Q_inv)
  e, _ = prior_error(goalb.reshape(B, d), thb[:, n - 1])
  err = err + 0.5 * (1.0 / p.K_g ** 2.0) * np.sum(e * e, axis=1, keepdims=True)
  e_o, _ = obstacle_error(thb, sdfb, epsb, p)
  nl = p.nlinks
  W = obs_w if obs_w.shape[-1] == nl and obs_w.shape[-2] == nl else obs_w.reshape(B, n, nl, 1) * np.eye(nl)
  err = err + _maha_sum(e_o, W.reshape(B, n, nl, nl))
  if p.non_holonomic:
    e_d, _ = nonholonomic_error(thb)
    err = err + 0.5 * (1.0 / p.K_d ** 2.0) * np.sum(e_d[..., 0, 0] ** 2, axis=1).reshape(B, 1, 1)
  if p.use_vel_limits:
    e_v, _ = vel_limit_error(thb, p)
    err = err + 0.5 * (1.0 / p.K_v ** 2.0) * np.sum(e_v[..., 0] ** 2, axis=(1, 2)).reshape(B, 1, 1)
  return err / p.M


def plan_layer_forward(thb, startb, goalb, sdfb, qc_inv_trajb, obscov_inv_trajb, eps_trajb, p, q_full=False):
  """plan_layer.py:87-99.  Returns (dtheta (B,n,d), err (B,1,1), err_ext (B,1,1)).
  q_full=True: qc_inv_trajb is a full (B,n-1,d,d) Q^-1 (plan_layer.py:90)."""
  B, n, d = thb.shape
  Q_inv = qc_inv_trajb if q_full else calc_Q_inv_batch(qc_inv_trajb, p.dt)
  A, b, K = construct_linear_system_batch(thb, startb, goalb, sdfb, Q_inv, obscov_inv_trajb, eps_trajb, p)
  dtheta = solve_linear_system_batch(A, b, K, p.reg, n, d)
  err = error_batch(thb, startb, goalb, sdfb, Q_inv, obscov_inv_trajb, eps_trajb, p)
  # fixed copies: plan_layer.py:70-81
  qc_fix, ow_fix, _ = p.static_covs(B)
  err_ext = error_batch(thb, startb, goalb, sdfb, calc_Q_inv_batch(qc_fix, p.dt), ow_fix, eps_trajb, p)
  return dtheta, err, err_ext


def triband(LAM, n, d):
  """Diagonal blocks (B,n,d,d) and super-diagonal blocks (B,n-1,d,d) of a dense (B,N,N) matrix,
  plus the max |entry| outside the block-tridiagonal band."""
  B = LAM.shape[0]
  Dg = np.stack([LAM[:, i * d:(i + 1) * d, i * d:(i + 1) * d] for i in range(n)], axis=1)
  Up = np.stack([LAM[:, i * d:(i + 1) * d, (i + 1) * d:(i + 2) * d] for i in range(n - 1)], axis=1)
  mask = np.ones(LAM.shape[1:], dtype=bool)
  for i in range(n):
    for j in (i - 1, i, i + 1):
      if 0 <= j < n: mask[i * d:(i + 1) * d, j * d:(j + 1) * d] = False
  off = np.abs(LAM[:, mask]).max() if mask.any() else 0.0
  return Dg, Up, off


# --------------------------------------------------------------------------------------------
# planner-level helpers (diff_gpmp2_planner.py, utils/planner_utils.py)
# --------------------------------------------------------------------------------------------
def straight_line_trajb(start_confb, goal_confb, traj_time, num_steps, dof):
  """utils/planner_utils.py:47-56.  confs (B,1,dof) -> (B,num_steps+1,2*dof)."""
  B = start_confb.shape[0]
  th = np.zeros((B, int(num_steps) + 1, 2 * dof))
  avg_vel = (goal_confb - start_confb) / traj_time * 1.0
  for i in range(int(num_steps) + 1):
    th[:, i, 0:dof] = start_confb[:, 0, 0:dof] * (num_steps - i) * 1.0 / num_steps * 1.0 \
                      + goal_confb[:, 0, 0:dof] * i * 1.0 / num_steps * 1.0
  th[:, :, dof:] = avg_vel
  return th


def planner_forward(th_initb, startb, goalb, sdfb, p, max_iters, tol_delta, covs=None):
  """diff_gpmp2_planner.py:92-174 (static covariances, no plan_time cut-off): per-sample GN loop;
  converged when ||dtheta||_F < tol_delta or j >= max_iters (utils/planner_utils.py:3-16); the last
  dtheta IS applied (:144,151).  Returns th_final, err_init, err_final, err_per_iter, err_ext_per_iter, iters."""
  B = th_initb.shape[0]
  th_out = np.zeros_like(th_initb)
  err_init, err_final, err_hist, err_ext_hist, iters = [], [], [], [], []
  for i in range(B):
    th = th_initb[i:i + 1].copy()
    s, g, sd = startb[i:i + 1], goalb[i:i + 1], sdfb[i:i + 1]
    qc, ow, eps = p.static_covs(1) if covs is None else [c[i:i + 1] for c in covs]
    j, eh, eeh = 0, [], []
    while True:
      dth, e_old, ee_old = plan_layer_forward(th, s, g, sd, qc, ow, eps, p)
      eh.append(float(e_old.item())); eeh.append(float(ee_old.item()))
      th = th + dth
      j += 1
      if np.linalg.norm(dth) < tol_delta or j >= max_iters:
        break
    th_out[i] = th[0]
    Q_inv = calc_Q_inv_batch(qc, p.dt)
    err_init.append(eh[0]); err_final.append(float(error_batch(th, s, g, sd, Q_inv, ow, eps, p).item()))
    err_hist.append(eh); err_ext_hist.append(eeh); iters.append(j)
  return th_out, err_init, err_final, err_hist, err_ext_hist, iters


def unweighted_errors_batch(thb, startb, goalb, sdfb, epsb, p):
  """diff_gpmp2_planner.py:229-237 -> plan_layer.py:374-388: mean-over-states unweighted L2 errors."""
  B, n, d = thb.shape
  e_p, _ = prior_error(startb.reshape(B, d), thb[:, 0])
  e_g, _ = prior_error(goalb.reshape(B, d), thb[:, n - 1])
  # torch.mean(..., dim=1) over a (B,1,1) tensor: the values unchanged, shape (B,1) (plan_layer.py:387)
  err_sg = (0.5 * np.sum(e_p * e_p, axis=1) + 0.5 * np.sum(e_g * e_g, axis=1)).reshape(B, 1)
  e_gp, _, _ = gp_factor_error(thb, p.dof, p.dt)
  err_gp = np.mean(0.5 * np.sum(e_gp[..., 0] ** 2, axis=-1), axis=1).reshape(B, 1, 1)
  e_o, _ = obstacle_error(thb, sdfb, epsb, p)
  err_obs = np.mean(0.5 * np.sum(e_o[..., 0] ** 2, axis=-1), axis=1).reshape(B, 1, 1)
  return err_sg, err_gp, err_obs


# --------------------------------------------------------------------------------------------
# synthetic inputs -- shared by the fixture generator, tests and bench
# --------------------------------------------------------------------------------------------
def circles_sdf(G, circles, x_lims=(-5.0, 5.0), y_lims=(-5.0, 5.0)):
  """Analytic union-of-circles SDF on a GxG grid, row 0 = y_max, col 0 = x_min (linspace endpoints)."""
  xs = np.linspace(x_lims[0], x_lims[1], G)
  ys = np.linspace(y_lims[1], y_lims[0], G)
  X, Y = np.meshgrid(xs, ys)
  sdf = np.full((G, G), np.inf)
  for (cx, cy, r) in circles:
    sdf = np.minimum(sdf, np.sqrt((X - cx) ** 2 + (Y - cy) ** 2) - r)
  return sdf


C2_CIRCLES = ((-2.0, -1.0, 1.0), (1.5, 2.0, 0.8), (0.0, 0.0, 0.7))
