"""Callers' helpers of the planner API, restated for torch tensors on any device.
Reference: diff_gpmp2/utils/planner_utils.py (check_convergence :3-16, check_convergence_batch :18-36,
straight_line_traj :38-45, straight_line_trajb :47-56)."""
import torch


def check_convergence(dtheta, j, err_delta, tol_err, tol_delta, max_iters, method='gauss_newton', verbose=False):
  """True when ||dtheta||_F < tol_delta or j >= max_iters (the err_delta / 'error increased' criteria are commented
  out in the reference, planner_utils.py:7-15, and stay inactive here)."""
  nrm = torch.norm(dtheta)
  if nrm < tol_delta:
    if verbose: print('Update got too small at iter %d: %f' % (j, nrm))
    return True
  if j >= max_iters:
    if verbose: print('Max iters done')
    return True
  return False


def check_convergence_batch(dthetab, j, err_delta, tol_err, tol_delta, max_iters, method='gauss_newton', device=None):
  """Per-sample convergence mask (B,1,1).  As in the reference (planner_utils.py:24-27) the second torch.where overwrites the
  first, so only the err_delta criterion survives: int64 ones/zeros from torch.where, or -- once j >= max_iters -- all ones
  as uint8 (the reference's torch.ones(...).byte(), created on the CPU whatever `device` says; here on dthetab's device)."""
  B = dthetab.shape[0]
  dev = dthetab.device if device is None else device
  err_delta_norm = torch.norm(err_delta.reshape(B, -1), dim=1, p=2)
  conv = torch.where(err_delta_norm < tol_err, torch.tensor(1, device=dev), torch.tensor(0, device=dev))
  if j >= max_iters:
    conv = torch.ones(B, 1, 1, dtype=torch.uint8, device=dthetab.device)
  return conv.view(B, 1, 1)


def straight_line_traj(start_conf, goal_conf, traj_time, num_steps, dof, device=None):
  """(1,dof) start/goal configurations -> (num_steps+1, 2*dof) constant-velocity straight line."""
  return straight_line_trajb(start_conf.reshape(1, 1, -1), goal_conf.reshape(1, 1, -1), traj_time, num_steps, dof, device)[0]


def straight_line_trajb(start_confb, goal_confb, traj_time, num_steps, dof, device=None):
  """(B,1,dof) start/goal configurations -> (B, num_steps+1, 2*dof).  Position i is
  start*(num_steps-i)/num_steps + goal*i/num_steps evaluated in the reference's operation order; velocity is the
  average velocity (goal-start)/traj_time at every state."""
  num_steps = int(num_steps)
  B = start_confb.shape[0]
  dev = start_confb.device if device is None else device
  s = start_confb[:, 0, 0:dof].to(dev); g = goal_confb[:, 0, 0:dof].to(dev)
  i = torch.arange(num_steps + 1, device=dev, dtype=s.dtype).view(1, -1, 1)
  pos = s.unsqueeze(1) * (num_steps - i) * 1.0 / num_steps * 1.0 + g.unsqueeze(1) * i * 1.0 / num_steps * 1.0
  vel = ((goal_confb.to(dev) - start_confb.to(dev)) / traj_time * 1.0)[:, :, 0:dof].expand(B, num_steps + 1, dof)
  return torch.cat((pos, vel), dim=-1).contiguous()
