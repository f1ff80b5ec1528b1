"""Python-3 restatement of the reference's TBPTT training iteration -- learning/train_planner.py:255-374 (one batch of `train()`)
and `one_step_loss` (:75-120) -- written against the PLANNER API only, so that the very same code drives
  * the reference's DiffGPMP2Planner on the CPU (tests/golden/make_golden.py::g7_tbptt, which stores the gradients), and
  * dgpmp2_amd's DiffGPMP2Planner on the GPU (tests/test_planner_api.py::test_tbptt_outer_loop_runs, which compares with them).
The reference file itself is Python-2 only (print statements, xrange) and cannot be imported; statement order, the buffers and the
chained backward calls follow it line by line (cited below).  Test infrastructure: pure torch, imports neither the reference nor
dgpmp2_amd.

Departures forced by Python 3 / devices, none of which changes a number:
  * `xrange` -> `range`, prints dropped;
  * one_step_loss builds its index tensors on the device of its input (the reference's `torch.tensor(0)` lives on the CPU and only
    works there);
  * the non-`use_inter_loss` branch of the reference references an undefined name (`th_currb`, :344) and cannot run at all; like
    every configuration the reference can execute, this restatement requires learn_params['dgpmp2']['use_inter_loss'] = True.
"""
import torch
import torch.nn as nn


class ConvStub(nn.Module):
  """Stands in for learning/learn_module_conv.py (out of this build's scope): image -> feature vector, no parameters."""

  def forward(self, im):
    return im.mean(dim=(2, 3)), None

  def print_gradients(self):
    pass


class FcnStub(nn.Module):
  """Stands in for learning/learn_module_fcn.py (whose first nn.Linear is sized with a py2 integer division and cannot be built
  under Python 3): (th, conv_out) -> (B, 1, out_dim), smooth in the trajectory and in the image features, one learnable vector."""

  def __init__(self, out_dim, lo=0.6, hi=1.4):
    super(FcnStub, self).__init__()
    self.w = nn.Parameter(torch.linspace(lo, hi, out_dim, dtype=torch.float64))

  def forward(self, th, conv_out):
    s = 1.0 + 0.02 * torch.tanh(th).mean(dim=(1, 2), keepdim=True) + 0.05 * conv_out.mean(dim=1).view(-1, 1, 1)
    return self.w.view(1, 1, -1) * s

  def print_gradients(self):
    pass


def one_step_loss(th_curr, th_opt, qc_inv_trajb, obscov_inv_traj_b, err_sg, err_gp, err_obs, criterion, learn_params, dof, epoch):
  """learning/train_planner.py:75-120."""
  dev = th_curr.device
  ix = lambda k: torch.tensor(k, device=dev)
  th_currx = torch.index_select(th_curr, -1, ix(0))
  th_curry = torch.index_select(th_curr, -1, ix(1))
  th_currpos = torch.cat((th_currx, th_curry), dim=-1)
  th_currvx = torch.index_select(th_curr, -1, ix(2))
  th_currvy = torch.index_select(th_curr, -1, ix(3))
  th_currvel = torch.cat((th_currvx, th_currvy), dim=-1)

  th_optx = torch.index_select(th_opt, -1, ix(0))
  th_opty = torch.index_select(th_opt, -1, ix(1))
  th_optpos = torch.cat((th_optx, th_opty), dim=-1)
  th_optvx = torch.index_select(th_opt, -1, ix(2))
  th_optvy = torch.index_select(th_opt, -1, ix(3))
  th_optvel = torch.cat((th_optvx, th_optvy), dim=-1)

  err_pos = (th_currpos - th_optpos).unsqueeze(-1)
  err_vel = (th_currvel - th_optvel).unsqueeze(-1)
  pos_loss = torch.mean(torch.einsum('bsij,bsjk->bsik', err_pos.transpose(2, 3), err_pos))
  vel_loss = torch.mean(torch.einsum('bsij,bsjk->bsik', err_vel.transpose(2, 3), err_vel))
  vel_lam = learn_params['optim']['vel_loss_lambda']
  expert_loss = pos_loss + vel_lam * vel_loss

  gp_loss = err_gp.mean()
  sg_loss = err_sg.mean()
  obs_loss = err_obs.mean()
  obs_lam = learn_params['optim']['ext_obs_lambda']
  ext_loss = gp_loss + sg_loss + obs_lam * obs_loss
  cov_loss = torch.tensor(0.0)
  total_loss = expert_loss + learn_params['optim']['ext_loss_weight'] * ext_loss
  return total_loss, pos_loss, vel_loss, cov_loss, gp_loss, sg_loss, obs_loss, ext_loss


def tbptt_batch(planner, sample, learn_params, planner_params, straight_line_trajb, device, optimizer=None, epoch=0):
  """One batch of train() (learning/train_planner.py:258-424), feed-forward model.  `sample`: dict with im, sdf, start, goal,
  th_opt (what the DataLoader yields, :259-263).  Returns the quantities the reference accumulates / leaves behind: the
  per-chunk losses and -- what this function exists for -- the gradients that `final_loss.backward()` and the chained
  `.backward(curr_grad)` calls deposit in the planner's parameters, in sdf_b and in th_init_b."""
  fixed_conv = learn_params['dgpmp2']['fixed_conv']
  dof = planner_params['dof']
  T = learn_params['dgpmp2']['T']                  # :216
  tk = learn_params['dgpmp2']['tk']                # :217
  tk2 = learn_params['dgpmp2']['tk2'] if 'tk2' in learn_params['dgpmp2'] else tk      # :221-222
  retain_graph = tk < tk2                          # :228
  if learn_params['dgpmp2']['use_inter_loss']: retain_graph = True                    # :229
  assert learn_params['dgpmp2']['use_inter_loss'], 'the reference cannot run the other branch (NameError at :344)'
  criterion = None

  im_b = sample['im'].to(device)                   # :259-263
  sdf_b = sample['sdf'].to(device)
  start_b = sample['start'].to(device)
  goal_b = sample['goal'].to(device)
  th_opt_b = sample['th_opt'].to(device)
  start_conf_b = start_b[:, :, 0:dof]
  goal_conf_b = goal_b[:, :, 0:dof]
  th_init_b = straight_line_trajb(start_conf_b, goal_conf_b, planner_params['total_time_sec'], planner_params['total_time_step'], dof, device)   # :266
  sdf_b.requires_grad_(True)                       # :267
  th_init_b.requires_grad_(True)                   # :268
  dthetab = torch.zeros_like(th_init_b)
  conv_out = None
  if fixed_conv:                                   # :272-274
    data = torch.cat((im_b, sdf_b), dim=1)
    conv_out, _ = planner.learn_module_conv(data)
  if optimizer is not None: optimizer.zero_grad()  # :276

  t = 0
  th_curr_b = th_init_b
  th_curr_buff = [(None, th_init_b)]               # :280
  final_loss = torch.tensor(0.0, device=device)    # :285
  log = {'final_loss': [], 'ext_loss': [], 'obs_loss': [], 'gp_loss': [], 'sg_loss': [], 'pos_loss': []}
  batch_total_loss = 0.0
  while t < T:                                     # :297
    th_curr_b = th_curr_buff[-1][1].detach()       # :299
    th_curr_b.requires_grad = True                 # :301
    dthetab, _, _, _, qc_inv_trajb, obscov_inv_traj_b, eps_traj_b = planner.step(th_curr_b, start_b, goal_b, im_b, sdf_b, conv_out, dthetab)   # :311
    th_new_b = th_curr_b + dthetab                 # :313
    th_curr_buff.append((th_curr_b, th_new_b))     # :314
    while len(th_curr_buff) > tk2:                 # :319-322
      del th_curr_buff[0]
    if learn_params['dgpmp2']['use_inter_loss']:   # :325-338
      err_sg, err_gp, err_obs = planner.unweighted_errors_batch(th_new_b, sdf_b)
      curr_total_loss, curr_pos_loss, curr_vel_loss, curr_cov_loss, curr_gp_loss, curr_sg_loss, curr_obs_loss, curr_ext_loss = one_step_loss(
          dthetab, th_opt_b - th_curr_b, qc_inv_trajb, obscov_inv_traj_b, err_sg, err_gp, err_obs, criterion, learn_params, dof, epoch)
      final_loss = final_loss + curr_total_loss
      batch_total_loss += curr_total_loss.item()
      log['ext_loss'].append(curr_ext_loss.item()); log['obs_loss'].append(curr_obs_loss.item())
      log['gp_loss'].append(curr_gp_loss.item()); log['sg_loss'].append(curr_sg_loss.item()); log['pos_loss'].append(curr_pos_loss.item())
    if (t + 1) % tk == 0:                          # :340
      final_loss = final_loss / tk * 1.0           # :355
      batch_total_loss = batch_total_loss / tk * 1.0
      final_loss.backward(retain_graph=retain_graph)                   # :366
      for j in range(tk2 - 1):                     # :367-374
        if th_curr_buff[-j - 2][0] is None:
          break
        curr_grad = th_curr_buff[-j - 1][0].grad
        th_curr_buff[-j - 2][1].backward(curr_grad, retain_graph=retain_graph)
      log['final_loss'].append(final_loss.item())
      if learn_params['dgpmp2']['optimize_tk'] and optimizer is not None:      # :397-403
        optimizer.step()
    t = t + 1
  with torch.no_grad():                            # :406-410
    _, _, errb, err_extb, _, _, _ = planner.step(th_new_b, start_b, goal_b, im_b, sdf_b, conv_out, dthetab)
  grads = {name: (None if p.grad is None else p.grad.detach().clone()) for name, p in planner.named_parameters()}
  return dict(log=log, batch_total_loss=batch_total_loss, param_grads=grads, sdf_grad=None if sdf_b.grad is None else sdf_b.grad.detach().clone(),
              th_init_grad=None if th_init_b.grad is None else th_init_b.grad.detach().clone(), th_final=th_new_b.detach().clone(),
              err=errb.detach().clone(), err_ext=err_extb.detach().clone(), th_curr_grad_last=None if th_curr_b.grad is None else th_curr_b.grad.detach().clone())
