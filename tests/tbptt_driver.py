"""Truncated back-propagation through a chain of Gauss-Newton steps, driven through the PLANNER API only
(DiffGPMP2Planner.step / unweighted_errors_batch) -- the way the reference's outer learning loop consumes the solver.

What the numbers are judged against: tests/golden/g7_tbptt.npz, produced in the build container by running the REFERENCE's own
training-loop text (learning/train_planner.py, read from /root/reference and exec'd at generation time by
tests/golden/make_golden.py::g7_tbptt -- nothing of it is stored in this repository) on the reference's DiffGPMP2Planner.  This
module is this build's own statement of the same procedure, used by tests/test_planner_api.py on the GPU:

  * a chain of links  x_k --step()--> y_k = x_k + dtheta_k,  x_{k+1} = y_k.detach()  (a fresh leaf per link);
  * every link contributes a loss (imitation of an expert update + the planner's unweighted factor errors at y_k);
  * every `every` links the running loss is averaged and back-propagated: first through the newest link, then the gradient that
    arrived at a link's input leaf is pushed into the output of the link before it, for at most `lookback` links.  Two properties
    of the procedure the fixtures pin down: the running loss is NOT cleared after a flush (it keeps accumulating, and is divided
    again at the next flush), and of a recurrent model's state (h, c) only h is chained backwards.

Test infrastructure: pure torch; imports neither the reference nor dgpmp2_amd.
"""
import collections

import torch
import torch.nn as nn

LossTerms = collections.namedtuple('LossTerms', 'total pos vel gp sg obs ext')


class ConvStub(nn.Module):
  """Stands in for the reference's CNN feature extractor (out of this build's scope): image stack -> feature vector, no parameters."""

  def forward(self, im):
    return im.mean(dim=(2, 3)), None

  def print_gradients(self):
    pass


class FcnStub(nn.Module):
  """Stands in for the reference's fully connected covariance predictor (its first nn.Linear is sized with a Python-2 integer
  division and cannot be constructed under Python 3): (th, conv_out) -> (B, 1, out_dim), smooth in the trajectory and in the image
  features, one learnable vector."""

  def __init__(self, out_dim, lo=0.6, hi=1.4):
    super(FcnStub, self).__init__()
    self.w = nn.Parameter(torch.linspace(lo, hi, out_dim, dtype=torch.float64))

  def forward(self, th, conv_out):
    s = 1.0 + 0.02 * torch.tanh(th).mean(dim=(1, 2), keepdim=True) + 0.05 * conv_out.mean(dim=1).view(-1, 1, 1)
    return self.w.view(1, 1, -1) * s

  def print_gradients(self):
    pass


class RecurrentFcnStub(nn.Module):
  """A recurrent predictor with the call shape the reference's planner expects of one (diff_gpmp2_planner.py:192):
  (th, conv_out, (h, c)) -> (out (B, 1, out_dim), (h', c')); init_hidden(B) -> (h0, c0).  Two learnable tensors, so that the
  gradient reaching `a` depends on the hidden state having been chained across steps."""

  def __init__(self, out_dim, hidden_dim=5):
    super(RecurrentFcnStub, self).__init__()
    self.hidden_dim = hidden_dim
    self.w = nn.Parameter(torch.linspace(0.7, 1.3, out_dim, dtype=torch.float64))
    self.a = nn.Parameter(torch.linspace(-0.4, 0.5, hidden_dim, dtype=torch.float64))

  def init_hidden(self, batch_size):
    z = torch.zeros(batch_size, self.hidden_dim, dtype=torch.float64, device=self.w.device)
    return (z + 0.1, z - 0.05)

  def forward(self, th, conv_out, hidden):
    h, c = hidden
    feat = torch.tanh(th).mean(dim=(1, 2)).view(-1, 1) + 0.3 * conv_out.mean(dim=1).view(-1, 1)
    c_new = 0.8 * c + 0.2 * torch.tanh(self.a.view(1, -1) * feat + 0.5 * h)
    h_new = torch.tanh(c_new) * 0.9
    s = 1.0 + 0.05 * h_new.mean(dim=1).view(-1, 1, 1)
    return self.w.view(1, 1, -1) * s, (h_new, c_new)

  def print_gradients(self):
    pass


def imitation_and_factor_loss(update, expert_update, e_sg, e_gp, e_obs, weights):
  """Loss of one link: mean squared distance between the GN update and the expert's (positions, and velocities weighted by
  `vel_loss_lambda`) + `ext_loss_weight` x (mean GP error + mean start/goal error + `ext_obs_lambda` x mean obstacle error)."""
  dof = update.shape[-1] // 2
  miss = update - expert_update
  pos = miss[..., :dof].pow(2).sum(-1).mean()
  vel = miss[..., dof:].pow(2).sum(-1).mean()
  gp, sg, obs = e_gp.mean(), e_sg.mean(), e_obs.mean()
  ext = gp + sg + weights['ext_obs_lambda'] * obs
  total = pos + weights['vel_loss_lambda'] * vel + weights['ext_loss_weight'] * ext
  return LossTerms(total, pos, vel, gp, sg, obs, ext)


class _Link(object):
  __slots__ = ('x', 'y', 'hin', 'hout')

  def __init__(self, x, y, hin=None, hout=None):
    self.x, self.y, self.hin, self.hout = x, y, hin, hout


def _leaf(t):
  return t.detach().requires_grad_(True)


def truncated_bptt(planner, batch, th_init, num_links, every, lookback, weights, recurrent=False, loss_fn=imitation_and_factor_loss, fused=False):
  """Run `num_links` GN steps from `th_init` on `batch` = dict(im, sdf, start, goal, th_opt) (sdf a leaf that requires grad), flushing
  gradients every `every` links through at most `lookback` links (see the module docstring); fused: planner.step_with_errors per link.  -> dict with the per-link loss terms, the
  final trajectory, and the input leaf of the last link (its .grad is what the last flush left there)."""
  im, sdf, start, goal, th_opt = batch['im'], batch['sdf'], batch['start'], batch['goal'], batch['th_opt']
  chain = collections.deque([_Link(None, th_init, None, planner.learn_module_fcn.init_hidden(th_init.shape[0]) if recurrent else None)],
                            maxlen=lookback)
  features = None
  if planner.fixed_conv:
    features, _ = planner.learn_module_conv(torch.cat((im, sdf), dim=1))
  update = torch.zeros_like(th_init)
  running = torch.zeros((), dtype=th_init.dtype, device=th_init.device)
  per_link = []
  for k in range(1, num_links + 1):
    x = _leaf(chain[-1].y)
    hin = tuple(_leaf(s) for s in chain[-1].hout) if recurrent else None
    if fused:          # dgpmp2_amd's addition: the step and the errors at x + update as one call (one autograd node)
      out, errors = planner.step_with_errors(x, start, goal, im, sdf, features, update, hin)
    else:
      out = planner.step(x, start, goal, im, sdf, features, update, hin) if recurrent else planner.step(x, start, goal, im, sdf, features, update)
    update, hout = out[0], (out[1] if recurrent else None)
    y = x + update
    chain.append(_Link(x, y, hin, hout))            # (the deque drops the oldest link: nothing older than `lookback` is reachable)
    if not fused: errors = planner.unweighted_errors_batch(y, sdf)
    terms = loss_fn(update, th_opt - x, *errors, weights)
    per_link.append(terms)
    running = running + terms.total
    if k % every == 0:
      running = running / every
      running.backward(retain_graph=True)
      links = list(chain)[::-1]                     # newest first
      for newer, older in zip(links, links[1:]):
        if older.x is None:                         # the entry that only holds the initial trajectory: nothing upstream of it
          break
        if recurrent:
          older.hout[0].backward(newer.hin[0].grad, retain_graph=True)
        older.y.backward(newer.x.grad, retain_graph=True)
  last = chain[-1]
  with torch.no_grad():
    if recurrent:
      fin = planner.step(last.y, start, goal, im, sdf, features, update, last.hout)
    else:
      fin = planner.step(last.y, start, goal, im, sdf, features, update)
  return dict(terms=per_link, th_final=last.y.detach(), last_input_leaf=last.x, err=fin[2].detach(), err_ext=fin[3].detach())
