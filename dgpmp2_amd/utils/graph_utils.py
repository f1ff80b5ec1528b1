"""Whole-iteration HIP-graph capture for host-bound loops around the planner (no counterpart in the reference).

The reference's outer loop (learning/train_planner.py:297-403) calls planner.step(), the error helpers, a loss and backward() eagerly, once per iteration.  On this
build the solver's share of such an iteration is 30-50 us of GPU time, while Python, the autograd engine and ~20 launch enqueues cost 85-215 us of host time: the loop is
host-bound.  Every entry point of the planner only enqueues launches on the current stream, so the iteration can be recorded ONCE as a HIP graph (forward launches,
the backward launches autograd issues, optionally the optimiser step) and replayed with one host call.  GraphedIteration does the bookkeeping torch asks for
(warm-up on a side stream, static input tensors, capture, re-capture when shapes change), so that the loop keeps its eager shape:

    def iteration(th, start, goal, im, sdf, th_opt):                  # a function of its tensor arguments (+ module parameters) only
      out, (e_sg, e_gp, e_obs) = planner.step_with_errors(th, start, goal, im, sdf, conv_out, dtheta)
      loss = criterion(out[0], th_opt - th) + e_gp.mean() + e_sg.mean() + e_obs.mean()
      grads = torch.autograd.grad(loss, params)                       # or loss.backward() into long-lived .grad tensors
      return (out[0], loss) + grads

    it = planner.graphed_iteration(iteration)
    for batch in loader:
      dtheta, loss, *grads = it(th, batch['start'], batch['goal'], batch['im'], batch['sdf'], batch['th_opt'])   # copies the inputs in, replays, returns the outputs

What must hold (torch's capture rules): no host synchronisation inside `iteration` (.item(), .cpu(), print of a tensor, planner.forward()'s per-sample lists), no
data-dependent Python control flow, every tensor the function reads is either an argument or a long-lived tensor updated in place (parameters, optimiser state).
The returned tensors are the graph's own output buffers: the next call overwrites them (clone what must survive; `clone_outputs=True` does it for you).
"""
import torch


def _is_tensor(a):
  return isinstance(a, torch.Tensor)


class GraphedIteration(object):
  """fn(*args) -> tensor or tuple of tensors, captured per signature (shapes / dtypes / devices / requires_grad of the tensor arguments, values of the others) in a HIP
  graph at the first call with that signature and replayed afterwards.  `warmup` eager runs precede the capture (allocator and lazy-initialisation warm-up, as
  torch.cuda.graphs asks); they and the capture itself consume the first call's inputs like ordinary calls (so an optimiser step inside `fn` is applied
  warmup + 1 times at the first call: pass warmup=0 and warm up yourself if that matters)."""

  def __init__(self, fn, warmup=3, clone_outputs=False, pool=None):
    self.fn, self.warmup, self.clone_outputs, self.pool = fn, int(warmup), bool(clone_outputs), pool
    self._entries = {}
    self.captures = 0

  @staticmethod
  def _signature(args):
    # (cheap on purpose -- this runs on every call: torch.Size and dtypes hash fast; the device and the tensor class are fixed per call site)
    return tuple([(a.shape, a.dtype, a.requires_grad) if isinstance(a, torch.Tensor) else ('value', a) for a in args])

  def _capture(self, sig, args):
    dev = next((a.device for a in args if _is_tensor(a) and a.is_cuda), None)
    if dev is None: raise RuntimeError('GraphedIteration: no CUDA/ROCm tensor among the arguments')
    statics = []
    for a in args:
      if _is_tensor(a):
        s = a.detach().clone()
        if a.requires_grad: s.requires_grad_(True)          # a fresh leaf per signature: gradients w.r.t. it are outputs of `fn` (or its .grad, see below)
        statics.append(s)
      else:
        statics.append(a)
    with torch.cuda.device(dev):
      side = torch.cuda.Stream(dev)
      side.wait_stream(torch.cuda.current_stream(dev))
      with torch.cuda.stream(side):
        for _ in range(self.warmup): self.fn(*statics)
      torch.cuda.current_stream(dev).wait_stream(side)
      graph = torch.cuda.CUDAGraph()
      with torch.cuda.graph(graph, pool=self.pool):
        out = self.fn(*statics)
    single = _is_tensor(out)
    outs = (out,) if single else tuple(out)
    self.captures += 1
    e = self._entries[sig] = (statics, graph, outs, single)
    return e

  def __call__(self, *args):
    sig = self._signature(args)
    e = self._entries.get(sig)
    if e is None:
      e = self._capture(sig, args)      # (the capture ran `fn` on copies of these very inputs, but a capture does not execute: replay below)
    statics, graph, outs, single = e
    dst, src = [], []
    for s, a in zip(statics, args):
      if isinstance(a, torch.Tensor) and s is not a: dst.append(s); src.append(a)
    if dst:
      with torch.no_grad():
        try: torch._foreach_copy_(dst, src)      # ONE multi-tensor launch for all inputs (four separate copy_ calls cost more host time than the replay)
        except (RuntimeError, AttributeError, TypeError):
          for s, a in zip(dst, src): s.copy_(a, non_blocking=True)
    graph.replay()
    if self.clone_outputs: outs = tuple(o.clone() if _is_tensor(o) else o for o in outs)
    return outs[0] if single else outs

  def static_inputs(self, *args):
    """The graph's own input tensors for this signature (after the first call): write into them in place and call replay() to skip the input copies."""
    return self._entries[self._signature(args)][0]

  def replay(self, *args):
    """Replay the graph captured for the signature of `args` WITHOUT copying inputs (the caller wrote into static_inputs(...) itself) -> the output buffers."""
    statics, graph, outs, single = self._entries[self._signature(args)]
    graph.replay()
    return outs[0] if single else outs
