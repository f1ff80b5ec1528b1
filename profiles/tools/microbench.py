#!/usr/bin/env python
"""Kernel-level timing (HIP events) of every C-ABI entry point on the benchmark workload: dgp_gn_step, dgp_gn_solve
(10 fused GN iterations), dgp_eval_errors, dgp_gn_step_backward.   usage: python profiles/tools/microbench.py [--reps 100]"""
import argparse, ctypes, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import make_inputs
from dgpmp2_amd import _capi
from dgpmp2_amd.gpmp2.plan_layer import solver_config


def timed(f, reps):
  import time
  t0 = time.time()
  while time.time() - t0 < 0.3:                 # steady clocks: a cold GPU runs the first milliseconds ~12 % slower
    for _ in range(50): f()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize(); e0.record()
  for _ in range(reps): f()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps * 1e3


def run(B, n, G, dtype, reps, percov=False):
  dev = torch.device('cuda:0')
  th0, start, goal, sdf = [t.to(dtype) for t in make_inputs(B, n, G, dev)]
  s = _capi.Solver(solver_config(n, 2, dtype))
  sa = s.sdf_arg(sdf.data_ptr(), G, G, 0)
  st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
  dth = torch.empty_like(th0); err = torch.empty(B, device=dev, dtype=dtype); eex = torch.empty_like(err)
  covs = None; keep = []
  if percov:
    qc = torch.eye(2, device=dev, dtype=dtype).expand(B, n - 1, 2, 2).contiguous(); ow = torch.full((B, n), 1e4, device=dev, dtype=dtype)
    ep = torch.full((B, n), 0.4, device=dev, dtype=dtype); keep = [qc, ow, ep]
    covs = s.covs_arg(_capi.DGP_QC_PERSTATE, qc.data_ptr(), ow.data_ptr(), ep.data_ptr())
  P = lambda t: t.data_ptr()
  out = dict(B=B, n=n, io=str(dtype).split('.')[-1], covs='per-state tensors' if percov else 'static', shape=s.launch_shape(B))
  out['gn_step_us'] = timed(lambda: s.gn_step(B, P(th0), P(start), P(goal), sa, covs, P(dth), P(err), P(eex), None, st), reps)
  tho = torch.empty_like(th0); it = torch.empty(B, dtype=torch.int32, device=dev); eh = torch.empty(B, 10, device=dev, dtype=dtype)
  out['gn_solve_10iters_us'] = timed(lambda: s.gn_solve(B, P(th0), P(start), P(goal), sa, covs, 10, 0.0, P(tho), P(it), P(eh), None, P(err), None, st), max(5, reps // 10))
  out['eval_errors_us'] = timed(lambda: s.eval_errors(B, P(th0), P(start), P(goal), sa, covs, P(err), P(eex), None, None, None, st), reps)
  g = torch.randn_like(th0); gth = torch.empty_like(th0); gst = torch.empty_like(start); ggo = torch.empty_like(goal)
  gs = torch.zeros_like(sdf); ge = torch.ones(B, device=dev, dtype=dtype)
  gq = torch.empty_like(keep[0]) if percov else None; gw = torch.empty(B, n, device=dev, dtype=dtype) if percov else None
  gp = torch.empty(B, n, device=dev, dtype=dtype) if percov else None
  PP = lambda t: None if t is None else t.data_ptr()
  out['gn_step_backward_us'] = timed(lambda: s.gn_step_backward(B, P(th0), P(start), P(goal), sa, covs, P(dth), P(g), P(ge), P(gth), P(gst), P(ggo),
                                                                P(gs), 0, PP(gq), PP(gw), PP(gp), st), reps)
  gs8 = torch.zeros(8, 1, G, G, device=dev, dtype=dtype)
  out['gn_step_backward_xcd_partial_sdf_grad_us'] = timed(lambda: s.gn_step_backward(B, P(th0), P(start), P(goal), sa, covs, P(dth), P(g), P(ge), P(gth), P(gst), P(ggo),
                                                                                     P(gs8), 0, PP(gq), PP(gw), PP(gp), st, g_sdf_copies=8), reps)
  gs16 = torch.zeros(16, 1, G, G, device=dev, dtype=dtype)
  out['gn_step_backward_16_partial_sdf_grads_us'] = timed(lambda: s.gn_step_backward(B, P(th0), P(start), P(goal), sa, covs, P(dth), P(g), P(ge), P(gth), P(gst), P(ggo),
                                                                                      P(gs16), 0, PP(gq), PP(gw), PP(gp), st, g_sdf_copies=16), reps)
  out['gn_step_backward_no_sdf_grad_us'] = timed(lambda: s.gn_step_backward(B, P(th0), P(start), P(goal), sa, covs, P(dth), P(g), P(ge), P(gth), P(gst), P(ggo),
                                                                            None, 0, PP(gq), PP(gw), PP(gp), st), reps)
  return {k: (round(v, 2) if isinstance(v, float) else v) for k, v in out.items()}


if __name__ == '__main__':
  ap = argparse.ArgumentParser(); ap.add_argument('--reps', type=int, default=100); a = ap.parse_args()
  import __graft_entry__; __graft_entry__.build()
  for (B, n, dt, pc) in [(4096, 64, torch.float32, False), (4096, 64, torch.float32, True), (4096, 64, torch.float64, False), (32768, 64, torch.float32, False)]:
    print(json.dumps(run(B, n, 256, dt, a.reps if B <= 4096 else 20, pc)), flush=True)
