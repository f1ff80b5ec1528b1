import ctypes, json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'profiles', 'tools'))
from bench import make_inputs
from dgpmp2_amd import _capi
from dgpmp2_amd.gpmp2.plan_layer import solver_config
from microbench import timed
B, n, G = 4096, 64, 256
dev = torch.device('cuda:0')
th0, start, goal, sdf = make_inputs(B, n, G, dev)
for name, kw in (('c2', {}), ('c3_vel_limits', dict(use_vel_limits=True, K_v=0.01, v_x=1.0, v_y=1.0))):
  s = _capi.Solver(solver_config(n, 2, torch.float32, **kw))
  sa = s.sdf_arg(sdf.data_ptr(), G, G, 0)
  st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
  dth = torch.empty_like(th0); err = torch.empty(B, device=dev); eex = torch.empty(B, device=dev)
  us = timed(lambda: s.gn_step(B, th0.data_ptr(), start.data_ptr(), goal.data_ptr(), sa, None, dth.data_ptr(), err.data_ptr(), eex.data_ptr(), None, st), 1000)
  print(json.dumps({'config': name, 'gn_step_us': round(us, 2)}))
