#!/usr/bin/env python
"""BASELINE config 1 plumbing on the HIP solver: 2-D point robot, one environment, one trajectory, GN to convergence, then
a backward pass through the whole plan -- the counterpart of the reference's examples/diff_gpmp2_2d_example.py:25-77,
with `dgpmp2_amd` swapped in for `diff_gpmp2`.

  python examples/diff_gpmp2_2d_example.py [--map tests/golden/g3_c1.npz] [--states 32] [--iters 50]

The default map is the signed distance field of the reference's env/simple_2d/5.png as stored in the golden fixture (the PNG
itself belongs to the reference and is not shipped); pass --circles to use an analytic three-circle map instead.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from dgpmp2_amd.robot_models import PointRobot2D                      # noqa: E402
from dgpmp2_amd.gpmp2.diff_gpmp2_planner import DiffGPMP2Planner      # noqa: E402
from dgpmp2_amd.utils.planner_utils import straight_line_traj         # noqa: E402
from dgpmp2_amd.utils.sdf_utils import circles_sdf, C2_CIRCLES        # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--map', default=os.path.join(ROOT, 'tests', 'golden', 'g3_c1.npz'))
  ap.add_argument('--circles', action='store_true')
  ap.add_argument('--states', type=int, default=32)
  ap.add_argument('--iters', type=int, default=50)
  args = ap.parse_args()
  torch.set_default_dtype(torch.float64)            # as the reference's example does (:26)
  device = torch.device('cuda')

  # parameters of examples/configs/{gpmp2_2d_params,robot_2d,env_2d_params}.yaml
  env_params = {'x_lims': [-5.0, 5.0], 'y_lims': [-5.0, 5.0]}
  planner_params = {'dof': 2, 'state_dim': 4, 'total_time_sec': 10.0, 'total_time_step': args.states - 1}
  gp_params = {'Q_c_inv': torch.eye(2), 'K_s': torch.tensor(0.01), 'K_g': torch.tensor(0.01)}
  obs_params = {'cost_sigma': torch.tensor(0.01), 'epsilon_dist': torch.tensor(0.4)}
  optim_params = {'method': 'gauss_newton', 'reg': 0.1, 'plan_time': float('inf'), 'max_iters': args.iters, 'tol_err': 1e-3,
                  'tol_delta': 1e-4}
  sdf_np = circles_sdf(256, C2_CIRCLES) if args.circles else np.load(args.map)['sdf']
  sdf = torch.from_numpy(sdf_np).to(device)
  im = (sdf > 0).double()

  robot = PointRobot2D(torch.tensor(0.4), use_cuda=True)
  start_conf = torch.tensor([[env_params['x_lims'][0] + 1.0, env_params['y_lims'][0] + 1.0]], device=device)
  goal_conf = torch.tensor([[env_params['x_lims'][1] - 1.0, env_params['y_lims'][1] - 1.0]], device=device)
  start = torch.cat((start_conf, torch.zeros(1, 2, device=device)), dim=1)
  goal = torch.cat((goal_conf, torch.zeros(1, 2, device=device)), dim=1)
  th_init = straight_line_traj(start_conf, goal_conf, planner_params['total_time_sec'], planner_params['total_time_step'], 2, device)

  planner = DiffGPMP2Planner(gp_params, obs_params, planner_params, optim_params, env_params, robot, use_cuda=True)
  t0 = time.time()
  th_final, _, err_init, err_final, err_per_iter, err_ext_per_iter, k, time_taken = planner.forward(
      th_init.unsqueeze(0), start.unsqueeze(0), goal.unsqueeze(0), im.unsqueeze(0).unsqueeze(0), sdf.unsqueeze(0).unsqueeze(0))
  torch.cuda.synchronize()
  print('Initial cost = %f' % err_init[0])
  print('Final cost = %f' % err_final[0])
  print('Iterations taken = %d' % k[0])
  print('Time taken = %f (seconds, fused GN loop incl. host round trip)' % (time.time() - t0))

  print('Calling .backward() through a 5-iteration plan')
  optim_params['max_iters'] = 5
  th_req = th_init.clone().requires_grad_(True)
  tb = time.time()
  th5 = planner.forward(th_req.unsqueeze(0), start.unsqueeze(0), goal.unsqueeze(0), im.unsqueeze(0).unsqueeze(0),
                        sdf.unsqueeze(0).unsqueeze(0))[0]
  th5.backward(torch.randn(th5.shape, device=device))
  torch.cuda.synchronize()
  print('Forward + backprop time = %f, |grad wrt th_init| = %f' % (time.time() - tb, float(th_req.grad.norm())))


if __name__ == '__main__':
  main()
