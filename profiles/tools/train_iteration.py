"""The training iteration with the gradient set the reference's loop asks for (bench.py: train_iteration_sdf_grad) on its own:
  python profiles/tools/train_iteration.py            -> one JSON line
  rocprofv3 --kernel-trace --stats -- python profiles/tools/train_iteration.py --profile per_sample   (names the kernels / memsets of ONE regime:
      30 eager iterations at B = 4096 with the per-sample or shared grid gradient; --profile replay / replay_static: 2000 HIP-graph replays of the iteration
      without the grid gradient -- the kernels back to back, the regime of the quoted replay figures)"""
import argparse, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench

ap = argparse.ArgumentParser()
ap.add_argument('--profile', default='')
ap.add_argument('--B', type=int, default=bench.B_PER_GPU)
a = ap.parse_args()
dev = torch.device('cuda:0')
if not a.profile:
  print(json.dumps(bench.train_iteration_sdf_grad(dev)))
  sys.exit(0)
B, n, G = a.B, bench.N_STATES, bench.GRID
planner = bench._bench_planner(B, n)
pl = planner.plan_layer
th0, start, goal, sdf = bench.make_inputs(B, n, G, dev)
g = torch.randn_like(th0); cw = torch.randn(B, 1, 1, device=dev); cws = cw.view(B, 1).contiguous()
thr = th0.clone().requires_grad_(True)
qc = torch.eye(2, device=dev).expand(B, n - 1, 2, 2).contiguous().requires_grad_(True)
ow = torch.full((B, n, 1, 1), 1e4, device=dev, requires_grad=True)
ep = torch.full((B, n, 1, 1), 0.4, device=dev, requires_grad=True)
if a.profile.startswith('replay'):
  # the iteration WITHOUT the grid gradient (learned per-state covariances: the figure quoted as "replayed training iteration"), captured once in a HIP graph and
  # replayed 2000 times: the kernels back to back, as the replay figures of bench.py / graph_replay_breakdown.py time them ('replay_static': static covariances)
  sdfb = sdf.expand(B, 1, G, G)
  static = a.profile == 'replay_static'
  leaves = (thr,) if static else (thr, qc, ow, ep)
  def it():
    dth, _, _, sg, gp_, ob = pl.forward_with_errors(thr, start, goal, None, sdfb, *((None, None, None) if static else (qc, ow, ep)))
    return torch.autograd.grad((dth, sg, gp_, ob), leaves, (g, cws, cw, cw))
  side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(side):
    for _ in range(3): it()
  torch.cuda.current_stream().wait_stream(side)
  gr = torch.cuda.CUDAGraph()
  with torch.cuda.graph(gr): it()
  for _ in range(2000): gr.replay()
  torch.cuda.synchronize()
  print('replayed 2000 iterations, B = %d, %s' % (B, a.profile))
  sys.exit(0)
leaf = (bench.make_per_sample_sdfs(B, G, dev, seed=1) if a.profile == 'per_sample' else sdf.clone()).requires_grad_(True)
sdfb = leaf if leaf.shape[0] == B else leaf.expand(B, 1, G, G)
for _ in range(30):
  dth, _, _, sg, gp_, ob = pl.forward_with_errors(thr, start, goal, None, sdfb, qc, ow, ep)
  torch.autograd.grad((dth, sg, gp_, ob), (thr, qc, ow, ep, leaf), (g, cws, cw, cw))
torch.cuda.synchronize()
print('profiled 30 iterations, B = %d, %s grid gradient' % (B, a.profile))
