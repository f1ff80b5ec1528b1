"""The training iteration under HIP-graph replay (torch.cuda.CUDAGraph), piece by piece, B = 4096, n = 64, d = 4, fp32 I/O, shared 256 x 256 grid: step alone, step + the
unweighted errors at th + dtheta (PlanLayer.forward_with_errors, no_grad), step + backward, the whole iteration (forward_with_errors + autograd.grad of all four outputs) --
with learned per-state covariance tensors (gradients w.r.t. all four inputs) and with static covariances.  python profiles/tools/graph_replay_breakdown.py on a GPU box;
output kept in profiles/r04_graph_replay.txt."""
import sys, os, time, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np
import test_planner_api as TP
from oracle import gpmp2_oracle as O
dev='cuda:0'
B, n, G = 4096, 64, 256
planner = TP.make_planner(n, B)
f32=torch.float32
sdf = torch.from_numpy(O.circles_sdf(G, O.C2_CIRCLES)).to(f32).to(dev)[None,None]
sdfb = sdf.expand(B,1,G,G)
start = torch.zeros(B,1,4,device=dev,dtype=f32); goal=torch.zeros(B,1,4,device=dev,dtype=f32)
start[:,0,:2]=torch.rand(B,2,device=dev)*8-4; goal[:,0,:2]=torch.rand(B,2,device=dev)*8-4
from dgpmp2_amd.utils.planner_utils import straight_line_trajb
th = straight_line_trajb(start[:,:,:2].double().cpu(), goal[:,:,:2].double().cpu(), 10.0, n-1, 2).to(f32).to(dev)
def mk(static):
  thr = th.clone().requires_grad_(True)
  if static: return thr, None, None, None
  qc = (torch.eye(2,device=dev,dtype=f32).expand(B,n-1,2,2).contiguous()*1.0).requires_grad_(True)
  ow = torch.full((B,n,1,1),1e4,device=dev,dtype=f32).requires_grad_(True)
  ep = torch.full((B,n,1,1),0.4,device=dev,dtype=f32).requires_grad_(True)
  return thr, qc, ow, ep
g = torch.randn(B,n,4,device=dev,dtype=f32); cw=torch.randn(B,1,1,device=dev,dtype=f32); cws=cw.view(B,1).contiguous()
def graphed(f):
  side=torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(side):
    for _ in range(3): f()
  torch.cuda.current_stream().wait_stream(side)
  gr=torch.cuda.CUDAGraph()
  with torch.cuda.graph(gr): f()
  return gr.replay
def wall(f, reps=300):
  for _ in range(30): f()
  best=1e9
  for _ in range(3):
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); best=min(best,(time.perf_counter()-t)/reps*1e6)
  return best
pl = planner.plan_layer
for static in (False, True):
  thr, qc, ow, ep = mk(static)
  leaves = (thr,) if static else (thr,qc,ow,ep)
  def fwd_only():
    with torch.no_grad(): pl.forward_with_errors(thr,start,goal,None,sdfb,qc,ow,ep)
  def step_only():
    with torch.no_grad(): pl.forward(thr,start,goal,None,sdfb,qc,ow,ep)
  def it():
    dth,_,_,sg,gp_,ob = pl.forward_with_errors(thr,start,goal,None,sdfb,qc,ow,ep)
    return torch.autograd.grad((dth,sg,gp_,ob), leaves, (g,cws,cw,cw))
  def it_step():
    dth = pl.forward(thr,start,goal,None,sdfb,qc,ow,ep)[0]
    return torch.autograd.grad(dth, leaves, g)
  print('static' if static else 'learned', 'graph replay us: step %.1f  step+errors %.1f  step+bwd %.1f  iteration %.1f' % (wall(graphed(step_only)), wall(graphed(fwd_only)), wall(graphed(it_step)), wall(graphed(it))))
# round 5: the reference's default learned mode (diag_identity) end to end -- the module output through get_covariances (tagged blocks -> DGP_QC_SCALAR) and handed to the kernels raw (DGP_COVS_SQUARED)
lm_out = torch.cat([torch.ones(B,1,n-1,device=dev,dtype=f32), torch.full((B,1,n),100.0,device=dev,dtype=f32)], dim=2).requires_grad_(True)
thr = th.clone().requires_grad_(True)
def it_di():
  qc_s, ow_s = planner.get_covariances(lm_out, 'diag_identity')
  dth,_,_,sg,gp_,ob = pl.forward_with_errors(thr,start,goal,None,sdfb,qc_s,ow_s,None)
  return torch.autograd.grad((dth,sg,gp_,ob), (thr,lm_out), (g,cws,cw,cw))
def it_raw():
  raw = pl.raw_covs(lm_out, 'diag_identity', False)
  dth,_,_,sg,gp_,ob = pl.forward_raw(thr,start,goal,None,sdfb,raw,with_errors=True)[:6]
  return torch.autograd.grad((dth,sg,gp_,ob), (thr,lm_out), (g,cws,cw,cw))
def fwd_raw():
  with torch.no_grad(): pl.forward_raw(thr,start,goal,None,sdfb,pl.raw_covs(lm_out.detach(), 'diag_identity', False),with_errors=True)
print('diag_identity graph replay us: iteration via get_covariances %.1f  iteration raw module output %.1f  forward raw %.1f   eager: %.1f / %.1f' % (wall(graphed(it_di)), wall(graphed(it_raw)), wall(graphed(fwd_raw)), wall(it_di), wall(it_raw)))
