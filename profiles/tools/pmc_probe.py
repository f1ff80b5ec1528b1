#!/usr/bin/env python
"""Workload for the rocprofv3 --pmc passes.  Calibration kernels of KNOWN size first (so that FETCH_SIZE / WRITE_SIZE can be
corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes: gfx950 FETCH_SIZE reads 1/2 of a wide coalesced stream, other
widths uncalibrated), then launches of dgp_gn_step on ONE workload:
   gn_step (BASELINE configs[1], shared SDF) | per_sample_sdf (one 256x256 grid per trajectory; 6 grid sets cycled so that the
   touched lines do not fit the 256 MiB Infinity Cache) | per_sample_sdf_tiled (the same as 4 x 4 tiles) | learned_covariances (per-state tensors) | config4_xyh (d = 6, 512x512)
   | config4_perstate (d = 6 with per-state covariance tensors: the learned mode)
usage: python profiles/tools/pmc_probe.py [workload]"""
import ctypes, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import make_inputs, make_per_sample_sdfs, B_PER_GPU, N_STATES, GRID
from dgpmp2_amd import _capi
from dgpmp2_amd.gpmp2.plan_layer import solver_config

workload = sys.argv[1] if len(sys.argv) > 1 else 'gn_step'
dev = torch.device('cuda:0')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
# calibration 1: 256 MiB fp32 copy (reads 256 MiB, writes 256 MiB)
a = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=dev).normal_()
b = torch.empty_like(a)
for _ in range(3): b.copy_(a)
torch.cuda.synchronize()
here = os.path.dirname(os.path.abspath(__file__))
so = '/tmp/libdgp_calib.so'
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', os.path.join(here, 'calib.hip'), '-o', so])
cal = ctypes.CDLL(so)
cal.calib_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]
cal.calib_gather_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p]
# calibration 2: the row access pattern (16 dword loads + 16 dword stores per lane, 64 B lane stride, 256 MiB each way)
for _ in range(3):
  assert cal.calib_launch(a.data_ptr(), b.data_ptr(), a.numel() // 16, st) == 0
torch.cuda.synchronize()
# calibration 3: the tap gather pattern on a 1 GiB array (8 Mi lines of 128 B): one 8-byte load per line, then both halves
big = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev).normal_()
for halves in (1, 2):
  for _ in range(3):
    assert cal.calib_gather_launch(big.data_ptr(), b.data_ptr(), big.numel() // 32, halves, st) == 0
torch.cuda.synchronize()
del big

B, n = B_PER_GPU, N_STATES
dof, G, kw, covs, keep = 2, GRID, {}, None, []
if workload in ('config4_xyh', 'config4_perstate', 'config4_auto_tile'):      # config4_perstate (round 6): the d = 6 learned mode, per-state Q_c^-1 tensors (the Kronecker kernels)
  dof, G, kw = 3, 512, dict(non_holonomic=True, K_d=0.01, epsilon_dist=0.2, reg=0.0)
th0, start, goal, sdf = make_inputs(B, n, G, dev, dof=dof)
s = _capi.Solver(solver_config(n, dof, torch.float32, **kw))
sas = [s.sdf_arg(sdf.data_ptr(), G, G, 0)]
if workload == 'per_sample_sdf':
  grids = [make_per_sample_sdfs(B, G, dev, seed=1 + i) for i in range(6)]
  sas = [s.sdf_arg(t.data_ptr(), G, G, G * G) for t in grids]
if workload == 'per_sample_sdf_tiled':      # the same six grid sets stored as 4 x 4 tiles (DgpSdf::layout = DGP_SDF_TILED4)
  from dgpmp2_amd.utils.sdf_utils import tile_sdf
  grids = [tile_sdf(make_per_sample_sdfs(B, G, dev, seed=1 + i)) for i in range(6)]
  sas = [s.sdf_arg(t.data_ptr(), G, G, G * G, layout=_capi.DGP_SDF_TILED4) for t in grids]
if workload in ('learned_covariances', 'config4_perstate'):
  qc = torch.eye(dof, device=dev).expand(B, n - 1, dof, dof).contiguous(); ow = torch.full((B, n), 1e4, device=dev); ep = torch.full((B, n), 0.2 if dof == 3 else 0.4, device=dev)
  keep = [qc, ow, ep]
  covs = s.covs_arg(_capi.DGP_QC_PERSTATE, qc.data_ptr(), ow.data_ptr(), ep.data_ptr())
dth = torch.empty_like(th0); err = torch.empty(B, device=dev); eex = torch.empty(B, device=dev); info = torch.zeros(B, dtype=torch.int32, device=dev)
ths = [th0]
for k in range(3):
  s.gn_step(B, ths[-1].data_ptr(), start.data_ptr(), goal.data_ptr(), sas[0], covs, dth.data_ptr(), err.data_ptr(), eex.data_ptr(), info.data_ptr(), st)
  ths.append(ths[-1] + dth)
torch.cuda.synchronize()
for k in range(24):
  s.gn_step(B, ths[k % 4].data_ptr(), start.data_ptr(), goal.data_ptr(), sas[k % len(sas)], covs, dth.data_ptr(), err.data_ptr(), eex.data_ptr(), info.data_ptr(), st)
torch.cuda.synchronize()
print('workload', workload, 'shape', s.launch_shape(B))
