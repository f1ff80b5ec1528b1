#!/usr/bin/env python
"""Workload for the rocprofv3 --pmc passes: a calibration copy of KNOWN size (so that FETCH_SIZE / WRITE_SIZE can be
corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes: gfx950 FETCH_SIZE reads 1/2 of a wide coalesced
stream, other widths uncalibrated) followed by launches of the benchmark kernel (BASELINE configs[1])."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import make_inputs, B_PER_GPU, N_STATES, GRID, DOF
from dgpmp2_amd import _capi
from dgpmp2_amd.gpmp2.plan_layer import solver_config

dev = torch.device('cuda:0')
# calibration: 256 MiB fp32 copy (reads 256 MiB, writes 256 MiB; larger than the 256 MiB Infinity Cache in total)
a = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=dev).normal_()
b = torch.empty_like(a)
for _ in range(3): b.copy_(a)
torch.cuda.synchronize()
# calibration in the kernel's own access pattern: 16 dword loads + 16 dword stores per lane, 64 B lane stride, 256 MiB each way
import subprocess
here = os.path.dirname(os.path.abspath(__file__))
so = '/tmp/libdgp_calib.so'
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', os.path.join(here, 'calib.hip'), '-o', so])
cal = ctypes.CDLL(so)
cal.calib_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]
for _ in range(3):
  rc = cal.calib_launch(a.data_ptr(), b.data_ptr(), a.numel() // 16, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
  assert rc == 0
torch.cuda.synchronize()
B, n = B_PER_GPU, N_STATES
th0, start, goal, sdf = make_inputs(B, n, GRID, dev)
s = _capi.Solver(solver_config(n, DOF, torch.float32))
sa = s.sdf_arg(sdf.data_ptr(), GRID, GRID, 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
dth = torch.empty_like(th0); err = torch.empty(B, device=dev); eex = torch.empty(B, device=dev)
for _ in range(20):
  s.gn_step(B, th0.data_ptr(), start.data_ptr(), goal.data_ptr(), sa, None, dth.data_ptr(), err.data_ptr(), eex.data_ptr(), None, st)
torch.cuda.synchronize()
print('shape', s.launch_shape(B))
