"""What a custom autograd node costs on this host before it does anything: a trivial Function (one empty_like each way) written in Python
against the same node as a C++ torch::autograd::Function, forward alone and forward + torch.autograd.grad, on a (4096, 64, 4) CUDA tensor.
Answers whether moving PlanLayer's autograd nodes (dgpmp2_amd/gpmp2/plan_layer.py) into a C++ extension would pay: see profiles/r04_autograd_floor.txt.

Build (host compiler only, ~25 s) and run on a GPU box:
  python profiles/tools/autograd_floor.py --build && python profiles/tools/autograd_floor.py
"""
import os, subprocess, sys, sysconfig, time
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '..', '..', 'dgpmp2_amd', 'lib')
SO = os.path.join(OUT, '_autograd_floor' + sysconfig.get_config_var('EXT_SUFFIX'))


def build():
  import torch.utils.cpp_extension as ce
  libdir = [l for l in ce.library_paths() if 'torch' in l][0]
  cmd = ['g++', '-O2', '-std=c++17', '-fPIC', '-shared', os.path.join(HERE, 'autograd_floor.cpp'), '-o', SO, '-DTORCH_EXTENSION_NAME=_autograd_floor',
         '-DTORCH_API_INCLUDE_EXTENSION_H', '-D_GLIBCXX_USE_CXX11_ABI=%d' % int(torch._C._GLIBCXX_USE_CXX11_ABI), '-I' + sysconfig.get_paths()['include']]
  cmd += ['-I' + i for i in ce.include_paths()] + ['-L' + libdir, '-lc10', '-ltorch_cpu', '-ltorch', '-ltorch_python', '-Wl,-rpath,' + libdir]
  subprocess.check_call(cmd)


class PyTriv(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x):
    ctx.save_for_backward(x)
    return torch.empty_like(x)

  @staticmethod
  def backward(ctx, g):
    x, = ctx.saved_tensors
    return torch.empty_like(x)


def bench(f, n=3000):
  for _ in range(300): f()
  torch.cuda.synchronize(); t = time.perf_counter()
  for _ in range(n): f()
  torch.cuda.synchronize()
  return (time.perf_counter() - t) / n * 1e6


if __name__ == '__main__':
  if '--build' in sys.argv:
    build(); sys.exit(0)
  sys.path.insert(0, OUT)
  import _autograd_floor as cc
  x = torch.zeros(4096, 64, 4, device='cuda', requires_grad=True)
  g = torch.ones_like(x)
  for r in range(5):
    print('python node: forward %.1f us, forward + autograd.grad %.1f us | C++ node: forward %.1f us, forward + autograd.grad %.1f us' % (
        bench(lambda: PyTriv.apply(x)), bench(lambda: torch.autograd.grad(PyTriv.apply(x), x, g)),
        bench(lambda: cc.triv(x)), bench(lambda: torch.autograd.grad(cc.triv(x), x, g))))
