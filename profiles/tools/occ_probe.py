import sys, json, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/profiles/tools')
from shape_sweep import time_step
for (n, shape) in ((16, '16,1'), (32, '16,2'), (64, '16,4')):
  for B in (2048, 4096, 8192, 16384):
    r = time_step(B, n, 256, torch.float32, 50, shape)
    print(json.dumps(dict(n=n, shape=shape, B=B, us=r['kernel_us'], us_per_4096=round(r['kernel_us'] * 4096 / B, 2))), flush=True)
