#!/usr/bin/env python
"""Where does a slow 20-launch region lose its time?  Reads a rocprofv3 --kernel-trace --hip-trace (csv) of
`DGP_BENCH_REGIONS=R DGP_BENCH_DUMP_REGIONS=1 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline`
and reports, for the groups of exactly K consecutive gn_kernel launches that are bracketed by idle time (= the timed regions):
the device-side span of each group, the largest gap between two consecutive kernels inside it, the delay from the first
hipLaunchKernel call of the group (host) to the first kernel's begin (device), and the HIP API calls that were in flight on the
host during the largest gap.

  python profiles/tools/region_gaps.py <dir with *_kernel_trace.csv and *_hip_api_trace.csv> [K]
"""
import csv
import glob
import os
import sys


def rows(path):
  with open(path) as f:
    for r in csv.DictReader(f):
      yield r


def main():
  d = sys.argv[1]
  K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
  kf = sorted(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True))
  hf = sorted(glob.glob(os.path.join(d, '**', '*hip_api_trace.csv'), recursive=True))
  if not kf: raise SystemExit('no kernel trace under %s' % d)
  ks = []
  for r in rows(kf[0]):
    ks.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
  ks.sort()
  api = []
  if hf:
    for r in rows(hf[0]):
      api.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Function']))
    api.sort()
  # groups of consecutive gn_kernel launches separated by >= 15 us of device idle time
  groups, cur = [], []
  for s, e, nm in ks:
    if 'gn_kernel' not in nm:
      if cur: groups.append(cur); cur = []
      continue
    if cur and s - cur[-1][1] > 15000:
      groups.append(cur); cur = []
    cur.append((s, e))
  if cur: groups.append(cur)
  reg = [g for g in groups if len(g) == K]
  print('%d kernel records, %d groups, %d of exactly %d launches' % (len(ks), len(groups), len(reg), K))
  if not reg: return
  import bisect
  starts = [a[0] for a in api]
  out = []
  for g in reg:
    span = (g[-1][1] - g[0][0]) / 1e3
    gaps = [(g[i + 1][0] - g[i][1]) / 1e3 for i in range(K - 1)]
    durs = [(e - s) / 1e3 for s, e in g]
    im = max(range(K - 1), key=lambda i: gaps[i])
    # host: the launch calls of this group are the K hipLaunchKernel-like calls that END before the kernels begin ... take the calls
    # in flight during the largest gap
    inflight = []
    if api:
      lo, hi = g[im][1], g[im + 1][0]
      j = bisect.bisect_left(starts, lo - 200000)
      while j < len(api) and api[j][0] < hi:
        if api[j][1] > lo: inflight.append('%s(%.1fus)' % (api[j][2], (api[j][1] - api[j][0]) / 1e3))
        j += 1
    out.append((span, max(gaps), im, sum(durs) / K, max(durs), inflight))
  spans = sorted(o[0] for o in out)
  med = spans[len(spans) // 2]
  print('device span of a region: min %.1f  median %.1f  p90 %.1f  max %.1f us' % (spans[0], med, spans[int(0.9 * len(spans))], spans[-1]))
  print('kernel duration inside regions: mean %.2f us' % (sum(o[3] for o in out) / len(out)))
  slow = [o for o in out if o[0] > 1.12 * med]
  print('%d of %d regions are > 12 %% above the median span' % (len(slow), len(out)))
  for o in sorted(slow, key=lambda o: -o[0])[:12]:
    print('  span %.1f us: largest gap %.1f us after launch %d, mean kernel %.2f us, longest kernel %.2f us; host calls in flight: %s'
          % (o[0], o[1], o[2], o[3], o[4], ', '.join(o[5][:6]) or '-'))
  fast = [o for o in out if o[0] <= 1.12 * med]
  if fast:
    print('typical region: largest gap %.2f us, mean kernel %.2f us, longest kernel %.2f us' %
          (sum(o[1] for o in fast) / len(fast), sum(o[3] for o in fast) / len(fast), sum(o[4] for o in fast) / len(fast)))


if __name__ == '__main__':
  main()
