#!/usr/bin/env python
"""Build-time guard against the spill level at which hipcc has miscompiled these kernels (six faults in round 2, every
one among the heaviest spillers, none among the spill-free kernels).

A kernel is in the DANGER ZONE when its gfx950 assembly shows >= 300 spilled VGPRs, >= 150 spilled SGPRs or >= 1 KB of scratch per lane --
the level of the six known faults (<3,16,4,double,STEP,general>: 355 VGPRs, 1.4 KB; the fused-loop kernels that "spill hundreds of SGPR lane
masks into VGPR lanes").  Such a kernel may only ship once a GPU run of tests/test_hip_every_kernel.py (every forward instantiation against
the C oracle, every backward one against the autograd oracle and the emulator) has been green ON THAT BUILD; the kernels of that build and
their spill counts are recorded in dgpmp2_amd/csrc/spill_baseline.json.  check() lists every danger-zone kernel that is not in the baseline
or whose spill counts have GROWN by more than 10 % since -- i.e. whose code generation has not been verified at that spill level.
__graft_entry__.build() prints the list loudly and stores it in kernel_stats.json under "_spill_guard"; tests/test_capi_load.py fails on it.

  python profiles/tools/spill_guard.py            -> check dgpmp2_amd/lib/kernel_stats.json against the baseline
  python profiles/tools/spill_guard.py --update   -> rewrite the baseline from the current build (ONLY after a green GPU every-kernel run)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
STATS = os.path.join(ROOT, 'dgpmp2_amd', 'lib', 'kernel_stats.json')
BASELINE = os.path.join(ROOT, 'dgpmp2_amd', 'csrc', 'spill_baseline.json')
VGPR_SPILL, SGPR_SPILL, SCRATCH = 300, 150, 1024
KEYS = ('scratch_bytes_per_lane', 'vgpr_spill', 'sgpr_spill')


def danger(v):
  return v.get('vgpr_spill', 0) >= VGPR_SPILL or v.get('sgpr_spill', 0) >= SGPR_SPILL or v.get('scratch_bytes_per_lane', 0) >= SCRATCH


def check(stats, baseline):
  """-> list of (kernel, current counts, baseline counts or None) for danger-zone kernels not covered by the baseline."""
  out = []
  for k, v in sorted(stats.items()):
    if k.startswith('_') or not isinstance(v, dict) or not danger(v): continue
    cur = [int(v.get(key, 0)) for key in KEYS]
    base = baseline.get(k)
    if base is None or any(c > 1.10 * b + 8 for c, b in zip(cur, base)):
      out.append((k, cur, base))
  return out


def load_baseline():
  try:
    return json.load(open(BASELINE)).get('kernels', {})
  except (OSError, ValueError):
    return {}


def main():
  stats = json.load(open(STATS))
  if '--update' in sys.argv:
    ks = {k: [int(v.get(key, 0)) for key in KEYS] for k, v in sorted(stats.items()) if not k.startswith('_') and isinstance(v, dict) and danger(v)}
    note = sys.argv[sys.argv.index('--update') + 1] if len(sys.argv) > sys.argv.index('--update') + 1 else ''
    json.dump({'note': 'danger-zone kernels (>= %d spilled VGPRs, >= %d spilled SGPRs or >= %d B scratch per lane) of a build whose GPU every-kernel tests were green; '
                       '[scratch bytes per lane, spilled VGPRs, spilled SGPRs].  %s' % (VGPR_SPILL, SGPR_SPILL, SCRATCH, note), 'kernels': ks},
              open(BASELINE, 'w'), indent=1, sort_keys=True)
    print('baseline rewritten: %d danger-zone kernels' % len(ks))
    return 0
  bad = check(stats, load_baseline())
  for k, cur, base in bad:
    print('UNVERIFIED SPILL LEVEL  %-44s scratch/vgpr/sgpr spills %s   baseline %s' % (k, cur, base))
  print('%d kernels, %d in the danger zone, %d not covered by the baseline' % (
      sum(1 for k in stats if not k.startswith('_')), sum(1 for k, v in stats.items() if not k.startswith('_') and isinstance(v, dict) and danger(v)), len(bad)))
  return 1 if bad else 0


if __name__ == '__main__':
  sys.exit(main())
