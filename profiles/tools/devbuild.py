#!/usr/bin/env python
"""Tuning build: compile only SOME kernel translation units (the full library takes ~5.5 min on 8 cores) into
dgpmp2_amd/lib/libdgpmp2_dev.so; the launch entry points of the units left out are stubs that fail with hipErrorInvalidValue.
Use it with DGP_LIB_PATH=dgpmp2_amd/lib/libdgpmp2_dev.so (dgpmp2_amd/_capi.py).  Never the product build.

  python profiles/tools/devbuild.py 2_f32_g0 2_f32_g1 [-D...] [-o name.so]     units: <dof>_<f32|f64>_g<0 static|1 general|2 backward|3 per-state Kronecker|4 chain backward>
Prints the ISA statistics (registers, scratch, instruction counts) of the kernels whose name contains --show (default ',16,4,').
"""
import argparse
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'profiles', 'tools'))
import isa_stats

CSRC = os.path.join(ROOT, 'dgpmp2_amd', 'csrc')
ALL = ['%s_%s_g%d' % (d, t, g) for d in ('2', '3', '2t', '3t', '2e', '3e') for t in ('f32', 'f64') for g in (0, 1, 2, 3, 4)]      # (2t / 3t: the tiled twins, -DDGP_TL=1)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('units', nargs='+')
  ap.add_argument('-D', action='append', default=[])
  ap.add_argument('--flag', action='append', default=[], help='extra hipcc argument, e.g. --flag=-mllvm --flag=-amdgpu-spill-sgpr-to-vgpr=false')
  ap.add_argument('-o', default='libdgpmp2_dev.so')
  ap.add_argument('--show', default=',16,4,')
  ap.add_argument('--raw', action='store_true', help='plain hipcc -c: no repair of the device assembly')
  a = ap.parse_args()
  for u in a.units: assert u in ALL, (u, ALL)
  work = os.path.join('/tmp', 'dgp_dev_' + a.o.replace('.', '_'))
  shutil.rmtree(work, ignore_errors=True); os.makedirs(work)
  hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
  base = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', '-save-temps=obj'] + ['-D' + d for d in a.D] + a.flag
  jobs = []
  for u in a.units:
    dof, t, g = u.split('_')
    d = os.path.join(work, u); os.makedirs(d)
    jobs.append(base + (['-DDGP_TL=1'] if dof.endswith('t') else (['-DDGP_STEP_ERRS=1'] if dof.endswith('e') else [])) + ['-DDGP_INST_DOF=' + dof.rstrip('te'), '-DDGP_INST_F64=%d' % (t == 'f64'), '-DDGP_INST_GROUP=' + g[1], os.path.join(CSRC, 'gn_inst.hip'),
                        '-o', os.path.join(d, u + '.o')])
  d = os.path.join(work, 'abi'); os.makedirs(d)
  jobs.append(base + [os.path.join(CSRC, 'dgpmp2_hip.hip'), '-o', os.path.join(d, 'abi.o')])
  d = os.path.join(work, 'long'); os.makedirs(d)
  jobs.append(base + [os.path.join(CSRC, 'gn_long_inst.hip'), '-o', os.path.join(d, 'long.o')])      # the long-trajectory kernels (small: always built)
  d = os.path.join(work, 'edt'); os.makedirs(d)
  jobs.append(base + [os.path.join(CSRC, 'sdf_edt.hip'), '-o', os.path.join(d, 'edt.o')])       # dgp_sdf_2d (the binding resolves every declared symbol)
  stub = os.path.join(work, 'stubs.hip')
  with open(stub, 'w') as f:
    f.write('#include "%s"\n' % os.path.join(CSRC, 'gn_device.h'))
    for u in ALL:
      if u not in a.units and not (u[1] == 'e' and u[-1] not in ('013' if u[0] == '2' else '03')):      # (the twin units the ABI references: groups 0, 3 and -- d = 4 -- 1)
        f.write('hipError_t dgp_launch_%s(DgpShape, int, const dgp::GnParams&, const dgp::GnGradParams*, hipStream_t) { return hipErrorInvalidValue; }\n' % u)
  d = os.path.join(work, 'stubs'); os.makedirs(d)
  jobs.append([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', stub, '-o', os.path.join(d, 'stubs.o')])
  # every unit through the product's own compile pipeline (__graft_entry__.compile_hip_unit: device assembly, the exec-join repair, assembler, bundler, host object);
  # --raw: plain hipcc -c -save-temps (the unrepaired compiler output, for the reproducer builds of profiles/r06_compiler_fault.md)
  sys.path.insert(0, ROOT)
  import __graft_entry__ as G
  from concurrent.futures import ThreadPoolExecutor
  def one(j):
    if a.raw or j[-3].endswith('stubs.hip'): return subprocess.call(j), None
    flags = [x for x in j[1:-3] if x not in ('--offload-arch=gfx950', '-c', '-save-temps=obj')]
    return 0, G.compile_hip_unit(hipcc, flags, j[-3], j[-1])
  with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex: res = list(ex.map(one, jobs))
  rcs = [r[0] for r in res]
  if any(rcs): raise SystemExit('hipcc failed: %s' % rcs)
  for j, r in zip(jobs, res):
    if r[1] and (r[1][0] or r[1][1]): print('%-28s exec-join repair: %d instruction(s) moved, %d finding(s) left' % (os.path.basename(j[-1]), r[1][0], r[1][1]))
  out = os.path.join(ROOT, 'dgpmp2_amd', 'lib', a.o)
  subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + [j[-1] for j in jobs] + ['-o', out])
  stats = {}
  for u in a.units:
    for fn in os.listdir(os.path.join(work, u)):
      if fn.endswith('gfx950.s'): stats.update(isa_stats.parse(os.path.join(work, u, fn)))
  json.dump(stats, open(out + '.stats.json', 'w'), indent=1, sort_keys=True)
  for k, v in sorted(stats.items()):
    if a.show in k:
      print('%-46s vgpr %3d agpr %3d scratch %4d occ %d valu %5d f64 %5d dpp %4d agprmov %4d lds %4d vmem %3d/%3d' % (
          k, v['vgpr'], v['agpr'], v['scratch_bytes_per_lane'], v['waves_per_simd'], v['valu'], v['fma_f64'] + v['mul_f64'] + v['add_f64'], v['dpp'],
          v['agpr_moves'], v['lds'], v['vmem_load'], v['vmem_store']))
  print('built', out)


if __name__ == '__main__':
  main()
