#!/usr/bin/env python
"""Round 6: an EXACT census of the exec-join spill placement (profiles/r06_compiler_fault.md) at the MIR level, for the translation units of the shipped library.

exec_join_check.py works on the final assembly, where tail duplication and block placement blur which blocks are joins (a body's tail with a duplicated restore looks like a
join top): it can prove the `store` class and must leave `reload`s in front of a restore as "reported".  Right after the VGPR run of the register allocator
(`-mllvm -stop-after=greedy,2`) the blocks are still the structured ones: every `$exec = S_OR_B64 $exec, ...` at the top of a block IS the restore of a join, and every vector
instruction in front of it in that block was put there by the allocator under the wrong mask.  Per unit: hipcc --cuda-device-only -S -mllvm -stop-after=greedy,2 (MIR instead of assembly) -> count, per
kernel, the vector COPYs / SI_SPILL_*_SAVE / SI_SPILL_*_RESTORE in front of an exec restore.

  python profiles/tools/r06_mir_census.py [unit ...]        units as in devbuild.py (2_f32_g1, 3e_f32_g0, 2t_f32_g2, ...; default: all of the product build)  -> one line per unit + totals
  python profiles/tools/r06_mir_census.py --check           all units, then compared with what the build repaired (dgpmp2_amd/lib/kernel_stats.json: _exec_join_repaired): the units
                                                            must be the same and every finding must be a copy into an A(V) register = two moved instructions; exit status 1 otherwise"""
import os, re, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, 'dgpmp2_amd', 'csrc')
LLVM = os.environ.get('LLVM_BIN', '/opt/rocm/lib/llvm/bin')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
ALL = ['%s_%s_g%d' % (d, t, g) for d in ('3', '2', '3t', '2t') for t in ('f32', 'f64') for g in (2, 3, 4, 0, 1)] + \
      ['%s_%s_g%d' % (d, t, g) for d in ('3e', '2e') for t in ('f32', 'f64') for g in (3, 0, 1) if not (d == '3e' and g == 1)] + ['long']
VEC_DEF = re.compile(r'^\s*(?:renamable |dead |early-clobber )*(%\d+(?:\.\w+)?:(?:vreg|av|areg|vgpr|agpr)\w*|\$[av]gpr\w+)(?:\(tied-def \d+\))? = (COPY|SI_SPILL_\w+_RESTORE|V_ACCVGPR_\w+|V_MOV_B\w+)')
VEC_SAVE = re.compile(r'^\s*SI_SPILL_(?:V|AV|A)\d+_SAVE')
BLOCK = re.compile(r'^  bb\.\d+')
NAME = re.compile(r'^name:\s+(\S+)')


def census(unit, work):
  if unit == 'long': flags, src = [], 'gn_long_inst.hip'
  else:
    dof, t, g = unit.split('_')
    flags = (['-DDGP_TL=1'] if dof.endswith('t') else (['-DDGP_STEP_ERRS=1'] if dof.endswith('e') else [])) + \
            ['-DDGP_INST_DOF=' + dof.rstrip('te'), '-DDGP_INST_F64=%d' % (t == 'f64'), '-DDGP_INST_GROUP=' + g[1:]]
    src = 'gn_inst.hip'
  mir = os.path.join(work, unit + '.mir')
  # hipcc's OWN code generation, stopped behind the VGPR run of the register allocator (the same driver and switches as the product build: llc on the emitted IR allocates
  # slightly differently and finds a different set of instances)
  subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '--cuda-device-only', '-S', '-Wno-unused-command-line-argument', '-mllvm', '-stop-after=greedy,2'] + flags +
                        [os.path.join(CSRC, src), '-o', mir], stderr=subprocess.DEVNULL)
  out = {}      # kernel -> [saves, restores, copies to A/AV, other vector copies]
  kernel, pending, in_body = None, [], False
  n_kernels = 0
  with open(mir) as f:
    for line in f:
      m = NAME.match(line)
      if m: kernel, pending = m.group(1), []; n_kernels += 1; continue
      if BLOCK.match(line): pending = []; continue
      if '$exec = S_OR_B64 $exec' in line or '$exec = S_OR_B64_term $exec' in line:
        for p in pending:
          c = out.setdefault(kernel, [0, 0, 0, 0])
          if VEC_SAVE.match(p): c[0] += 1
          else:
            m = VEC_DEF.match(p)
            if 'RESTORE' in m.group(2): c[1] += 1
            elif re.search(r':(av|areg)_|\$agpr', m.group(1)): c[2] += 1
            else: c[3] += 1
        pending = []; continue
      s = line.strip()
      if not s or s.startswith((';', 'successors', 'liveins', 'predecessors')): continue
      if VEC_SAVE.match(line) or VEC_DEF.match(line): pending.append(line)
      elif '$exec' in line.split('=')[0] if '=' in line else False: pending = []      # another exec write: not a join prologue
  os.remove(mir)
  return unit, n_kernels, out


def main():
  check = '--check' in sys.argv[1:]
  units = [u for u in sys.argv[1:] if u != '--check'] or ALL
  work = tempfile.mkdtemp(prefix='dgp_census_')
  tot = [0, 0, 0, 0]; nk = 0; hit = 0
  found = {}
  print('# unit: kernels | kernels with vector code in front of a join\'s exec restore | SI_SPILL saves / SI_SPILL restores / copies into A(V) registers / other vector copies')
  with ThreadPoolExecutor(max_workers=min(len(units), os.cpu_count() or 1)) as ex:
    for unit, n, out in ex.map(lambda u: census(u, work), units):
      c = [sum(v[i] for v in out.values()) for i in range(4)]
      nk += n; hit += len(out)
      for i in range(4): tot[i] += c[i]
      if any(c): found[unit] = c
      print('%-10s %3d | %3d | %d / %d / %d / %d' % (unit, n, len(out), c[0], c[1], c[2], c[3]), flush=True)
      for k, v in sorted(out.items()):
        print('    %s: saves %d restores %d copies->A(V) %d other copies %d' % (k, *v))
  print('# total: %d kernels, %d with findings | saves %d restores %d copies->A(V) %d other vector copies %d' % (nk, hit, *tot))
  if check:
    import json
    rep = json.load(open(os.path.join(ROOT, 'dgpmp2_amd', 'lib', 'kernel_stats.json'))).get('_exec_join_repaired', {})
    rep = {k.replace('gn_inst_', ''): v for k, v in rep.items()}
    want = {u: 2 * (c[0] + c[2]) for u, c in found.items()}      # a 64-bit copy / save = two instructions in the assembly
    odd = {u: c for u, c in found.items() if c[1] or c[3]}
    ok = want == rep and not odd
    print('# check against the build\'s repairs %s: %s' % (rep, 'SAME' if ok else 'DIFFERENT (census %s%s)' % (want, ', reloads / plain copies: %s' % odd if odd else '')))
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
  main()
