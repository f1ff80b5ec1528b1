#!/usr/bin/env python
"""Repair of the miscompile signature exec_join_check.py finds (profiles/r06_compiler_fault.md): register-allocator spill code (VGPR -> AGPR copies, scratch stores,
and reloads) that hipcc left at the top of a JOIN block in front of the `s_or_b64 exec, exec, sN` that re-enables the lanes of a divergent `if` is moved directly
BEHIND that instruction.  Nothing else changes: the moved instructions read / write vector registers and scratch only, the instructions they cross are scalar, and
behind the restore they do for EVERY lane what they did for the lanes of the branch -- which is what a spill of a live-through value means.
  python profiles/tools/exec_join_patch.py in.s out.s      -> prints the number of instructions moved; out.s == in.s when there is nothing to repair"""
import sys
import exec_join_check as C


def patch(src, dst):
  lines = open(src).read().split('\n')
  finds = [f[:5] for f in C.classify(src) if f[5] == 'store']      # only the definite signature: a store whose slot holds nothing else for the other lanes
  if not finds:
    if dst != src: open(dst, 'w').write('\n'.join(lines))
    return 0
  by_restore = {}
  for kernel, label, ln, inst, rest in finds: by_restore.setdefault((kernel, label), []).append(ln - 1)
  moved = 0
  for key, idxs in by_restore.items():
    idxs = sorted(idxs)
    r = idxs[-1] + 1
    while not C.RESTORE.match(lines[r]): r += 1          # the restore these instructions were found in front of
    block = [lines[i] for i in idxs]
    for i in idxs: lines[i] = None
    lines[r] = lines[r] + '\n' + '\n'.join(block) + '\t; (moved behind the exec restore: exec_join_patch.py)'
    moved += len(idxs)
  open(dst, 'w').write('\n'.join(l for l in lines if l is not None))
  return moved


if __name__ == '__main__':
  n = patch(sys.argv[1], sys.argv[2])
  print('%d instruction(s) moved' % n)
