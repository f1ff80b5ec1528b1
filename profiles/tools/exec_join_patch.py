#!/usr/bin/env python
"""Repair of the miscompile signature exec_join_check.py finds (profiles/r06_compiler_fault.md): register-allocator spill code (VGPR -> AGPR copies, scratch stores,
and reloads) that hipcc left at the top of a JOIN block in front of the `s_or_b64 exec, exec, sN` that re-enables the lanes of a divergent `if` is moved directly
BEHIND that instruction.  Nothing else changes: the moved instructions read / write vector registers and scratch only, the instructions they cross are scalar (checked:
a store is left alone -- and the build fails -- if anything in between rewrites the register it saves, touches its slot or is a memory operation), and behind the restore they do for
EVERY lane what they did for the lanes of the branch -- which is what a spill of a live-through value means.
  python profiles/tools/exec_join_patch.py in.s out.s      -> prints the number of instructions moved; out.s == in.s when there is nothing to repair"""
import re
import sys
import exec_join_check as C


def _vregs(text):
  out = set()
  for tok in re.findall(r'v\[\d+:\d+\]|\bv\d+\b', text): out |= C._regs(tok)
  return out


def movable(lines, i, r):
  """May the spill store in line i move behind the restore in line r?  Only if nothing in between rewrites the register it saves or touches the slot it writes: the
  instructions it crosses are normally scalar (SGPR copies, lane-mask merges); anything else must not depend on it."""
  inst = lines[i].split(';')[0]
  mw = C.SLOT_W.match(inst)
  if not mw: return False
  slot = mw.group(1)                      # an AGPR name, or None for a scratch slot
  srcs = _vregs(inst.split(',', 1)[1]) if slot else _vregs(inst)
  for k in range(i + 1, r):
    l = lines[k]
    if l is None: continue
    l = l.split(';')[0]
    if not l.strip() or l.lstrip().startswith('.'): continue
    head = l.split()[0]
    if head.startswith('s_'): continue
    if head.startswith(('scratch_', 'buffer_', 'global_', 'flat_', 'ds_')): return False      # would change the order of memory operations (and their wait counts)
    ops = l.split(None, 1)[1] if len(l.split(None, 1)) > 1 else ''
    first = ops.split(',')[0]
    writes = _vregs(first) if not head.startswith('v_accvgpr_write') else set()
    if writes & srcs: return False                       # the saved register is rewritten in front of the restore: moving the store would save the new value
    if slot and re.search(r'\b%s\b' % slot, l): return False      # the slot is read or written in between
  return True


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
    idxs = [i for i in idxs if movable(lines, i, r)]     # (what cannot move stays, is found again by the caller's check and fails the build)
    if not idxs: continue
    block = [lines[i] for i in idxs]
    for i in idxs: lines[i] = None
    lines[r] = lines[r] + '\n' + '\n'.join(block) + '\t; (moved behind the exec restore: exec_join_patch.py)'
    moved += len(idxs)
  open(dst, 'w').write('\n'.join(l for l in lines if l is not None))
  return moved


if __name__ == '__main__':
  n = patch(sys.argv[1], sys.argv[2])
  print('%d instruction(s) moved' % n)
