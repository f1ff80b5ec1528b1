#!/usr/bin/env python
"""Static check of gfx950 assembly for ONE miscompile signature (found in round 6, profiles/r06_compiler_fault.md):

  .LBBx_y:                       ; %Flow...           <- join block of a divergent `if`
      v_accvgpr_write_b32 a122, v212                 <- register-allocator copy / spill, executed under the PARTIAL exec mask of the `if`
      s_or_b64 exec, exec, s[0:1]                    <- ... because it sits in front of the instruction that re-enables the lanes

Lanes that skipped the `if` never execute the copy: they later read a stale AGPR / scratch slot.  The checker reports, per kernel, every VGPR -> AGPR copy or scratch store (spill code)
between a block label and an exec-restoring `s_or_b64 exec, exec, ...` that follows it in the same block (before any other exec write / branch).
  python profiles/tools/exec_join_check.py file.s [...]      -> one line per finding; exit status 1 if any"""
import re, sys

LABEL = re.compile(r'^(\.LBB\d+_\d+):|^; (%bb\.\d+):')      # a block the assembler labels, or a fall-through block (comment only)
KERNEL = re.compile(r'^(_Z\w+):')
RESTORE = re.compile(r'^\s+s_or_b64 exec, exec, ')
# What the register allocator inserts and nothing else in these kernels does (no MFMA: AGPRs are spill space only): VGPR -> AGPR copies and scratch stores.
# (Plain v_mov / arithmetic in front of the restore can be legitimate: the phi copies of values the `if` body defined, which only its lanes must take.)
VEC = re.compile(r'^\s+(v_accvgpr_write_b32|scratch_store_|buffer_store_dword.*offen|buffer_store_dword.*s\[\d+:\d+\], 0 offset'
                 # ... and the mirror image: a RELOAD in front of the restore fills the register of the branch's lanes only
                 r'|v_accvgpr_read_b32|scratch_load_)')
IGNORES_EXEC = re.compile(r'^\s+(v_readlane_b32|v_writelane_b32|v_readfirstlane_b32)')
OTHER_EXEC = re.compile(r'exec')      # any other instruction that names exec (s_and_saveexec, s_mov exec, s_andn2 ... exec): the region in front of it is not a join prologue
BRANCH = re.compile(r'^\s+(s_cbranch|s_branch|s_endpgm|s_setpc)')


def check(path):
  out = []
  kernel, label, pending = None, None, []
  for ln, line in enumerate(open(path), 1):
    m = KERNEL.match(line)
    if m: kernel, label, pending = m.group(1), None, []; continue
    m = LABEL.match(line)
    if m: label, pending = (m.group(1) or m.group(2)), []; continue
    if label is None: continue
    if RESTORE.match(line):
      for l2, t in pending: out.append((kernel, label, l2, t.strip(), line.strip()))
      label, pending = None, []
      continue
    if line.lstrip().startswith(';'): continue
    if BRANCH.match(line) or OTHER_EXEC.search(line):
      label, pending = None, []      # another exec write / the block ends: whatever was collected ran under a mask that belongs to it
      continue
    if VEC.match(line) and not IGNORES_EXEC.match(line): pending.append((ln, line))
  return out


if __name__ == '__main__':
  bad = 0
  for f in sys.argv[1:]:
    for kernel, label, ln, inst, rest in check(f):
      bad += 1
      print('%s:%d  %s  %s:  `%s`  in front of  `%s`' % (f, ln, kernel, label, inst, rest))
  print('%d finding(s)' % bad)
  sys.exit(1 if bad else 0)
