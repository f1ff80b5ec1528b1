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


SAVEEXEC = re.compile(r'^\s+s_(?:and|andn2|or|xor)_saveexec_b64 (s\[\d+:\d+\]|vcc)')
SAVE_COPY = re.compile(r'^\s+s_mov_b64 (s\[\d+:\d+\]), exec\s*$')
EXEC_WRITE = re.compile(r'^\s+s_(?:mov|and|andn2)_b64 exec, ')
RESTORE_SRC = re.compile(r'^\s+s_or_b64 exec, exec, (s\[\d+:\d+\]|vcc)')


def check(path):
  """-> [(kernel, label, line, instruction, restore)]: spill-like vector instructions between the label of a JOIN block and the exec restore that follows it.
  A block is a join of a divergent `if` (or the exit of a divergent loop) when the lanes that SKIPPED the body arrive there:
    * it is the target of an `s_cbranch_execz` (the wavefront jumps over the body when no lane is left), or
    * it is the fall-through of an `s_cbranch_execnz` (out-of-line body, or the back edge of a divergent loop), or
    * the `if` has no skip branch at all (short bodies: `s_and_saveexec sN` -- or `s_mov_b64 sN, exec` ... `s_mov_b64 exec, sM` -- and straight on) and the block is a fall-through block behind it that restores from that sN.
  A block that is only entered from inside the body of an `if` WITH a skip branch -- the body's last block in front of a tail-duplicated restore, uniform branches inside the
  body -- is NOT a join: what it holds in front of the restore is the body's own code, meant for the body's lanes.  (Round 6, second pass: the first version of this checker took
  every labelled block in front of a restore for a join, and the repair moved two register saves of a body's exit shuffle in gn_backward_kernel<2,16,4,float,general>[tiled]
  behind the instructions that reuse their source registers; the MIR census profiles/tools/r06_mir_census.py showed the difference.)"""
  out = []
  text = open(path).read()
  lines = text.split('\n')
  skip_targets = set(re.findall(r's_cbranch_execz (\.LBB\d+_\d+)', text))
  kernel, label, pending, prev, allowed = None, None, [], '', None
  saved = set()        # `s_mov_b64 sN, exec` seen, the exec write that narrows the mask not yet
  open_nb = set()      # saved-exec registers of `if`s without a skip branch whose restore has not been seen yet (straight-line code only)
  for i, line in enumerate(lines):
    ln = i + 1
    m = KERNEL.match(line)
    if m: kernel, label, pending, prev, allowed = m.group(1), None, [], '', None; open_nb.clear(); saved.clear(); continue
    m = LABEL.match(line)
    if m:
      name, asm_label = (m.group(1) or m.group(2)), m.group(1) is not None
      join = (asm_label and name in skip_targets) or bool(re.match(r'^\s+s_cbranch_execnz', prev))
      if asm_label: open_nb.clear(); saved.clear()            # a branch target: other paths arrive here, the straight-line bookkeeping ends
      if join: label, allowed = name, None
      elif open_nb: label, allowed = name, set(open_nb)
      else: label, allowed = None, None
      pending = []
      continue
    if line.lstrip().startswith(';') or not line.strip(): continue
    m, mc, mw = SAVEEXEC.match(line), SAVE_COPY.match(line), EXEC_WRITE.match(line)
    if mc: saved.add(mc.group(1))            # `s_mov_b64 sN, exec`: the other way of opening an `if` (... s_and_b64 sM, sN, cond; s_mov_b64 exec, sM)
    if m or mw:
      ahead = [l for l in lines[i + 1:i + 12] if l.strip() and not l.lstrip().startswith(';')][:4]
      if not any(re.match(r'^\s+s_cbranch_exec', l) for l in ahead):
        if m: open_nb.add(m.group(1))
        else: open_nb |= saved
      saved.clear()
    if label is None:
      prev = line
      if BRANCH.match(line): open_nb.clear()
      src = RESTORE_SRC.match(line)
      if src: open_nb.discard(src.group(1))      # (restored inside a block that is no candidate: that `if` is closed)
      continue
    if RESTORE.match(line):
      src = RESTORE_SRC.match(line)
      if allowed is None or (src and src.group(1) in allowed):
        for l2, t in pending: out.append((kernel, label, l2, t.strip(), line.strip()))
      if src: open_nb.discard(src.group(1))
      label, pending, prev, allowed = None, [], line, None
      continue
    prev = line
    if BRANCH.match(line) or OTHER_EXEC.search(line):
      label, pending, allowed = None, [], None      # another exec write / the block ends: whatever was collected ran under a mask that belongs to it
      if BRANCH.match(line): open_nb.clear()
      continue
    if VEC.match(line) and not IGNORES_EXEC.match(line): pending.append((ln, line))
  return out


SLOT_W = re.compile(r'^\s+(?:v_accvgpr_write_b32 (a\d+),|scratch_store_\w+ off, v\S+, off(?: offset:(\d+))?)')
SLOT_R = re.compile(r'^\s+(?:v_accvgpr_read_b32 v\d+, (a\d+)\s*$|scratch_load_\w+ v\S+, off, off(?: offset:(\d+))?)')


def _regs(tok):
  """'v12' / 'v[12:15]' -> set of VGPR numbers"""
  m = re.match(r'v\[(\d+):(\d+)\]', tok)
  if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
  m = re.match(r'v(\d+)$', tok)
  return {int(m.group(1))} if m else set()


def _reload_kind(lines, i, inst):
  """A reload in front of the exec restore fills its register for the branch's lanes only.  'reload-live': the first thing that happens to the register BEHIND the
  restore (same block) is a read -- the other lanes consume a register nobody filled for them: the mirror image of the `store` signature;
  'reload': overwritten first, or not touched again before the block ends (the value served the branch's lanes in front of the restore)."""
  m = re.match(r'\s*(?:v_accvgpr_read_b32|scratch_load_\w+) (v\d+|v\[\d+:\d+\])', inst)
  if not m: return 'reload'
  mine = _regs(m.group(1))
  j = i + 1
  while j < len(lines) and not RESTORE.match(lines[j]): j += 1
  # (consumed in front of the restore too?  that serves the branch's lanes; what matters is whether anybody reads the register BEHIND the restore before it is rewritten.
  #  Registers rewritten in front of the restore drop out.)
  for k in range(i + 1, j):
    l = lines[k].split(';')[0]
    ops = re.findall(r'v\[\d+:\d+\]|v\d+', l)
    if ops and l.lstrip().startswith('v_') and not any(_regs(o) & mine for o in ops[1:]): mine -= _regs(ops[0])
  if not mine: return 'reload'
  j += 1
  while j < len(lines):
    l = lines[j].split(';')[0]
    if LABEL.match(lines[j]) or BRANCH.match(l) or KERNEL.match(lines[j]): return 'reload'
    ops = re.findall(r'v\[\d+:\d+\]|v\d+', l)
    if ops and l.strip() and not l.lstrip().startswith(('s_', '.')):
      stores = l.lstrip().startswith(('global_store', 'scratch_store', 'buffer_store', 'flat_store', 'ds_write', 'ds_store'))
      srcs = ops if stores else ops[1:]
      if any(_regs(o) & mine for o in srcs): return 'reload-live'
      if not stores and (_regs(ops[0]) & mine):
        mine -= _regs(ops[0])
        if not mine: return 'reload'
    j += 1
  return 'reload'


def classify(path, findings=None):
  """-> [(kernel, label, line, instruction, restore, kind)], kind:
       'store'   spill store at a join top whose slot holds NOTHING else for the other lanes: no write of the slot since its last read -- the miscompile (every lane
                 reads the slot later, only the branch's lanes wrote it);
       'merge'   spill store at a join top behind an earlier, not yet consumed write of the same slot (the two writes may complement each other: the slot of a
                 value defined on both paths): reported, not counted;
       'reload'  reload in front of the restore: wrong only if the register is read by other lanes afterwards -- reported, not counted (the every-kernel GPU tests cover them)."""
  findings = check(path) if findings is None else findings
  if not findings: return []
  lines = open(path).read().split('\n')
  out = []
  for kernel, label, ln, inst, rest in findings:
    mw = SLOT_W.match('\t' + inst)
    if not mw:
      out.append((kernel, label, ln, inst, rest, _reload_kind(lines, ln - 1, inst))); continue
    slot = mw.group(1) or ('scratch+%s' % (mw.group(2) or '0'))
    kind = 'store'
    i = ln - 2
    while i >= 0 and not KERNEL.match(lines[i]):
      l = lines[i]
      w, r = SLOT_W.match(l), SLOT_R.match(l)
      if r and (r.group(1) or ('scratch+%s' % (r.group(2) or '0'))) == slot: break           # consumed: whatever was written before is dead
      if w and (w.group(1) or ('scratch+%s' % (w.group(2) or '0'))) == slot: kind = 'merge'; break
      i -= 1
    out.append((kernel, label, ln, inst, rest, kind))
  return out


if __name__ == '__main__':
  bad = 0
  for f in sys.argv[1:]:
    for kernel, label, ln, inst, rest, kind in classify(f):
      bad += kind == 'store'
      print('%-6s %s:%d  %s  %s:  `%s`  in front of  `%s`' % (kind, f, ln, kernel, label, inst, rest))
  print('%d finding(s) of the miscompile signature' % bad)
  sys.exit(1 if bad else 0)
