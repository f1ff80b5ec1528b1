#!/usr/bin/env python
"""Histogram of one kernel's gfx950 assembly by source line (build with -gline-tables-only -save-temps; profiles/tools/kprobe.sh NAME ... -gline-tables-only).
   python profiles/tools/isa_lines.py /tmp/kp/NAME/*gfx950.s [regex of mnemonics, default: all VALU]   -> instruction counts per file:line, most first
Found with it (round 3): three speculated fp64 square roots in the static Woodbury step (DESIGN.md section 5 "(i)")."""
import collections
import os
import re
import sys


def main():
  path = sys.argv[1]
  pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else r'v_')
  files, cur, infn, cnt = {}, None, False, collections.Counter()
  for l in open(path):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
    if re.match(r'_ZN7dgp_dev\w+:', l): infn = os.environ.get('ISA_KERNEL', '') in l      # ISA_KERNEL=<mangled-name substring>: one kernel of a unit with many
    if infn and 's_endpgm' in l: infn = False
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m: cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
    m = re.match(r'\s*([a-z][a-z_0-9]+)\s', l)
    if infn and m and pat.match(m.group(1)): cnt[(m.group(1) if len(sys.argv) > 2 else '',) + (cur or ('?', 0))] += 1
  for k, v in cnt.most_common(40): print('%5d  %s %s:%d' % ((v,) + k))
  print('%5d  total' % sum(cnt.values()))


if __name__ == '__main__':
  main()
