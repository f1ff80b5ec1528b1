#!/usr/bin/env python
"""Static per-kernel statistics from the gfx950 assembly hipcc leaves behind with -save-temps=obj: registers, scratch, and
instruction histogram (fp64 FMA / MUL / ADD, DPP moves, all VALU, memory instructions).  __graft_entry__.build() runs this
over every translation unit and writes dgpmp2_amd/lib/kernel_stats.json; bench.py takes the fp64 flop count of the kernel
it launches from there (the kernels are straight-line apart from wave-uniform branches, so the static count is the executed
count to within the branch arms: SQ_INSTS_VALU / SQ_WAVES measured 3 882-4 104 against 4 104 static on the headline kernel).

  python profiles/tools/isa_stats.py <file.s> [...]      -> JSON on stdout, one entry per kernel
"""
import json
import re
import sys

def short_name(mangled):
  """'_ZN7dgp_dev9gn_kernelILi2ELi16ELi4EfLi0ELb1EEEvN3dgp8GnParamsE' -> 'gn_kernel<2,16,4,float,0,true>' (the template
  arguments of these kernels are ints, bools and float/double: decoded here, no external demangler needed)."""
  m = re.search(r'\d+(gn_\w+?)I((?:Li\d+E|Lb[01]E|[fd])+)E', mangled)
  if not m: return mangled
  args = []
  for t in re.findall(r'Li\d+E|Lb[01]E|[fd]', m.group(2)):
    args.append({'f': 'float', 'd': 'double'}.get(t) or (('true' if t[2] == '1' else 'false') if t[1] == 'b' else t[2:-1]))
  # the trailing TL argument of gn_kernel / gn_backward_kernel (the grid layout the translation unit is compiled for, csrc/gn_device.h): the standard kernels keep the
  # names every table, baseline and tool knows them by; the tiled twins get a suffix
  tiled = ''
  if m.group(1) in ('gn_kernel', 'gn_backward_kernel') and len(args) == (7 if m.group(1) == 'gn_kernel' else 7):
    tiled = {'0': '', '1': '[tiled]', '2': '[errs]'}.get(args[-1], '[twin %s]' % args[-1])
    args = args[:-1]
  return '%s<%s>%s' % (m.group(1), ','.join(args), tiled)


def parse(path):
  kernels = {}
  cur, body = None, None
  meta_for = None
  for line in open(path, errors='replace'):
    s = line.strip()
    m = re.match(r'^(_Z\w+):\s*(;.*)?$', s)
    if m and cur is None:
      cur, body = m.group(1), {'valu': 0, 'fma_f64': 0, 'mul_f64': 0, 'add_f64': 0, 'rcp_f64': 0, 'other_f64': 0, 'dpp': 0, 'salu': 0, 'smem': 0,
                              'vmem_load': 0, 'vmem_store': 0, 'atomic': 0, 'lds': 0, 'branch': 0, 'agpr_moves': 0, 'scratch_ops': 0, 'mfma': 0,
                              'total': 0}
      continue
    if cur is not None:
      if s.startswith('.Lfunc_end'):
        kernels[cur] = body
        meta_for, cur, body = cur, None, None
        continue
      if not s or s[0] in '.;' or s.endswith(':'):
        continue
      op = s.split()[0]
      body['total'] += 1
      if op.startswith('v_'):
        if op.startswith('v_mfma') or op.startswith('v_smfma'): body['mfma'] += 1
        elif op.startswith('v_accvgpr'): body['agpr_moves'] += 1; body['valu'] += 1
        else:
          body['valu'] += 1
          if op.endswith('_dpp') or ' row_' in s or 'quad_perm' in s: body['dpp'] += 1
          if re.match(r'v_(fma|fmac)_f64', op): body['fma_f64'] += 1
          elif op.startswith('v_mul_f64'): body['mul_f64'] += 1
          elif op.startswith('v_add_f64'): body['add_f64'] += 1
          elif op.startswith('v_rcp_f64'): body['rcp_f64'] += 1
          elif op.endswith('_f64') or '_f64_' in op: body['other_f64'] += 1
      elif op.startswith('s_load') or op.startswith('s_buffer_load'): body['smem'] += 1
      elif op.startswith('s_cbranch') or op.startswith('s_branch'): body['branch'] += 1
      elif op.startswith('s_'): body['salu'] += 1
      elif op.startswith('scratch_'): body['scratch_ops'] += 1
      elif op.startswith('global_atomic') or op.startswith('flat_atomic') or op.startswith('buffer_atomic'): body['atomic'] += 1
      elif op.startswith('global_load') or op.startswith('flat_load') or op.startswith('buffer_load'): body['vmem_load'] += 1
      elif op.startswith('global_store') or op.startswith('flat_store') or op.startswith('buffer_store'): body['vmem_store'] += 1
      elif op.startswith('ds_'): body['lds'] += 1
      continue
    if meta_for is not None:
      m = re.match(r'^; (NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|Occupancy|NumSgprs|LDSByteSize|codeLenInByte)\s*[:=]\s*(\d+)', s)
      if m:
        kernels[meta_for][{'NumVgprs': 'vgpr', 'NumAgprs': 'agpr', 'TotalNumVgprs': 'vgpr_total', 'ScratchSize': 'scratch_bytes_per_lane',
                           'Occupancy': 'waves_per_simd', 'NumSgprs': 'sgpr', 'LDSByteSize': 'lds_bytes', 'codeLenInByte': 'code_bytes'}[m.group(1)]] = int(m.group(2))
  # register spill counts from the code-object metadata at the end of the file (.sgpr_spill_count: SGPRs -- lane masks of selects and
  # ballots -- spilled into VGPR lanes; hipcc 7.0 has produced wrong code in d = 6 kernels that do a lot of it)
  text = open(path, errors='replace').read()
  for blk in text.split('  - .agpr_count:')[1:]:
    nm = re.search(r'\.name:\s+(\S+)', blk)
    if not nm or nm.group(1) not in kernels: continue
    for key, field in (('sgpr_spill', 'sgpr_spill_count'), ('vgpr_spill', 'vgpr_spill_count')):
      m = re.search(r'\.%s:\s+(\d+)' % field, blk)
      if m: kernels[nm.group(1)][key] = int(m.group(1))
  return {short_name(n): v for n, v in kernels.items()}


if __name__ == '__main__':
  out = {}
  for f in sys.argv[1:]:
    out.update(parse(f))
  json.dump(out, sys.stdout, indent=1, sort_keys=True)
