#!/bin/bash
# Round 6: the exec-join miscompile (profiles/r06_compiler_fault.md) as llc-only reproducers: (1) a 90-line MIR function through -run-pass=greedy, with its control (the same
# function without the SGPR COPY in front of the exec restore); (2) ONE real kernel's optimised LLVM IR, 0.3 s per run.
#   profiles/tools/r06_llc_repro.sh            run llc on the committed IR, run the checker, show the join block before / after the register allocator
#   profiles/tools/r06_llc_repro.sh regen      re-derive the IR from the source first (hipcc -emit-llvm of the unit, opt internalize + globaldce around the one kernel)
set -u
R=$(cd "$(dirname "$0")/../.." && pwd)
B=${LLVM_BIN:-/opt/rocm/lib/llvm/bin}
W=${TMPDIR:-/tmp}/dgp_llc_repro; mkdir -p "$W"
K=_ZN7dgp_dev9gn_kernelILi2ELi16ELi2EfLi0ELi0ELi0EEEvN3dgp8GnParamsE
IRGZ=$R/profiles/r06_exec_join_repro/gn_kernel_2_16_2_float_step_general.ll.gz
if [ "${1:-}" = regen ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC --cuda-device-only -emit-llvm -S -DDGP_INST_DOF=2 -DDGP_INST_F64=0 -DDGP_INST_GROUP=1 "$R/dgpmp2_amd/csrc/gn_inst.hip" -o "$W/unit.ll" 2>/dev/null || exit 1
  "$B/opt" -S -passes='internalize,globaldce' -internalize-public-api-list=$K "$W/unit.ll" -o "$W/one.ll" || exit 1
  gzip -9 -c "$W/one.ll" > "$IRGZ"
fi
M=$R/profiles/r06_exec_join_repro/exec_join_prologue.mir
echo "== (1) exec_join_prologue.mir through -run-pass=greedy: the join block bb.2"
"$B/llc" -mtriple=amdgcn-amd-amdhsa -mcpu=gfx950 -run-pass=greedy -verify-machineinstrs "$M" -o - 2>/dev/null | awk '/^  bb.2:/,/S_BRANCH/' | grep -E "SI_SPILL|S_OR_B64|COPY" | sed 's/, implicit.*//; s/ :: .*//'
echo "== control: the same function without the SGPR COPY in front of the exec restore"
grep -v 'sgpr12_sgpr13 = COPY' "$M" | sed 's/\$sgpr12_sgpr13/$sgpr6_sgpr7/g' > "$W/control.mir"
"$B/llc" -mtriple=amdgcn-amd-amdhsa -mcpu=gfx950 -run-pass=greedy -verify-machineinstrs "$W/control.mir" -o - 2>/dev/null | awk '/^  bb.2:/,/S_BRANCH/' | grep -E "SI_SPILL|S_OR_B64|COPY" | sed 's/, implicit.*//; s/ :: .*//'
echo "== the MIR function compiled to assembly (-start-before=greedy) and the checker on it"
"$B/llc" -mtriple=amdgcn-amd-amdhsa -mcpu=gfx950 -start-before=greedy "$M" -o "$W/mir.s" 2>/dev/null; python "$R/profiles/tools/exec_join_check.py" "$W/mir.s"
echo "== (2) the IR of gn_kernel<2,16,2,float,STEP,general>"
gzip -dc "$IRGZ" > "$W/one.ll"
"$B/llc" --version | grep -i "version" | head -2
"$B/llc" -mtriple=amdgcn-amd-amdhsa -mcpu=gfx950 -O3 "$W/one.ll" -o "$W/one.s" || exit 1
python "$R/profiles/tools/exec_join_check.py" "$W/one.s"
echo "== which allocator runs it takes (llc switches; code size as a proxy for what the switch costs)"
for o in "" "-sgpr-regalloc=basic" "-sgpr-regalloc=fast" "-vgpr-regalloc=basic"; do
  "$B/llc" -mtriple=amdgcn-amd-amdhsa -mcpu=gfx950 -O3 $o "$W/one.ll" -o "$W/alt.s" 2>/dev/null
  echo "  [${o:-default: greedy / greedy}]  $(python "$R/profiles/tools/exec_join_check.py" "$W/alt.s" | tail -1)  $(grep -E '^; codeLenInByte' "$W/alt.s" | tr -d ';')"
done
echo "== the join block in the MIR: after the pass in front of the VGPR allocation run, and after that run (print-after=greedy, third dump)"
"$B/llc" -mtriple=amdgcn-amd-amdhsa -mcpu=gfx950 -O3 "$W/one.ll" -o /dev/null -print-after=amdgpu-reserve-wwm-regs -print-after=greedy 2> "$W/pa.txt"
python - "$W/pa.txt" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
parts = re.split(r'^# \*\*\* IR Dump After (.*?) \*\*\*:\n', txt, flags=re.M)
names, bodies = parts[1::2], parts[2::2]
for want in ('AMDGPU Reserve WWM', 'Greedy'):
  i = [k for k, n in enumerate(names) if n.startswith(want)][-1]
  lines = bodies[i].split('\n')
  for j, l in enumerate(lines):
    if 'S_OR_B64 $exec' in l:
      k = j
      while k > 0 and not re.match(r'\d+B\tbb\.\d+', lines[k]): k -= 1
      blk = [x for x in lines[k:j + 1] if not x.lstrip().startswith((';', 'successors', 'liveins'))]
      if want == 'Greedy' and not any('COPY' in x and ('av_' in x or 'areg' in x or 'agpr' in x) for x in blk): continue
      if want != 'Greedy' and 'bb.97.' not in lines[k]: continue
      print('--', names[i]); print('\n'.join(x[:170] for x in blk)); break
PY
