"""CPU-only: the C-ABI shared library loads and exports every symbol include/dgpmp2_hip.h declares; host-side
argument validation works without a GPU (no compute calls here)."""
import ctypes as C
import os
import re
import pytest
from dgpmp2_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def api():
  if not os.path.exists(_capi.LIB_PATH):
    import __graft_entry__
    __graft_entry__.build()
  return _capi.get_api()


def test_exports_every_declared_symbol(api):
  hdr = open(os.path.join(ROOT, 'include', 'dgpmp2_hip.h')).read()
  hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
  declared = set(re.findall(r'\b(dgp_[a-z0-9_]+)\s*\(', hdr))
  assert declared == set('dgp_' + s for s in _capi.CApi.SYMBOLS)
  for name in declared:
    assert hasattr(api.lib, name), name
  assert api.abi_version() == _capi.DGP_ABI_VERSION


def test_struct_layout_matches_header():
  # sizes implied by the header on LP64: see DgpConfig/DgpSdf/DgpCovs in include/dgpmp2_hip.h
  assert C.sizeof(_capi.DgpConfig) == 6 * 4 + 8 * (1 + 2 + 2 + 2 + 1 + 1 + 9 + 1 + 1 + 1 + 3)
  assert C.sizeof(_capi.DgpSdf) == 40 and C.sizeof(_capi.DgpCovs) == 32


def _cfg(**kw):
  base = dict(num_states=64, dof=2, io_dtype=_capi.DGP_F32, total_time_sec=10.0, x_lims=(-5, 5), y_lims=(-5, 5), K_s=0.01,
              K_g=0.01, reg=0.1, sphere_radius=0.4, Q_c_inv=[[1, 0], [0, 1]], cost_sigma=0.01, epsilon_dist=0.4)
  base.update(kw)
  return _capi.make_config(**base)


def test_create_destroy_and_M(api):
  s = _capi.Solver(_cfg())
  assert s.M == 4 * (63 + 2) + 64                          # plan_layer.py:43
  s2 = _capi.Solver(_cfg(use_vel_limits=True, K_v=0.01, v_x=1.0, v_y=1.0))
  assert s2.M == 4 * 65 + 64 + 2 * 64                       # plan_layer.py:45
  s3 = _capi.Solver(_cfg(dof=3, Q_c_inv=[[1, 0, 0], [0, 1, 0], [0, 0, 1]], non_holonomic=True, K_d=0.01))
  assert s3.M == 6 * 65 + 64 + 64                           # plan_layer.py:44


@pytest.mark.parametrize('kw,code', [
    (dict(dof=4, Q_c_inv=[[1, 0, 0, 0]] * 4), _capi.DGP_EUNSUPPORTED),
    (dict(nlinks=2), _capi.DGP_EUNSUPPORTED),
    (dict(num_states=1), _capi.DGP_EINVAL),
    (dict(non_holonomic=True, K_d=0.01), _capi.DGP_EINVAL),       # needs dof == 3
    (dict(K_s=0.0), _capi.DGP_EINVAL),
    (dict(x_lims=(5, -5)), _capi.DGP_EINVAL),
])
def test_create_rejects_bad_config(api, kw, code):
  with pytest.raises(_capi.DgpError) as e:
    _capi.Solver(_cfg(**kw))
  assert e.value.code == code and len(str(e.value)) > 20


def test_calls_validate_arguments_without_gpu(api):
  s = _capi.Solver(_cfg())
  sdf = s.sdf_arg(None, 256, 256, 0)
  with pytest.raises(_capi.DgpError) as e:
    s.gn_step(8, 0x1000, 0x1000, 0x1000, sdf, None, 0x1000)
  assert e.value.code == _capi.DGP_EINVAL and 'sdf' in str(e.value)
  sdf = s.sdf_arg(0x1000, 256, 256, 0)
  with pytest.raises(_capi.DgpError):
    s.gn_step(0, 0x1000, 0x1000, 0x1000, sdf, None, 0x1000)
  with pytest.raises(_capi.DgpError):
    s.gn_step(8, 0x1000, 0x1000, 0x1000, sdf, s.covs_arg(_capi.DGP_QC_PERSTATE, None), 0x1000)
  with pytest.raises(_capi.DgpError) as e:                       # single-column grids: taps are fetched as column pairs
    s.gn_step(8, 0x1000, 0x1000, 0x1000, s.sdf_arg(0x1000, 64, 1, 0), None, 0x1000)
  assert e.value.code == _capi.DGP_EUNSUPPORTED


def test_launch_shape_choice(api):
  """dgp_launch_shape (host logic, dgp_host::choose_shape): the shape always covers n, the benchmark batch gets 16 lanes x 4
  states, short trajectories get one state per lane, long ones the only shape that fits."""
  for n in (2, 7, 16, 33, 64, 101, 128, 200, 256):
    s = _capi.Solver(_cfg(num_states=n))
    for B in (1, 3, 256, 1024, 4096, 65536):
      lpt, c = s.launch_shape(B)
      assert lpt in (16, 32, 64) and c in (1, 2, 4) and lpt * c >= n, (n, B, lpt, c)
  assert _capi.Solver(_cfg(num_states=64)).launch_shape(4096) == (16, 4)
  assert _capi.Solver(_cfg(num_states=64)).launch_shape(32768) == (16, 4)
  assert _capi.Solver(_cfg(num_states=16)).launch_shape(4096) == (16, 1)
  assert _capi.Solver(_cfg(num_states=101)).launch_shape(4096) == (32, 4)
  assert _capi.Solver(_cfg(num_states=256)).launch_shape(8) == (64, 4)
  s6 = _capi.Solver(_cfg(dof=3, Q_c_inv=[[1, 0, 0], [0, 1, 0], [0, 0, 1]]))
  assert s6.launch_shape(4096) == (16, 4)                   # d = 6, BASELINE configs[3]: 29.0 us against 45.8 us with (32,2)
  assert s6.launch_shape(32768) == (16, 4)
  # n > 256: the loop kernels of gn_long.h, one trajectory per wavefront, ceil(n / 64) rows per lane (reported as C), whatever the batch
  assert _capi.Solver(_cfg(num_states=257)).launch_shape(4096) == (64, 5)
  assert _capi.Solver(_cfg(num_states=512)).launch_shape(1) == (64, 8)
  assert _capi.Solver(_cfg(num_states=1024)).launch_shape(7) == (64, 16)
  assert _capi.Solver(_cfg(num_states=640, dof=3, Q_c_inv=[[1, 0, 0], [0, 1, 0], [0, 0, 1]])).launch_shape(7) == (64, 10)
  assert _capi.Solver(_cfg(num_states=512)).step_kernel_variant(64) == 0             # (generic rows: no static / Woodbury specialisation)
  for kw in (dict(num_states=1025), dict(num_states=641, dof=3, Q_c_inv=[[1, 0, 0], [0, 1, 0], [0, 0, 1]])):      # LDS capacity of those kernels
    with pytest.raises(_capi.DgpError) as e:
      _capi.Solver(_cfg(**kw))
    assert e.value.code == _capi.DGP_EUNSUPPORTED


def test_step_kernel_variant_choice(api, monkeypatch):
  """dgp_step_kernel_variant (host logic): the Woodbury kernels (3) exactly where gn_woodbury.h applies -- Q_c_inv = c I, no velocity
  limits, four states per lane, at least four states (3 when the trajectory fills the shape, 4 otherwise) -- the block elimination (1) elsewhere, the general kernels (0) for a
  non-diagonal Q_c_inv; DGP_NO_WOODBURY=1 keeps the block elimination."""
  monkeypatch.delenv('DGP_NO_WOODBURY', raising=False)
  monkeypatch.delenv('DGP_FORCE_SHAPE', raising=False)
  assert _capi.Solver(_cfg(num_states=64)).step_kernel_variant(4096) == 3            # BASELINE configs[1]
  assert _capi.Solver(_cfg(num_states=64)).step_kernel_variant(256) == 1             # (32,2) at small batches
  assert _capi.Solver(_cfg(num_states=63)).step_kernel_variant(4096) == 4            # a missing row: the goal row is an interior row of the last lane -> the 'ragged' instantiation
  assert _capi.Solver(_cfg(num_states=101)).step_kernel_variant(4096) == 4           # the reference YAML's length, shape (32,4)
  assert _capi.Solver(_cfg(num_states=16)).step_kernel_variant(4096) == 1            # one state per lane
  assert _capi.Solver(_cfg(num_states=256)).step_kernel_variant(8) == 3
  assert _capi.Solver(_cfg(num_states=64, use_vel_limits=True, K_v=0.01)).step_kernel_variant(4096) == 1
  assert _capi.Solver(_cfg(num_states=64, Q_c_inv=[[1, 0], [0, 2]])).step_kernel_variant(4096) == 1
  assert _capi.Solver(_cfg(num_states=64, Q_c_inv=[[1, 0.1], [0.1, 1]])).step_kernel_variant(4096) == 0
  assert _capi.Solver(_cfg(dof=3, Q_c_inv=[[1, 0, 0], [0, 1, 0], [0, 0, 1]], non_holonomic=True, K_d=0.01)).step_kernel_variant(4096) == 3
  monkeypatch.setenv('DGP_NO_WOODBURY', '1')
  assert _capi.Solver(_cfg(num_states=64)).step_kernel_variant(4096) == 1


def test_spill_guard_every_heavy_spiller_was_verified_on_a_gpu():
  """hipcc has miscompiled these kernels six times, every time among the heaviest spillers.  A kernel at that spill
  level (profiles/tools/spill_guard.py: >= 300 spilled VGPRs, >= 150 spilled SGPRs or >= 1 KB scratch per lane) must be listed, with its spill
  counts, in dgpmp2_amd/csrc/spill_baseline.json -- which is only ever rewritten after tests/test_hip_every_kernel.py ran green on a GPU
  against the build that produced those counts.  A new or grown heavy spiller fails HERE, before it ships unverified."""
  import json, os, sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  sys.path.insert(0, os.path.join(root, 'profiles', 'tools'))
  import spill_guard
  stats = json.load(open(os.path.join(root, 'dgpmp2_amd', 'lib', 'kernel_stats.json')))
  base = spill_guard.load_baseline()
  assert base, 'dgpmp2_amd/csrc/spill_baseline.json is missing or empty'
  bad = spill_guard.check(stats, base)
  assert not bad, 'heavy spillers without a GPU-verified baseline entry:\n' + '\n'.join('%s %s (baseline %s)' % b for b in bad)
  # the guard itself: a grown spill count and an unknown kernel are both reported
  k = next(iter(base))
  grown = {k: dict(scratch_bytes_per_lane=base[k][0] * 2 + 2048, vgpr_spill=base[k][1], sgpr_spill=base[k][2]), 'gn_kernel<9,9,9,float,0,0>': dict(scratch_bytes_per_lane=4096, vgpr_spill=0, sgpr_spill=0)}
  assert len(spill_guard.check(grown, base)) == 2


def test_exec_join_checker_and_repair(tmp_path):
  """Round 6: the static checker of the hipcc miscompile signature (profiles/r06_compiler_fault.md) and the repair the build applies, on a synthetic kernel:
  a spill store at the top of a join block in front of the exec restore is found and moved behind it; the body of the `if`, an out-of-line body, a slot that
  holds an earlier unconsumed write (`merge`) and a reload whose register is rewritten are left alone; the built library's record carries no surviving finding."""
  import json, sys
  sys.path.insert(0, os.path.join(ROOT, 'profiles', 'tools'))
  import exec_join_check as CK, exec_join_patch as P
  asm = '''
_Z6kernelv:
	v_accvgpr_write_b32 a7, v9
	s_and_saveexec_b64 s[0:1], vcc
	s_cbranch_execz .LBB0_2
; %bb.1:
	v_accvgpr_write_b32 a5, v3
	global_store_dword v[0:1], v2, off
.LBB0_2:
	v_accvgpr_write_b32 a1, v10
	v_accvgpr_write_b32 a7, v11
	s_mov_b64 s[4:5], s[6:7]
	s_or_b64 exec, exec, s[0:1]
	v_accvgpr_read_b32 v4, a1
	s_and_saveexec_b64 s[0:1], vcc
	s_cbranch_execz .LBB0_4
; %bb.3:
	global_store_dword v[0:1], v4, off
.LBB0_4:
	v_accvgpr_read_b32 v20, a2
	v_mov_b32_e32 v20, 0
	s_or_b64 exec, exec, s[0:1]
	s_and_saveexec_b64 s[0:1], vcc
	s_cbranch_execnz .LBB0_6
.LBB0_5:
	s_or_b64 exec, exec, s[0:1]
	s_endpgm
.LBB0_6:
	v_accvgpr_write_b32 a9, v1
	s_branch .LBB0_5
.Lfunc_end0:
'''
  src, dst = str(tmp_path / 'k.s'), str(tmp_path / 'k_fixed.s')
  open(src, 'w').write(asm)
  kinds = sorted((f[3].split()[1].rstrip(','), f[5]) for f in CK.classify(src))
  assert kinds == [('a1', 'store'), ('a7', 'merge'), ('v20', 'reload')], kinds      # a5 (the body) and a9 (the out-of-line body) are not join blocks
  assert P.patch(src, dst) == 1
  fixed = open(dst).read().split('\n')
  i_or = next(i for i, l in enumerate(fixed) if 's_or_b64 exec, exec, s[0:1]' in l)
  assert 'v_accvgpr_write_b32 a1, v10' in fixed[i_or + 1] and not any(f[5] == 'store' for f in CK.classify(dst))
  assert sum('v_accvgpr_write_b32 a7, v11' in l for l in fixed[:i_or]) == 1       # the merge-class store stays where it was
  stats = os.path.join(ROOT, 'dgpmp2_amd', 'lib', 'kernel_stats.json')
  if os.path.exists(stats):
    s = json.load(open(stats))
    if '_exec_join' in s: assert not [j for j in s['_exec_join'] if j['kind'] in ('store', 'reload-live')]


def test_exec_join_checker_tells_joins_from_body_tails(tmp_path):
  """Second pass of round 6: (1) the last block of an `if` body in front of a tail-duplicated restore is NOT a join -- its register saves belong to the body's lanes (the first
  checker `repaired` such an exit shuffle in a shipped kernel); (2) an `if` without a skip branch joins in the fall-through block that restores from its saved mask;
  (3) a store whose source register is rewritten in front of the restore is not moved -- the finding stays, which fails the build."""
  import sys
  sys.path.insert(0, os.path.join(ROOT, 'profiles', 'tools'))
  import exec_join_check as CK, exec_join_patch as P
  asm = '''
_Z2k1v:
	s_and_saveexec_b64 s[0:1], vcc
	s_cbranch_execz .LBB0_4
; %bb.1:
	s_cmp_eq_u64 s[20:21], 0
	s_cbranch_scc1 .LBB0_3
; %bb.2:
	global_store_dword v[0:1], v2, off
.LBB0_3:
	v_accvgpr_write_b32 a7, v71
	v_mov_b32_e32 v71, v64
	s_or_b64 exec, exec, s[0:1]
	s_branch .LBB0_5
.LBB0_4:
	s_or_b64 exec, exec, s[0:1]
.LBB0_5:
	s_and_saveexec_b64 s[2:3], s[4:5]
	v_readlane_b32 s24, v254, 51
; %bb.6:
	v_mov_b32_e32 v8, 0
; %bb.7:
	v_accvgpr_write_b32 a30, v236
	s_mov_b64 s[6:7], s[86:87]
	s_or_b64 exec, exec, s[2:3]
	s_and_saveexec_b64 s[8:9], vcc
	s_cbranch_execz .LBB0_9
; %bb.8:
	global_store_dword v[0:1], v2, off
.LBB0_9:
	v_accvgpr_write_b32 a31, v237
	v_mov_b32_e32 v237, v3
	s_or_b64 exec, exec, s[8:9]
	s_endpgm
.Lfunc_end0:
'''
  src, dst = str(tmp_path / 'k.s'), str(tmp_path / 'k_fixed.s')
  open(src, 'w').write(asm)
  found = sorted((f[1], f[3].split()[1].rstrip(','), f[5]) for f in CK.classify(src))
  assert found == [('%bb.7', 'a30', 'store'), ('.LBB0_9', 'a31', 'store')], found      # .LBB0_3 (the body's tail, entered from inside the body only) is not reported
  assert P.patch(src, dst) == 1                                                        # a31's source register v237 is rewritten in front of the restore: not movable
  left = [(f[1], f[5]) for f in CK.classify(dst)]
  assert left == [('.LBB0_9', 'store')], left


def test_exec_join_repair_on_the_compilers_own_output(tmp_path):
  """Round 6: the llc-only reproducer (profiles/r06_exec_join_repro: the optimised IR of gn_kernel<2,16,2,float,STEP,general>) through the installed llc -- the checker finds the
  misplaced spill copies of profiles/r06_compiler_fault.md in the compiler's own output, the patch moves them behind the exec restore, and the patched text still assembles.
  Skipped where llc is absent or no longer shows the signature (a fixed compiler)."""
  import gzip, subprocess, sys
  llvm = os.environ.get('LLVM_BIN', '/opt/rocm/lib/llvm/bin')
  llc, clang = os.path.join(llvm, 'llc'), os.path.join(llvm, 'clang')
  irgz = os.path.join(ROOT, 'profiles', 'r06_exec_join_repro', 'gn_kernel_2_16_2_float_step_general.ll.gz')
  if not (os.path.exists(llc) and os.path.exists(clang)): pytest.skip('no llc / clang under %s' % llvm)
  sys.path.insert(0, os.path.join(ROOT, 'profiles', 'tools'))
  import exec_join_check as CK, exec_join_patch as P
  ir, raw, fixed = str(tmp_path / 'one.ll'), str(tmp_path / 'one.s'), str(tmp_path / 'one_fixed.s')
  open(ir, 'wb').write(gzip.open(irgz).read())
  subprocess.check_call([llc, '-mtriple=amdgcn-amd-amdhsa', '-mcpu=gfx950', '-O3', ir, '-o', raw])
  stores = [f for f in CK.classify(raw) if f[5] == 'store']
  if not stores: pytest.skip('this llc does not place spill code in front of the exec restore any more')
  assert all('v_accvgpr_write_b32' in f[3] or 'scratch_store' in f[3] for f in stores)
  assert P.patch(raw, fixed) == len(stores)
  assert not [f for f in CK.classify(fixed) if f[5] in ('store', 'reload-live')]
  # same instructions, same count: only their order inside the join blocks changed
  body = lambda path: sorted(l.split(';')[0].strip() for l in open(path) if l.startswith('\t') and not l.strip().startswith(('.', ';')))
  assert body(raw) == body(fixed)
  subprocess.check_call([clang, '-x', 'assembler', '-target', 'amdgcn-amd-amdhsa', '-mcpu=gfx950', '-c', fixed, '-o', str(tmp_path / 'one.o')])


def test_exec_join_mir_reproducer_and_its_control(tmp_path):
  """The 90-line MIR reproducer of the spill placement (profiles/r06_exec_join_repro/exec_join_prologue.mir) through `llc -run-pass=greedy`: with an SGPR COPY in front of
  the exec restore of the join block the VGPR save lands in front of both; without the COPY it lands behind the restore.  Skipped without llc or on a fixed compiler."""
  import subprocess
  llc = os.path.join(os.environ.get('LLVM_BIN', '/opt/rocm/lib/llvm/bin'), 'llc')
  if not os.path.exists(llc): pytest.skip('no llc')
  mir = open(os.path.join(ROOT, 'profiles', 'r06_exec_join_repro', 'exec_join_prologue.mir')).read()
  control = '\n'.join(l for l in mir.split('\n') if 'sgpr12_sgpr13 = COPY' not in l).replace('$sgpr12_sgpr13', '$sgpr6_sgpr7')

  def join_block(text, name):
    path = str(tmp_path / name)
    open(path, 'w').write(text)
    out = subprocess.run([llc, '-mtriple=amdgcn-amd-amdhsa', '-mcpu=gfx950', '-run-pass=greedy', '-verify-machineinstrs', path, '-o', '-'], check=True, capture_output=True, text=True).stdout
    blk = out[out.index('  bb.2:'):]
    blk = blk[:blk.index('S_BRANCH')]
    return [('save' if 'SI_SPILL' in l else 'restore' if 'S_OR_B64' in l else 'copy') for l in blk.split('\n') if any(k in l for k in ('SI_SPILL', 'S_OR_B64 $exec', '= COPY'))]
  got = join_block(mir, 'a.mir')
  if got == ['copy', 'restore', 'save']: pytest.skip('this llc places the save behind the exec restore: fixed compiler')
  assert got == ['save', 'copy', 'restore'], got
  assert join_block(control, 'b.mir') == ['restore', 'save']
