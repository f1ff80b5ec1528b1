#!/bin/bash
# Round 6: issue / wait counters of dgp_sdf_2d's two kernels (4096 x 256^2), one counter group per rocprofv3 run: is the row pass bound by LDS reads, as profiles/r06_sdf_edt_ab.txt argues?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/edt_pmc; rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
P="python $R/profiles/tools/edt_bench.py"
export DGP_EDT_CASES=4096x256
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d "$O" -o sq1 -- $P > "$O/sq1.log" 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d "$O" -o sq2 -- $P > "$O/sq2.log" 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL -d "$O" -o sq3 -- $P > "$O/sq3.log" 2>&1
python - "$O" <<'PY'
import sys, sqlite3, glob, collections
out = sys.argv[1]
for db in sorted(glob.glob(out + '/**/*_results.db', recursive=True)):
  con = sqlite3.connect(db)
  tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
  pmc = [t for t in tabs if t.startswith('rocpd_pmc_event')]; ks = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')]; sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')]; info = [t for t in tabs if t.startswith('rocpd_info_pmc')]
  if not (pmc and ks and sym and info): print('#', db, 'no counter tables', tabs[:6]); continue
  q = ("select s.kernel_name, i.name, count(*), avg(e.value) from %s e join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id join %s i on e.pmc_id = i.id "
       "where s.kernel_name like '%%edt_%%' group by s.kernel_name, i.name" % (pmc[0], ks[0], sym[0], info[0]))
  try: rows = con.execute(q).fetchall()
  except Exception as ex: print('#', db, 'query failed:', ex); continue
  print('#', db.split('/')[-1])
  for k, n, c, v in rows: print('%-44s %-26s dispatches=%d avg=%.1f' % (k.split('(')[0][-44:], n, c, v))
PY
