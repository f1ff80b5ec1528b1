#!/usr/bin/env python
"""Turn a rocprofv3 rocpd sqlite result (--kernel-trace --stats [--pmc ...]) into a small text summary for profiles/."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
print('# %s' % (sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]))
print('## kernel stats (ns): name | calls | total | avg | min | max | vgpr | agpr | sgpr | scratch | lds | grid | wg')
for r in cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(accum_vgpr_count), "
                     "max(sgpr_count), max(scratch_size), max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc limit 12"):
  print(' | '.join(str(round(x, 1)) if isinstance(x, float) else str(x) for x in r))
try:
  rows = list(cur.execute("select k.name, p.name, count(*), avg(e.value) from pmc_events e join kernels k on k.dispatch_id = e.dispatch_id "
                          "join pmc_info p on p.id = e.pmc_id group by k.name, p.name order by k.name"))
  if rows:
    print('## counters: kernel | counter | dispatches | avg value per dispatch')
    for r in rows: print(' | '.join(str(x) for x in r))
except sqlite3.Error as e:
  print('## no counters (%s)' % e)
