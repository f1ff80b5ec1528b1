#!/usr/bin/env python
"""Read the rocpd databases written by pmc_traffic.sh and print per-kernel average counter values per dispatch; derive the
HBM traffic of the GN kernel with the calibration copy's correction factors and write profiles/traffic.json."""
import glob, json, os, sqlite3, sys
d = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc'
vals = {}
for f in sorted(glob.glob(os.path.join(d, '*_results.db'))):
  cur = sqlite3.connect(f).cursor()
  try:
    rows = cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
  except sqlite3.Error as ex:
    print('# %s: %s' % (f, ex)); continue
  for kname, cname, cnt, avg in rows:
    short = 'gn_kernel' if 'gn_kernel' in kname else ('copy' if 'copyBuffer' in kname else ('calib' if 'calib_dword_rw' in kname else None))
    if short: vals.setdefault(short, {}).setdefault(cname, []).append((cnt, avg, kname[:90]))
  for r in cur.execute("select name, count(*), avg(duration) from kernels where name like '%gn_kernel%' group by name"):
    print('# %s: %s x%d avg %.1f ns' % (os.path.basename(f), r[0][:80], r[1], r[2]))
for short, cs in vals.items():
  for c, lst in sorted(cs.items()):
    for cnt, avg, kn in lst: print('%-10s %-22s dispatches=%-4d avg=%.1f   [%s]' % (short, c, cnt, avg, kn))
try:
  known = 64 * 1024 * 1024 * 4            # calibration kernels: bytes read == bytes written == 256 MiB
  wide_f = known / (max(v[1] for v in vals['copy']['FETCH_SIZE']) * 1024.0)
  print('# wide float4 copy: FETCH_SIZE correction %.3f (the guide 2x), WRITE_SIZE correction %.3f' %
        (wide_f, known / (max(v[1] for v in vals['copy']['WRITE_SIZE']) * 1024.0)))
  cf = max(v[1] for v in vals['calib']['FETCH_SIZE']); cw = max(v[1] for v in vals['calib']['WRITE_SIZE'])
  kf = known / (cf * 1024.0); kw = known / (cw * 1024.0)
  print('# calibration kernel in the GN kernel own row access pattern: FETCH_SIZE correction %.3f, WRITE_SIZE correction %.3f' % (kf, kw))
  gf = vals['gn_kernel']['FETCH_SIZE'][0][1]; gw = vals['gn_kernel']['WRITE_SIZE'][0][1]
  out = {'fetch_size_kb_raw': gf, 'write_size_kb_raw': gw, 'fetch_correction': kf, 'write_correction': kw,
         'hbm_bytes_per_launch': gf * 1024.0 * kf + gw * 1024.0 * kw,
         'note': 'FETCH_SIZE/WRITE_SIZE (KB) of gn_kernel per dispatch, each scaled by known_bytes/reported_bytes of a 256 MiB '
                 'calibration kernel with the GN kernel\'s own access pattern (four float4 loads + four float4 stores per lane, 64 B lane stride), '
                 'measured in the same rocprofv3 passes, as MI355X_MICROARCH.md prescribes for non-wide access widths'}
  print(json.dumps(out))
  if len(sys.argv) > 2: json.dump(out, open(sys.argv[2], 'w'), indent=1)
except (KeyError, ValueError) as ex:
  print('# traffic not derivable:', ex)
