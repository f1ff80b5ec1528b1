#!/usr/bin/env python
"""Read the rocpd databases written by pmc_traffic.sh (gpurun_out/pmc/<workload>/*_results.db) and print per-kernel average
counter values per dispatch; derive the HBM traffic of the GN kernel of every workload with the calibration kernels' correction
factors and write profiles/traffic.json (read by bench.py as roofline.traffic).
  python profiles/tools/pmc_report.py [gpurun_out/pmc] [profiles/traffic.json]"""
import glob, json, os, sqlite3, sys
root = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc'
MiB = 1024.0 * 1024.0
KNOWN_STREAM = 256 * MiB                 # copy and row-pattern calibration kernels: bytes read == bytes written
GATHER_LINES = 8 * 1024 * 1024           # calib_gather8: one (two) 8-byte load(s) in each of 8 Mi 128-byte lines
traffic = {}
for wdir in sorted(glob.glob(os.path.join(root, '*'))):
  if not os.path.isdir(wdir): continue
  w = os.path.basename(wdir)
  vals = {}
  print('## workload %s' % w)
  for f in sorted(glob.glob(os.path.join(wdir, '*_results.db'))):
    cur = sqlite3.connect(f).cursor()
    try:
      rows = cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
    except sqlite3.Error as ex:
      print('# %s: %s' % (f, ex)); continue
    for kname, cname, cnt, avg in rows:
      short = None
      for key, tag in (('gn_kernel', 'gn_kernel'), ('copyBuffer', 'copy'), ('calib_dword_rw', 'calib_rows'), ('calib_gather8ILi1', 'gather1'), ('calib_gather8ILi2', 'gather2'),
                       ('calib_gather8<1>', 'gather1'), ('calib_gather8<2>', 'gather2')):
        if key in kname: short = tag; break
      if short: vals.setdefault(short, {}).setdefault(cname, []).append((cnt, avg, kname[:90]))
    for r in cur.execute("select name, count(*), avg(duration) from kernels where name like '%gn_kernel%' group by name"):
      print('# %s: %s x%d avg %.1f ns' % (os.path.basename(f), r[0][:80], r[1], r[2]))
  for short, cs in sorted(vals.items()):
    for c, lst in sorted(cs.items()):
      for cnt, avg, kn in lst: print('%-10s %-18s dispatches=%-4d avg=%.1f   [%s]' % (short, c, cnt, avg, kn))
  try:
    mx = lambda tag, c: max(v[1] for v in vals[tag][c])
    wide_f = KNOWN_STREAM / (mx('copy', 'FETCH_SIZE') * 1024.0); wide_w = KNOWN_STREAM / (mx('copy', 'WRITE_SIZE') * 1024.0)
    kf = KNOWN_STREAM / (mx('calib_rows', 'FETCH_SIZE') * 1024.0); kw = KNOWN_STREAM / (mx('calib_rows', 'WRITE_SIZE') * 1024.0)
    print('# wide float4 copy: FETCH_SIZE correction %.3f (the guide: 2x), WRITE_SIZE correction %.3f' % (wide_f, wide_w))
    print('# row pattern of the GN kernel: FETCH_SIZE correction %.3f, WRITE_SIZE correction %.3f' % (kf, kw))
    g1 = mx('gather1', 'FETCH_SIZE') * 1024.0; g2 = mx('gather2', 'FETCH_SIZE') * 1024.0
    print('# tap-gather pattern: FETCH_SIZE reports %.1f B per line touched once, %.1f B per line touched in both 64-byte halves' % (g1 / GATHER_LINES, g2 / GATHER_LINES))
    gf = vals['gn_kernel']['FETCH_SIZE'][0][1]; gw = vals['gn_kernel']['WRITE_SIZE'][0][1]
    out = {'fetch_size_kb_raw': gf, 'write_size_kb_raw': gw, 'fetch_correction': kf, 'write_correction': kw,
           'gather_fetch_bytes_per_line_raw': g1 / GATHER_LINES, 'gather_fetch_bytes_per_line_both_halves_raw': g2 / GATHER_LINES,
           'hbm_bytes_per_launch': gf * 1024.0 * kf + gw * 1024.0 * kw,
           'note': 'FETCH_SIZE/WRITE_SIZE (KB) of gn_kernel per dispatch, each scaled by known_bytes/reported_bytes of a 256 MiB calibration kernel '
                   'with the GN kernel\'s own row access pattern measured in the same rocprofv3 passes (MI355X_MICROARCH.md, HBM section)'}
    if 'TCC_HIT_sum' in vals.get('gn_kernel', {}):
      h = vals['gn_kernel']['TCC_HIT_sum'][0][1]; m = vals['gn_kernel']['TCC_MISS_sum'][0][1]
      out['tcc_hit_rate'] = h / (h + m); out['tcc_hit'] = h; out['tcc_miss'] = m
    print(json.dumps(out))
    traffic[w] = out
  except (KeyError, ValueError) as ex:
    print('# traffic not derivable:', ex)
if len(sys.argv) > 2 and traffic:
  if 'gn_step' in traffic: traffic['hbm_bytes_per_launch'] = traffic['gn_step']['hbm_bytes_per_launch']      # (the key round 1 wrote)
  import subprocess, time      # provenance: bench.py quotes it next to roofline.traffic
  try: commit = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, universal_newlines=True).stdout.strip() or None
  except OSError: commit = None
  traffic['collected'] = {'date': time.strftime('%Y-%m-%d', time.gmtime()), 'commit': commit or os.environ.get('DGP_COLLECT_COMMIT', 'round 6')}
  json.dump(traffic, open(sys.argv[2], 'w'), indent=1, sort_keys=True)
