"""Dev tool: kernel timeline of the last timed steps from a rocprofv3 sqlite database.  python tools/r5_tl.py x_results.db [steps [skip]]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; sym = [t for t in tabs if 'info_kernel_symbol' in t][0]
rows = list(cur.execute(f"select s.kernel_name, d.start, d.end, d.queue_id from {kd} d join {sym} s on d.kernel_id=s.id order by d.start"))
idx = [i for i, r in enumerate(rows) if 'pfb_spec' in r[0]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
off = int(sys.argv[3]) if len(sys.argv) > 3 else 8          # channelizer launches behind the timed steps (checks, one-lane runs)
i0 = idx[-(k + off)]; i1 = idx[-off]; t0 = rows[i0][1]
for r in rows[i0:i1 + 1]:
    nm = r[0].replace('_ZN5snout', '')[:14]
    if (r[2] - r[1]) < 30000: continue
    print('%-16s q%-3d start %8.3f end %8.3f dur %7.3f ms' % (nm, r[3], (r[1] - t0) / 1e6, (r[2] - t0) / 1e6, (r[2] - r[1]) / 1e6))
