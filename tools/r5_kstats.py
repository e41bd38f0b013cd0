"""Dev tool: per-kernel durations from a rocprofv3 sqlite database (rocprofv3 --kernel-trace without --output-format csv).
    python tools/r5_kstats.py gpurun_out/x/y_results.db [name filter]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; sym = [t for t in tabs if 'info_kernel_symbol' in t][0]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
rows = list(cur.execute(f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, max(d.end-d.start)/1e3, sum(d.end-d.start)/1e3 from {kd} d join {sym} s on d.kernel_id=s.id group by s.kernel_name order by 6 desc"))
for r in rows:
    if flt in r[0]:
        print('%-48s n %4d avg %9.1f min %9.1f max %9.1f us' % (r[0].replace('_ZN5snout', '')[:48], r[1], r[2], r[3], r[4]))
