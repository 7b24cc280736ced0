#!/usr/bin/env python
"""print the kernel timeline of the last full step in a rocprofv3 (rocpd sqlite) kernel trace:
   start offset, duration, queue/stream and name of every dispatch between two consecutive gather-pass launches"""
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + '/*/*.db')[0]
con = sqlite3.connect(db)
cur = con.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tables if 'kernel_dispatch' in t and not t.startswith('rocpd_info')][0]
ks = [t for t in tables if t.startswith('rocpd_info_kernel_symbol')][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
q = f"select d.start, d.end, d.queue_id, d.stream_id, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"
rows = list(cur.execute(q))
marks = [i for i, r in enumerate(rows) if 'ray_encode' in r[4]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(marks) - 3
a, b = marks[k], marks[k + 1]
t0 = rows[a][0]
for st, en, qid, sid, name in rows[a:b]:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{(st - t0) / 1e3:9.1f} {(en - st) / 1e3:8.1f}  q{qid} s{sid}  {name[:90]}")
print("step span us:", (rows[b][0] - t0) / 1e3)
